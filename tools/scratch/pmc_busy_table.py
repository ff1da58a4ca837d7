"""profiles/<round>_pmc_step_busy.csv from a rocprofv3 --kernel-trace --pmc pass of bench.py: per device kernel, mean per dispatch of the SQ
busy counters (summed over the chip) and the ratios that say which pipe the waves were on.
    python tools/scratch/pmc_busy_table.py gpurun_out/r03_prof_step_busy profiles/r03_pmc_step_busy.csv"""
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
names = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
rows = []
for k, c in acc.items():
    n = len(disp[k]); m = {x: c.get(x, 0.0) / n for x in names}
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    rows.append((c.get("SQ_BUSY_CYCLES", 0.0), k, n, m, m["SQ_VALU_MFMA_BUSY_CYCLES"] / wc, m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_ACTIVE_INST_LDS"] / wc))
rows.sort(reverse=True)
with open(dst, "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc " + " ".join(names) + " -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0\n")
    o.write("# the " + rnd + " build; mean per dispatch, counters summed over the chip; sorted by total SQ_BUSY_CYCLES; ratios: share of the wave-cycles with the matrix pipe busy / a VALU instruction active / an LDS instruction active\n")
    o.write("kernel,dispatches," + ",".join(names) + ",mfma_busy_over_wave_cycles,valu_active_over_wave_cycles,lds_active_over_wave_cycles\n")
    for _, k, n, m, a, b, c in rows[:40]:
        o.write('"' + k.replace('"', "")[:150] + '",' + str(n) + "," + ",".join(f"{m[x]:.4g}" for x in names) + f",{a:.3f},{b:.3f},{c:.3f}\n")
print("wrote", dst, len(rows), "kernels")
