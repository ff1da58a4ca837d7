# Turns the rocprofv3 directories of tools/gpu_profile_session.sh (merged back into gpurun_out/) into the committed summaries under
# profiles/ for round $R (CPU is enough):  bash tools/make_round_profiles.sh
set -e
R=${R:-r04}
python tools/make_profiles.py step gpurun_out/${R}_prof_step ensemble_stats_kernel profiles/${R}_bench_step_kernels.csv \
  "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 (the shipped build of the round, ingestion on)"
python tools/make_profiles.py kernel-table gpurun_out/${R}_prof_step gpurun_out/${R}_prof_step_fetch gpurun_out/${R}_prof_step_write profiles/${R}_bench_step_stats_and_traffic.csv \
  "bench.py under rocprofv3: --kernel-trace --stats pass + separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; mean per dispatch over the launches of the timed steps" mean
[ -d gpurun_out/${R}_prof_step_busy ] && python tools/scratch/pmc_busy_table.py gpurun_out/${R}_prof_step_busy profiles/${R}_pmc_step_busy.csv ${R}
python tools/make_profiles.py kernel-table gpurun_out/${R}_prof_kb_trace gpurun_out/${R}_prof_kb_fetch gpurun_out/${R}_prof_kb_write profiles/${R}_kernel_bench_stats_and_traffic.csv \
  "tools/kernel_bench.py --iters 2 --only winograd,gemm_l4,gemm_aspp,combine,fog,night,rain,snow,normalize,segformer_head,stats,ece,aspp_dep,dwconv,depth_head_fused,upconv_forms,mixffn under rocprofv3 (kernel trace + separate FETCH_SIZE / WRITE_SIZE passes), median per dispatch"
python tools/make_profiles.py kernel-table gpurun_out/${R}_prof_b5 gpurun_out/${R}_prof_b5_fetch gpurun_out/${R}_prof_b5_write profiles/${R}_bench_b5_step_stats_and_traffic.csv \
  "bench.py --model b5_r101 (BASELINE config 5: SegFormer-B5 + DeepLabV3+-R101, bf16 MFMA path) under rocprofv3: --kernel-trace --stats pass + separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; mean per dispatch over the launches of the timed steps" mean
python tools/make_profiles.py kernel-table gpurun_out/${R}_prof_train gpurun_out/${R}_prof_train_fetch gpurun_out/${R}_prof_train_write profiles/${R}_train_step_stats_and_traffic.csv \
  "bench.py --mode train (BASELINE config 4, 1024x2048, batch 8) under rocprofv3: --kernel-trace --stats pass (3 steps) + separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (2 steps); hand-written kernels only, mean per dispatch" mean
cp gpurun_out/${R}_kernel_bench_hip_events.log gpurun_out/${R}_bench_line_default.json gpurun_out/${R}_bench_line_b5_r101_bf16.json gpurun_out/${R}_bench_line_train_1024x2048_bs8.json profiles/
python tools/make_profiles.py check-log profiles/${R}_kernel_bench_hip_events.log | tail -1
R=$R python - <<'PY'
import csv, json, os
R = os.environ["R"]
rows = list(csv.DictReader(open(f"gpurun_out/{R}_prof_train/train_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"profiles/{R}_train_step_kernels.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --kernel-steps 0   (1024x2048, batch 8, the shipped build: 1x1 convolutions as F.linear)\n")
    f.write(f"# 3 optimisation steps in the trace (1 warm-up + 2 timed): {tot / 1e6:.1f} ms of kernel time = {tot / 3e6:.0f} ms per step; wall time per step: profiles/{R}_bench_line_train_1024x2048_bs8.json\n")
    f.write("kernel,calls,total_ms,avg_us,percent\n")
    for r in rows[:60]:
        n = r["Name"].replace('"', "")[:160]
        f.write(f"\"{n}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.1f},{float(r['TotalDurationNs']) / tot * 100:.2f}\n")
for f in ("default", "b5_r101_bf16", "train_1024x2048_bs8"):
    d = json.loads(open(f"profiles/{R}_bench_line_{f}.json").read().strip().splitlines()[-1])
    rf = d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], rf.get("kernel"), rf.get("frac"), rf.get("traffic"))
PY
