set -x
timeout -k 10 120 tools/scratch/build/mall_probe > gpurun_out/s22_mall.log 2>&1; cat gpurun_out/s22_mall.log
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_bf16.py -x -q -m gpu -k "upsample or attention or rowdot or segformer or deeplab or mit or ensemble or aspp or bf16 or b5 or r101" > gpurun_out/s22_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s22_t1.log; tail -5 gpurun_out/s22_t1.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > gpurun_out/s22_bench.json 2> gpurun_out/s22_bench.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/s22_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof22 -o s22 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof22/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(int(r["Calls"]) for r in rows)
print("kernel launches in 9 steps:", tot, "per step ~", tot / 9.0)
for r in rows:
    if "Cijk" in r["Name"] or "at::" in r["Name"] or "rocclr" in r["Name"]:
        print(r["Calls"], r["Name"][:110])
PY
