set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s3_tests.log; tail -4 gpurun_out/s3_tests.log
timeout -k 10 400 python tools/kernel_bench.py --iters 10 --only "depth_head_fused,upconv_forms" > gpurun_out/s3_kb.log 2>&1
grep -v '^{' gpurun_out/s3_kb.log | tail -4
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/s3_bench.json") if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d["roofline"])
for k in d["kernels"][:14]: print(k["kernel"], k["launches_per_step"], k["avg_ms"], k["frac"], k["time_share_of_step"])
PY
