set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python tools/kernel_bench.py --iters 10 --only "depth_head_fused,upconv_forms,upconv3x3_bn_relu,winograd 128->64" > gpurun_out/s2_kb.log 2>&1
grep -v '^{' gpurun_out/s2_kb.log | tail -12
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > gpurun_out/s2_bench_fused.json 2> gpurun_out/s2_bench_fused.err && \
AWSEG_DEPTH_FUSED=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s2_bench_unfused.json 2> gpurun_out/s2_bench_unfused.err
python - <<'PY'
import json
for f in ("gpurun_out/s2_bench_fused.json","gpurun_out/s2_bench_unfused.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("resident_frames"))
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s2_tests.log; tail -4 gpurun_out/s2_tests.log
