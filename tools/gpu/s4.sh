set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mixffn or maxpool or forms or depth_head" -rP > gpurun_out/s4_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s4_t1.log; grep -n "mixffn fused\|passed\|failed\|rc=" gpurun_out/s4_t1.log | tail -12
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err && \
AWSEG_MIXFFN_FUSED=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s4_bench_nomix.json 2> gpurun_out/s4_bench_nomix.err
python - <<'PY'
import json
for f in ("gpurun_out/s4_bench.json","gpurun_out/s4_bench_nomix.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("resident_frames"))
        for k in d["kernels"][:22]: print("   ", k["kernel"], k["launches_per_step"], k["avg_ms"], k["frac"], k["time_share_of_step"])
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s4_tests.log; tail -4 gpurun_out/s4_tests.log
