set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "weather or fog or rain or snow or night or normalize" > gpurun_out/s14_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s14_t1.log; tail -4 gpurun_out/s14_t1.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > gpurun_out/s14_bench.json 2> gpurun_out/s14_bench.err && \
AWSEG_WEATHER_BATCH=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --no-parity-pass > gpurun_out/s14_bench_nowb.json 2> gpurun_out/s14_bench_nowb.err
python - <<'PY'
import json
for f in ("gpurun_out/s14_bench.json","gpurun_out/s14_bench_nowb.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("resident_frames"), d["miou"].get("overall_miou"))
        for k in d["kernels"]:
            if any(x in k["kernel"] for x in ("weather","fog","rain","snow","night","normalize")): print("   ", k["kernel"], k["launches_per_step"], k["avg_ms"], k["frac"], k["time_share_of_step"])
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s14_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s14_tests.log; tail -4 gpurun_out/s14_tests.log
