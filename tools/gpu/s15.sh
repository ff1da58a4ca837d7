set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
AWSEG_TWO_STREAMS=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 > gpurun_out/s15_bench_2s.json 2> gpurun_out/s15_bench_2s.err
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 > gpurun_out/s15_bench_1s.json 2> gpurun_out/s15_bench_1s.err
AWSEG_TWO_STREAMS=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s15_bench_2s_b.json 2> gpurun_out/s15_bench_2s_b.err
python - <<'PY'
import json
for f in ("gpurun_out/s15_bench_2s.json","gpurun_out/s15_bench_1s.json","gpurun_out/s15_bench_2s_b.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("resident_frames"), d["miou"])
    except Exception as e: print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1200:])
PY
AWSEG_TWO_STREAMS=1 timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s15_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s15_tests.log; tail -3 gpurun_out/s15_tests.log
