set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "aspp or stem or deeplab or ensemble or upsample or rowdot" > gpurun_out/s29_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s29_t1.log; tail -5 gpurun_out/s29_t1.log
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s29_bench.json 2> gpurun_out/s29_bench.err && python - <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s29_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"])
PY
done
cd /tmp
export AWSEG_TWO_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof29 -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > /dev/null 2>&1; echo "prof step exit $?"
cd $GRAFT_REPO_ROOT
python tools/scratch/step_timeline.py $(find /tmp/prof29 -name "*kernel_trace.csv" | head -1) > gpurun_out/s29_timeline.log 2>&1
tail -2 gpurun_out/s29_timeline.log
