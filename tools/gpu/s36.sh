set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "k_blocked or aspp or deeplab or ensemble or gemm_split" > gpurun_out/s36_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s36_t1.log; tail -5 gpurun_out/s36_t1.log
for cfg in 1 0 1 0; do
AWSEG_ASPP_KBLOCKED=$cfg timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s36_bench.json 2> gpurun_out/s36_bench.err && python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s36_bench.json").read().strip().splitlines()[-1])
print("aspp_kblocked", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
cd /tmp
for cfg in 1 0; do
AWSEG_TWO_STREAMS=0 AWSEG_ASPP_KBLOCKED=$cfg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof36_$cfg -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/scratch/step_timeline.py $(find /tmp/prof36_$cfg -name "*kernel_trace.csv" | head -1) | grep -A12 "aspp_dw3" | cut -c1-110
done
