set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "pieces or dual or aspp or deeplab or ensemble or gemm_split" > gpurun_out/s32_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s32_t1.log; tail -5 gpurun_out/s32_t1.log
for cfg in 1 0 1 0; do
AWSEG_ASPP_PIECES=$cfg timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s32_bench.json 2> gpurun_out/s32_bench.err && python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s32_bench.json").read().strip().splitlines()[-1])
print("aspp_pieces", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
