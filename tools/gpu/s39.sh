set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 0 1; do
AWSEG_HEAD_SCALAR_FMA=$v python -m adverse_weather_semantic_segmentation_robustness_benchmark_amd.csrc.build > gpurun_out/s39_build_$v.log 2>&1; tail -1 gpurun_out/s39_build_$v.log
AWSEG_HEAD_SCALAR_FMA=$v timeout -k 10 300 python tools/kernel_bench.py --iters 10 --only segformer_head > gpurun_out/s39_kb_$v.log 2>&1; grep "split" gpurun_out/s39_kb_$v.log | grep -v "^{"
done
AWSEG_HEAD_SCALAR_FMA=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "head or segformer or ensemble" > gpurun_out/s39_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s39_t1.log; tail -3 gpurun_out/s39_t1.log
AWSEG_HEAD_SCALAR_FMA=1 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s39_bench.json 2> gpurun_out/s39_bench.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/s39_bench.json").read().strip().splitlines()[-1])
print("bench scalar", d["value"], d["ms_per_step"])
PY
