set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bn_relu_dropout2d" > gpurun_out/s18_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s18_t1.log; tail -3 gpurun_out/s18_t1.log
timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/s18_train.json 2> gpurun_out/s18_train.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/s18_train.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["peak_hbm_gb"], d["losses"], d["roofline"]["traffic"])
    except Exception as e: print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py -x -q -m gpu -k "c4 or train" -rP > gpurun_out/s18_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s18_tests.log; grep -n "C4\|passed\|failed\|rc=" gpurun_out/s18_tests.log | tail -5
