set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/s8_train_nchw.json 2> gpurun_out/s8_train_nchw.err; echo "rc=$?"
AWSEG_NCHW_GRADS=0 timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 > gpurun_out/s8_train_old.json 2> gpurun_out/s8_train_old.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/s8_train_nchw.json","gpurun_out/s8_train_old.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["peak_hbm_gb"], d["losses"], d.get("replicas_identical"))
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c4" -rP > gpurun_out/s8_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s8_tests.log; grep -n "C4\|N=2\|passed\|failed\|rc=" gpurun_out/s8_tests.log | tail
