set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/s17_train.json 2> gpurun_out/s17_train.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/s17_train.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["peak_hbm_gb"], d["losses"], d["roofline"]["traffic"])
    except Exception as e: print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py tests/test_gpu_bf16.py -x -q -m gpu -k "c4 or train or as_written_graph" -rP > gpurun_out/s17_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s17_tests.log; grep -n "C4\|N=2\|passed\|failed\|rc=\|b5+r101" gpurun_out/s17_tests.log | tail
