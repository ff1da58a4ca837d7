set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
AWSEG_TRAIN_KEEP_CL=1 timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 > gpurun_out/s11_train_cl.json 2> gpurun_out/s11_train_cl.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/s11_train_cl.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["peak_hbm_gb"], d["losses"])
    except Exception as e: print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
