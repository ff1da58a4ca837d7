set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "0 deeplab" "-1 deeplab" "0 segformer" "-1 segformer" "0 deeplab"; do
set -- $cfg
AWSEG_SIDE_PRIORITY=$1 AWSEG_SIDE_MEMBER=$2 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s37_bench.json 2> gpurun_out/s37_bench.err && python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s37_bench.json").read().strip().splitlines()[-1])
print("side prio/member", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
