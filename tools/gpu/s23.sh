set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 1" "0 0" "1 1" "0 1" "1 0"; do
set -- $cfg
AWSEG_KV_PACKED=$1 AWSEG_SMALL_CONV_SPLIT=$2 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fp32-steps 0 --kernel-steps 0 --no-parity-pass > gpurun_out/s23_bench.json 2> gpurun_out/s23_bench.err && python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s23_bench.json").read().strip().splitlines()[-1])
print("kv_packed small_conv", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof23 -o s23 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof23/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(int(r["Calls"]) for r in rows)
print("kernel launches in 9 steps:", tot, "per step ~", tot / 9.0)
for r in rows:
    if "Cijk" in r["Name"] or "at::" in r["Name"] or "rocclr" in r["Name"]:
        print(r["Calls"], r["Name"][:110])
PY
