set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s10_prof_train -o train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --kernel-steps 0 > $O/s10_prof_train.log 2>&1; echo "prof train exit $?"
python3 - <<'PY'
import csv, glob, os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/s10_prof_train/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per step", tot/3e6)
for r in rows[:28]: print(r["Calls"], round(float(r["TotalDurationNs"])/3e6,1), "ms/step", r["Name"][:110])
PY
rm -rf $O/s10_prof_train/*/*kernel_trace* 2>/dev/null
