set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s38_smoke.log 2>&1; tail -2 gpurun_out/s38_smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s38_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s38_tests.log; tail -3 gpurun_out/s38_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --fp32-steps 0 > gpurun_out/s38_bench.json 2> gpurun_out/s38_bench.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/s38_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
