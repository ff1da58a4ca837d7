set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s33_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s33_tests.log; tail -4 gpurun_out/s33_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s33_smoke.log 2>&1; tail -2 gpurun_out/s33_smoke.log
