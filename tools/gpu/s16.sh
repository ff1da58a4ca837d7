set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s16_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s16_tests.log; tail -4 gpurun_out/s16_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s16_smoke.log 2>&1; tail -2 gpurun_out/s16_smoke.log
R=r04 PART=a bash tools/gpu_profile_session.sh
