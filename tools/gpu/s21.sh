set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python tools/scratch/subbatch_probe.py > gpurun_out/s21_probe.log 2>&1; echo "rc=$?" >> gpurun_out/s21_probe.log; tail -20 gpurun_out/s21_probe.log
