set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python tools/scratch/leftovers_probe.py > gpurun_out/s20_probe.log 2>&1; echo "rc=$?" >> gpurun_out/s20_probe.log; tail -60 gpurun_out/s20_probe.log
