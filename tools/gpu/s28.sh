set -x
cd /tmp
export TMPDIR=/tmp
export AWSEG_TWO_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof28 -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > /dev/null 2>&1; echo "prof step exit $?"
cd $GRAFT_REPO_ROOT
python tools/scratch/step_timeline.py $(find /tmp/prof28 -name "*kernel_trace.csv" | head -1) > gpurun_out/s28_timeline.log 2>&1
tail -3 gpurun_out/s28_timeline.log
