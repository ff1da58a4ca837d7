set -x
cd $GRAFT_REPO_ROOT
R=r03
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout -k 10 500 python bench.py --model b5_r101 > gpurun_out/${R}_bench_line_b5_r101_bf16.json 2> gpurun_out/${R}_bench_line_b5.err; echo "bench b5 exit $?"
cd /tmp && export TMPDIR=/tmp
B5="$GRAFT_REPO_ROOT/bench.py --model b5_r101 --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_b5 -o step -- python3 $B5 > $O/${R}_prof_b5.log 2>&1; echo "b5 trace exit $?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_prof_b5_fetch -o step -- python3 $B5 > $O/${R}_prof_b5_fetch.log 2>&1; echo "b5 fetch exit $?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_prof_b5_write -o step -- python3 $B5 > $O/${R}_prof_b5_write.log 2>&1; echo "b5 write exit $?"
