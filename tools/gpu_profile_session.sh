# One GPU-box session that produces every file under profiles/ for this round (run through gpurun; outputs land in gpurun_out/,
# tools/make_profiles.py turns the rocprofv3 directories into the committed CSV summaries).
set -x
cd $GRAFT_REPO_ROOT
R=${R:-r04}
PART=${PART:-all}          # a: HIP-event lines + the step's trace / counters; b: kernel_bench trace / counters, B5, train (a gpurun call is capped at 20 min)
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0"
K="$GRAFT_REPO_ROOT/tools/kernel_bench.py --iters 2 --only winograd,gemm_l4,gemm_aspp,combine,fog,night,rain,snow,normalize,segformer_head,stats,ece,aspp_dep,dwconv,depth_head_fused,upconv_forms,mixffn"
O=$GRAFT_REPO_ROOT/gpurun_out
if [ "$PART" != "b" ]; then
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/kernel_bench.py --iters 10 > gpurun_out/${R}_kernel_bench_hip_events.log 2>&1; echo "kernel_bench exit $?"
timeout -k 10 400 python bench.py > gpurun_out/${R}_bench_line_default.json 2> gpurun_out/${R}_bench_line_default.err; echo "bench exit $?"
cd /tmp
export AWSEG_TWO_STREAMS=0      # the profiler passes time / count each kernel with the chip to itself (the bench lines above run the two ensemble members on two streams)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0 > $O/${R}_prof_step.log 2>&1; echo "prof step exit $?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_prof_step_fetch -o step -- python3 $B > $O/${R}_prof_step_fetch.log 2>&1; echo "step fetch exit $?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_prof_step_write -o step -- python3 $B > $O/${R}_prof_step_write.log 2>&1; echo "step write exit $?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${R}_prof_step_busy -o step -- python3 $B > $O/${R}_prof_step_busy.log 2>&1; echo "step busy exit $?"
rm -f $O/${R}_prof_step_busy/*/*kernel_trace* 2>/dev/null
unset AWSEG_TWO_STREAMS
fi
if [ "$PART" != "a" ]; then
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --model b5_r101 > gpurun_out/${R}_bench_line_b5_r101_bf16.json 2> gpurun_out/${R}_bench_line_b5.err; echo "bench b5 exit $?"
cd /tmp
export AWSEG_TWO_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_kb_trace -o kb -- python3 $K > $O/${R}_prof_kb_trace.log 2>&1; echo "kb trace exit $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_prof_kb_fetch -o kb -- python3 $K > $O/${R}_prof_kb_fetch.log 2>&1; echo "kb fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_prof_kb_write -o kb -- python3 $K > $O/${R}_prof_kb_write.log 2>&1; echo "kb write exit $?"
B5="$GRAFT_REPO_ROOT/bench.py --model b5_r101 --steps 3 --warmup 2 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass --resident-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_b5 -o step -- python3 $B5 > $O/${R}_prof_b5.log 2>&1; echo "b5 trace exit $?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_prof_b5_fetch -o step -- python3 $B5 > $O/${R}_prof_b5_fetch.log 2>&1; echo "b5 fetch exit $?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_prof_b5_write -o step -- python3 $B5 > $O/${R}_prof_b5_write.log 2>&1; echo "b5 write exit $?"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_train -o train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --kernel-steps 0 > $O/${R}_prof_train.log 2>&1; echo "prof train exit $?"
TR="$GRAFT_REPO_ROOT/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline --kernel-steps 0"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_prof_train_fetch -o train -- python3 $TR > $O/${R}_prof_train_fetch.log 2>&1; echo "train fetch exit $?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_prof_train_write -o train -- python3 $TR > $O/${R}_prof_train_write.log 2>&1; echo "train write exit $?"
unset AWSEG_TWO_STREAMS
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --mode train --steps 3 --warmup 2 > gpurun_out/${R}_bench_line_train_1024x2048_bs8.json 2> gpurun_out/${R}_bench_line_train.err; echo "bench train exit $?"
fi
du -sh $O/${R}_prof_* | tail -8
# Afterwards, in the repo (CPU is enough):
#   bash tools/make_round_profiles.sh        # or, step by step:
#   python tools/make_profiles.py step gpurun_out/r03_prof_step ensemble_stats_kernel profiles/r03_bench_step_kernels.csv "<header>"
#        (marker = a kernel that runs once per step: the one-pass confusion + statistics kernel)
#   python tools/make_profiles.py kernel-table gpurun_out/r03_prof_step gpurun_out/r03_prof_step_fetch gpurun_out/r03_prof_step_write \
#        profiles/r03_bench_step_stats_and_traffic.csv "<header>" mean
#   python tools/make_profiles.py kernel-table gpurun_out/r03_prof_kb_trace gpurun_out/r03_prof_kb_fetch gpurun_out/r03_prof_kb_write \
#        profiles/r03_kernel_bench_stats_and_traffic.csv "<header>"
#   python tools/make_profiles.py kernel-table gpurun_out/r03_prof_b5 gpurun_out/r03_prof_b5_fetch gpurun_out/r03_prof_b5_write \
#        profiles/r03_bench_b5_step_stats_and_traffic.csv "<header>" mean
#   cp gpurun_out/r03_kernel_bench_hip_events.log gpurun_out/r03_bench_line_default.json gpurun_out/r03_bench_line_b5_r101_bf16.json profiles/
#   python tools/make_profiles.py check-log profiles/r03_kernel_bench_hip_events.log
