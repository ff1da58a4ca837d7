set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "philox or training_forward_backward" 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 500 python tools/kernel_bench.py --iters 10 > gpurun_out/r02_kernel_bench_hip_events.log 2>&1; echo "kernel_bench exit $?"
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_line_default.json 2> gpurun_out/r02_bench_line_default.err; echo "bench exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --kernel-steps 0 --fp32-steps 0 --no-parity-pass > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_step.log 2>&1; echo "prof step exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_trace -o kb -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py --iters 3 --only "winograd,gemm l4,gemm aspp,combine,fog,night,rain,snow,normalize" > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_trace.log 2>&1; echo "kb trace exit $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_fetch -o kb -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py --iters 2 --only "winograd,gemm l4,gemm aspp,combine,fog,night,rain,snow,normalize" > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_fetch.log 2>&1; echo "kb fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_write -o kb -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py --iters 2 --only "winograd,gemm l4,gemm aspp,combine,fog,night,rain,snow,normalize" > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_write.log 2>&1; echo "kb write exit $?"
ls $GRAFT_REPO_ROOT/gpurun_out/r02_prof_kb_fetch | head
