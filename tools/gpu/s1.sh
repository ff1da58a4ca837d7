set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "forms or depth_head or ragged or stem_pad or fog_night or normalize" -rP > gpurun_out/s1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s1_tests.log
tail -5 gpurun_out/s1_tests.log
