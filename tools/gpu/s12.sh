set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py -x -q -m gpu -k "c4 or train" -rP > gpurun_out/s12_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s12_tests.log; grep -n "C4\|N=2\|passed\|failed\|rc=" gpurun_out/s12_tests.log | tail
