set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "upconv3x3_train or bn_relu or training or c4 or heads_on_hip" > gpurun_out/s19_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s19_t1.log; tail -3 gpurun_out/s19_t1.log
R=r04 PART=b bash tools/gpu_profile_session.sh
