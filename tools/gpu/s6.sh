set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
echo "== default FAST" > gpurun_out/s6_wrw.log
timeout -k 10 200 python tools/scratch/wrw_probe.py >> gpurun_out/s6_wrw.log 2>&1
for v in MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_WRW_XDLOPS MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_WRW_V4R4_XDLOPS MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS MIOPEN_DEBUG_CONV_IMPLICIT_GEMM; do
  echo "== $v=0" >> gpurun_out/s6_wrw.log
  env $v=0 timeout -k 10 200 python tools/scratch/wrw_probe.py >> gpurun_out/s6_wrw.log 2>&1
done
echo "== MIOPEN_FIND_MODE=NORMAL (find, benchmark candidates) first shapes only" >> gpurun_out/s6_wrw.log
grep -v "^+" gpurun_out/s6_wrw.log | grep -v amdgpu.ids | tail -40
