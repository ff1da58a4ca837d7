"""Cycle stamps of gemm_split3 (block 0, waves 0 and 4): build the library with -DAWSEG_G3_STAMP (see the end of this file), then
    python tools/scratch/g3_stamps.py            # on the GPU box
prints, per K tile: wait + barrier | DMA issue at the top | first 16-deep step | (late DMA +) second step."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops, _native as N

lib = N.lib() if hasattr(N, "lib") else N._LIB
fn = lib.awseg_debug_g3_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for (m, n, k) in [(65536, 512, 2048), (65536, 256, 1024), (1048576, 256, 256), (1048576, 256, 64), (262144, 512, 128)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    ws = ops.gemm_split_weights(w)
    for _ in range(2): ops.gemm_split_bias_act(x, ws, b, 1)
    torch.cuda.synchronize(); fn(buf, 1)
    ops.gemm_split_bias_act(x, ws, b, 1); torch.cuda.synchronize(); fn(buf, 1)
    for g in range(2):
        v = [buf[8 * g + i] for i in range(6)]; t = max(v[4], 1)
        print(f"M={m} N={n} K={k} wave {4 * g}: per K tile: wait+barrier {v[0] / t:.0f} | DMA at top {v[1] / t:.0f} | step 0 {v[2] / t:.0f} | step 1 {v[3] / t:.0f} | sum {(v[0] + v[1] + v[2] + v[3]) / t:.0f}  ({t} K tiles, block total {v[5]})")
# hipcc ... -DAWSEG_G3_STAMP -c gemm_split3.hip, linked with the other objects of csrc/build into a copy of libawseg_hip.so (tools/ab_lib.sh)
