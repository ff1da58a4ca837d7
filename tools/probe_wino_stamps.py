import ctypes, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops, _native as N
B, H, W, cin, cout = 8, 64, 128, 2048, 256
x = torch.randn(B, H, W, cin, device="cuda"); u = ops.winograd_weights(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05); sh = torch.randn(cout, device="cuda")
for _ in range(3):
    ops.conv3x3_winograd(x, u, sh, act=1)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 16)()
N.lib().awseg_debug_wino_stamps.argtypes = [ctypes.c_void_p]
print("rc", N.lib().awseg_debug_wino_stamps(buf))
nch = cin // 8
for w in range(4):
    v = [buf[w * 4 + i] for i in range(4)]
    print(f"wave {w}: per chunk: issue-phase {v[0]/nch:.0f}  glds_wait {v[1]/nch:.0f}  barrier {v[2]/nch:.0f}  total loop {v[3]/nch:.0f} (clock64 ticks)")
