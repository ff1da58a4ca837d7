import sys, torch, numpy as np
sys.path.insert(0, '.')
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(32768, 256, 64), (32768, 256, 128), (65536, 256, 1024)]:
    x = torch.randint(-4, 5, (M, K), device='cuda').float()
    w = torch.randint(-4, 5, (N, K), device='cuda').float()
    b = torch.zeros(N, device='cuda')
    ws = ops.gemm_split_weights(w)
    got = ops.gemm_split_bias_act(x, ws, b, 0)
    ref = x.double() @ w.double().t()
    bad = (got.double() - ref).abs() > 1e-3
    print(M, N, K, 'bad frac', bad.float().mean().item())
    if bad.any():
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        print(' bad rows', rows.numel(), rows[:20].tolist(), ' row%256 hist', torch.bincount(rows % 256, minlength=256).nonzero().flatten()[:40].tolist())
        print(' bad cols', cols.numel(), cols[:40].tolist())
        tiles = torch.bincount(rows // 256)
        print(' tiles with bad rows', (tiles > 0).sum().item(), 'of', M // 256)
        r0 = rows[0].item()
        print(' row', r0, 'bad cols', bad[r0].nonzero().flatten()[:64].tolist())
        print(' got', got[r0, bad[r0]][:8].tolist(), 'ref', ref[r0, bad[r0]][:8].tolist())
    # second run: deterministic?
    got2 = ops.gemm_split_bias_act(x, ws, b, 0)
    print(' rerun equal', torch.equal(got, got2))
