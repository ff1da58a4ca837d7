"""How accurate would Winograd F(4x4,3x3) be with this repo's split-operand products?  numpy emulation of one output tile column:
U = G g G^T and V = B^T d B in float32, both split into f16 high + f16 low parts (U normalised into [2^13, 2^14) as csrc/wino_split.hip
does), products hi*hi + hi*lo + lo*hi accumulated in float32 over Cin, inverse transform in float32 — against the float64 direct
convolution, next to the same emulation of F(2x2,3x3).  python tools/scratch/wino43_error.py"""
import numpy as np

rng = np.random.default_rng(0)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
B2T = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
A2T = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
B4T = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
A4T = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)


def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def run(G, BT, AT, cin, cout, scale, tiles=64):
    n, m = BT.shape[0], AT.shape[0]
    g = rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)
    d = np.maximum(rng.standard_normal((tiles, cin, n, n)), 0) * scale          # post-ReLU activations
    ref = np.zeros((tiles, cout, m, m))
    for y in range(m):
        for x in range(m):
            ref[:, :, y, x] = np.einsum("tcij,ocij->to", d[:, :, y:y + 3, x:x + 3], g)
    U = np.einsum("ij,ocjk,lk->ocil", G, g, G).astype(np.float32)
    eu = int(np.floor(np.log2(np.abs(U).max()))) - 13
    Us = (U * np.float32(2.0 ** -eu)).astype(np.float32)
    V = np.einsum("ij,tcjk,lk->tcil", BT.astype(np.float32), d.astype(np.float32), BT.astype(np.float32)).astype(np.float32)
    uh, ul = split(Us)
    vh, vl = split(V)
    M = np.zeros((tiles, cout, n, n), np.float32)
    for c0 in range(0, cin, 16):                                              # float32 accumulation, 16 channels a step as the MFMA
        sl = slice(c0, c0 + 16)
        part = (np.einsum("tcij,ocij->toij", vh[:, sl].astype(np.float64), uh[:, sl].astype(np.float64))
                + np.einsum("tcij,ocij->toij", vh[:, sl].astype(np.float64), ul[:, sl].astype(np.float64))
                + np.einsum("tcij,ocij->toij", vl[:, sl].astype(np.float64), uh[:, sl].astype(np.float64)))
        M = (M + part.astype(np.float32)).astype(np.float32)
    out = np.einsum("ij,tojk,lk->toil", AT.astype(np.float32), M, AT.astype(np.float32)).astype(np.float32) * np.float32(2.0 ** eu)
    exact = np.einsum("ij,tojk,lk->toil", AT, np.einsum("tcij,ocij->toij", V.astype(np.float64), U.astype(np.float64)), AT)
    return np.abs(out - ref).max(), np.abs(exact - ref).max(), np.abs(ref).max(), np.abs(V).max() / max(np.abs(d).max(), 1e-30)


for cin, cout in ((64, 64), (128, 64), (512, 64), (2048, 64)):
    for scale in (1.0, 30.0):
        e2 = run(G2, B2T, A2T, cin, cout, scale)
        e4 = run(G4, B4T, A4T, cin, cout, scale)
        print(f"Cin {cin:5d} act x{scale:4.0f}: F(2,3) split |err| {e2[0]:.2e} (transforms alone in f32 {e2[1]:.2e}), F(4,3) split {e4[0]:.2e} "
              f"(transforms alone {e4[1]:.2e}); |out| max {e4[2]:.1f}; V growth x{e2[3]:.0f} / x{e4[3]:.0f}")
