"""CPU prototype (float64) of the bilinear-FORMS restatement of relu(shift + conv3x3(interpolate_x32(f))):

Inside one cell of the x32 bilinear upsampling the pre-activation is an exact bilinear form A + B t + C s + D t s of the local
pixel coordinates; pixels whose 3x3 window crosses a cell boundary or the image border are single columns / rows (forms E + F s,
E' + F' t) or single pixels (constants).  This script builds the two tables the fused depth-head kernel reads (F4 / F2, the
layout of awseg_upconv_forms) and evaluates them per 16x16 tile exactly the way the kernel's patch generator does, against
torch's interpolate -> conv2d in float64.  Pure test / design infrastructure: nothing in the product imports it.
"""
import numpy as np
import torch
import torch.nn.functional as F

S = 32          # upsampling factor
TILE = 16


def cell_of(x):
    return -1 if x < 16 else (x - 16) // S


def cell_origin(k):
    return 16 + S * k


def desc_free(k, d, n):
    """1-D descriptor of tap d for a free interior pixel of cell k (n low-res samples): (i0, i1, p, q), lambda = p + q * t."""
    if k < 0:
        return (0, 0, 0.0, 0.0)
    if k >= n - 1:
        return (n - 1, n - 1, 0.0, 0.0)
    return (k, k + 1, (d + 0.5) / S, 1.0 / S)


def desc_fixed(x, d, n):
    """1-D descriptor of tap d for the fixed pixel x; None = the tap is outside the image (zero padding)."""
    xx = x + d
    if xx < 0 or xx >= S * n:
        return None
    k = cell_of(xx)
    if k < 0:
        return (0, 0, 0.0, 0.0)
    if k >= n - 1:
        return (n - 1, n - 1, 0.0, 0.0)
    t = xx - cell_origin(k)
    return (k, k + 1, (t + 0.5) / S, 0.0)


def entry(G, rows, cols):
    """sum over the valid taps of the bilinear form; G [h, w, 9, C]; rows / cols: three 1-D descriptors (dy / dx = -1, 0, 1).
    Returns (const, t, s, ts) coefficient vectors [C]."""
    C = G.shape[-1]
    out = [np.zeros(C) for _ in range(4)]
    for iy, ry in enumerate(rows):
        if ry is None:
            continue
        for ix, rx in enumerate(cols):
            if rx is None:
                continue
            tap = iy * 3 + ix
            y0, y1, py, qy = ry
            x0, x1, px, qx = rx
            g00, g01, g10, g11 = G[y0, x0, tap], G[y0, x1, tap], G[y1, x0, tap], G[y1, x1, tap]
            gx, gy, gxy = g01 - g00, g10 - g00, g00 - g01 - g10 + g11
            out[0] += g00 + gx * px + gy * py + gxy * px * py
            out[1] += gx * qx + gxy * qx * py
            out[2] += gy * qy + gxy * px * qy
            out[3] += gxy * qx * qy
    return out


def special_coords(n):
    """special pixel coordinates along one axis with n low-res samples, in id order"""
    N = S * n
    ids = [0]
    for k in range(n):
        ids += [15 + S * k, 16 + S * k]
    ids.append(N - 1)
    return ids


def special_id(x, n):
    N = S * n
    if x == 0:
        return 0
    if x == N - 1:
        return 2 * n + 1
    if x % S == 15:
        return 1 + 2 * (x // S)
    if x % S == 16:
        return 2 + 2 * (x // S)
    return -1


def build_tables(G, shift):
    """F4 [3h+3][w+1][4][C], F2 [3h+3][2w+2][2][C] (float64 here)."""
    h, w, _, C = G.shape
    nrs = 3 * h + 3
    F4 = np.zeros((nrs, w + 1, 4, C))
    F2 = np.zeros((nrs, 2 * w + 2, 2, C))
    sy, sx = special_coords(h), special_coords(w)
    rowdescs = []
    for ky in range(-1, h):
        rowdescs.append(([desc_free(ky, d, h) for d in (-1, 0, 1)], True))
    for y in sy:
        rowdescs.append(([desc_fixed(y, d, h) for d in (-1, 0, 1)], False))
    for rs, (rows, rfree) in enumerate(rowdescs):
        for kx in range(-1, w):
            cols = [desc_free(kx, d, w) for d in (-1, 0, 1)]
            c0, ct, cs, cts = entry(G, rows, cols)
            F4[rs, kx + 1] = np.stack([c0 + shift, ct, cs, cts])            # special rows: cs = cts = 0 by construction (qy = 0)
        for ci, x in enumerate(sx):
            cols = [desc_fixed(x, d, w) for d in (-1, 0, 1)]
            c0, ct, cs, cts = entry(G, rows, cols)
            assert np.all(ct == 0) and np.all(cts == 0)
            F2[rs, ci] = np.stack([c0 + shift, cs])
    return F4, F2


def generate_tile(F4, F2, h, w, my, mx):
    """the 18 x 18 patch (post-ReLU, zero outside the image) of tile (my, mx) the way the kernel's generator builds it"""
    H, W = S * h, S * w
    C = F4.shape[-1]
    y0, x0 = TILE * my, TILE * mx
    ky, kx = cell_of(y0), cell_of(x0)
    patch = np.zeros((18, 18, C))
    for r in range(18):
        y = y0 - 1 + r
        if y < 0 or y >= H:
            continue
        rid = special_id(y, h)
        if rid >= 0:
            assert r in (0, 1, 16, 17)
            rs, s = h + 1 + rid, 0.0
        else:
            assert cell_of(y) == ky
            rs, s = ky + 1, float(y - cell_origin(ky))
        A, B, Cc, D = F4[rs, kx + 1]
        R, Sl = A + Cc * s, B + D * s
        for pc in range(18):
            x = x0 - 1 + pc
            if x < 0 or x >= W:
                continue
            cid = special_id(x, w)
            if cid >= 0:
                assert pc in (0, 1, 16, 17)
                E, Fs = F2[rs, cid]
                v = E + Fs * s
            else:
                assert cell_of(x) == kx
                v = R + Sl * float(x - cell_origin(kx))
            patch[r, pc] = np.maximum(v, 0.0)
    return patch


def reference(G, shift):
    """relu(shift + sum_tap shift_tap(up(G_tap))) with zero padding, float64, via torch"""
    h, w, _, C = G.shape
    g = torch.from_numpy(G).permute(2, 3, 0, 1)                                # [9, C, h, w]
    up = F.interpolate(g, size=(S * h, S * w), mode="bilinear", align_corners=False)   # [9, C, H, W]
    pad = F.pad(up, (1, 1, 1, 1))
    H, W = S * h, S * w
    out = torch.zeros(C, H, W, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            out += pad[ky * 3 + kx, :, ky:ky + H, kx:kx + W]
    out += torch.from_numpy(shift).view(C, 1, 1)
    return torch.relu(out).permute(1, 2, 0).numpy()                            # [H, W, C]


def main():
    rs = np.random.RandomState(0)
    for (h, w) in [(1, 1), (1, 2), (2, 3), (3, 2)]:
        C = 4
        G = rs.randn(h, w, 9, C)
        shift = rs.randn(C)
        ref = reference(G, shift)
        F4, F2 = build_tables(G, shift)
        H, W = S * h, S * w
        refp = np.zeros((H + 2, W + 2, C))
        refp[1:-1, 1:-1] = ref
        worst = 0.0
        for my in range(H // TILE):
            for mx in range(W // TILE):
                p = generate_tile(F4, F2, h, w, my, mx)
                want = refp[TILE * my:TILE * my + 18, TILE * mx:TILE * mx + 18]
                worst = max(worst, np.abs(p - want).max())
        print(f"h={h} w={w}: max |forms - interpolate+conv| = {worst:.3e}")
        assert worst < 1e-12


if __name__ == "__main__":
    main()
