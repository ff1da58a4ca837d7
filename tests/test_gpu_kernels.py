"""-m gpu: the HIP kernels, called through the C ABI, against the CPU oracle on seeded inputs
and against the golden vectors the reference produced.  Integer / byte outputs are compared
bit-exactly; float32 outputs within the tolerance stated at each assert."""
import numpy as np
import pytest

from tests.conftest import maxconf_flips_are_rounding_ties
import torch

pytestmark = pytest.mark.gpu

C = 19


@pytest.fixture(scope="module")
def ops(native):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
    return ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ------------------------------------------------------------------ confusion / argmax
@pytest.mark.parametrize("pdt", [torch.int64, torch.uint8])
def test_confusion_golden(ops, golden_metrics, pdt):
    g = golden_metrics
    for n in range(int(g["n_cases"])):
        counts = ops.new_counts(C, "cuda")
        oob = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.confusion_accumulate(dev(g[f"pred{n}"], pdt), dev(g[f"label{n}"]), C, counts, oob)
        assert oob.item() == 0
        assert np.array_equal(counts[0].cpu().numpy(), g[f"counts{n}"])


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4096 + 5, 1 << 20])
@pytest.mark.parametrize("ldt", ["uint8", "int64"])
def test_confusion_vs_oracle_ragged(ops, oracle, n, ldt):
    rs = np.random.RandomState(n)
    pred = rs.randint(0, C, n).astype(np.int64)
    # piecewise-constant labels exercise the in-register run merging
    lab = np.repeat(rs.randint(0, C, n // 7 + 1), 7)[:n]
    lab[rs.rand(n) < 0.05] = 255
    lab = lab.astype(ldt)
    counts = ops.new_counts(C, "cuda")
    oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.confusion_accumulate(dev(pred), dev(lab), C, counts, oob)
    assert np.array_equal(counts[0].cpu().numpy(), oracle.confusion(pred, lab, C))


def test_confusion_accumulates_and_flags_oob(ops):
    pred = torch.zeros(64, dtype=torch.int64, device="cuda")
    lab = torch.full((64,), 3, dtype=torch.int64, device="cuda")
    lab[5] = 19                                            # out of range for int64 labels
    counts = ops.new_counts(C, "cuda")
    oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.confusion_accumulate(pred, lab, C, counts, oob)
    ops.confusion_accumulate(pred, lab, C, counts, oob)
    assert oob.item() == 2 and counts[0, 3 * C].item() == 126


def test_argmax_golden_ties_nan(ops, golden_metrics):
    g = golden_metrics
    for dt in (torch.int64, torch.uint8):
        pred = ops.argmax(dev(g["am_logits"]), dt)
        assert np.array_equal(pred.cpu().numpy().astype(np.int64), g["am_pred"])


@pytest.mark.parametrize("shape", [(2, 19, 24, 40), (1, 19, 7, 9), (3, 5, 16, 16), (1, 32, 8, 12)])
def test_argmax_vs_oracle(ops, oracle, shape):
    rs = np.random.RandomState(sum(shape))
    x = rs.randn(*shape).astype(np.float32)
    x[rs.rand(*shape) < 0.01] = np.nan
    x = np.round(x * 2) / 2                                 # many exact ties
    assert np.array_equal(ops.argmax(dev(x)).cpu().numpy(), oracle.argmax(x))


def test_combine_golden(ops, golden_model):
    g = golden_model
    w = dev(g["ens_w"])
    t = torch.tensor([float(g["ens_t"])], device="cuda")
    for n in range(int(g["n_combine"])):
        mode, ts = [int(v) for v in g[f"combine_cfg{n}"]]
        out, pred = ops.combine_argmax_confusion(dev(g["seg1"]), dev(g["seg2"]), mode, w if mode == 0 else None,
                                                 t if ts else None, want_pred=True)
        ref = g[f"combine{n}"]
        if mode == 1:
            maxconf_flips_are_rounding_ties(out.cpu().numpy(), ref, g["seg1"], g["seg2"])   # stated tolerance: exact except float32 rounding ties of the two confidences
        else:
            assert np.array_equal(out.cpu().numpy(), ref)                   # bit-exact float32
            assert np.array_equal(pred.cpu().numpy(), ref.argmax(axis=1))


@pytest.mark.parametrize("hw", [(24, 40), (7, 9)])
@pytest.mark.parametrize("ldt", ["uint8", "int64"])
def test_fused_combine_argmax_confusion_slots(ops, oracle, hw, ldt):
    rs = np.random.RandomState(3)
    B = 5
    s1 = rs.randn(B, C, *hw).astype(np.float32)
    s2 = rs.randn(B, C, *hw).astype(np.float32)
    lab = rs.randint(0, C, (B,) + hw)
    lab[rs.rand(*lab.shape) < 0.05] = 255
    lab = lab.astype(ldt)
    cond = np.array([0, 1, 4, 1, -1], dtype=np.int32)
    counts = ops.new_counts(C, "cuda", 6)
    oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    w = torch.tensor([0.4, 0.6], device="cuda")
    t = torch.tensor([1.3], device="cuda")
    out, pred = ops.combine_argmax_confusion(dev(s1), dev(s2), 0, w, t, want_logits=True, want_pred=True,
                                             pred_dtype=torch.uint8, label=dev(lab), counts=counts, oob=oob, cond=dev(cond))
    ref_logits = oracle.combine(s1, s2, 0, float(np.float32(0.4)), float(np.float32(0.6)), float(np.float32(1.3)))
    assert np.array_equal(out.cpu().numpy(), ref_logits)
    ref_pred = oracle.argmax(ref_logits)
    assert np.array_equal(pred.cpu().numpy().astype(np.int64), ref_pred)
    got = counts.cpu().numpy()
    assert np.array_equal(got[0], oracle.confusion(ref_pred, lab, C))
    for slot, c in ((1, 0), (2, 1), (5, 4)):
        sel = cond == c
        assert np.array_equal(got[slot], oracle.confusion(ref_pred[sel], lab[sel], C))
    assert got[3].sum() == 0 and got[4].sum() == 0 and oob.item() == 0


def test_ece_bins(ops, oracle, golden_metrics):
    g = golden_metrics
    bins = ops.new_ece_bins(15, "cuda")
    edges = torch.linspace(0, 1, 16).cuda()
    ops.ece_accumulate(dev(g["ece_logits"]), dev(g["ece_label"]), bins, edges)
    got = ops.ece_bins_to_numpy(bins)[0]
    cnt, sconf, scorr = oracle.ece_bins(g["ece_logits"], g["ece_label"])
    # bin membership can differ from the oracle only where expf differs in the last ulp at an edge
    assert np.abs(got["count"] - cnt).sum() <= 2
    assert abs(oracle.ece_from_bins(got["count"], got["sum_conf"], got["sum_correct"]) - float(g["ece"])) < 1e-5


def test_ece_bins_do_not_depend_on_pointer_alignment(ops):
    """awseg_ece_accumulate picks the 4-pixels-per-lane kernel or the scalar one from hw % 4 and the pointer's
    16-byte alignment; both use the same exponential and summation order, so the same logits land in the same bins
    whether they are passed aligned or as an offset view (label / logits shifted by one element)."""
    g = torch.Generator(device="cuda").manual_seed(12)
    B, C_, hw = 3, 19, 64 * 100
    base = torch.randn(B * C_ * hw + 1, device="cuda", generator=g) * 3
    lab = torch.randint(0, C_, (B, hw), device="cuda", generator=g).to(torch.uint8)
    lab[:, ::53] = 255
    edges = torch.linspace(0, 1, 16).cuda()
    aligned = base[:B * C_ * hw].clone().view(B, C_, hw)
    shifted = base[1:]                                                  # 4-byte offset: the scalar kernel
    shifted.copy_(aligned.view(-1))
    a, b = ops.new_ece_bins(15, "cuda"), ops.new_ece_bins(15, "cuda")
    ops.ece_accumulate(aligned, lab, a, edges)
    N_ = ops.N
    ws = N_.workspace.get(aligned.device, N_.lib().awseg_metrics_workspace(B, C_, hw))
    assert shifted.data_ptr() % 16 != 0
    N_.call("awseg_ece_accumulate", shifted.data_ptr(), B, C_, hw, N_.ptr(lab), N_.label_dtype(lab), None, N_.ptr(edges), 15, N_.ptr(b), 1,
            N_.ptr(ws), N_.stream())
    assert torch.equal(a, b) and int(a[0, :, 0].sum()) == int((lab != 255).sum())


# ------------------------------------------------------------------ normalise / weather
def test_normalize(ops, oracle):
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 255, (3, 16, 24, 3), dtype=np.uint8)
    out = ops.normalize(dev(imgs)).cpu().numpy()
    for b in range(3):
        assert np.array_equal(out[b], oracle.normalize(imgs[b]))           # 3 float32 roundings, bit-exact
    sel = torch.tensor([2, 0], dtype=torch.int32, device="cuda")
    out2 = torch.zeros(3, 3, 16, 24, device="cuda")
    ops.normalize(dev(imgs), out=out2, sel=sel)
    assert np.array_equal(out2[2].cpu().numpy(), out[2]) and out2[1].abs().sum().item() == 0


@pytest.mark.parametrize("hw_", [(17, 23), (5, 7), (1, 1), (3, 1), (33, 2)])
def test_normalize_and_night_ragged_sizes(ops, oracle, hw_):
    """H*W % 4 != 0 (preprocessing.py:61-92 takes any H x W): per-image bases of a batch are then not dword aligned; the
    kernels take their scalar accesses and still give the oracle's bytes, for every frame of a batch and with `sel`."""
    h, w = hw_
    rs = np.random.RandomState(h * 100 + w)
    B = 3
    imgs = rs.randint(0, 255, (B, h, w, 3), dtype=np.uint8)
    out = ops.normalize(dev(imgs)).cpu().numpy()
    for b in range(B):
        assert np.array_equal(out[b], oracle.normalize(imgs[b]))
    sel = torch.tensor([2, 1], dtype=torch.int32, device="cuda")
    out2 = torch.zeros(B, 3, h, w, device="cuda")
    ops.normalize(dev(imgs), out=out2, sel=sel)
    assert np.array_equal(out2[1:].cpu().numpy(), out[1:]) and out2[0].abs().sum().item() == 0
    # night on frames 2 and 1 with host-drawn noise (parity mode), byte output + fused normalised output
    noise = rs.normal(0, 5.0 / 255.0, (2, h, w, 3))
    nj = ops.night_jobs([2, 1], [0.7, 0.9], [0.5, 0.8])
    nout = dev(imgs).clone()
    nnorm = torch.zeros(B, 3, h, w, device="cuda")
    ops.night(dev(imgs), nj, noise=dev(noise), out=nout, norm_out=nnorm)
    for j, b in enumerate((2, 1)):
        ref = oracle.night(imgs[b], noise[j], (0.7, 0.9)[j], (0.5, 0.8)[j])
        assert np.array_equal(nout[b].cpu().numpy(), ref)
        assert np.array_equal(nnorm[b].cpu().numpy(), oracle.normalize(ref))
    assert np.array_equal(nout[0].cpu().numpy(), imgs[0])


def test_fog_night_depth_golden(ops, golden_weather):
    """Bit-exact uint8 (and float64 depth) against what the reference itself produced."""
    g = golden_weather
    for k in range(int(g["n_cases"])):
        img = g[f"img{k}"]
        h, w = img.shape[:2]
        imgs = dev(img[None])
        jobs = ops.fog_jobs([0], [float(g[f"fog_intensity{k}"])])
        noise = dev(g[f"fog_noise{k}"][None])
        depth = ops.synthetic_depth(h, w, jobs, "cuda", noise)
        assert np.array_equal(depth[0].cpu().numpy(), g[f"depth{k}"])
        out = torch.empty_like(imgs)
        dout = torch.empty(1, h, w, dtype=torch.float64, device="cuda")
        ops.fog(imgs, jobs, noise=noise, out=out, depth_out=dout)
        assert np.array_equal(dout[0].cpu().numpy(), g[f"depth{k}"])
        mism = (out[0].cpu().numpy() != g[f"fog{k}"]).sum()
        assert mism == 0, f"fog case {k}: {mism} bytes differ"
        out2 = torch.empty_like(imgs)
        ops.fog(imgs, jobs, depth=depth, out=out2)
        assert torch.equal(out, out2)
        # every size the reference accepts, the ragged 17 x 23 fixture included (scalar accesses when H*W % 4 != 0)
        nj = ops.night_jobs([0], [float(g[f"night_brightness{k}"])], [float(g[f"night_intensity{k}"])])
        nout = torch.empty_like(imgs)
        ops.night(imgs, nj, noise=dev(g[f"night_noise{k}"][None]), out=nout)
        assert np.array_equal(nout[0].cpu().numpy(), g[f"night{k}"])


def test_weather_batched_and_fused_normalise(ops, oracle):
    rs = np.random.RandomState(5)
    B, h, w = 4, 40, 72
    imgs = rs.randint(0, 255, (B, h, w, 3), dtype=np.uint8)
    d = dev(imgs)
    # fog on images 1 and 3 only
    noise = rs.normal(0, 10, (2, h, w))
    jobs = ops.fog_jobs([1, 3], [0.4, 0.8])
    out = d.clone()
    norm = torch.zeros(B, 3, h, w, device="cuda")
    ops.fog(d, jobs, noise=dev(noise), out=out, norm_out=norm)
    for j, b in enumerate((1, 3)):
        ref = oracle.fog(imgs[b], oracle.synthetic_depth(noise[j]), (0.4, 0.8)[j])
        assert np.array_equal(out[b].cpu().numpy(), ref)
        assert np.array_equal(norm[b].cpu().numpy(), oracle.normalize(ref))
    assert torch.equal(out[0], d[0]) and torch.equal(out[2], d[2])
    # night on image 2
    nz = rs.normal(0, 5 / 255, (1, h, w, 3))
    nj = ops.night_jobs([2], [0.8], [0.6])
    ops.night(d, nj, noise=dev(nz), out=out, norm_out=norm)
    ref = oracle.night(imgs[2], nz[0], 0.8, 0.6)
    assert np.array_equal(out[2].cpu().numpy(), ref)
    assert np.array_equal(norm[2].cpu().numpy(), oracle.normalize(ref))


@pytest.mark.parametrize("hw", [(40, 72), (33, 70), (16, 64)])
def test_rain_snow_vs_oracle(ops, oracle, hw):
    """Rain / snow: bit-exact against the CPU restatement (which is itself unpinned: no cv2)."""
    h, w = hw
    rs = np.random.RandomState(h * w)
    imgs = rs.randint(0, 255, (2, h, w, 3), dtype=np.uint8)
    np.random.seed(3)
    i0, drops0 = oracle.draw_rain(h, w, 0.5)
    i1, drops1 = oracle.draw_rain(h, w, None)
    jobs, prims = ops.prim_jobs([0, 1], [i0, i1], [drops0, drops1])
    out = torch.zeros(2, h, w, 3, dtype=torch.uint8, device="cuda")
    # prepass=True: coverage bit map rasterised once per frame + pure-stencil blur; False: rasteriser inside every tile
    for prepass in (True, False):
        out.zero_()
        ops.rain(dev(imgs), jobs, prims, out=out, prepass=prepass)
        for b, (i, dr) in enumerate(((i0, drops0), (i1, drops1))):
            ref = oracle.rain(imgs[b], i, dr)
            assert (out[b].cpu().numpy() != ref).sum() == 0, prepass
    s0, fl0, k0 = oracle.draw_snow(h, w, 0.5)
    s1, fl1, _ = oracle.draw_snow(h, w, None)
    for ks in ((3, 7), (7, 3)):
        jobs, prims = ops.prim_jobs([0, 1], [s0, s1], [fl0, fl1], ks)
        for prepass in (True, False):
            out.zero_()
            ops.snow(dev(imgs), jobs, prims, out=out, prepass=prepass)
            assert np.array_equal(out[0].cpu().numpy(), oracle.snow(imgs[0], s0, fl0, ks[0])), prepass
            assert np.array_equal(out[1].cpu().numpy(), oracle.snow(imgs[1], s1, fl1, ks[1])), prepass


def test_rain_snow_coverage_prepass_equals_in_tile_rasteriser_at_full_size(ops):
    """1024x2048 frames, interior tiles with the aligned staging path, thick and thin drops, both flake sizes, fused normalise:
    the two forms of the kernel must agree byte for byte (and float for float)."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data import preprocessing as P
    B, h, w = 3, 1024, 2048
    np.random.seed(11)
    imgs = torch.randint(0, 255, (B, h, w, 3), dtype=torch.uint8, device="cuda")
    rd = [P.draw_rain(h, w, v) for v in (0.2, 0.5, 0.8)]
    rj, rp = ops.prim_jobs(list(range(B)), [d[0] for d in rd], [d[1] for d in rd])
    a, b = torch.zeros_like(imgs), torch.zeros_like(imgs)
    na, nb = torch.zeros(B, 3, h, w, device="cuda"), torch.zeros(B, 3, h, w, device="cuda")
    ops.rain(imgs, rj, rp, out=a, norm_out=na, prepass=True)
    ops.rain(imgs, rj, rp, out=b, norm_out=nb, prepass=False)
    assert torch.equal(a, b) and torch.equal(na, nb) and not torch.equal(a, imgs)
    sd = [P.draw_snow(h, w, v) for v in (0.2, 0.5, 0.8)]
    sj, sp = ops.prim_jobs(list(range(B)), [d[0] for d in sd], [d[1] for d in sd], [3, 7, 3])
    ops.snow(imgs, sj, sp, out=a, norm_out=na, prepass=True)
    ops.snow(imgs, sj, sp, out=b, norm_out=nb, prepass=False)
    assert torch.equal(a, b) and torch.equal(na, nb)


def test_philox_modes_are_deterministic_and_plausible(ops):
    h, w = 64, 128
    imgs = torch.randint(0, 255, (2, h, w, 3), dtype=torch.uint8, device="cuda")
    jobs = ops.fog_jobs([0, 1], [0.5, 0.5], seeds=[11, 12])
    a = torch.empty_like(imgs); b = torch.empty_like(imgs)
    d = torch.empty(2, h, w, dtype=torch.float64, device="cuda")
    ops.fog(imgs, jobs, out=a, depth_out=d)
    ops.fog(imgs, jobs, out=b)
    assert torch.equal(a, b)
    assert d.min().item() >= 1.0 and 30 < d.mean().item() < 70          # (y/H)*100 + smoothed N(0,10)
    nj = ops.night_jobs([0, 1], [0.8, 0.8], [0.6, 0.6], seeds=[1, 2])
    ops.night(imgs, nj, out=a); ops.night(imgs, nj, out=b)
    assert torch.equal(a, b) and (a.float().mean() < imgs.float().mean())
    f = ops.fog_density_field(["fog", "rain", "clean"], h, w, "cuda", 7)
    assert 0.5 <= f[0].min().item() and f[0].max().item() < 1.0 and abs(f[0].mean().item() - 0.75) < 0.01
    assert 0.2 <= f[1].min().item() and f[1].max().item() < 0.5 and f[2].max().item() < 0.1


def _philox4x32_7(seed, ctr, stream):
    """numpy restatement of awseg_philox::gen (csrc/awseg_common.h): Philox4x32 with 7 rounds; ctr is a uint64 array."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1 = ctr & mask, ctr >> np.uint64(32)
    c2, c3 = np.full_like(ctr, stream), np.full_like(ctr, 0x9E3779B9)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    for _ in range(7):
        p0, p1 = M0 * c0, M1 * c2
        n0, n2 = (p1 >> np.uint64(32)) ^ c1 ^ k0, (p0 >> np.uint64(32)) ^ c3 ^ k1
        c0, c1, c2, c3 = n0, p1 & mask, n2, p0 & mask
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return [c0, c1, c2, c3]


@pytest.mark.parametrize("geom", [(70, 256, 32), (200, 512, 32), (64, 128, 7), (37, 16, 32)])
def test_philox_fog_strip_kernel_depth_equals_scipy_on_the_same_samples(ops, geom, monkeypatch):
    """The strip form of throughput-mode fog (lane = 4 pixels, wave shifts for the horizontal taps, LDS column for the
    vertical ones, scipy 'reflect' by mirrored quads / reflected rows): its depth output against scipy.ndimage on the field
    rebuilt from the SAME Philox bytes — strips that end inside the image, waves that end inside a row, every border."""
    from scipy.ndimage import gaussian_filter
    h, w, rows = geom
    monkeypatch.setenv("AWSEG_FOG_STRIP_ROWS", str(rows))
    imgs = torch.randint(0, 255, (2, h, w, 3), dtype=torch.uint8, device="cuda")
    seeds = [0x1234567 + h, (7 << 40) + w]
    a = torch.empty_like(imgs)
    d = torch.empty(2, h, w, dtype=torch.float64, device="cuda")
    ops.fog(imgs, ops.fog_jobs([0, 1], [0.5, 0.3], seeds=seeds), out=a, depth_out=d)
    wq = w // 4
    gy, q = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(wq, dtype=np.uint64), indexing="ij")
    for b in range(2):
        words = _philox4x32_7(seeds[b], (gy >> np.uint64(1)) * np.uint64(wq) + q, 0x0F07)
        odd = (gy & np.uint64(1)).astype(bool)
        w0, w1 = np.where(odd, words[2], words[0]), np.where(odd, words[3], words[1])
        byte = lambda x, k: ((x >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.float64)
        n = np.stack([byte(w0, 0) - byte(w0, 1), byte(w0, 2) - byte(w0, 3), byte(w1, 0) - byte(w1, 1), byte(w1, 2) - byte(w1, 3)], axis=-1)
        field = (np.arange(h)[:, None] * (100.0 / h)) + 10.0 * 0.009584116 * n.reshape(h, w)
        ref = np.maximum(gaussian_filter(field, sigma=2, mode="reflect", truncate=4.0), 1.0)
        err = np.abs(d[b].cpu().numpy() - ref).max()
        assert err < 2e-3, (geom, b, err)                                # float32 filter on values up to 100


def test_philox_fog_noise_field_has_the_reference_moments(ops):
    """Throughput-mode fog synthesises its depth noise in the kernel from cheap non-Gaussian white noise (differences of
    random bytes); what the transform uses is that noise through scipy's 17-tap sigma-2 Gaussian on both axes.  The field
    must have the reference's second moments (white N(0, 10) through the same filter: variance 100 (sum w^2)^2) and
    Gaussian marginals (kurtosis 3: central limit theorem over ~50 effective terms)."""
    h, w = 512, 1024
    imgs = torch.randint(0, 255, (2, h, w, 3), dtype=torch.uint8, device="cuda")
    a = torch.empty_like(imgs)
    d = torch.empty(2, h, w, dtype=torch.float64, device="cuda")
    ops.fog(imgs, ops.fog_jobs([0, 1], [0.5, 0.5], seeds=[21, 22]), out=a, depth_out=d)
    taps = torch.from_numpy(ops.gaussian_taps()).double()
    want_std = 10.0 * float((taps ** 2).sum())
    ramp = (torch.arange(h, device="cuda", dtype=torch.float64) / h * 100.0)[None, :, None]
    z = (d - ramp)[:, h // 4: h - 16, 16: w - 16].reshape(-1)          # away from the max(., 1) clip and the reflected borders
    std, kurt = z.std().item(), ((z - z.mean()) ** 4).mean().item() / z.var().item() ** 2
    print(f"philox fog noise field: std {std:.4f} (theory {want_std:.4f}), mean {z.mean().item():+.4f}, kurtosis {kurt:.3f}")
    assert abs(std - want_std) < 0.03 * want_std and abs(z.mean().item()) < 0.05 and abs(kurt - 3.0) < 0.15


def test_fog_density_field_parity_mode_is_the_reference(ops, oracle, golden_trainer):
    """A16 parity mode: fed the reference's torch.rand draws, the kernel returns the reference's field bit for bit
    (fixture: the reference method itself, tests/golden/make_golden.py::gen_trainer); and the trainer's
    density_rng="torch" mode replays torch's CPU generator in the reference's order."""
    g = golden_trainer
    conds = [str(c) for c in g["conditions"]]
    b, h, w = g["uniform"].shape
    got = ops.fog_density_field(conds, h, w, "cuda", 0, uniform=dev(g["uniform"]))
    assert np.array_equal(got.cpu().numpy(), g["density"])
    assert np.array_equal(oracle.trainer_fog_density(conds, g["uniform"]), g["density"])
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.training.trainer import AdverseWeatherTrainer
    tr = AdverseWeatherTrainer.__new__(AdverseWeatherTrainer)          # the method reads only these attributes
    tr.device, tr.density_rng, tr._density_seed = torch.device("cuda"), "torch", 0
    torch.manual_seed(11)
    d = tr._estimate_fog_density({"weather_condition": conds, "image": torch.zeros(b, 3, h, w)})
    assert np.array_equal(d.cpu().numpy(), g["density"])
    assert tr._estimate_fog_density({"weather_condition": [], "image": torch.zeros(1, 3, h, w)}) is None
    # ragged pixel counts (hw % 4 != 0) and one sample
    u = torch.rand(1, 5, 7)
    f = ops.fog_density_field(["snow"], 5, 7, "cuda", 0, uniform=u.cuda())
    assert np.array_equal(f.cpu().numpy(), oracle.trainer_fog_density(["snow"], u.numpy()))


def test_depth_head_winograd_size_matches_reference_fixture(ops, golden_model):
    """DepthEstimationHead (PKG/models/model.py:16-78) at the shapes the Winograd kernels take (in 64, hidden 128 -> 64):
    output of the REFERENCE module on seeded parameters (re-drawn here by the generator's recipe) vs the fused eval
    path (Winograd 3x3 + BN + ReLU, Winograd 3x3 + 1x1 + sigmoid) — 1e-4 abs (north_star), measured ~1e-6."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("make_golden_recipe", Path(__file__).parent / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)                                        # module level only defines functions (no reference import)
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import DepthEstimationHead
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused
    g = golden_model
    head = DepthEstimationHead(in_channels=64, hidden_channels=128).eval()
    head.load_state_dict(mg.dhead64_params(head.state_dict()))
    head = head.cuda()
    x = dev(g["dhead64_in"])
    ref = g["dhead64_out"]
    h = head.depth_head
    assert fused._is_winograd(h[0]) and fused._is_winograd(h[4])
    with torch.no_grad():
        got = head.forward_fused(x.contiguous(memory_format=torch.channels_last))
        # and the single-launch tail: Conv3x3 -> BN -> ReLU -> Conv1x1 -> Sigmoid on the hidden map
        mid = fused.conv_bn_act(x.contiguous(memory_format=torch.channels_last), h[0], h[1], 1)
        u, shift = fused.winograd_conv_bn(h[4], h[5])
        tail = ops.conv3x3_winograd(fused.nhwc_view(mid), u, shift, w2=h[7].weight.view(-1), b2=h[7].bias).unsqueeze(1)
    e1 = np.abs(got.cpu().numpy() - ref).max()
    e2 = np.abs(tail.cpu().numpy() - ref).max()
    print(f"depth head 64->128->64->1 vs reference fixture: fused path {e1:.2e}, single-launch tail {e2:.2e}")
    assert e1 <= 1e-4 and e2 <= 1e-4


# ------------------------------------------------------------------ loss
def test_loss_golden(ops, golden_model):
    g = golden_model
    for n in range(int(g["n_loss"])):
        base, ldt, variant = [str(v) for v in g[f"loss_cfg{n}"]]
        lab = dev(g["loss_label"].astype(np.dtype(ldt)))
        dens = None
        if variant in ("density", "density_depth_target"):
            dens = dev(g["loss_density"])
        elif variant == "from_depth":
            dens = ops.fog_density_from_depth(dev(g["loss_dpred"][:, 0]))
        oob = torch.zeros(1, dtype=torch.int64, device="cuda")
        mean, _ = ops.fog_ce_forward(dev(g["loss_logits"]), lab, dens, base == "focal", 2.0, oob)
        assert abs(mean.item() - float(g[f"loss_seg{n}"])) < 1e-4          # north_star: 1e-4 abs on the loss
        if f"loss_grad{n}" in g.files and variant != "from_depth":
            grad = ops.fog_ce_backward(dev(g["loss_logits"]), lab, dens, base == "focal", 2.0, torch.ones(1, device="cuda"))
            assert np.abs(grad.cpu().numpy() - g[f"loss_grad{n}"]).max() < 1e-6


@pytest.mark.parametrize("shape", [(2, 19, 16, 24), (1, 19, 7, 9), (2, 5, 8, 8)])
def test_loss_vs_oracle(ops, oracle, shape):
    rs = np.random.RandomState(1)
    x = (rs.randn(*shape) * 3).astype(np.float32)
    lab = rs.randint(0, shape[1], (shape[0],) + shape[2:]).astype(np.uint8)
    dens = rs.rand(shape[0], *shape[2:]).astype(np.float32)
    for focal in (False, True):
        oob = torch.zeros(1, dtype=torch.int64, device="cuda")
        mean, pix = ops.fog_ce_forward(dev(x), dev(lab), dev(dens), focal, 2.0, oob, want_pixel=True)
        ref_mean, ref_pix = oracle.fog_ce(x, lab, dens, focal=focal, want_pixel=True)
        assert abs(mean.item() - ref_mean) < 1e-5
        assert np.abs(pix.cpu().numpy() - ref_pix).max() < 1e-4
        grad = ops.fog_ce_backward(dev(x), dev(lab), dev(dens), focal, 2.0, torch.full((1,), 0.7, device="cuda"))
        assert np.abs(grad.cpu().numpy() - oracle.fog_ce_grad(x, lab, dens, focal=focal, g=0.7)).max() < 1e-6


def test_loss_label_out_of_range_flag(ops):
    x = torch.randn(1, 19, 4, 4, device="cuda")
    lab = torch.full((1, 4, 4), 255, dtype=torch.uint8, device="cuda")     # CE has no ignore for 255 -> IndexError
    oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.fog_ce_forward(x, lab, None, False, 2.0, oob)
    assert oob.item() == 16


def test_depth_estimate_golden_and_ragged(ops, oracle, golden_depth):
    """DepthEstimationPreprocessor.estimate_depth: float64 bit-exact against the scipy-made
    fixtures and the C oracle (tile edges, images smaller than the 8-pixel halo, batch > 1)."""
    g = golden_depth
    for k in range(4):
        got = ops.depth_estimate(dev(g[f"img{k}"][None]), dtype=torch.float64)[0].cpu().numpy()
        assert np.array_equal(got, g[f"depth{k}"]), f"case {k}"
    rs = np.random.RandomState(5)
    for (b, h, w) in [(3, 33, 65), (2, 70, 130), (1, 3, 4), (2, 31, 64), (1, 97, 61)]:
        imgs = rs.randint(0, 256, (b, h, w, 3), dtype=np.uint8)
        d64 = ops.depth_estimate(dev(imgs), dtype=torch.float64).cpu().numpy()
        d32 = ops.depth_estimate(dev(imgs), dtype=torch.float32).cpu().numpy()
        for i in range(b):
            want = oracle.depth_estimate(imgs[i])
            assert np.array_equal(d64[i], want), (b, h, w, i)
            assert np.array_equal(d32[i], want.astype(np.float32))               # loader.py:290 .float()


def test_depth_estimate_abi_edges(ops, native):
    N = native
    imgs = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = torch.zeros(1, 8, 8, dtype=torch.float32, device="cuda")
    taps = N.host(np.full(17, 1.0 / 17))
    f = N.lib().awseg_depth_estimate
    assert f(N.ptr(imgs), 0, 8, 8, taps, N.ptr(ws), None, N.ptr(out), N.stream()) == 0    # empty batch
    assert f(N.ptr(imgs), 1, 8, 8, taps, N.ptr(ws), None, None, N.stream()) == -1          # no output: AWSEG_EINVAL
    assert f(N.ptr(imgs), 1, 0, 8, taps, N.ptr(ws), None, N.ptr(out), N.stream()) == -1
    assert f(None, 1, 8, 8, taps, N.ptr(ws), None, N.ptr(out), N.stream()) == -1
    assert N.lib().awseg_depth_estimate_workspace(5) == 20
    assert ops.depth_estimate(torch.zeros(0, 8, 8, 3, dtype=torch.uint8, device="cuda")).shape == (0, 8, 8)


def test_style_transfer_lut(ops, oracle):
    """WeatherAugmentationPipeline._apply_style_transfer as a device LUT, byte-exact against the
    per-pixel numpy restatement; mixed batch with a pass-through frame, ragged size, in place."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import WeatherAugmentationPipeline, style_lut
    rs = np.random.RandomState(11)
    pipe = WeatherAugmentationPipeline()
    for (h, w) in [(16, 32), (7, 12), (33, 20)]:
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        for t in ("fog", "rain", "snow", "night", "clean"):
            got = pipe._apply_style_transfer(img, t)
            assert np.array_equal(got, oracle.style_transfer(img, t)), (h, w, t)
    imgs = rs.randint(0, 256, (3, 16, 32, 3), dtype=np.uint8)
    luts = torch.from_numpy(np.stack([style_lut("rain"), style_lut("night")])).cuda()
    d = dev(imgs)
    ops.lut3_apply(d, luts, torch.tensor([1, -1, 0], dtype=torch.int32, device="cuda"), out=d)       # in place
    got = d.cpu().numpy()
    assert np.array_equal(got[0], oracle.style_transfer(imgs[0], "night")) and np.array_equal(got[1], imgs[1])
    assert np.array_equal(got[2], oracle.style_transfer(imgs[2], "rain"))


def test_fog_density_map_local_contrast(ops, oracle):
    """get_fog_density_map: the local-contrast kernel against the numpy restatement (float32, same
    tap order -> exact), then the percentile / depth weighting within 1e-6."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms
    rs = np.random.RandomState(21)
    tf = WeatherDegradationTransforms(device="cuda")
    for (h, w) in [(40, 72), (33, 31), (6, 9), (64, 64)]:
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        depth = rs.uniform(1.0, 100.0, (h, w))
        want, want_c = oracle.fog_density_map(img, depth)
        got_c = ops.local_contrast(dev(img[None]))[0].cpu().numpy()
        assert np.abs(got_c - want_c).max() <= 1e-7, (h, w)
        got = tf.get_fog_density_map(img, depth)
        assert got.shape == (h, w) and np.abs(got - want).max() < 1e-6
        assert got.min() >= 0.0 and got.max() <= 1.0


def test_density_from_depth(ops, golden_model):
    g = golden_model
    got = ops.fog_density_from_depth(dev(g["loss_dpred"][:, 0])).cpu().numpy()
    assert (np.abs(got - g["density_from_depth"]) > 1e-6).mean() < 1e-3


# ------------------------------------------------------------------ heads
def _fold_head(g):
    inv = 1.0 / np.sqrt(g["head_bn_var"].astype(np.float64) + float(g["head_bn_eps"]))
    scale = (g["head_bn_w"] * inv).astype(np.float32)
    shift = ((g["head_b1"] - g["head_bn_mean"]) * g["head_bn_w"] * inv + g["head_bn_b"]).astype(np.float32)
    return scale, shift


def test_segformer_head_fused_golden(ops, golden_model):
    """conv3x3(up(f)) restructured: within 1e-4 of torch's interpolate -> conv -> BN -> ReLU -> conv."""
    g = golden_model
    scale, shift = _fold_head(g)
    feat = torch.from_numpy(g["head_feat"]).cuda()                        # [1,Cin,h,w]
    w1 = torch.from_numpy(g["head_w1"]).cuda()                            # [Cmid,Cin,3,3]
    cmid, cin = w1.shape[0], w1.shape[1]
    g9 = torch.einsum("bchw,ockl->bhwklo", feat, w1).reshape(1, feat.shape[2], feat.shape[3], 9, cmid).contiguous()
    H, W = [int(v) for v in g["head_size"]]
    out = ops.segformer_head_fused(g9, dev(scale), dev(shift), dev(g["head_w2"]), dev(g["head_b2"]), H, W)
    assert np.abs(out.cpu().numpy() - g["head_out"]).max() < 1e-4


@pytest.mark.parametrize("shape", [(2, 20, 36, 32, 64, 1), (1, 16, 16, 16, 64, 1), (2, 37, 29, 64, 128, 1), (1, 24, 40, 48, 64, 2),
                                   (1, 33, 47, 32, 128, 2), (1, 50, 34, 16, 64, 3)])
def test_conv3x3_winograd_matches_direct(ops, shape):
    """Winograd F(2x2,3x3) MFMA convolution vs torch's direct conv (fp64 reference): bias, ReLU,
    residual, dilation, ragged edges.  Tolerance 1e-4 abs on O(1) outputs (transform rounding)."""
    B, H, W, Cin, Cout, d = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    wt = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (3.0 * Cin ** 0.5)
    scale = torch.rand(Cout, device="cuda", generator=g) + 0.5
    shift = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    u = ops.winograd_weights(wt, scale)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), (wt * scale.view(-1, 1, 1, 1)).double(), None, 1, d, d)
    ref = ref.permute(0, 2, 3, 1) + shift.double()
    got = ops.conv3x3_winograd(x, u, shift, act=0, dilation=d)
    assert (got.double() - ref).abs().max().item() < 1e-4
    got = ops.conv3x3_winograd(x, u, shift, act=1, dilation=d, residual=res)
    assert (got.double() - (ref + res.double()).clamp_min(0)).abs().max().item() < 1e-4
    if Cout == 64:
        w2 = torch.randn(64, device="cuda", generator=g) * 0.2
        b2 = torch.randn(1, device="cuda", generator=g)
        got = ops.conv3x3_winograd(x, u, shift, dilation=d, w2=w2, b2=b2)
        want = torch.sigmoid((ref.clamp_min(0) * w2.double()).sum(-1) + b2.double())
        assert got.shape == (B, H, W) and (got.double() - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("shape", [(2, 20, 36, 32, 64, 1), (1, 16, 16, 16, 64, 1), (2, 37, 29, 64, 128, 1), (1, 24, 40, 48, 64, 2),
                                   (1, 33, 47, 32, 128, 2), (1, 50, 34, 16, 64, 3), (1, 40, 72, 256, 256, 1), (2, 16, 32, 2048, 256, 1)])
@pytest.mark.parametrize("xscale,wscale", [(1.0, 1.0), (3e4, 1e-3), (1e-6, 50.0)])
def test_conv3x3_winograd_split_float32_grade(ops, shape, xscale, wscale):
    """Winograd F(2x2,3x3) on split-operand f16 MFMA against a float64 direct convolution, next to the float32-MFMA
    Winograd kernel on the same inputs: bias, ReLU, residual, dilation, ragged edges, the fused 1x1 + sigmoid head; and
    activations / filters far outside the f16 range (the in-kernel range guard redoes those tiles scaled).  1e-4 abs on
    O(1) outputs (north_star), and never worse than a small multiple of the float32 kernel's error."""
    B, H, W, Cin, Cout, d = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + 17)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g) * xscale
    x[:, ::3, 1::4] *= 1e-3                                        # small activations next to O(1) ones
    wt = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (3.0 * Cin ** 0.5) * wscale
    scale = torch.rand(Cout, device="cuda", generator=g) + 0.5
    shift = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    u32 = ops.winograd_weights(wt, scale)
    us = ops.winograd_split_weights(wt, scale)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), (wt * scale.view(-1, 1, 1, 1)).double(), None, 1, d, d)
    ref = ref.permute(0, 2, 3, 1) + shift.double()
    mag = max(1.0, ref.abs().max().item())
    got = ops.conv3x3_winograd_split(x, us, Cout, shift, act=0, dilation=d)
    f32 = ops.conv3x3_winograd(x, u32, shift, act=0, dilation=d)
    e_s, e_f = (got.double() - ref).abs().max().item() / mag, (f32.double() - ref).abs().max().item() / mag
    print(f"winograd {shape} x*{xscale:g} w*{wscale:g}: rel err float32 kernel {e_f:.3e}, split kernel {e_s:.3e}")
    assert e_s < 1e-4 and e_s < 4 * e_f + 2e-6
    got = ops.conv3x3_winograd_split(x, us, Cout, shift, act=1, dilation=d, residual=res)
    assert (got.double() - (ref + res.double()).clamp_min(0)).abs().max().item() / mag < 1e-4
    if Cout == 64:
        w2 = torch.randn(64, device="cuda", generator=g) * 0.2 / mag
        b2 = torch.randn(1, device="cuda", generator=g)
        got = ops.conv3x3_winograd_split(x, us, Cout, shift, dilation=d, w2=w2, b2=b2)
        want = torch.sigmoid((ref.clamp_min(0) * w2.double()).sum(-1) + b2.double())
        assert got.shape == (B, H, W) and (got.double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("dil", [1, 2])
def test_conv3x3_winograd_split_range_guard_sees_every_input_element(ops, dil):
    """The symmetric eight-wave kernel takes max|x| once per input row of the 4x4 tiles (row 1 by the threads that build V row 1,
    row 3 by V row 3's, rows 0 and 2 by V row 0's): ONE out-of-range activation anywhere — block corners, tile seams, the halo
    rows / columns a block shares with its neighbours, the image border, any channel chunk — must send its block through the
    rescaled pass.  Unnoticed, a 3e5 input saturates the f16 high part and the output is off by orders of magnitude."""
    B, H, W, Cin, Cout = 1, 40, 52, 48, 64
    g = torch.Generator(device="cuda").manual_seed(101 + dil)
    wt = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (3.0 * Cin ** 0.5)
    shift = torch.randn(Cout, device="cuda", generator=g)
    us = ops.winograd_split_weights(wt, None)
    base = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    spots = [(0, 0, 0), (H - 1, W - 1, Cin - 1), (15, 15, 5), (15, 16, 17), (16, 15, 31), (16, 16, 32), (17, 17, 47), (14, 31, 16), (31, 32, 15),
             (33, 1, 20), (2, 49, 40), (39, 0, 3), (0, 51, 44), (20, 20, 0)]
    spots += [(int(y), int(x), int(c)) for y, x, c in zip(torch.randint(0, H, (10,), generator=torch.Generator().manual_seed(5)).tolist(),
                                                          torch.randint(0, W, (10,), generator=torch.Generator().manual_seed(6)).tolist(),
                                                          torch.randint(0, Cin, (10,), generator=torch.Generator().manual_seed(7)).tolist())]
    for (y, x, c) in spots:
        xin = base.clone()
        xin[0, y, x, c] = 3.0e5
        ref = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2).double(), wt.double(), None, 1, dil, dil).permute(0, 2, 3, 1) + shift.double()
        got = ops.conv3x3_winograd_split(xin, us, Cout, shift, act=0, dilation=dil)
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, f"spot {(y, x, c)} dilation {dil}: relative error {err:.3e}"


def test_conv3x3_winograd_split_abi_checks(ops, native):
    N = native
    sh = torch.zeros(64, device="cuda")
    us = ops.winograd_split_weights(torch.zeros(64, 16, 3, 3, device="cuda"))
    x = torch.zeros(1, 8, 8, 16, device="cuda")
    assert us.numel() == N.lib().awseg_winograd_split_weight_halfs(16, 64) + 8
    assert N.lib().awseg_winograd_split_weight_halfs(8, 64) < 0 and N.lib().awseg_winograd_split_weight_halfs(16, 32) < 0
    with pytest.raises(N.AwsegError):
        ops.conv3x3_winograd_split(torch.zeros(1, 8, 8, 32, device="cuda"), us, 64, sh)                                  # weights of another shape
    with pytest.raises(N.AwsegError):
        ops.conv3x3_winograd_split(x, us, 64, sh, w2=torch.zeros(64, device="cuda"))                                     # w2 without b2
    assert ops.conv3x3_winograd_split(x[:0], us, 64, sh).shape == (0, 8, 8, 64)
    out = ops.conv3x3_winograd_split(x, us, 64, sh)                                                                       # all-zero input and filters
    assert (out == 0).all()


def test_conv3x3_winograd_abi_checks(ops, native):
    N = native
    x = torch.zeros(1, 8, 8, 16, device="cuda"); u = torch.zeros(2, 16, 2, 64, 4, device="cuda"); sh = torch.zeros(64, device="cuda")
    with pytest.raises(N.AwsegError):
        ops.conv3x3_winograd(torch.zeros(1, 8, 8, 8, device="cuda"), torch.zeros(1, 16, 2, 64, 4, device="cuda"), sh)      # Cin % 16
    with pytest.raises(N.AwsegError):
        ops.conv3x3_winograd(x, torch.zeros(2, 16, 2, 32, 4, device="cuda"), sh[:32])                                        # Cout % 64
    with pytest.raises(N.AwsegError):
        ops.conv3x3_winograd(x, u, sh, w2=torch.zeros(64, device="cuda"))                                               # w2 without b2
    assert ops.conv3x3_winograd(x[:0], u, sh).shape == (0, 8, 8, 64)


@pytest.mark.parametrize("shape", [(4096, 64, 256), (1000, 48, 304), (8 * 64 * 128, 256, 1024), (37, 19, 64)])
@pytest.mark.parametrize("split", [False, None])
def test_gemm_bias_act_epilogues(ops, shape, split):
    """1x1 convolution as one launch — split=False: the hipBLASLt float32 call; None: whatever the dispatcher picks
    (the split-operand kernel for the wide shapes) — bias, residual (also in place) and ReLU in the epilogue, against a
    float64 reference; 1e-4 abs on O(1) outputs."""
    M, Nn, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(Nn, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(Nn, device="cuda", generator=g)
    r = torch.randn(M, Nn, device="cuda", generator=g)
    ref = x.double() @ w.double().t() + b.double()
    assert (ops.gemm_bias_act(x, w, b, 0, split=split).double() - ref).abs().max().item() < 1e-4
    assert (ops.gemm_bias_act(x, w, b, 1, split=split).double() - ref.clamp_min(0)).abs().max().item() < 1e-4
    assert (ops.gemm_bias_act(x, w, b, 1, residual=r, split=split).double() - (ref + r.double()).clamp_min(0)).abs().max().item() < 1e-4
    r2 = r.clone()
    out = ops.gemm_bias_act(x, w, b, 0, residual=r2, out=r2, split=split)     # accumulate over the residual's buffer
    assert out.data_ptr() == r2.data_ptr() and (r2.double() - (ref + r.double())).abs().max().item() < 1e-4
    assert ops.gemm_bias_act(x[:0], w, b, 1, split=split).shape == (0, Nn)


def test_aspp_depthwise3(ops):
    torch.manual_seed(0)
    B, h, w, Cc = 2, 20, 28, 16
    x = torch.randn(B, Cc, h, w, device="cuda")
    wdw = torch.randn(3, Cc, 1, 3, 3, device="cuda")
    rates = (3, 6, 9)
    out = ops.aspp_depthwise3(x.permute(0, 2, 3, 1).contiguous(), wdw.reshape(3, Cc, 9).permute(0, 2, 1).contiguous(), rates)
    for r in range(3):
        ref = torch.nn.functional.conv2d(x, wdw[r], padding=rates[r], dilation=rates[r], groups=Cc)
        assert (out[r].permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg", [(2, 256, 4, 6, 128, 192, 19), (1, 256, 3, 5, 96, 160, 19), (1, 128, 5, 3, 150, 100, 7),
                                 (1, 64, 2, 2, 64, 64, 32), (1, 256, 32, 64, 1024, 2048, 19)])
def test_segformer_head_mfma_vs_v1_and_torch(ops, cfg, monkeypatch):
    """MFMA head (v2) vs the as-written torch op sequence and vs the VALU kernel (v1), incl. image
    borders (zero padding of the 3x3), non-multiple-of-32 sizes and the full 1024x2048 geometry."""
    B, cmid, h, w, H, W, cout = cfg
    torch.manual_seed(sum(cfg))
    cin = 32
    feat = torch.randn(B, cin, h, w, device="cuda")
    conv1 = torch.nn.Conv2d(cin, cmid, 3, padding=1).cuda()
    bn = torch.nn.BatchNorm2d(cmid).cuda().eval()
    conv2 = torch.nn.Conv2d(cmid, cout, 1).cuda()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight * inv).contiguous()
        shift = ((conv1.bias - bn.running_mean) * scale + bn.bias).contiguous()
        g9 = torch.einsum("bchw,ockl->bhwklo", feat, conv1.weight).reshape(B, h, w, 9, cmid).contiguous()
        w2 = conv2.weight.view(cout, cmid).contiguous()
        got = ops.segformer_head_fused(g9, scale, shift, w2, conv2.bias, H, W)
        if H * W <= 256 * 256:
            up = torch.nn.functional.interpolate(feat, size=(H, W), mode="bilinear", align_corners=False)
            mid_ref = torch.relu(bn(conv1(up)))
            ref = conv2(mid_ref)
            assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
            if cmid in (32, 64, 128, 256) and min(H / h, W / w) >= 17:
                mid = ops.upconv3x3_bn_relu(g9, scale, shift, H, W)
                assert (mid - mid_ref).abs().max().item() < 1e-4 * max(1.0, mid_ref.abs().max().item())
        monkeypatch.setenv("AWSEG_HEAD_V1", "1")
        v1 = ops.segformer_head_fused(g9, scale, shift, w2, conv2.bias, H, W)
        monkeypatch.delenv("AWSEG_HEAD_V1")
        assert (got - v1).abs().max().item() < 1e-4 * max(1.0, v1.abs().max().item())
        # BatchNorm scale folded into g9 (scale=None): the 16-byte-gather kernel (v3) when Cmid is 128 / 256
        g9f = torch.einsum("bchw,ockl->bhwklo", feat, conv1.weight * scale.view(-1, 1, 1, 1)).reshape(B, h, w, 9, cmid).contiguous()
        if cmid % 32 == 0 and min(H / h, W / w) >= 17:
            v3 = ops.segformer_head_fused(g9f, None, shift, w2, conv2.bias, H, W, split=False)
            assert (v3 - v1).abs().max().item() < 1e-4 * max(1.0, v1.abs().max().item())
            # the split-operand f16-MFMA kernel (v4) where it applies (Cmid 128 / 256), else the same float32 kernel
            v4 = ops.segformer_head_fused(g9f, None, shift, w2, conv2.bias, H, W, split=True)
            assert (v4 - v1).abs().max().item() < 1e-4 * max(1.0, v1.abs().max().item())


@pytest.mark.parametrize("cmid", [128, 256])
def test_segformer_head_split_is_float32_grade_and_guards_its_operand_range(ops, cmid):
    """v4 (split f16 operands) against the as-written op sequence in FLOAT64, next to the float32-MFMA kernel on the same
    inputs; then with a patch of the feature map scaled to 1e5 — beyond f16 — where the rows it reaches are recomputed on the
    float32 instruction inside the kernel, and with an infinity (propagates as in torch)."""
    B, h, w, H, W, cout, cin = 2, 5, 7, 160, 224, 19, 32
    torch.manual_seed(cmid)
    feat = torch.randn(B, cin, h, w, device="cuda")
    conv1 = torch.nn.Conv2d(cin, cmid, 3, padding=1).cuda()
    conv2 = torch.nn.Conv2d(cmid, cout, 1).cuda()
    shift = (torch.randn(cmid, device="cuda") * 0.3).contiguous()
    w2 = conv2.weight.view(cout, cmid).contiguous()

    def both(f):
        with torch.no_grad():
            g9 = torch.einsum("bchw,ockl->bhwklo", f, conv1.weight).reshape(B, h, w, 9, cmid).contiguous()
            up = torch.nn.functional.interpolate(f.double(), size=(H, W), mode="bilinear", align_corners=False)
            mid = torch.relu(torch.nn.functional.conv2d(up, conv1.weight.double(), None, padding=1) + shift.double().view(1, -1, 1, 1))
            ref = torch.nn.functional.conv2d(mid, conv2.weight.double(), conv2.bias.double())
            f32 = ops.segformer_head_fused(g9, None, shift, w2, conv2.bias, H, W, split=False)
            spl = ops.segformer_head_fused(g9, None, shift, w2, conv2.bias, H, W, split=True)
        return ref, f32, spl

    ref, f32, spl = both(feat)
    mag = ref.abs().max().item()
    e32, esp = (f32.double() - ref).abs().max().item() / mag, (spl.double() - ref).abs().max().item() / mag
    print(f"head Cmid {cmid}: float32 MFMA {e32:.2e}, split f16 {esp:.2e} of max |logit| {mag:.2f}")
    assert esp < 1e-5 and esp < 8 * max(e32, 2e-7)
    big = feat.clone(); big[0, :, 1:3, 2:4] *= 1e5
    ref, f32, spl = both(big)
    mag = ref.abs().max().item()
    esp = (spl.double() - ref).abs().max().item() / mag
    assert torch.isfinite(spl).all() and esp < 1e-5, esp
    # away from the patch (rows of image 1 never see it) the result is the small-magnitude one
    assert (spl[1].double() - ref[1]).abs().max().item() < 1e-5 * ref[1].abs().max().item()
    inf = feat.clone(); inf[1, 0, 2, 3] = float("inf")
    ref, f32, spl = both(inf)
    assert torch.equal(torch.isfinite(spl), torch.isfinite(f32))


def test_dwconv3x3_nhwc_and_bias_act(ops):
    torch.manual_seed(0)
    B, H, W, Cc = 2, 17, 23, 24
    x = torch.randn(B, Cc, H, W, device="cuda")
    conv = torch.nn.Conv2d(Cc, Cc, 3, padding=1, groups=Cc).cuda()
    xl = x.permute(0, 2, 3, 1).contiguous()
    w9 = conv.weight.view(Cc, 9).t().contiguous()
    with torch.no_grad():
        ref = conv(x)
        got = ops.dwconv3x3_nhwc(xl, w9, conv.bias, 0).permute(0, 3, 1, 2)
        assert (got - ref).abs().max().item() < 1e-5
        got = ops.dwconv3x3_nhwc(xl, w9, conv.bias, 2).permute(0, 3, 1, 2)
        assert (got - torch.nn.functional.gelu(ref)).abs().max().item() < 1e-5
        ref_d = torch.nn.functional.conv2d(x, conv.weight, None, padding=3, dilation=3, groups=Cc)
        got = ops.dwconv3x3_nhwc(xl, w9, None, 1, dilation=3).permute(0, 3, 1, 2)
        assert (got - torch.relu(ref_d)).abs().max().item() < 1e-5
        y = torch.randn(B, H, W, Cc, device="cuda"); r = torch.randn_like(y); b = torch.randn(Cc, device="cuda")
        exp = torch.relu(y + b + r)
        assert torch.equal(ops.bias_act_nhwc_(y.clone(), b, r, 1), exp)
        assert torch.equal(ops.bias_act_nhwc_(y.clone(), b, None, 0), y + b)


def test_upconv_channels_last_matches_nchw(ops):
    torch.manual_seed(1)
    B, cmid, h, w, H, W = 1, 128, 3, 4, 96, 128
    g9 = torch.randn(B, h, w, 9, cmid, device="cuda")
    sc, sf = torch.rand(cmid, device="cuda") + 0.5, torch.randn(cmid, device="cuda") * 0.1
    a = ops.upconv3x3_bn_relu(g9, sc, sf, H, W, channels_last=False)
    b = ops.upconv3x3_bn_relu(g9, sc, sf, H, W, channels_last=True)
    assert b.shape == a.shape and b.is_contiguous(memory_format=torch.channels_last) and torch.equal(a, b.contiguous())


def test_fog_throughput_mode_statistics_and_consistency(ops, oracle):
    """Philox/float32 fog (noise=None): noise statistics match N(0,10) through the sigma=2 filter, and
    the bytes equal the float64 oracle applied to the kernel's own depth up to 1 LSB."""
    h, w = 256, 512
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 255, (2, h, w, 3), dtype=np.uint8)
    jobs = ops.fog_jobs([0, 1], [0.5, 0.8], seeds=[5, 6])
    out = torch.empty(2, h, w, 3, dtype=torch.uint8, device="cuda")
    norm = torch.empty(2, 3, h, w, device="cuda")
    d = torch.empty(2, h, w, dtype=torch.float64, device="cuda")
    ops.fog(dev(imgs), jobs, out=out, norm_out=norm, depth_out=d)
    depth = d.cpu().numpy()
    base = (np.arange(h)[:, None] / h) * 100.0
    resid = (depth[0] - base)[40:-8, 8:-8]                     # rows where the max(.,1) clamp is inactive
    taps = ops.gaussian_taps()
    expect_std = 10.0 * float((taps ** 2).sum())               # std of separably filtered white noise
    assert abs(resid.mean()) < 0.05 and abs(resid.std() - expect_std) / expect_std < 0.05
    assert abs(np.corrcoef(depth[0].ravel(), depth[1].ravel())[0, 1]) > 0.9  # same ramp, different noise
    assert not np.array_equal(depth[0], depth[1])
    for b, inten in ((0, 0.5), (1, 0.8)):
        ref = oracle.fog(imgs[b], depth[b], inten)
        diff = np.abs(out[b].cpu().numpy().astype(np.int16) - ref.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
        assert np.array_equal(norm[b].cpu().numpy(), oracle.normalize(out[b].cpu().numpy()))


@pytest.mark.parametrize("shape", [(1000, 32), (257, 64), (130, 160), (64, 256), (33, 320), (20, 512), (7, 1024)])
def test_layernorm_rows(ops, shape):
    torch.manual_seed(shape[1])
    x = torch.randn(*shape, device="cuda") * 3 + 1
    g, b = torch.randn(shape[1], device="cuda"), torch.randn(shape[1], device="cuda")
    ref = torch.nn.functional.layer_norm(x, (shape[1],), g, b, 1e-5)
    assert (ops.layernorm_rows(x, g, b, 1e-5) - ref).abs().max().item() < 2e-5


def test_abi_argument_checks_and_edge_shapes(ops, native):
    """Error conventions of the C ABI (negative AWSEG_E* codes -> AwsegError, nothing launched) and the
    smallest / ragged shapes."""
    N = native
    lib = N.lib()
    # empty pixel list: a no-op, not an error
    counts = ops.new_counts(19, "cuda"); oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.confusion_accumulate(torch.zeros(0, dtype=torch.int64, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 19, counts, oob)
    assert counts.sum().item() == 0
    # one pixel, one class
    c1 = ops.new_counts(1, "cuda")
    ops.confusion_accumulate(torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.uint8, device="cuda").contiguous(), 1, c1, oob)
    assert c1.item() == 1
    # more classes than the register budget
    with pytest.raises(N.AwsegError, match="invalid argument"):
        ops.argmax(torch.zeros(1, 33, 4, 4, device="cuda"))
    # odd H*W: every size the reference's transforms accept is taken (scalar accesses; test_normalize_and_night_ragged_sizes)
    z = ops.normalize(torch.zeros(1, 3, 5, 3, dtype=torch.uint8, device="cuda"))
    assert z.shape == (1, 3, 3, 5) and torch.isfinite(z).all()
    # ... as do the scalar fall-back paths of the logit kernels
    x = torch.randn(1, 19, 3, 5, device="cuda")
    assert torch.equal(ops.argmax(x), x.argmax(dim=1))
    # non-contiguous / wrong dtype labels
    with pytest.raises(N.AwsegError, match="uint8 or int64"):
        ops.confusion_accumulate(torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda"), 19, counts, oob)
    # rain must not run in place (the blur reads neighbours)
    imgs = torch.zeros(1, 32, 64, 3, dtype=torch.uint8, device="cuda")
    jobs, prims = ops.prim_jobs([0], [0.5], [np.zeros((0, 5), np.int32)])
    with pytest.raises(N.AwsegError, match="invalid argument"):
        ops.rain(imgs, jobs, prims, out=imgs)
    # zero primitives = haze + blur only, identical to the oracle
    assert lib.awseg_abi_version() == 1 and lib.awseg_device_count() >= 1


def test_rain_zero_primitives_and_many_jobs(ops, oracle):
    rs = np.random.RandomState(1)
    B, h, w = 20, 32, 64                                   # more jobs than one kernel-argument pack (16)
    imgs = rs.randint(0, 255, (B, h, w, 3), dtype=np.uint8)
    jobs, prims = ops.prim_jobs(list(range(B)), [0.3 + 0.02 * i for i in range(B)], [np.zeros((0, 5), np.int32)] * B)
    out = torch.zeros(B, h, w, 3, dtype=torch.uint8, device="cuda")
    ops.rain(dev(imgs), jobs, prims, out=out)
    for b in (0, 15, 16, 19):
        assert np.array_equal(out[b].cpu().numpy(), oracle.rain(imgs[b], 0.3 + 0.02 * b, np.zeros((0, 5), np.int32)))
    nz = rs.normal(0, 5 / 255, (B, h, w, 3))
    nj = ops.night_jobs(list(range(B)), [0.8] * B, [0.5] * B)
    ops.night(dev(imgs), nj, noise=dev(nz), out=out)
    for b in (0, 16, 19):
        assert np.array_equal(out[b].cpu().numpy(), oracle.night(imgs[b], nz[b], 0.8, 0.5))
    fz = rs.normal(0, 10, (B, h, w))
    fj = ops.fog_jobs(list(range(B)), [0.5] * B)
    ops.fog(dev(imgs), fj, noise=dev(fz), out=out)
    for b in (0, 16, 19):
        assert np.array_equal(out[b].cpu().numpy(), oracle.fog(imgs[b], oracle.synthetic_depth(fz[b]), 0.5))


def test_gemm_tune_keeps_results(ops, native):
    """awseg_gemm_tune (opt-in, synchronising) times hipBLASLt's candidates; whichever it keeps, results stay right."""
    N = native
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(8192, 128, device="cuda", generator=g); w = torch.randn(64, 128, device="cuda", generator=g) / 11.0
    b = torch.randn(64, device="cuda", generator=g); scratch = torch.empty(8192, 64, device="cuda")
    ws = N.workspace.get(x.device, ops.GEMM_WORKSPACE_BYTES, tag="gemm")
    rc = N.lib().awseg_gemm_tune(N.ptr(x), N.ptr(w), N.ptr(b), 0, 1, N.ptr(scratch), 8192, 64, 128, N.ptr(ws), ops.GEMM_WORKSPACE_BYTES, N.stream())
    assert rc >= 1
    ref = (x.double() @ w.double().t() + b.double()).clamp_min(0)
    assert (ops.gemm_bias_act(x, w, b, 1).double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("shape", [(2, 17, 23, 8), (1, 64, 128, 64), (3, 9, 10, 4), (1, 1, 7, 12)])
def test_maxpool3x3s2_nhwc_equals_torch(ops, shape):
    B, H, W, C = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool3x3s2_nhwc(x), ref)
    # NaN propagates as in nn.MaxPool2d (fmaxf dropped it): NaNs in a window's first, middle and last taps
    x[0, 0, 0, 0] = float("nan"); x[-1, H // 2, W // 2, C - 1] = float("nan"); x[0, H - 1, W - 1, 1] = float("nan")
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = ops.maxpool3x3s2_nhwc(x)
    assert ref.isnan().sum().item() >= 3 and torch.equal(got.isnan(), ref.isnan()) and torch.equal(got.nan_to_num(0.0), ref.nan_to_num(0.0))
    # the stem's epilogue in the pooling kernel's store == the pooling followed by awseg_bias_act_nhwc, bit for bit
    x2 = torch.randn(B, H, W, C, device="cuda", generator=g)
    shift = torch.randn(C, device="cuda", generator=g)
    two = ops.maxpool3x3s2_nhwc(x2)
    ops.bias_act_nhwc_(two, shift, None, 1)
    assert torch.equal(ops.maxpool3x3s2_nhwc(x2, shift=shift), two)


@pytest.mark.parametrize("cfg", [(2, 19, 16, 32, 64, 128, True), (1, 3, 7, 5, 28, 20, True), (2, 1, 4, 8, 64, 128, False),
                                 (1, 2, 5, 9, 33, 70, False), (1, 19, 256, 512, 1024, 2048, True), (2, 4, 8, 8, 16, 16, False),
                                 (1, 2, 9, 12, 18, 24, True), (1, 1, 6, 6, 5, 4, True)])
def test_upsample_bilinear_matches_torch(ops, cfg):
    """awseg_upsample_bilinear against F.interpolate (both corner conventions, non-multiple-of-4 widths, the full DeepLab size)."""
    B, C, h, w, H, W, align = cfg
    g = torch.Generator(device="cuda").manual_seed(h * w)
    x = torch.randn(B, C, h, w, device="cuda", generator=g)
    ref = torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=align)
    got = ops.upsample_bilinear(x, (H, W), align)
    err = (got - ref).abs().max().item()
    assert err < 1e-6 * max(1.0, ref.abs().max().item()), err      # same expression; torch's build may contract a*b+c into FMAs


@pytest.mark.parametrize("shape", [(2, 5, 7, 8, 4), (1, 16, 32, 256, 48), (1, 3, 3, 4, 4), (2, 9, 13, 256, 8), (1, 4, 21, 512, 48)])
def test_dwconv3x3_upcat_matches_torch(ops, shape):
    """Decoder fusion: depthwise3x3(cat(UpsamplingBilinear2d(x4)(a), hi)) against the torch ops it replaces (1e-5)."""
    B, h, w, Ca, Ch = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    a = torch.randn(B, Ca, h, w, device="cuda", generator=g)
    hi = torch.randn(B, Ch, 4 * h, 4 * w, device="cuda", generator=g)
    wdw = torch.randn(Ca + Ch, 1, 3, 3, device="cuda", generator=g)
    cat = torch.cat([torch.nn.UpsamplingBilinear2d(scale_factor=4)(a), hi], dim=1)
    ref = torch.nn.functional.conv2d(cat, wdw, padding=1, groups=Ca + Ch)
    got = ops.dwconv3x3_upcat(a.permute(0, 2, 3, 1).contiguous(), hi.permute(0, 2, 3, 1).contiguous(), wdw.view(Ca + Ch, 9).t().contiguous())
    assert (got.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("geom", [(2, 40, 52, 128), (1, 64, 20, 64), (3, 10, 7, 64), (1, 37, 75, 192)])
def test_aspp_depthwise3_xcd_sliced_path(ops, geom):
    """The XCD-aware (image, 64-channel slice) work order of the ASPP depthwise kernel (C % 64 == 0), real rates: row classes
    with 1..6 rows, maps lower than a rate (every row its own class), widths below a rate (no horizontal neighbours)."""
    torch.manual_seed(1)
    B, h, w, Cc = geom
    x = torch.randn(B, Cc, h, w, device="cuda")
    wdw = torch.randn(3, Cc, 1, 3, 3, device="cuda")
    rates = (12, 24, 36)
    out = ops.aspp_depthwise3(x.permute(0, 2, 3, 1).contiguous(), wdw.reshape(3, Cc, 9).permute(0, 2, 1).contiguous(), rates)
    for r in range(3):
        ref = torch.nn.functional.conv2d(x, wdw[r], padding=rates[r], dilation=rates[r], groups=Cc)
        assert (out[r].permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("shape", [(2, 1, 200, 64), (1, 2, 128, 32), (2, 5, 300, 2048), (1, 8, 64, 96)])
def test_attention_d32_matches_sdpa(ops, shape):
    """fp32 flash attention (head_dim 32, token-major layout) against a float64 softmax(QK^T)V; ragged query counts."""
    B, nh, nq, nkv = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    C = nh * 32
    q = torch.randn(B, nq, C, device="cuda", generator=g)
    k = torch.randn(B, nkv, C, device="cuda", generator=g)
    v = torch.randn(B, nkv, C, device="cuda", generator=g)
    scale = 32 ** -0.5
    got = ops.attention_d32(q, k, v, nh, scale)
    qh, kh, vh = (t.double().view(B, -1, nh, 32).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, nq, C)
    assert (got.double() - ref).abs().max().item() < 2e-5
    # large logits: the running-maximum rescaling must hold
    got = ops.attention_d32(q * 8, k * 8, v, nh, scale)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * (64 * scale), dim=-1) @ vh).transpose(1, 2).reshape(B, nq, C)
    assert (got.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("shape", [(2, 1, 200, 64), (1, 2, 128, 32), (2, 5, 300, 2048), (1, 8, 64, 96)])
def test_attention_d32_split_operands_float32_grade(ops, shape):
    """The split-operand f16-MFMA attention (22-bit operands, float32 accumulation) against float64, next to the
    float32-MFMA kernel on the same inputs: same gates, and its error may not exceed a small multiple of the float32
    kernel's.  Includes values that are subnormal as f16 (|x| < 6.1e-5) and large logits."""
    B, nh, nq, nkv = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + 1)
    C = nh * 32
    q = torch.randn(B, nq, C, device="cuda", generator=g)
    k = torch.randn(B, nkv, C, device="cuda", generator=g)
    v = torch.randn(B, nkv, C, device="cuda", generator=g)
    k[:, ::3, ::5] *= 1e-5                       # f16-subnormal high parts
    v[:, 1::4, 3::7] *= 3e-6
    scale = 32 ** -0.5
    qh, kh, vh = (t.double().view(B, -1, nh, 32).transpose(1, 2) for t in (q, k, v))
    for mult, gate in ((1.0, 2e-5), (8.0, 1e-4)):
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * (mult * mult * scale), dim=-1) @ vh).transpose(1, 2).reshape(B, nq, C)
        e32 = (ops.attention_d32(q * mult, k * mult, v, nh, scale, split=False).double() - ref).abs().max().item()
        esp = (ops.attention_d32(q * mult, k * mult, v, nh, scale, split=True).double() - ref).abs().max().item()
        print(f"attention {shape} x{mult}: |err| float32 kernel {e32:.3e}, split kernel {esp:.3e}")
        assert esp < gate
        assert esp < 4 * e32 + 1e-6


def test_attention_d32_split_flat_softmax(ops):
    """2048 keys with nearly equal scores: every probability is ~5e-4 of the row sum — the case that would lose mass if
    small probabilities were flushed in f16."""
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(1, 256, 32, device="cuda", generator=g) * 0.01
    k = torch.randn(1, 2048, 32, device="cuda", generator=g)
    v = torch.randn(1, 2048, 32, device="cuda", generator=g)
    k[:, 5] *= 400.0                             # one dominant key: the others sit ~e^-20 below the maximum for some rows
    ref = (torch.softmax(q.double() @ k.double().transpose(-1, -2) * 32 ** -0.5, dim=-1) @ v.double())
    got = ops.attention_d32(q, k, v, 1, 32 ** -0.5, split=True)
    assert (got.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", [(300, 256, 128), (1000, 19, 304), (257, 2048, 512), (128, 128, 2048), (4100, 320, 72), (200, 160, 264), (1100, 384, 1024), (38400, 256, 128), (38500, 512, 104),
                                   (131100, 256, 128), (65600, 512, 104), (65700, 128, 512), (70000, 384, 136), (66000, 64, 256), (40000, 320, 96),
                                   (270100, 48, 256), (300000, 19, 256), (262200, 32, 128), (262300, 32, 32), (262400, 56, 40), (66000, 160, 160), (65600, 160, 640), (131000, 200, 72)])
@pytest.mark.parametrize("res_act", [(False, 0), (True, 1)])
def test_gemm_split_float32_grade(ops, shape, res_act):
    """Split-operand f16-MFMA GEMM against float64, next to the hipBLASLt float32 GEMM on the same inputs: ragged M and
    N, K not a multiple of the 32-wide K tile, bias / residual / ReLU epilogue, residual aliasing the output; the last
    two shapes before the end have enough tiles for the 128 x 256 block-tile configuration (N % 256 == 0, >= one tile per CU),
    the next two for the 256 x 256 single-accumulator one (ragged M, K tail), the next two for the 256 x 128 one (N % 128 == 0); the
    last eight are N < 64 on 2^18+ rows and N % 64 != 0 above it (160 = 2.5 tiles, 200): masked 64-column tiles of the LDS-DMA kernel (columns past N never stored, a single K tile
    for K = 32, a K tail for K = 40), with the element behind the last column of every row checked untouched."""
    M, Nn, K = shape
    has_res, act = res_act
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0
    w = torch.randn(Nn, K, device="cuda", generator=g) * 0.05
    x[::7, ::3] *= 1e-5                                         # f16-subnormal high parts
    bias = torch.randn(Nn, device="cuda", generator=g)
    res = torch.randn(M, Nn, device="cuda", generator=g) if has_res else None
    ref = x.double() @ w.double().t() + bias.double()
    if has_res:
        ref = ref + res.double()
    if act:
        ref = ref.clamp_min(0)
    ws = ops.gemm_split_weights(w)
    assert ws.shape == (2, Nn, K)
    out = res.clone() if has_res else None
    if Nn % 64 and M > 60000 and not has_res:
        # the output as a view of a buffer with one more row: a store past column N of row m would land in row m + 1 and be
        # overwritten, one past the last row would not — fill with a sentinel and look at the row behind the matrix
        buf = torch.full((M + 1, Nn), 12345.0, device="cuda")
        out = buf[:M]
        got = ops.gemm_split_bias_act(x, ws, bias, act, out=out)
        assert got.data_ptr() == buf.data_ptr() and torch.all(buf[M] == 12345.0)
    else:
        got = ops.gemm_split_bias_act(x, ws, bias, act, residual=out, out=out)
    lib = ops.gemm_bias_act(x, w, bias, act, residual=res, split=False)
    e_split = (got.double() - ref).abs().max().item()
    e_lib = (lib.double() - ref).abs().max().item()
    print(f"gemm {shape} res={has_res}: |err| hipBLASLt f32 {e_lib:.3e}, split {e_split:.3e}")
    assert e_split < 1e-5 * max(1.0, ref.abs().max().item())
    assert e_split < 4 * e_lib + 1e-6


@pytest.mark.parametrize("cfg", [(8, 512, 1024, 64), (3, 200, 328, 64), (2, 1024, 2048, 128)])
def test_stem_rows_gemm_matches_conv7x7(ops, cfg):
    """awseg_conv_rows_gemm_split_bias_act (the 7x7 / stride-2 / padding-3 stem on 3 channels, A operand gathered as runs of 8
    padded pixels per kernel row) against torch's convolution in float64: odd sizes (the last run ends in the right padding),
    bias + ReLU epilogue, an image border on every side."""
    B, H, W, Nn = cfg
    g = torch.Generator(device="cuda").manual_seed(sum(cfg))
    x = torch.randn(B, 3, H, W, device="cuda", generator=g) * 2.0
    wt = torch.randn(Nn, 3, 7, 7, device="cuda", generator=g) * 0.1
    bias = torch.randn(Nn, device="cuda", generator=g)
    wo = (W + 6 - 7) // 2 + 1
    wp = max(W + 3, (wo - 1) * 2 + 8)
    xp = torch.zeros(B, H, wp, 4, device="cuda")
    xp[:, :, 3:3 + W, :3] = x.permute(0, 2, 3, 1)
    ws = ops.gemm_split_weights(ops.stem_rows_weights(wt))
    got = ops.conv_rows_gemm_split(xp, ws, bias, 1, 7, 2, 3, wo)
    assert got is not None, "the LDS-DMA kernel declined a stem shape with thousands of tiles"
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), 2, 3).clamp_min(0).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item()
    print(f"stem rows {cfg}: max abs err {err:.3e} at magnitude {ref.abs().max().item():.1f}")
    assert err < 1e-5 * max(1.0, ref.abs().max().item())
    # a shape with too few tiles is declined (None), not computed some other way
    small = torch.zeros(1, 16, 35, 4, device="cuda")
    assert ops.conv_rows_gemm_split(small, ws, bias, 1, 7, 2, 3, 14) is None


def test_stem_pad_cache_is_keyed_on_width_and_channels(ops):
    """fused._stem_rows keeps one zero-padded [B,H,W+pads,4] image per shape.  W = 2k and W = 2k - 1 need the same padded
    width, so a cache keyed on the padded width alone would hand the narrower input the wider one's last column where its
    zero padding belongs (round-3 advisor finding): run W = 2k, then W = 2k - 1 at the same B and H, against F.conv2d."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused
    g = torch.Generator(device="cuda").manual_seed(77)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(64, 3, 7, 7, device="cuda", generator=g) * 0.1)
    fused._stem_pad.clear()
    for W in (328, 327, 328, 326):
        x = torch.randn(3, 3, 200, W, device="cuda", generator=g) * 2.0 + 5.0      # far from zero: a stale column would show
        got = fused._stem_rows(x.contiguous(memory_format=torch.channels_last), conv, conv.weight)
        assert got is not None
        ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, 2, 3).permute(0, 2, 3, 1)
        err = (got.double() - ref).abs().max().item()
        assert err < 1e-5 * max(1.0, ref.abs().max().item()), (W, err)


@pytest.mark.parametrize("cfg", [(2, 37, 53, 64, 128, 3, 3, 2, 1), (1, 64, 96, 32, 160, 2, 2, 2, 0), (3, 40, 40, 128, 256, 1, 1, 2, 0),
                                 (8, 128, 256, 128, 128, 3, 3, 2, 1), (2, 128, 256, 256, 512, 1, 1, 2, 0), (4, 256, 512, 64, 256, 3, 3, 2, 1)])
def test_conv_gemm_split_equals_im2col_plus_gemm(ops, cfg):
    """awseg_conv_gemm_split_bias_act (A operand gathered from the NHWC image while the K tiles are staged) against the
    im2col matrix + the same GEMM — bit-identical — and against torch's convolution in float64: stride-2 3x3 with padding,
    kernel == stride patches, stride-2 1x1 downsample; ragged tile edges, tiles spanning two images, every block-tile shape."""
    B, H, W, C, Nn, kh, kw, st, pd = cfg
    g = torch.Generator(device="cuda").manual_seed(sum(cfg))
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    wt = torch.randn(Nn, C, kh, kw, device="cuda", generator=g) * 0.05
    bias = torch.randn(Nn, device="cuda", generator=g)
    w2 = wt.permute(0, 2, 3, 1).reshape(Nn, kh * kw * C).contiguous()
    ws = ops.gemm_split_weights(w2)
    got = ops.conv_gemm_split(x, ws, bias, 1, kh, kw, st, pd)
    cols, ho, wo = ops.im2col_nhwc(x, kh, kw, st, pd, 1, kh * kw * C)
    ref_same = ops.gemm_split_bias_act(cols, ws, bias, 1).view(B, ho, wo, Nn)
    assert got.shape == ref_same.shape and torch.equal(got, ref_same)
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=st, padding=pd))
    err = (got.permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err


def test_conv_gemm_split_dilated_and_large_operands(ops):
    """The gathered-A convolution with dilation (atrous 3x3, stride 1) and with activations beyond the f16 range (the GEMM's
    second, scaled pass re-gathers the tile): against torch's convolution in float64."""
    g = torch.Generator(device="cuda").manual_seed(77)
    B, H, W, C, Nn = 2, 33, 47, 64, 128
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    x[0, 5:9, 7:20] *= 1e5
    wt = torch.randn(Nn, C, 3, 3, device="cuda", generator=g) * 0.05
    bias = torch.randn(Nn, device="cuda", generator=g)
    ws = ops.gemm_split_weights(wt.permute(0, 2, 3, 1).reshape(Nn, 9 * C).contiguous())
    for dil in (2, 3):
        got = ops.conv_gemm_split(x, ws, bias, 0, 3, 3, 1, dil, dil)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=1, padding=dil, dilation=dil)
        assert got.shape == (B, H, W, Nn)
        rowmag = ref.abs().amax(dim=1).clamp_min(1.0)                                  # per pixel: rows differ by 1e5
        err = ((got.permute(0, 3, 1, 2).double() - ref).abs().amax(dim=1) / rowmag).max().item()
        assert torch.isfinite(got).all() and err < 1e-5, (dil, err)


@pytest.mark.parametrize("shape", [(300, 256, 128), (38400, 256, 128), (4100, 320, 72), (131100, 256, 128)])
@pytest.mark.parametrize("xscale,wscale", [(1e5, 0.05), (3e4, 1e-9), (1e30, 1e-28), (2.0, 1e6), (1e-30, 1e4)])
def test_gemm_split_large_operands(ops, shape, xscale, wscale):
    """operands outside the f16 range must not give silently wrong products.  Activations at 1e5 / 1e30
    (a few rows only, or all of them), weights at 1e6 / 1e-9 / 1e-28: the split GEMM must stay float32-grade against
    float64 — same gate as test_gemm_split_float32_grade, relative to the result's magnitude."""
    M, Nn, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K + 5)
    x = torch.randn(M, K, device="cuda", generator=g)
    x[::5] *= xscale                                            # every fifth row is large: tiles with and without a second pass
    w = torch.randn(Nn, K, device="cuda", generator=g) * wscale
    bias = torch.randn(Nn, device="cuda", generator=g)
    res = torch.randn(M, Nn, device="cuda", generator=g)
    ref = (x.double() @ w.double().t() + bias.double() + res.double()).clamp_min(0)
    ws = ops.gemm_split_weights(w)
    out = res.clone()
    got = ops.gemm_split_bias_act(x, ws, bias, 1, residual=out, out=out)
    lib = ops.gemm_bias_act(x, w, bias, 1, residual=res, split=False)
    assert torch.isfinite(got).all()
    # errors per ROW, relative to the row's magnitude (rows differ by xscale)
    rowmag = ref.abs().amax(dim=1).clamp_min(1.0)
    e_split = ((got.double() - ref).abs().amax(dim=1) / rowmag).max().item()
    e_lib = ((lib.double() - ref).abs().amax(dim=1) / rowmag).max().item()
    print(f"gemm {shape} x*{xscale:g} w*{wscale:g}: rel err hipBLASLt f32 {e_lib:.3e}, split {e_split:.3e}")
    assert e_split < 1e-5
    assert e_split < 4 * e_lib + 1e-6


def test_gemm_split_propagates_inf_nan(ops):
    x = torch.randn(256, 64, device="cuda"); w = torch.randn(128, 64, device="cuda")
    x[3, 5] = float("inf"); x[77, 0] = float("nan")
    got = ops.gemm_split_bias_act(x, ops.gemm_split_weights(w), None, 0)
    assert not torch.isfinite(got[3]).any() and torch.isnan(got[77]).all()
    ok = torch.ones(256, dtype=torch.bool, device="cuda"); ok[3] = ok[77] = False
    ref = x[ok].double() @ w.double().t()
    assert (got[ok].double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("qs,ks,vs", [(1.0, 1.0, 1e5), (1e5, 1e-4, 1.0), (300.0, 300.0, 7e4), (1e5, 1e5, 1e5), (1e-3, 5e4, 1e20)])
def test_attention_d32_split_large_operands(ops, qs, ks, vs):
    """q / k / v outside the f16 range (|x| up to 1e5 and beyond): the split-operand kernel redoes the tile with
    power-of-two scaled operands; error against float64 within a small multiple of the float32 kernel's."""
    B, nh, nq, nkv = 2, 2, 300, 256
    g = torch.Generator(device="cuda").manual_seed(123)
    C = nh * 32
    q = torch.randn(B, nq, C, device="cuda", generator=g) * qs
    k = torch.randn(B, nkv, C, device="cuda", generator=g) * ks
    v = torch.randn(B, nkv, C, device="cuda", generator=g) * vs
    scale = 32 ** -0.5
    qh, kh, vh = (t.double().view(B, -1, nh, 32).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B, nq, C)
    mag = ref.abs().max().item()
    o32 = ops.attention_d32(q, k, v, nh, scale, split=False)
    osp = ops.attention_d32(q, k, v, nh, scale, split=True)
    assert torch.isfinite(osp).all()
    e32 = (o32.double() - ref).abs().max().item() / mag
    esp = (osp.double() - ref).abs().max().item() / mag
    print(f"attention q*{qs:g} k*{ks:g} v*{vs:g}: rel err float32 kernel {e32:.3e}, split kernel {esp:.3e}")
    assert esp < 4 * e32 + 2e-6


def test_gemm_split_abi_checks(ops):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native as N
    x = torch.zeros(8, 12, device="cuda"); ws = torch.zeros(2 * 4 * 16 + 8, dtype=torch.int16, device="cuda"); o = torch.zeros(8, 4, device="cuda")
    rc = N.lib().awseg_gemm_split_bias_act(N.ptr(x), N.ptr(ws), None, None, 0, N.ptr(o), 8, 4, 12, N.stream())
    assert rc != 0                                              # K % 8 != 0
    rc = N.lib().awseg_gemm_split_bias_act(N.ptr(x), N.ptr(ws), None, None, 0, N.ptr(o), 0, 4, 16, N.stream())
    assert rc == 0                                              # empty problem


def test_attention_d32_split_tiny_operands(ops):
    """Whole tensors below the f16 normal range (|x| ~ 1e-5): the high parts are f16 SUBNORMALS, so this fails by
    ~100 % if the f16 matrix cores flushed subnormal inputs; the relative error must stay at float32 level."""
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(1, 128, 64, device="cuda", generator=g)
    k = torch.randn(1, 256, 64, device="cuda", generator=g)
    v = torch.randn(1, 256, 64, device="cuda", generator=g) * 1e-5
    qh, kh, vh = (t.double().view(1, -1, 2, 32).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 32 ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(1, 128, 64)
    got = ops.attention_d32(q, k, v, 2, 32 ** -0.5, split=True).double()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 5e-6
    # tiny keys: scores ~1e-5, the softmax is uniform up to 1e-5 — flushing would make it exactly uniform
    k2 = k * 1e-5
    v2 = torch.randn(1, 256, 64, device="cuda", generator=g)
    ref = torch.softmax(qh @ (kh * 1e-5).transpose(-1, -2) * 32 ** -0.5, dim=-1) @ v2.double().view(1, -1, 2, 32).transpose(1, 2)
    uni = v2.double().view(1, -1, 2, 32).transpose(1, 2).mean(dim=2, keepdim=True)
    got = ops.attention_d32(q, k2, v2, 2, 32 ** -0.5, split=True).double().view(1, 128, 2, 32).transpose(1, 2)
    assert (got - ref).abs().max().item() < 0.05 * (ref - uni).abs().max().item()


@pytest.mark.parametrize("shape", [(2, 4, 8, 64, 128), (1, 3, 5, 37, 61), (1, 64, 128, 1024, 2048)])
def test_depth_upsample_combine_matches_torch(ops, shape):
    B, h, w, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    d1 = torch.rand(B, 1, H, W, device="cuda", generator=g); lo = torch.rand(B, 1, h, w, device="cuda", generator=g)
    wts = torch.softmax(torch.tensor([0.3, -0.2], device="cuda"), 0)
    ref2 = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)
    d2, d = ops.depth_upsample_combine(d1, lo, wts)
    assert (d2 - ref2).abs().max().item() < 1e-6 and (d - (wts[0] * d1 + wts[1] * ref2)).abs().max().item() < 1e-6
    d2, d = ops.depth_upsample_combine(d1, lo, None)
    assert (d - (d1 + ref2) / 2).abs().max().item() < 1e-6



@pytest.mark.parametrize("case", [(2, 17, 23, 8, 3, 3, 2, 1), (1, 32, 64, 32, 8, 8, 8, 0), (2, 16, 20, 160, 2, 2, 2, 0), (1, 9, 11, 4, 3, 3, 2, 1),
                                  (2, 12, 16, 128, 3, 3, 2, 1)])
def test_im2col_patch_gemm_is_the_convolution(ops, case):
    """awseg_im2col_nhwc + one GEMM == F.conv2d for the strided / patch convolutions (ResNet stride-2 3x3, MiT patch
    embeddings and sequence reductions): the patch matrix bit-exact against torch's unfold, the product within fp32
    summation-order noise of the convolution; K padded to a multiple of 8 with zero columns."""
    import torch.nn.functional as F
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused
    b, h, w, c, kh, kw, s, p = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    x = torch.randn(b, h, w, c, device="cuda", generator=g)
    kp = (kh * kw * c + 7) // 8 * 8
    cols, ho, wo = ops.im2col_nhwc(x, kh, kw, s, p, 1, kp)
    un = F.unfold(x.permute(0, 3, 1, 2), (kh, kw), dilation=1, padding=p, stride=s)             # [B, C*kh*kw, L], (c, ky, kx) order
    un = un.view(b, c, kh * kw, ho * wo).permute(0, 3, 2, 1).reshape(b * ho * wo, kh * kw * c)    # -> (ky, kx, c)
    assert torch.equal(cols[:, :kh * kw * c], un) and (cols[:, kh * kw * c:] == 0).all()
    conv = torch.nn.Conv2d(c, 24, (kh, kw), stride=s, padding=p).cuda()
    w2 = fused.patch_weights(conv)
    assert w2.shape == (24, kp)
    y = fused.conv_gemm_nhwc(x, conv, w2, conv.bias, 0)
    ref = conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and (y - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case", [(2, 4, 6, 64, 128, 128, 192), (1, 5, 8, 32, 256, 160, 256), (1, 3, 4, 16, 32, 100, 130),
                                  (2, 5, 8, 256, 256, 160, 256), (2, 5, 8, 256, 128, 160, 256)])
def test_upconv3x3_train_forward_and_adjoint_vs_torch_autograd(ops, case):
    """ops.upconv3x3_train (awseg_upconv3x3_linear + awseg_upconv3x3_adjoint) against torch autograd of the expression it
    replaces in training — F.conv2d(F.interpolate(f, (H, W), bilinear, align_corners=False), w, b, padding=1), PKG/models/
    model.py:209-214 — in float64: forward, and the gradients with respect to the features, the filters and the bias."""
    import torch.nn.functional as F
    B, h, w, cin, cmid, H, W = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    f = torch.randn(B, cin, h, w, device="cuda", generator=g)
    wt = torch.randn(cmid, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)
    bias = torch.randn(cmid, device="cuda", generator=g)
    dz = torch.randn(B, cmid, H, W, device="cuda", generator=g)
    assert ops.upconv3x3_train_supported(cin, cmid, h, w, H, W)
    tok = f.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    w1, b1 = wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    z = ops.upconv3x3_train(tok, w1, b1, H, W)
    (z * dz).sum().backward()
    f64, w64, b64 = f.double().requires_grad_(True), wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = F.conv2d(F.interpolate(f64, size=(H, W), mode="bilinear", align_corners=False), w64, b64, padding=1)
    (ref * dz.double()).sum().backward()

    def rel(a, b):
        return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    e = {"z": rel(z, ref), "d features": rel(tok.grad.permute(0, 3, 1, 2), f64.grad), "d filters": rel(w1.grad, w64.grad), "d bias": rel(b1.grad, b64.grad)}
    print(f"upconv3x3 train {case}: " + ", ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert z.shape == (B, cmid, H, W) and all(v < 2e-5 for v in e.values()), e


# ------------------------------------------------------------------ the SegFormer depth head as ONE launch (depthfuse.hip)
@pytest.mark.parametrize("hwc", [(1, 1, 16), (1, 2, 32), (2, 3, 128), (3, 2, 48)])
def test_upconv_forms_table_matches_float64_restatement(ops, hwc):
    """awseg_upconv_forms against tests/forms_ref.py (the float64 derivation of the bilinear forms, itself checked against
    interpolate -> conv2d to 1e-14 on the CPU): every entry of F4 / F2, borders and half cells included."""
    from tests import forms_ref
    h, w, C = hwc
    rs = np.random.RandomState(h * 100 + w * 10 + C)
    G = rs.randn(2, h, w, 9, C)
    shift = rs.randn(C)
    forms = ops.upconv_forms(dev(G, torch.float32), dev(shift, torch.float32)).cpu().numpy().astype(np.float64)
    for b in range(2):
        F4, F2 = forms_ref.build_tables(G[b].astype(np.float32).astype(np.float64), shift.astype(np.float32).astype(np.float64))
        ref = np.concatenate([F4.reshape(-1), F2.reshape(-1)])
        assert forms[b].shape == ref.shape
        err = np.abs(forms[b] - ref).max()
        assert err < 2e-6 * max(1.0, np.abs(ref).max()), err          # float64 sums rounded once to float32


@pytest.mark.parametrize("cfg", [(1, 1, 1, 128), (2, 1, 2, 128), (2, 2, 3, 128), (1, 3, 2, 64), (2, 4, 5, 128)])
def test_depth_head_fused_matches_as_written_module(ops, cfg):
    """DepthEstimationHead.forward_from_lowres — ONE full-resolution launch, hidden map generated inside the Winograd kernel —
    against the as-written module on F.interpolate(features) in float64 (PKG/models/model.py:16-78, :211, :219-221): every
    cell / border class (h or w = 1: half cells only), non-trivial BatchNorm statistics; and against the two-launch path
    (awseg_upconv3x3_bn_relu + Winograd).  1e-4 abs is north_star's gate; measured ~1e-6."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import DepthEstimationHead
    B, h, w, hidden = cfg
    torch.manual_seed(B * 1000 + h * 100 + w * 10)
    head = DepthEstimationHead(in_channels=256, hidden_channels=hidden).eval()
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.3)
            if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
                m.bias.normal_(0, 0.2)
    feats = torch.randn(B, h, w, 256)
    ref_mod = DepthEstimationHead(in_channels=256, hidden_channels=hidden).eval().double()
    ref_mod.load_state_dict(head.state_dict())
    with torch.no_grad():
        up = torch.nn.functional.interpolate(feats.permute(0, 3, 1, 2).double(), size=(32 * h, 32 * w), mode="bilinear", align_corners=False)
        ref = ref_mod(up).numpy()
    head = head.cuda()
    saved = ops.DEPTH_FUSED
    try:
        with torch.no_grad():
            ops.DEPTH_FUSED = True
            got = head.forward_from_lowres(feats.cuda(), 32 * h, 32 * w).cpu().numpy()
            ops.DEPTH_FUSED = False
            two = head.forward_from_lowres(feats.cuda(), 32 * h, 32 * w).cpu().numpy()
    finally:
        ops.DEPTH_FUSED = saved
    e_f, e_2 = np.abs(got - ref).max(), np.abs(two - ref).max()
    print(f"depth head fused {cfg}: |err| one launch {e_f:.2e}, two launches {e_2:.2e}, between them {np.abs(got - two).max():.2e}")
    assert got.shape == ref.shape
    assert e_f <= 1e-5 and e_2 <= 1e-5


def test_depth_head_fused_range_guard(ops):
    """The generated patch goes through the same operand-range guard as a fetched one: hidden activations of 3e4 (the block reruns
    its tile scaled) and of 1e-3 — against float64."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import DepthEstimationHead
    torch.manual_seed(5)
    for gain in (3e4, 1e-3):
        head = DepthEstimationHead(in_channels=256, hidden_channels=128).eval()
        with torch.no_grad():
            head.depth_head[0].weight.mul_(gain)
            head.depth_head[4].weight.mul_(1.0 / gain)
        feats = torch.randn(1, 2, 2, 256)
        ref_mod = DepthEstimationHead(in_channels=256, hidden_channels=128).eval().double()
        ref_mod.load_state_dict(head.state_dict())
        with torch.no_grad():
            up = torch.nn.functional.interpolate(feats.permute(0, 3, 1, 2).double(), size=(64, 64), mode="bilinear", align_corners=False)
            ref = ref_mod(up).numpy()
            got = head.cuda().forward_from_lowres(feats.cuda(), 64, 64).cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5, (gain, np.abs(got - ref).max())


# ------------------------------------------------------------------ MiT Mix-FFN as one tile kernel (mixffn.hip)
@pytest.mark.parametrize("cfg", [(2, 13, 37, 32), (1, 6, 30, 64), (1, 20, 70, 64), (3, 1, 1, 32), (1, 64, 96, 32)])
def test_mixffn_fused_matches_float64_ops(ops, cfg):
    """awseg_mixffn_fused (LayerNorm -> fc1 -> depthwise 3x3 -> GELU -> fc2 -> + residual in one launch, hidden map in LDS)
    against the same five torch operations in float64 (transformers' SegformerLayer second half): ragged tiles (H % 6, W % 30),
    frames smaller than a tile, tile borders inside the frame, both token widths.  1e-5 of the output magnitude."""
    B, H, W, C = cfg
    g = torch.Generator(device="cuda").manual_seed(sum(cfg))
    tok = torch.randn(B, H, W, C, device="cuda", generator=g) * 2.0 + 0.3
    gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.2
    w1, b1 = torch.randn(4 * C, C, device="cuda", generator=g) / C ** 0.5, torch.randn(4 * C, device="cuda", generator=g) * 0.1
    wd, bd = torch.randn(4 * C, 1, 3, 3, device="cuda", generator=g) * 0.3, torch.randn(4 * C, device="cuda", generator=g) * 0.1
    w2, b2 = torch.randn(C, 4 * C, device="cuda", generator=g) / (4 * C) ** 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    taps = wd.view(4 * C, 9).t().contiguous()
    got = ops.mixffn_fused(tok, gamma, beta, 1e-6, w1, b1, taps, bd, w2, b2)
    assert got is not None and got.data_ptr() != tok.data_ptr()
    F = torch.nn.functional
    td = tok.double()
    h = F.linear(F.layer_norm(td, (C,), gamma.double(), beta.double(), 1e-6), w1.double(), b1.double())       # [B,H,W,4C]
    h = F.conv2d(h.permute(0, 3, 1, 2), wd.double(), bd.double(), 1, 1, 1, 4 * C).permute(0, 2, 3, 1)
    ref = td + F.linear(F.gelu(h), w2.double(), b2.double())
    err = (got.double() - ref).abs().max().item()
    print(f"mixffn fused {cfg}: max abs err {err:.2e} at magnitude {ref.abs().max().item():.1f}")
    assert err < 1e-5 * max(1.0, ref.abs().max().item())
    # GELU outputs beyond the f16 operand range: the chunk's fc2 runs on the float32-input MFMA (block-uniform branch) — same gate
    big = ops.mixffn_fused(tok, gamma, beta, 1e-6, w1, b1 + 3.0e5, taps, bd, w2 * 1e-3, b2)
    hb = F.linear(F.layer_norm(td, (C,), gamma.double(), beta.double(), 1e-6), w1.double(), b1.double() + 3.0e5)
    hb = F.conv2d(hb.permute(0, 3, 1, 2), wd.double(), bd.double(), 1, 1, 1, 4 * C).permute(0, 2, 3, 1)
    refb = td + F.linear(F.gelu(hb), (w2 * 1e-3).double(), b2.double())
    errb = (big.double() - refb).abs().max().item()
    print(f"mixffn fused {cfg}, hidden activations ~3e5: max abs err {errb:.2e} at magnitude {refb.abs().max().item():.1f}")
    assert errb < 1e-5 * max(1.0, refb.abs().max().item())
    # parameters beyond the f16 operand range are declined (the caller keeps its separate launches)
    assert ops.mixffn_fused(tok, gamma, beta, 1e-6, w1 * 1e6, b1, taps, bd, w2, b2) is None
    # widths the kernel does not take are declined, not computed some other way
    assert ops.mixffn_fused(torch.zeros(1, 4, 4, 160, device="cuda"), torch.ones(160, device="cuda"), torch.zeros(160, device="cuda"), 1e-6,
                            torch.zeros(640, 160, device="cuda"), torch.zeros(640, device="cuda"), torch.zeros(9, 640, device="cuda"),
                            torch.zeros(640, device="cuda"), torch.zeros(160, 640, device="cuda"), torch.zeros(160, device="cuda")) is None


# ------------------------------------------------------------------ depthwise 3x3 under autograd (training step, dwtrain.hip)
@pytest.mark.parametrize("cfg", [(2, 9, 13, 8, 1, True), (1, 33, 47, 128, 1, True), (2, 20, 24, 64, 12, False), (1, 64, 96, 304, 1, False), (3, 5, 5, 4, 2, True)])
def test_depthwise_conv3x3_train_matches_torch_autograd(ops, cfg):
    """_DepthwiseConv3x3NHWC (forward / input gradient on awseg_dwconv3x3_nhwc, weight + bias gradient on awseg_dwconv3x3_wgrad_nhwc)
    against torch's float64 autograd of the same nn.Conv2d: dilation 1 / 2 / 12 (larger than the frame's half: every tap clipped
    somewhere), ragged chunks, channel counts below and above one 32-quad slice."""
    B, H, W, C, d, has_bias = cfg
    g = torch.Generator(device="cuda").manual_seed(sum(cfg[:5]))
    conv = torch.nn.Conv2d(C, C, 3, 1, d, d, groups=C, bias=has_bias).cuda()
    x = torch.randn(B, H, W, C, device="cuda", generator=g, requires_grad=True)
    gy = torch.randn(B, H, W, C, device="cuda", generator=g)
    assert ops.depthwise_conv3x3_train_ok(conv, x)
    y = ops.depthwise_conv3x3_nhwc_train(x, conv)
    y.backward(gy)
    got = (y.detach(), x.grad.clone(), conv.weight.grad.clone(), None if not has_bias else conv.bias.grad.clone())
    ref_conv = torch.nn.Conv2d(C, C, 3, 1, d, d, groups=C, bias=has_bias).cuda().double()
    ref_conv.load_state_dict(conv.state_dict())
    xd = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = ref_conv(xd)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    ref = (yr.detach().permute(0, 2, 3, 1), xd.grad.permute(0, 2, 3, 1), ref_conv.weight.grad, None if not has_bias else ref_conv.bias.grad)
    for name, a, b in zip(("forward", "input gradient", "weight gradient", "bias gradient"), got, ref):
        if a is None:
            continue
        err = (a.double() - b).abs().max().item()
        assert err < 2e-5 * max(1.0, b.abs().max().item()), (cfg, name, err)


# ------------------------------------------------------------------ BatchNorm2d -> ReLU -> Dropout2d of the training heads (bntrain.hip)
@pytest.mark.parametrize("cfg", [(2, 8, 6, 10, 0.1), (3, 128, 32, 64, 0.1), (1, 5, 4, 4, 0.0), (2, 64, 130, 66, 0.5)])
def test_bn_relu_dropout2d_train_matches_the_module_graph(ops, cfg):
    """ops._BNReLUDropout2d against nn.BatchNorm2d(train) -> nn.ReLU -> nn.Dropout2d in float64 on the SAME Dropout2d draw (same
    seed: the Function draws its mask exactly as F.dropout2d does): output, input gradient, gamma / beta gradients, running
    statistics (unbiased variance, momentum), num_batches_tracked."""
    B, C, H, W, p = cfg
    g = torch.Generator(device="cuda").manual_seed(sum(int(v * 10) for v in cfg))
    x = (torch.randn(B, C, H, W, device="cuda", generator=g) * 2.0 + 0.5).requires_grad_(True)
    gy = torch.randn(B, C, H, W, device="cuda", generator=g)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    ref_bn = torch.nn.BatchNorm2d(C).cuda().double().train()
    ref_bn.load_state_dict(bn.state_dict())
    relu, drop = torch.nn.ReLU(), (torch.nn.Dropout2d(p).train() if p > 0 else None)
    assert ops.bn_relu_dropout2d_train_ok(x, bn, relu, drop)
    torch.manual_seed(1234)
    y = ops.bn_relu_dropout2d_train(x, bn, drop)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    torch.manual_seed(1234)
    noise = torch.empty(B, C, 1, 1, device="cuda").bernoulli_(1.0 - p).div_(1.0 - p).double() if p > 0 else 1.0
    yr = torch.relu(ref_bn(xd)) * noise
    yr.backward(gy.double())
    for name, a, b in (("output", y.detach(), yr.detach()), ("dx", x.grad, xd.grad), ("dgamma", bn.weight.grad, ref_bn.weight.grad),
                       ("dbeta", bn.bias.grad, ref_bn.bias.grad), ("running_mean", bn.running_mean, ref_bn.running_mean),
                       ("running_var", bn.running_var, ref_bn.running_var)):
        err = (a.double() - b).abs().max().item()
        assert err < 3e-5 * max(1.0, b.abs().max().item()), (cfg, name, err)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    if C % 32 == 0:
        # the same backward with the input gradient handed back over channels-last memory (what ops._UpConv3x3's adjoint reads): same values
        x2 = x.detach().clone().requires_grad_(True)
        torch.manual_seed(1234)
        y2 = ops.bn_relu_dropout2d_train(x2, bn, drop, dx_channels_last=True)
        y2.backward(gy)
        assert x2.grad.shape == x.grad.shape and torch.equal(x2.grad.contiguous(), x.grad)     # (autograd lays a leaf's .grad out as the leaf)
        xi = x.detach().clone().requires_grad_(True)
        torch.manual_seed(1234)
        mid = xi * 1.0                                                # a non-leaf in front: its incoming gradient keeps the Function's layout
        seen = []
        mid.register_hook(lambda gr: seen.append(gr.permute(0, 2, 3, 1).is_contiguous()))
        ops.bn_relu_dropout2d_train(mid, bn, drop, dx_channels_last=True).backward(gy)
        assert seen == [True] and torch.equal(xi.grad.contiguous(), x.grad)


# ------------------------------------------------------------------ one launch for a batch of mixed conditions (awseg_weather_batch)
@pytest.mark.parametrize("shape", [(8, 64, 128), (5, 40, 72), (7, 96, 260)])
def test_weather_batch_one_launch_equals_the_per_kind_launchers(ops, shape, monkeypatch):
    """WeatherDegradationTransforms.apply_batch in throughput mode (Philox noise, per-frame streams): the ONE-launch path
    (awseg_weather_batch: a job per frame, the per-kind kernels' bodies) writes byte-identical uint8 frames and bit-identical
    normalised tensors to the per-kind launchers on the same draws — every condition, 3x3 and 7x7 snow, ragged tile edges."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms
    B, H, W = shape
    rs = np.random.RandomState(B * 1000 + H)
    imgs = dev(rs.randint(0, 255, (B, H, W, 3), dtype=np.uint8))
    conds = [["clean", "fog", "rain", "snow", "night"][(3 * i + 1) % 5] for i in range(B)]
    ids = list(range(100, 100 + B))
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "WEATHER_BATCH", mode)
        tf = WeatherDegradationTransforms(seed=None, rng="philox", device=torch.device("cuda"))
        tf._frame_seed = 77
        out = torch.zeros_like(imgs)
        norm = torch.zeros(B, 3, H, W, device="cuda")
        calls = []
        orig = ops.N.call
        tf.apply_batch(imgs, conds, out=out, norm_out=norm, frame_ids=ids)
        res[mode] = (out.cpu().numpy(), norm.cpu().numpy())
    assert np.array_equal(res[True][0], res[False][0])
    assert np.array_equal(res[True][1], res[False][1])
    assert not np.array_equal(res[True][0], imgs.cpu().numpy())          # something was corrupted at all


def test_upsample_bilinear_reads_nhwc_rows_through_strides(ops):
    """The 19-class head's GEMM writes NHWC rows; the x4 upsampling reads them through the strides of the NCHW view — same
    values, bit for bit, as on a planar copy (and a non-x4 scale, which takes planar maps only, makes that copy itself)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randn(2, 24, 40, 19, device="cuda", generator=g)              # [B,h,w,C]
    view = rows.permute(0, 3, 1, 2)
    assert not view.is_contiguous()
    prev = ops.STRIDED_UPSAMPLE
    try:
        ops.STRIDED_UPSAMPLE = True                                # (off by default: the planar copy is the faster way at the bench shape)
        for size, align in (((96, 160), True), ((96, 160), False), ((48, 80), True)):
            ref = ops.upsample_bilinear(view.contiguous(), size, align)
            got = ops.upsample_bilinear(view, size, align)
            assert torch.equal(got, ref)
    finally:
        ops.STRIDED_UPSAMPLE = prev
    assert ops.N.lib().awseg_upsample_bilinear_strided(ops.N.ptr(rows), 2, 19, 24, 40, 24 * 40 * 19, 1, 40 * 19, 19, 48, 80, 1,
                                                      ops.N.ptr(torch.empty(2, 19, 48, 80, device="cuda")), None) == -2     # AWSEG_ERANGE


@pytest.mark.parametrize("shape", [(2, 1, 200, 64), (1, 2, 128, 32), (2, 5, 300, 2048)])
def test_attention_d32_packed_keys_and_values(ops, shape):
    """Keys and values packed per token ([key | value] rows, as one GEMM over the stacked projection weights writes them): the
    three kernels read them with the doubled row pitch and return what they return on separate tensors."""
    B, nh, nq, nkv = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + 3)
    C = nh * 32
    q = torch.randn(B, nq, C, device="cuda", generator=g)
    kv = torch.randn(B, nkv, 2 * C, device="cuda", generator=g)
    k, v = kv[..., :C].contiguous(), kv[..., C:].contiguous()
    for split in (False, True):
        assert torch.equal(ops.attention_d32_packed_kv(q, kv, nh, 32 ** -0.5, split=split), ops.attention_d32(q, k, v, nh, 32 ** -0.5, split=split))
    with ops.precision("bf16"):
        assert torch.equal(ops.attention_d32_packed_kv(q, kv, nh, 32 ** -0.5), ops.attention_d32(q, k, v, nh, 32 ** -0.5))


def test_rowdot_sigmoid_and_aspp_pool_branch_vs_float64(ops):
    """The two small tails of the DeepLab member: 1x1 convolution to one channel + Sigmoid on NHWC rows, and the ASPP pooling
    branch (1x1 + folded BatchNorm + ReLU on one row per image, then its slice of the projection) — against float64; the pooled
    branch twice and with a batch above one pass of the kernel."""
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(1000, 128, device="cuda", generator=g)
    w = torch.randn(128, device="cuda", generator=g) * 0.1
    b = torch.randn(1, device="cuda", generator=g)
    ref = torch.sigmoid(x.double() @ w.double() + b.double())
    assert (ops.rowdot_sigmoid(x, w, b).double() - ref).abs().max().item() < 1e-6
    assert (ops.rowdot_sigmoid(x, w, None, sigmoid=False).double() - x.double() @ w.double()).abs().max().item() < 1e-5
    for batch, cin, cmid, cout in ((8, 2048, 256, 256), (11, 520, 30, 19)):
        mean = torch.randn(batch, cin, device="cuda", generator=g)
        w1 = torch.randn(cmid, cin, device="cuda", generator=g) / cin ** 0.5
        b1 = torch.randn(cmid, device="cuda", generator=g)
        w2 = torch.randn(cout, cmid, device="cuda", generator=g) / cmid ** 0.5
        b2 = torch.randn(cout, device="cuda", generator=g)
        ref = torch.relu(mean.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
        for _ in range(2):
            got = ops.aspp_pool_branch(mean, w1, b1, w2, b2)
            assert (got.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("case", [(5000, 64, 64, 256, 0), (300, 512, 1024, 2048, 0), (1, 128, 256, 512, 2), (40000, 64, 64, 256, 0), (3, 256, 512, 1024, 2),
                                  (2, 32, 96, 64, 2)])
def test_gemm_split_dual_operand_vs_float64(ops, case):
    """The split-operand GEMM with its A operand in two pieces along K (conv3 + downsample branch of a stage's first bottleneck as one
    product): plain second rows and a stride-2 gather from an NHWC image (odd sizes: the last row / column of the grid), ragged M,
    bias + ReLU, against float64 and next to the two separate split GEMMs it replaces."""
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    mb, k1, k2, n, st = case
    if st == 0:
        m = mb
        x2 = torch.randn(m, k2, device="cuda", generator=g)
        rows2 = x2
    else:
        B, H, W = mb, 21, 27
        x2 = torch.randn(B, H, W, k2, device="cuda", generator=g)
        rows2 = x2[:, ::st, ::st].reshape(-1, k2)
        m = rows2.shape[0]
    x = torch.randn(m, k1, device="cuda", generator=g)
    w = torch.randn(n, k1 + k2, device="cuda", generator=g) / (k1 + k2) ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    ref = torch.relu(torch.cat([x, rows2], dim=1).double() @ w.double().t() + b.double())
    got = ops.gemm_split_dual(x, x2, ops.gemm_split_weights(w), b, 1, stride=st)
    assert got is not None
    err = (got.double() - ref).abs().max().item()
    two = ops.gemm_bias_act(x, w[:, :k1].contiguous(), b, 1, residual=ops.gemm_bias_act(rows2.contiguous(), w[:, k1:].contiguous(), torch.zeros_like(b), 0, split=True), split=True)
    err2 = (two.double() - ref).abs().max().item()
    print(f"dual {case}: |err| {err:.2e} (two launches {err2:.2e})")
    assert err < 2e-5 and err < 4 * err2 + 1e-6
    # large activations in the second source only: the range guard must see them
    got = ops.gemm_split_dual(x, x2 * 3e4, ops.gemm_split_weights(w), b, 1, stride=st)
    ref = torch.relu(torch.cat([x, rows2 * 3e4], dim=1).double() @ w.double().t() + b.double())
    assert (got.double() - ref).abs().max().item() < 2e-5 * 3e4


@pytest.mark.parametrize("shape", [(2, 16, 24, 64, (3, 5, 7)), (1, 64, 128, 2048, (12, 24, 36)), (3, 9, 8, 32, (12, 24, 36))])
def test_aspp_depthwise3_with_mean_output(ops, shape):
    """The ASPP depthwise pass that also leaves the pooling branch's global average: same three maps as awseg_aspp_depthwise3 bit for
    bit, mean against float64 (and twice: the reduction order is fixed, so the bits are)."""
    B, h, w, C, rates = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape[:4]))
    x = torch.randn(B, h, w, C, device="cuda", generator=g)
    wdw = torch.randn(3, 9, C, device="cuda", generator=g)
    ref = ops.aspp_depthwise3(x, wdw, rates)
    got, mean = ops.aspp_depthwise3_mean(x, wdw, rates)
    assert mean is not None
    assert torch.equal(got, ref)
    assert (mean.double() - x.double().mean(dim=(1, 2))).abs().max().item() < 1e-6
    assert torch.equal(ops.aspp_depthwise3_mean(x, wdw, rates)[1], mean)
    # a width the LDS-staged walk does not take: the maps still come, the mean is left to the caller
    xs = torch.randn(1, 5, 4, 32, device="cuda", generator=g)
    got, mean = ops.aspp_depthwise3_mean(xs, wdw[:, :, :32].contiguous(), rates)
    assert mean is None and torch.equal(got, ops.aspp_depthwise3(xs, wdw[:, :, :32].contiguous(), rates))


def test_stem_image_fill_matches_strided_copy(ops):
    """awseg_stem_image against the torch expression it replaces (planar frames -> columns 3.. of the zero-padded NHWC image), for a
    ragged width, a channels-last input and a 4-channel input; untouched cells keep what the buffer held."""
    g = torch.Generator(device="cuda").manual_seed(3)
    for B, C, H, W, cl in ((2, 3, 9, 37, False), (1, 3, 8, 64, True), (2, 4, 5, 16, False)):
        x = torch.randn(B, C, H, W, device="cuda", generator=g)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        wp = W + 11
        ref = torch.full((B, H, wp, 4), 7.0, device="cuda")
        got = ref.clone()
        ref[:, :, 3:3 + W, :C] = x.permute(0, 2, 3, 1)
        if C < 4:
            ref[:, :, 3:3 + W, C:] = 0.0                      # channels past C of the interior are written as zeros
        ops.stem_image_fill(x, got)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("scales", [(1.0, 1.0, 1.0), (1e5, 1.0, 1.0), (1.0, 3e5, 1.0), (1.0, 1.0, 7e4), (2e5, 1e5, 1e6)])
def test_attention_d32_prepared_key_value_image_equals_in_block_staging(ops, scales):
    """The split-operand attention that reads key / value tiles prepared once per launch (awseg_attention_d32_split_ws: LDS-DMA of
    f16 high | low images) against the kernel whose query blocks split the tiles themselves: the same values bit for bit, in range
    and with each of q, k, v (and all three) beyond the f16 range — the image kernel settles the key / value exponents, the query
    blocks the query's."""
    qs, ks, vs = scales
    B, nh, nq, nkv = 2, 2, 2100, 256                            # (>= 8 queries per key: the shapes ops routes to the image form)
    g = torch.Generator(device="cuda").manual_seed(int(qs + ks + vs) % 1000)
    C = nh * 32
    q = torch.randn(B, nq, C, device="cuda", generator=g) * qs
    k = torch.randn(B, nkv, C, device="cuda", generator=g) * ks
    v = torch.randn(B, nkv, C, device="cuda", generator=g) * vs
    prev = ops.ATTN_KV_IMAGE
    try:
        ops.ATTN_KV_IMAGE = False
        ref = ops.attention_d32(q, k, v, nh, 32 ** -0.5, split=True)
        ops.ATTN_KV_IMAGE = True
        got = ops.attention_d32(q, k, v, nh, 32 ** -0.5, split=True)
    finally:
        ops.ATTN_KV_IMAGE = prev
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("case", [(70000, 256, 256, 4), (5000, 256, 64, 3), (300, 512, 128, 2), (65536, 256, 256, 4)])
def test_gemm_split_pieces_vs_float64(ops, case):
    """The split-operand GEMM over an A operand in 2 .. 4 equally wide pieces (the ASPP projection over its branches without the
    concatenated map): against float64 with bias, ReLU and a residual that aliases the output."""
    m, n, kp, npieces = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    pieces = [torch.randn(m, kp, device="cuda", generator=g) for _ in range(npieces)]
    w = torch.randn(n, kp * npieces, device="cuda", generator=g) / (kp * npieces) ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    r = torch.randn(m, n, device="cuda", generator=g)
    ref = torch.relu(torch.cat(pieces, dim=1).double() @ w.double().t() + b.double() + r.double())
    acc = r.clone()
    got = ops.gemm_split_pieces(pieces, ops.gemm_split_weights(w), b, 1, residual=acc, out=acc)
    assert got is not None and got.data_ptr() == acc.data_ptr()
    assert (got.double() - ref).abs().max().item() < 2e-5
