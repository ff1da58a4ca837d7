"""-m gpu: BASELINE.json configs[2] .. [4] at their STATED shapes, so that the driver's run of this suite — not a
builder-run bench line — is the record for them.

* C5 (configs[4]): SegFormer-B5 + DeepLabV3+-R101, bf16 MFMA path, ONE 2048x1024 frame, against the float32-grade
  path on the same weights: max-abs logit error, logit magnitude and argmax agreement printed; asserted at the stated
  bf16 tolerance.
* C4 (configs[3]): ONE AdverseWeatherTrainer optimisation step at 1024x2048 (batch 2 — the smallest the graph accepts: the ASPP
  image-pooling BatchNorm sees one value per channel and sample and refuses a training batch of 1, in smp as here;
  the reference's op graph under autograd keeps ~25 GB per frame alive): loss dict against the as-written graph (F.interpolate -> Conv2d -> ...)
  <= 1e-4, every gradient finite, the weights move.
* C3 (configs[2]): bench.py --gpus 2 on backend nccl (= RCCL) when the lease has two devices: rccl_ranks == 2 and the
  mIoU dict equals the N=1 run's key for key.  Skips on a one-GPU lease, so the first multi-GPU lease proves it.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def P(native):
    import adverse_weather_semantic_segmentation_robustness_benchmark_amd as pkg
    return pkg


def _calibrate(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 1.5 + 0.5)
            mod.running_mean.copy_((torch.rand(mod.running_mean.shape, generator=g) - 0.5) * 0.2)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.25)
    return model


# stated bf16 tolerance for a whole B5 + R101 forward: operands carry 8 significant bits (2^-9 relative rounding), the
# networks are ~100 contractions deep with residual paths, errors add like a random walk: 100^0.5 * 2^-9 ~ 2e-2 of the
# logit magnitude; asserted at 5e-2 of the magnitude, argmax agreement >= 97 % (flips sit on near-ties)
C5_REL_TOL, C5_AGREE = 5e-2, 0.97


def test_c5_b5_r101_bf16_at_2048x1024(P):
    torch.manual_seed(55)
    m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False,
                        segformer_name="nvidia/segformer-b5-finetuned-cityscapes-1024-1024", deeplab_backbone="resnet101", compute_dtype="bf16")
    m = _calibrate(m).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(1, 3, 1024, 2048, device="cuda", generator=g)            # 2048 x 1024 Cityscapes full resolution, NCHW
    out_bf = m(x)
    m.compute_dtype = m.segformer.compute_dtype = m.deeplabv3plus.compute_dtype = None
    out_32 = m(x)
    for k in ("segformer_seg", "deeplabv3plus_seg", "segmentation"):
        a, b = out_bf[k], out_32[k]
        assert a.shape == (1, 19, 1024, 2048) and torch.isfinite(a).all()
        err, mag = (a - b).abs().max().item(), b.abs().max().item()
        agree = (a.argmax(1) == b.argmax(1)).float().mean().item()
        print(f"C5 b5+r101 bf16 vs float32-grade path at 2048x1024, {k}: max abs err {err:.3e} at logit magnitude {mag:.3g} "
              f"(relative {err / mag:.3e}), argmax agreement {agree:.4f}")
        assert err < C5_REL_TOL * mag and agree > C5_AGREE and not torch.equal(a, b), k
    for k in ("depth", "segformer_depth", "deeplabv3plus_depth"):
        err = (out_bf[k] - out_32[k]).abs().max().item()
        print(f"C5 {k}: max abs diff {err:.3e} (sigmoid outputs in [0, 1])")
        assert err < C5_REL_TOL, k


def test_c4_trainer_step_at_1024x2048(P, tmp_path):
    """PKG/training/trainer.py:280-375 at BASELINE configs[3]'s resolution.  Dropout off (the HIP heads and the as-written
    graph consume the dropout stream differently), lr > 0 so the step really updates; the reference values are the same
    model BEFORE the step run through the as-written op graph (fused_train = False) on the same batch and density draws."""
    import copy
    import torch.nn.functional as F
    H, W = 1024, 2048
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")       # no solver benchmarking of the backward convolutions (bench.py does the same)
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(16)
    model = _calibrate(P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False))
    for mod in model.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.p = 0.0
    g = torch.Generator().manual_seed(4)
    NB = 2
    batch = {"image": torch.randn(NB, 3, H, W, generator=g), "label": torch.randint(0, 19, (NB, H, W), generator=g).to(torch.uint8),
             "weather_condition": ["fog", "rain"], "depth": torch.rand(NB, H, W, generator=g), "dataset": ["synthetic"] * NB}
    config = {"epochs": 1, "optimizer": {"type": "sgd", "learning_rate": 1e-3, "momentum": 0.0, "weight_decay": 0.0},
              "loss": {"type": "fog_density_aware"}, "density_rng": "torch", "grad_clip": 1.0}
    ref_model = copy.deepcopy(model).cuda().train()
    ref_model.segformer.fused_train = False
    t = P.AdverseWeatherTrainer(model, [batch], None, config, torch.device("cuda"), checkpoint_dir=str(tmp_path / "ck"), log_dir=str(tmp_path / "lg"))
    before = [p.detach().clone() for p in model.parameters()]
    torch.manual_seed(99)
    tm = t.train_epoch()
    grads_ok = all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    n_grads = sum(p.grad is not None for p in model.parameters())
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, model.parameters()))
    # the as-written graph on the same batch, density draws (torch.rand order of trainer.py:501-509) and weights
    torch.manual_seed(99)
    with torch.no_grad():
        img = batch["image"].cuda()
        dens = torch.stack([torch.rand(H, W) * 0.5 + 0.5, torch.rand(H, W) * 0.3 + 0.2]).cuda()   # 'fog': U(.5, 1), 'rain': U(.2, .5), in sample order
        out = ref_model(img)
        ce = F.cross_entropy(out["segmentation"], batch["label"].cuda().long(), reduction="none")
        seg = (ce * (1.0 + 2.0 * dens)).mean().item()
        dl = F.mse_loss(out["depth"].squeeze(1), batch["depth"].cuda(), reduction="none").mean().item()
    ref = {"train_loss": seg + 0.1 * dl, "train_seg_loss": seg, "train_depth_loss": dl}
    print(f"C4 trainer step at {H}x{W}, batch {NB}: {tm}; as-written graph: {ref}; {n_grads} gradients, {moved} tensors moved, "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GB")
    assert tm["train_samples"] == NB and grads_ok and n_grads > 300 and moved > 300
    for k, r in ref.items():
        assert abs(tm[k] - r) <= 1e-4, (k, tm[k], r)                        # north_star: 1e-4 abs on the loss


def _bench_line(*argv, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "AWSEG_DIST_BACKEND")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_c3_two_gpus_over_rccl_reproduce_the_single_gpu_miou(native):
    """BASELINE configs[2] on the smallest multi-GPU lease: two ranks, one device each, counters all-reduced over RCCL."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one-GPU lease: RCCL needs one device per rank (the 2-rank gloo tests cover the sharding logic)")
    common = ["--steps", "2", "--warmup", "1", "--frames", "20", "--batch", "2", "--no-cpu-baseline", "--fp32-steps", "0", "--kernel-steps", "0"]
    one = _bench_line("--gpus", "1", *common)
    two = _bench_line("--gpus", "2", *common)
    print("N=1:", one["value"], one["miou"]); print("N=2:", two["value"], two["miou"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["config"]["dist_backend"] == "nccl"
    assert one["miou"] == two["miou"], "pooled mIoU dict differs between 1 and 2 GPUs"


def test_c4_two_ranks_on_one_gpu_over_gloo_keep_the_replicas_identical(native):
    """The same check on a one-GPU lease: two ranks share the device, gradients averaged over gloo (bench.py falls back to it when
    there are more ranks than devices) — the trainer's bucketed all-reduce path end to end, minus RCCL itself."""
    two = _bench_line("--gpus", "2", "--mode", "train", "--steps", "1", "--warmup", "1", "--batch", "2", "--height", "128", "--width", "256",
                      "--no-cpu-baseline", "--kernel-steps", "0")
    print("N=2 train (gloo, one GPU):", two["value"], two["losses"], two["replica_parameter_sums"])
    assert two["n_gpus"] == 2 and two["config"]["dist_backend"] == "gloo" and two["rccl_ranks"] == 0
    assert two["replicas_identical"] and len(two["replica_parameter_sums"]) == 2
    assert all(v == v for v in two["losses"].values())


def test_c4_two_gpus_over_rccl_keep_the_replicas_identical(native):
    """BASELINE configs[3] on the smallest multi-GPU lease: two ranks, one device each, AdverseWeatherTrainer steps with the bucketed
    gradient all-reduce over RCCL — after the steps both replicas hold bit-identical parameters (the gloo twin of this check:
    tests/test_abi_and_host.py::test_two_rank_gloo_counter_allreduce_matches_single_process)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one-GPU lease: RCCL needs one device per rank (the 2-rank gloo test covers the gradient buckets)")
    two = _bench_line("--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "256", "--width", "512",
                      "--no-cpu-baseline", "--kernel-steps", "0")
    print("N=2 train:", two["value"], two["losses"], two["replica_parameter_sums"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["config"]["dist_backend"] == "nccl"
    assert two["replicas_identical"] and len(two["replica_parameter_sums"]) == 2
    assert all(v == v for v in two["losses"].values())                    # finite losses


@pytest.mark.parametrize("env", [{"AWSEG_WINO8": "0"}, {"AWSEG_WINO8": "1"}, {"AWSEG_WINO8": "2"}, {"AWSEG_WINO8_TPB": "3"}, {"AWSEG_WINO8_TPB": "64"}, {"AWSEG_GEMM_SPLIT_V3": "0"}, {"AWSEG_G3_STAGGER": "0"}, {"AWSEG_G3_HALF": "0"}, {"AWSEG_G3_HALF": "2"}, {"AWSEG_G3_HALF": "2", "AWSEG_G3_THREE": "0"},
                                 {"AWSEG_ASPP_LDS": "0"}, {"AWSEG_ASPP_LDS": "0", "AWSEG_ASPP_ROWS": "0"}, {"AWSEG_STATS_WIDE": "0"}])
def test_round2_kernels_stay_selectable_and_correct(env):
    """The earlier kernels (four-wave, alternating-role and non-persistent symmetric Winograd; register-staged split GEMM; gemm_split3 without the staggered
    DMA issue, with 256-row tiles only and with 128-row tiles (two blocks per CU) on every shape; the ASPP depthwise walk without LDS staging, one class or all classes per lane; the one-pass statistics on 256 threads x 4 pixels) remain behind environment switches, and AWSEG_WINO8_TPB forces the persistent Winograd blocks onto the small test maps (3 tiles a block: ragged last blocks; 64: blocks that own the whole map) for A/B measurements (tools/ab_kernel.sh): their own parity tests run in a
    child process with the switch set (the launchers read it once per process)."""
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_kernels.py"), str(ROOT / "tests" / "test_gpu_models.py"), "-q", "-x", "-m", "gpu", "-k",
                        "winograd_split or gemm_split_float32_grade or conv_gemm_split_equals or aspp or eval_stats or confusion_stats"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600, cwd=str(ROOT))
    tail = r.stdout.decode()[-600:]
    assert r.returncode == 0 and " passed" in tail, tail
