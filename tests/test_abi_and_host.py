"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/awseg.h
declares (no compute calls without a GPU), host-side logic, and the multi-rank reduction on gloo."""
import ctypes
import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
PKG = "adverse_weather_semantic_segmentation_robustness_benchmark_amd"


@pytest.fixture(scope="module")
def built_lib():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.csrc import build
    return build.build()


def header_symbols():
    text = (ROOT / "include" / "awseg.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awseg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(str(built_lib))
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/awseg.h but not exported"
    lib.awseg_abi_version.restype = ctypes.c_int
    assert lib.awseg_abi_version() == 1
    lib.awseg_error_string.restype = ctypes.c_char_p
    assert b"invalid argument" in lib.awseg_error_string(-1)


def test_python_binding_covers_header():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native
    assert sorted(_native.SIGNATURES) == header_symbols()
    _native.lib()                                          # binds every symbol; raises on drift


def test_library_built_from_another_header_is_refused(monkeypatch):
    """The library carries 60 bits of sha256(include/awseg.h) from its build; the binding refuses a library whose hash is not the
    hash of the header beside it (a stale .so next to newer sources would be called with the wrong arguments otherwise)."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native as N
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.csrc import build
    assert N.lib().awseg_header_hash() == build.header_hash() != 0
    monkeypatch.setattr(build, "header_hash", lambda: 12345)
    monkeypatch.setattr(N, "_lib", None)
    with pytest.raises(N.AwsegError, match="ABI drift"):
        N.lib()
    monkeypatch.undo()
    N.lib()


def test_job_struct_layouts_match_header():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native as N
    assert N.FOG_JOB.fields["beta"][1] == 8 and N.FOG_JOB.fields["seed"][1] == 24
    assert N.NIGHT_JOB.fields["intensity"][1] == 16 and N.PRIM_JOB.fields["intensity"][1] == 16


def test_no_cpu_fallback_for_host_tensors():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native as N, ops
    with pytest.raises(N.AwsegError, match="no CPU fallback"):
        ops.argmax(torch.zeros(1, 19, 4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(N.AwsegError, match="is missing"):
        N.lib()


def test_product_never_imports_oracle():
    for path in (ROOT / PKG).rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle", path.read_text(), flags=re.M), f"{path} imports the oracle"


def test_config_semantics(tmp_path, monkeypatch):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils import config as C
    cfg = C.create_default_config()
    assert cfg.get("model.num_classes") == 19 and cfg.get("data.image_size") == [512, 1024]
    assert cfg.get("nope.deeper", 7) == 7 and "model.type" in cfg and "model.zzz" not in cfg
    cfg.set("a.b.c", 3); assert cfg["a.b.c"] == 3
    cfg.update({"model": {"num_classes": 5}}); assert cfg.get("model.num_classes") == 5 and cfg.get("model.type") == "ensemble"
    C.validate_config(cfg)
    bad = C.create_default_config(); bad.set("training.batch_size", 0)
    with pytest.raises(ValueError):
        C.validate_config(bad)
    p = tmp_path / "c.yaml"
    C.save_config(cfg, p)
    monkeypatch.setenv("CONFIG_TRAINING__BATCH_SIZE", "16")
    monkeypatch.setenv("CONFIG_MLFLOW__ENABLED", "false")
    monkeypatch.setenv("CONFIG_OPTIMIZER__LEARNING_RATE", "0.5")
    loaded = C.load_config(p)
    assert loaded.get("training.batch_size") == 16 and loaded.get("mlflow.enabled") is False
    assert loaded.get("optimizer.learning_rate") == 0.5
    with pytest.raises(FileNotFoundError):
        C.load_config(tmp_path / "missing.yaml")
    assert C.get_device_config("cpu") == "cpu" and C.get_device_config("auto") in ("cpu", "cuda")


def test_iou_from_counts_matches_reference_goldens(golden_metrics):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import iou_from_counts, RobustnessMetrics
    g = golden_metrics
    for n in range(int(g["n_cases"])):
        res = iou_from_counts(torch.from_numpy(g[f"counts{n}"]), 19)
        if np.isnan(g[f"miou{n}"]):
            assert np.isnan(res["mean_iou"])
        else:
            assert res["mean_iou"] == float(g[f"miou{n}"])
        assert np.array_equal(res["per_class_iou"], g[f"per_class{n}"])
    rm = RobustnessMetrics()
    assert [rm.compute_robustness_degradation_ratio(a, b) for a, b in g["deg_pairs"]] == list(g["deg_ratio"])
    s = rm.create_robustness_summary({"clean": {"mean_iou": 0.5}, "fog": {"mean_iou": 0.4}, "rain": {"mean_iou": 0.6}})
    assert abs(s["robustness_degradation_fog"] - 0.2) < 1e-12 and s["robustness_degradation_rain"] == 0.0


def test_ece_from_bins_matches_reference(oracle, golden_metrics):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import ConfidenceCalibration
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.ops import ECE_BIN_DTYPE
    g = golden_metrics
    cnt, sconf, scorr = oracle.ece_bins(g["ece_logits"], g["ece_label"])
    b = np.zeros(15, dtype=ECE_BIN_DTYPE)
    b["count"], b["sum_conf"], b["sum_correct"] = cnt, sconf, scorr
    assert abs(ConfidenceCalibration.ece_from_bins(b) - float(g["ece"])) < 1e-6


def test_weather_draw_order_replays_reference(golden_weather):
    """Host-side draws of the product follow the reference's RNG call order exactly."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data import preprocessing as P
    g = golden_weather
    for k in range(int(g["n_cases"])):
        h, w, seed, inten = g[f"case{k}"]
        h, w = int(h), int(w)
        inten = None if inten < 0 else float(inten)
        np.random.seed(int(seed))
        noise, i_used = P.draw_fog(h, w, inten)
        assert np.array_equal(noise, g[f"fog_noise{k}"]) and i_used == float(g[f"fog_intensity{k}"])
        np.random.seed(int(seed))
        i_used, bf, nz = P.draw_night(h, w, inten)
        assert i_used == float(g[f"night_intensity{k}"]) and bf == float(g[f"night_brightness{k}"])
        assert np.array_equal(nz, g[f"night_noise{k}"])


def test_rain_snow_draws_match_oracle_draws(oracle):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data import preprocessing as P
    np.random.seed(9); a = P.draw_rain(40, 72)
    np.random.seed(9); b = oracle.draw_rain(40, 72)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert set(np.unique(a[1][:, 4])) <= {1, 3}                # thickness is drawn from {1, 3}
    np.random.seed(9); a = P.draw_snow(40, 72)
    np.random.seed(9); b = oracle.draw_snow(40, 72)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[2] in (3, 7)


def test_fog_jobs_parameters():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
    j = ops.fog_jobs([3], [0.5])
    assert j[0]["image"] == 3 and j[0]["beta"] == 0.005 + 0.5 * (0.05 - 0.005) and j[0]["atmos"] == 0.7 + 0.5 * (1.0 - 0.7)
    assert np.array_equal(ops.gaussian_taps(), __import__("oracle.cpu_oracle", fromlist=["x"]).gaussian_taps())


def test_early_stopping_and_trainer_setup(tmp_path):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.training.trainer import EarlyStopping, AdverseWeatherTrainer
    es = EarlyStopping(patience=2, min_delta=0.01)
    m = torch.nn.Linear(2, 2)
    assert not es(1.0, m) and not es(0.995, m) and es(0.999, m)      # two non-improvements -> stop
    assert es.early_stop and es.best_loss == 1.0
    for kind, cls in (("adamw", torch.optim.AdamW), ("sgd", torch.optim.SGD), ("other", torch.optim.Adam)):
        for sched in ("cosine", "step", "plateau"):
            t = AdverseWeatherTrainer(torch.nn.Conv2d(3, 19, 1), [], [], {"optimizer": {"type": kind}, "scheduler": {"enabled": True, "type": sched},
                                                                         "loss": {"type": "fog_density_aware"}}, torch.device("cpu"),
                                      checkpoint_dir=str(tmp_path / "c"), log_dir=str(tmp_path / "l"))
            assert isinstance(t.optimizer, cls) and t.scheduler is not None and t.current_epoch == 0
            assert type(t.loss_fn).__name__ == "FogDensityAwareLoss" and t.metrics.num_classes == 19
    t = AdverseWeatherTrainer(torch.nn.Conv2d(3, 19, 1), [], [], {"loss": {"type": "cross_entropy"}}, torch.device("cpu"),
                              checkpoint_dir=str(tmp_path / "c"), log_dir=str(tmp_path / "l"))
    assert isinstance(t.loss_fn, torch.nn.CrossEntropyLoss) and t.scheduler is None
    assert t._estimate_fog_density({"image": torch.zeros(1, 3, 4, 4)}) is None      # trainer.py:490-492


def test_model_state_dict_prefixes():
    import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
    m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False)
    keys = set(m.state_dict())
    for k in ("ensemble_weights", "temperature", "segformer.segmentation_head.0.weight", "segformer.segmentation_head.4.bias",
              "segformer.depth_head.depth_head.7.weight", "deeplabv3plus.model.encoder.layer4.2.conv3.weight",
              "deeplabv3plus.model.decoder.aspp.0.convs.1.0.0.weight", "deeplabv3plus.model.decoder.aspp.0.project.0.weight",
              "deeplabv3plus.model.decoder.block2.0.1.weight", "deeplabv3plus.model.segmentation_head.0.bias",
              "deeplabv3plus.depth_head.depth_head.0.weight"):
        assert k in keys, k
    assert any(k.startswith("segformer.segformer.") for k in keys)
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 36.0) < 0.5            # ~36 M parameters
    m2 = P.EnsembleModel(temperature_scaling=False, include_depth=False, pretrained=False)
    assert "temperature" not in m2.state_dict() and not hasattr(m2.segformer, "depth_head")
    assert hasattr(m, "segformer") and hasattr(m, "deeplabv3plus")                    # evaluate.py:194 probes these


def test_training_mode_graph_matches_reference_contract():
    """Training-mode forward = the reference's op graph on torch (autograd-capable): keys + shapes."""
    import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
    torch.manual_seed(0)
    m = P.EnsembleModel(num_classes=5, include_depth=True, pretrained=False).train()
    out = m(torch.randn(2, 3, 64, 96))
    assert set(out) == {"segmentation", "segformer_seg", "deeplabv3plus_seg", "depth", "segformer_depth", "deeplabv3plus_depth"}
    assert out["segmentation"].shape == (2, 5, 64, 96) and out["depth"].shape == (2, 1, 64, 96)
    assert 0.0 <= out["depth"].min().item() and out["depth"].max().item() <= 1.0
    out["segmentation"].mean().backward()
    assert m.ensemble_weights.grad is not None and m.temperature.grad is not None


def test_training_epochs_redraw_their_samples_and_validation_does_not():
    """the reference redraws pixels, the weather choice and every corruption parameter on each __getitem__
    (PKG/data/loader.py:206, 231, 265); here the draws are keyed by (seed, split, EPOCH, index) — fresh per training
    epoch, rank-independent, frozen for val / test.  Host-side keys only (no GPU)."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, _Loader
    tr = CityscapesKITTIDataset(split="train", image_size=(8, 8), num_samples=40, device="cpu", include_depth=False)
    va = CityscapesKITTIDataset(split="val", image_size=(8, 8), num_samples=40, device="cpu", include_depth=False)
    k0, c0, f0 = [tr._sample_key(i) for i in range(40)], tr.choose_conditions(0, 40), tr.weather_transforms._frame_seed
    tr.set_epoch(1)
    k1, c1, f1 = [tr._sample_key(i) for i in range(40)], tr.choose_conditions(0, 40), tr.weather_transforms._frame_seed
    assert not set(k0) & set(k1) and c0 != c1 and f0 != f1
    a, _ = tr.synth_raw(0, 2)
    tr.set_epoch(0)
    b, _ = tr.synth_raw(0, 2)
    assert not torch.equal(a, b) and [tr._sample_key(i) for i in range(40)] == k0          # a pure function of the epoch
    v0 = [va._sample_key(i) for i in range(40)]
    va.set_epoch(3)
    assert [va._sample_key(i) for i in range(40)] == v0 and va.weather_transforms._frame_seed == va.seed
    # the loader advances the epoch once per pass, identically on every rank

    class Probe(CityscapesKITTIDataset):
        def make_batch(self, start, n, raw=None):
            return (self.epoch, start, n)
    ds = Probe(split="train", image_size=(8, 8), num_samples=6, device="cpu", include_depth=False)
    ld = _Loader(ds, 2, False, 0, 1)
    assert [list(ld), list(ld)] == [[(0, 0, 2), (0, 2, 2), (0, 4, 2)], [(1, 0, 2), (1, 2, 2), (1, 4, 2)]]


def test_shard_range_covers_everything():
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.parallel import shard_range
    for n in (0, 1, 7, 20, 160):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(n, r, world)]
            assert got == list(range(n))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import parallel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import iou_from_counts
from oracle import cpu_oracle as O
rank, local, world = parallel.init_from_env(backend="gloo")
C, N = 19, 10
rs = np.random.RandomState(0)
pred = rs.randint(0, C, (N, 16, 24)); lab = rs.randint(0, C, (N, 16, 24)).astype(np.uint8)
cond = np.arange(N) % 5
mine = list(parallel.shard_range(N, rank, world))
counts = torch.zeros(6, C * C, dtype=torch.int64)
for i in mine:                                   # per-rank accumulation (the oracle stands in for the HIP kernel on CPU)
    c = torch.from_numpy(O.confusion(pred[i], lab[i], C))
    counts[0] += c; counts[1 + cond[i]] += c
sums = torch.tensor([float(len(mine)), 1.5 * len(mine)], dtype=torch.float64)
parallel.all_reduce_sum_([counts, sums])
full = torch.from_numpy(O.confusion(pred, lab, C))
assert torch.equal(counts[0], full), "pooled confusion differs from single-process"
for k in range(5):
    assert torch.equal(counts[1 + k], torch.from_numpy(O.confusion(pred[cond == k], lab[cond == k], C)))
assert iou_from_counts(counts[0], C)["mean_iou"] == O.iou_from_counts(full.numpy(), C)["mean_iou"]
assert sums.tolist() == [float(N), 1.5 * N]
# gradient buckets average gradients across ranks
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 2))
for p in net.parameters():
    p.grad = torch.full_like(p, float(rank + 1))
parallel.GradientBuckets(list(net.parameters()), bucket_mb=0.0001).all_reduce_()
for p in net.parameters():
    assert torch.allclose(p.grad, torch.full_like(p, (1 + world) / 2.0))
# --- the overlapped path: identical replicas after broadcast, hooks fire per bucket during backward, one rank lacks a
# gradient for a parameter (control flow differs per rank): message sizes stay fixed, nothing hangs, and after one
# optimizer step the state_dicts are equal on every rank
torch.manual_seed(100 + rank)                                   # DIFFERENT initial weights per rank on purpose
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16); self.b = torch.nn.Linear(16, 4); self.unused = torch.nn.Linear(8, 3)
        self.dead = torch.nn.Linear(8, 3)                        # takes part in NO rank's graph
        self.bn = torch.nn.BatchNorm1d(4)
    def forward(self, x, use_extra):
        y = self.bn(self.b(torch.relu(self.a(x))))
        return y.sum() + (self.unused(x).sum() if use_extra else 0.0)
net = Net()
parallel.broadcast_module_(net)
ref0 = [t.clone() for t in net.state_dict().values()]
gathered = [None] * world
dist.all_gather_object(gathered, [t.tolist() for t in ref0])
assert gathered[0] == gathered[1], "broadcast_module_ left the replicas different"
buckets = parallel.GradientBuckets(list(net.parameters()), bucket_mb=0.0002)
assert len(buckets.buckets) >= 2
opt = torch.optim.SGD(net.parameters(), lr=0.1)
torch.manual_seed(7 + rank)
x = torch.randn(5, 8)
buckets.zero_grad()
net(x, use_extra=(rank == 0)).backward()                        # rank 1 never touches `unused`
buckets.finish()
# the averaged gradient equals the mean of the per-rank gradients computed without any bucket machinery
net2 = Net(); net2.load_state_dict({k: v for k, v in zip(net.state_dict().keys(), ref0)})
net2(x, use_extra=(rank == 0)).backward()
for (n1, p1), (n2, p2) in zip(net.named_parameters(), net2.named_parameters()):
    g = p2.grad.clone() if p2.grad is not None else torch.zeros_like(p2)
    dist.all_reduce(g); g /= world
    if n1.startswith("dead."):
        # no rank produced a gradient: .grad is None, as after the single-process optimizer.zero_grad() — the optimizer
        # skips the parameter (no weight decay / moment update) at any GPU count
        assert p1.grad is None, n1
        continue
    assert p1.grad is not None and torch.allclose(p1.grad, g, atol=1e-6), n1   # `unused`: fired on rank 0 only, averaged everywhere
dead_before = [p.clone() for p in net.dead.parameters()]
torch.optim.AdamW(net.parameters(), lr=0.1, weight_decay=0.5).step()          # would shrink `dead` if it carried a zero gradient
assert all(torch.equal(a, b) for a, b in zip(dead_before, net.dead.parameters()))
buckets.zero_grad()                                             # the next step re-attaches every view, `dead` included
assert all(p.grad is not None for p in net.parameters())
net(x, use_extra=(rank == 0)).backward()
buckets.finish()
opt.step()
after = [None] * world
dist.all_gather_object(after, [p.detach().tolist() for p in net.parameters()])
assert after[0] == after[1], "replicas diverged after one data-parallel step"
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_counter_allreduce_matches_single_process(tmp_path):
    """world_size 2 on gloo: sharded accumulation + one SUM all-reduce == single-process counts,
    bit-identical mIoU; bucketed gradient averaging."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()[-2000:]


def _bench(*argv, timeout=180):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                          env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")})


def test_bench_self_launches_its_ranks_and_shards_the_global_set():
    """`python bench.py --gpus N` with no launcher starts N rank processes itself; every frame of the
    fixed global sample set is owned by exactly one rank; a failing rank makes the command fail."""
    import json
    for n in (1, 2, 3):
        r = _bench("--gpus", str(n), "--dry-run")
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == n and line["frames"] == 160 and line["frames_owned_once"] is True
        assert line["frames_rank0"] == -(-160 // n)
    r = _bench("--gpus", "2", "--dry-run", "--fail-rank", "1")
    assert r.returncode == 3 and b"rank 1 exited" in r.stderr


def test_checkpoint_key_translation_between_transformers_spellings():
    """`segformer.segformer.*` keys are transformers-version dependent; a checkpoint in
    the legacy spelling (the releases REF/requirements.txt pins) must load into the live model."""
    import torch
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils import checkpoint as K
    cur = ["segformer.segformer.stages.0.patch_embeddings.proj.weight", "segformer.segformer.stages.1.patch_embeddings.layer_norm.bias",
           "segformer.segformer.stages.2.blocks.1.layernorm_before.weight", "segformer.segformer.stages.0.blocks.0.attention.q_proj.bias",
           "segformer.segformer.stages.0.blocks.0.attention.k_proj.weight", "segformer.segformer.stages.0.blocks.0.attention.v_proj.weight",
           "segformer.segformer.stages.0.blocks.0.attention.o_proj.weight",
           "segformer.segformer.stages.0.blocks.0.attention.sequence_reduction.sequence_reduction.weight",
           "segformer.segformer.stages.0.blocks.0.attention.sequence_reduction.layer_norm.weight",
           "segformer.segformer.stages.3.blocks.1.layernorm_after.bias", "segformer.segformer.stages.3.blocks.1.mlp.fc1.weight",
           "segformer.segformer.stages.3.blocks.1.mlp.dwconv.dwconv.weight", "segformer.segformer.stages.3.blocks.1.mlp.fc2.bias",
           "segformer.segformer.stages.3.layer_norm.weight", "segformer.segmentation_head.0.weight", "ensemble_weights", "temperature"]
    legacy = ["segformer.segformer.encoder.patch_embeddings.0.proj.weight", "segformer.segformer.encoder.patch_embeddings.1.layer_norm.bias",
              "segformer.segformer.encoder.block.2.1.layer_norm_1.weight", "segformer.segformer.encoder.block.0.0.attention.self.query.bias",
              "segformer.segformer.encoder.block.0.0.attention.self.key.weight", "segformer.segformer.encoder.block.0.0.attention.self.value.weight",
              "segformer.segformer.encoder.block.0.0.attention.output.dense.weight", "segformer.segformer.encoder.block.0.0.attention.self.sr.weight",
              "segformer.segformer.encoder.block.0.0.attention.self.layer_norm.weight", "segformer.segformer.encoder.block.3.1.layer_norm_2.bias",
              "segformer.segformer.encoder.block.3.1.mlp.dense1.weight", "segformer.segformer.encoder.block.3.1.mlp.dwconv.dwconv.weight",
              "segformer.segformer.encoder.block.3.1.mlp.dense2.bias", "segformer.segformer.encoder.layer_norm.3.weight",
              "segformer.segmentation_head.0.weight", "ensemble_weights", "temperature"]
    sd_legacy = {k: torch.full((1,), float(i)) for i, k in enumerate(legacy)}
    got = K.remap_segformer_keys(sd_legacy, cur)
    assert list(got.keys()) == cur and all(got[k].item() == i for i, k in enumerate(cur))
    back = K.remap_segformer_keys(got, legacy)                         # and the other direction
    assert list(back.keys()) == legacy
    same = K.remap_segformer_keys(got, cur)                            # already matching: untouched
    assert list(same.keys()) == cur
    stray = K.remap_segformer_keys({"foo.encoder.block.0.0.unknown.weight": torch.zeros(1)}, cur)
    assert list(stray.keys()) == ["foo.encoder.block.0.0.unknown.weight"]   # unknown entries pass through (strict load reports them)


def test_checkpoint_loads_legacy_spelling_into_live_segformer():
    import torch
    from transformers import SegformerConfig, SegformerModel
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils import checkpoint as K
    cfg = SegformerConfig(num_encoder_blocks=4, depths=[1, 1, 1, 1], sr_ratios=[8, 4, 2, 1], hidden_sizes=[8, 16, 24, 32],
                          num_attention_heads=[1, 2, 3, 4])
    torch.manual_seed(0)
    a, b = SegformerModel(cfg), SegformerModel(cfg)
    other = [k.replace("stages.0.", "STAGE0.") for k in a.state_dict()]      # force the translation direction
    legacy_names = {}
    for k in a.state_dict():
        nk = K._current_to_legacy(k)
        assert nk is not None and K._legacy_to_current(nk) == k, k            # every MiT entry has both spellings
        legacy_names[nk] = a.state_dict()[k]
    K.load_model_state(b, {"model_state_dict": legacy_names})
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k])
    del other


def test_evaluation_report_files(tmp_path):
    """REF/scripts/evaluate.py:277-392: json + markdown with the reference's sections and formats."""
    import json
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.report import generate_evaluation_report
    res = {"overall_miou": 0.5, "miou_clean": 0.8, "miou_fog": 0.6, "miou_night": 0.4, "robustness_degradation_fog": 0.25,
           "robustness_degradation_ratio": 0.2, "expected_calibration_error": 0.04, "ensemble_disagreement_auroc": 0.9}
    generate_evaluation_report(res, tmp_path / "out")
    assert json.loads((tmp_path / "out" / "evaluation_results.json").read_text()) == res
    md = (tmp_path / "out" / "evaluation_report.md").read_text().splitlines()
    assert md[0] == "# Adverse Weather Semantic Segmentation Evaluation Report"
    assert "| miou_clean | 0.780 | 0.800 | ✓ |" in md and "| miou_fog | 0.650 | 0.600 | ✗ |" in md
    assert "| miou_rain | 0.620 | 0.000 | ✗ |" in md                          # missing metric counts as 0.0
    assert "| expected_calibration_error | 0.050 | 0.040 | ✗ |" in md          # the reference marks with >= for every row
    assert "- **Fog**: mIoU = 0.600" in md and "- **Night**: mIoU = 0.400" in md and not any("Rain**: mIoU" in l for l in md)
    assert "- **Overall Degradation Ratio**: 0.200" in md and "- **Fog Degradation**: 0.250" in md
    assert "- **Expected Calibration Error**: 0.040" in md and "- **Disagreement AUROC**: 0.900" in md
