"""Checkpoint format of the reference (SURVEY §8(f) #3, §8(b) "state_dict / checkpoint").

Layout (PKG/training/trainer.py:620-627): {'epoch', 'model_state_dict', 'optimizer_state_dict',
'scheduler_state_dict', 'metrics', 'config'} saved as latest.pth / best.pth / epoch_N.pth and read
back by REF/scripts/evaluate.py:80-81 and trainer.py:651-659.

The `segformer.segformer.*` part of the state_dict is named by whatever `transformers` wrote the
checkpoint.  Releases the reference pins (>=4.30, REF/requirements.txt:20) call the MiT encoder
`encoder.patch_embeddings.{s}`, `encoder.block.{s}.{b}.attention.self.query`, ...; the release in
this image calls it `stages.{s}.patch_embeddings`, `stages.{s}.blocks.{b}.attention.q_proj`, ....
`remap_segformer_keys` translates either spelling into the one the live model uses, so a
checkpoint written by the reference stack loads here and vice versa.  Tensors are untouched.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, Mapping

import torch

# (legacy spelling, current spelling) of one MiT block's sub-modules
_BLOCK_PARTS = [
    ("layer_norm_1", "layernorm_before"),
    ("attention.self.query", "attention.q_proj"),
    ("attention.self.key", "attention.k_proj"),
    ("attention.self.value", "attention.v_proj"),
    ("attention.self.sr", "attention.sequence_reduction.sequence_reduction"),
    ("attention.self.layer_norm", "attention.sequence_reduction.layer_norm"),
    ("attention.output.dense", "attention.o_proj"),
    ("layer_norm_2", "layernorm_after"),
    ("mlp.dense1", "mlp.fc1"),
    ("mlp.dwconv.dwconv", "mlp.dwconv.dwconv"),
    ("mlp.dense2", "mlp.fc2"),
]
_LEGACY_RE = re.compile(r"^(?P<pre>.*?)encoder\.(?P<kind>patch_embeddings|block|layer_norm)\.(?P<s>\d+)\.(?P<rest>.+)$")
_CURRENT_RE = re.compile(r"^(?P<pre>.*?)stages\.(?P<s>\d+)\.(?P<rest>.+)$")


def _legacy_to_current(key: str):
    m = _LEGACY_RE.match(key)
    if not m:
        return None
    pre, kind, s, rest = m["pre"], m["kind"], m["s"], m["rest"]
    if kind == "patch_embeddings":
        return f"{pre}stages.{s}.patch_embeddings.{rest}"
    if kind == "layer_norm":
        return f"{pre}stages.{s}.layer_norm.{rest}"
    b, _, tail = rest.partition(".")
    for old, new in _BLOCK_PARTS:
        if tail.startswith(old + "."):
            return f"{pre}stages.{s}.blocks.{b}.{new}{tail[len(old):]}"
    return None


def _current_to_legacy(key: str):
    m = _CURRENT_RE.match(key)
    if not m:
        return None
    pre, s, rest = m["pre"], m["s"], m["rest"]
    if rest.startswith("patch_embeddings."):
        return f"{pre}encoder.patch_embeddings.{s}.{rest[len('patch_embeddings.'):]}"
    if rest.startswith("layer_norm."):
        return f"{pre}encoder.layer_norm.{s}.{rest[len('layer_norm.'):]}"
    if rest.startswith("blocks."):
        b, _, tail = rest[len("blocks."):].partition(".")
        for old, new in _BLOCK_PARTS:
            if tail.startswith(new + "."):
                return f"{pre}encoder.block.{s}.{b}.{old}{tail[len(new):]}"
    return None


def remap_segformer_keys(state_dict: Mapping[str, torch.Tensor], target_keys: Iterable[str]) -> Dict[str, torch.Tensor]:
    """Rename MiT encoder entries of `state_dict` to the spelling found in `target_keys` (the live
    model's state_dict keys).  Keys that already match, and everything that is not a MiT encoder
    entry, pass through unchanged."""
    target = set(target_keys)
    out = {}
    for k, v in state_dict.items():
        if k in target:
            out[k] = v
            continue
        for conv in (_legacy_to_current, _current_to_legacy):
            nk = conv(k)
            if nk is not None and nk in target:
                out[nk] = v
                break
        else:
            out[k] = v
    return out


def load_model_state(model: torch.nn.Module, checkpoint: Mapping, strict: bool = True):
    """`model.load_state_dict(checkpoint['model_state_dict'])` (evaluate.py:80-81, trainer.py:653)
    with the transformers-version key translation applied first."""
    sd = checkpoint["model_state_dict"] if "model_state_dict" in checkpoint else checkpoint
    return model.load_state_dict(remap_segformer_keys(sd, model.state_dict().keys()), strict=strict)


def make_checkpoint(epoch: int, model: torch.nn.Module, optimizer, scheduler, metrics: Mapping[str, float], config) -> Dict:
    """The dict trainer.py:620-627 writes."""
    return {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
            "scheduler_state_dict": scheduler.state_dict() if scheduler else None, "metrics": dict(metrics), "config": config}
