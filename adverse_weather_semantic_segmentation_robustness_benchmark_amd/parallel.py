"""One-process-per-GPU data parallelism for the hot path.

Evaluation shards the image index range over ranks with NO data-path collective; the only
exchange is one SUM all-reduce of the int64 counters (6 x 19 x 19 confusion + ECE bins, ~20 KB —
latency-bound, xGMI topology irrelevant) at the end.  Integer sums are order-independent, so
the pooled mIoU is bit-identical at any GPU count.  Training all-reduces flattened gradient
buckets (RCCL ring/tree over the 7 xGMI links) — see `GradientBuckets`.
Backend "nccl" IS RCCL on ROCm; CPU tests use "gloo".
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world_size); initialises the default group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # AWSEG_DIST_BACKEND=gloo: rehearse the N>1 path where RCCL cannot run (CPU, or several ranks on ONE GPU)
            backend = os.environ.get("AWSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(n: int, rank: int, world: int) -> range:
    """Contiguous block of sample indices owned by `rank` (ceil-sized blocks, tail rank may be short)."""
    per = (n + world - 1) // world
    return range(min(rank * per, n), min((rank + 1) * per, n))


def all_reduce_sum_(tensors: Iterable[torch.Tensor]) -> None:
    """In-place SUM all-reduce of several small counter tensors as ONE message per dtype."""
    if not is_dist():
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def barrier() -> None:
    if is_dist():
        dist.barrier()


def broadcast_module_(module: torch.nn.Module, src: int = 0) -> None:
    """Every parameter and buffer of `module` from rank `src` to all ranks (one message per dtype).  Data parallelism
    averages GRADIENTS only, so the replicas must start identical; the backbones are random-init offline."""
    if not is_dist():
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for group in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src=src)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


class GradientBuckets:
    """Bucketed gradient averaging for the trainer (classic DP; BatchNorm stays per rank, as the
    reference has no SyncBN), overlapped with backward.

    The trainable parameters — a FIXED list, the same on every rank — are packed in reverse registration
    order (≈ the order backward produces them) into flat float32 buckets of ~`bucket_mb`; each parameter's
    `.grad` is a VIEW into its bucket, so autograd accumulates straight into the message buffer and nothing
    is packed or scattered.  A post-accumulate-grad hook per parameter counts its bucket down; the bucket
    whose last gradient has just landed is all-reduced asynchronously (RCCL's own stream) while backward
    continues with earlier layers.  `finish()` launches whatever has not fired (parameters that took no part
    in this step contribute zeros, so message sizes never depend on a rank's control flow), waits, scales by
    1/world, and detaches (.grad = None) the parameters no rank produced a gradient for.  With ~36 M fp32 parameters (144 MB) and 7 x 153 GB/s xGMI links, 25-50 MB buckets
    keep every link busy without serialising behind one giant message."""

    def __init__(self, params: List[torch.nn.Parameter], bucket_mb: float = 32.0) -> None:
        self.params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024 / 4)
        groups, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel()
            if size >= cap:
                groups.append(cur)
                cur, size = [], 0
        if cur:
            groups.append(cur)
        self.buckets = groups
        self.flat, self._bucket_of, self._views, self._view = [], {}, [], {}
        for bi, group in enumerate(groups):
            dt, dev = group[0].dtype, group[0].device
            flat = torch.zeros(sum(p.numel() for p in group), dtype=dt, device=dev)
            off = 0
            for p in group:
                n = p.numel()
                self._views.append((p, flat[off:off + n].view_as(p)))
                self._view[id(p)] = self._views[-1][1]
                self._bucket_of[id(p)] = bi
                off += n
            self.flat.append(flat)
        self._pending = [0] * len(groups)
        self._next = 0
        self._work = [None] * len(groups)
        self._armed = False
        self._fired = set()
        # host-side control messages (which parameters received a gradient on ANY rank) travel over gloo: a CPU tensor, so
        # reading the answer does not synchronise the GPU stream.  Created here — every rank constructs its buckets once.
        self._ctl = None
        if is_dist() and dist.get_backend() != "gloo":
            self._ctl = dist.new_group(backend="gloo")
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def zero_grad(self) -> None:
        """Zero every bucket and (re-)attach the parameters' .grad views; arms the hooks for one backward."""
        for flat in self.flat:
            flat.zero_()
        for p, view in self._views:
            p.grad = view
        self._pending = [len(g) for g in self.buckets]
        self._work = [None] * len(self.buckets)
        self._next = 0
        self._fired = set()
        self._armed = True

    def _launch(self, bi: int) -> None:
        if self._work[bi] is None and is_dist():
            self._work[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._armed:
            return
        bi = self._bucket_of[id(p)]
        view = self._view[id(p)]
        self._fired.add(id(p))
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
            # autograd replaced the view instead of accumulating into it: copy the gradient back into the bucket
            view.copy_(p.grad)
            p.grad = view
        self._pending[bi] -= 1
        # collectives must be issued in the SAME order on every rank: launch strictly by bucket index, so a bucket that
        # fills early waits for its predecessors (a rank whose control flow skipped a parameter launches that bucket,
        # and everything behind it, in finish())
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def finish(self) -> None:
        """All buckets reduced and averaged; call after backward, before clipping / the optimizer step."""
        self._armed = False
        if not is_dist():
            return
        world = dist.get_world_size()
        for bi in range(len(self.buckets)):
            self._launch(bi)                       # buckets with parameters that got no gradient this step
        for bi, work in enumerate(self._work):
            work.wait()
            self.flat[bi].div_(world)
        # A parameter NO rank produced a gradient for (a depth head on batches without depth targets, `ensemble_weights`
        # under the mean strategy) gets .grad = None, exactly what the single-process path's optimizer.zero_grad() leaves:
        # the optimizer then skips it (no weight decay, no moment update) on 1 and on N GPUs alike.  A parameter that fired
        # on SOME rank keeps its averaged gradient everywhere.
        fired = torch.tensor([1 if id(p) in self._fired else 0 for p in self.params], dtype=torch.int32)
        dist.all_reduce(fired, op=dist.ReduceOp.MAX, group=self._ctl)
        for p, f in zip(self.params, fired.tolist()):
            if not f:
                p.grad = None

    def all_reduce_(self) -> None:
        """Non-overlapped form for callers that ran backward without zero_grad(): gather whatever .grad holds (zeros
        for missing ones — fixed message sizes on every rank), reduce, scatter."""
        if not is_dist():
            return
        if not self._armed:
            self._fired = {id(p) for p, _ in self._views if p.grad is not None}
            for p, view in self._views:
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
            self._work = [None] * len(self.buckets)
        self.finish()
