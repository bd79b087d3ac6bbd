"""Data-parallel gradient exchange for the GPS step: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; graphs are independent
units, so the batch shards across ranks with ONE exchange per step: an averaging all-reduce of
the gradients (19.4 M fp32 = 77.7 MB for GPS-medium), here bucketed per GPS layer (~7.7 MB at
d = 384) and launched from autograd hooks as soon as a bucket's last gradient has been
accumulated, so the collective of layer l overlaps the backward of layers < l.

Design points (MI355X: 8 GPUs, xGMI point-to-point, 7 links x ~153 GB/s per GPU):
  * buckets are per layer, not fixed-size chunks: 7.7 MB is already past the latency knee of a
    direct reduce-scatter + all-gather over 7 links, and layer granularity is what gives
    backward overlap;
  * a bucket is packed by ONE multi-tensor copy when it completes (autograd keeps producing
    ordinary per-parameter gradients: no per-parameter accumulate kernels), reduced in place,
    and after ``finish()`` every ``p.grad`` IS a view of the reduced flat buffer (no unpack);
  * the average uses RCCL's native AVG reduction (one pass); gloo falls back to SUM + scale;
  * BatchNorm statistics stay per rank (standard DDP semantics; the reference has no SyncBN).
Works with ``backend='nccl'`` (= RCCL on ROCm) and with ``gloo`` (CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def broadcast_state(model: nn.Module, src: int = 0, process_group=None, bucket_bytes: int = 64 << 20) -> int:
    """Replica initialisation ("running BN stats / Performer projection buffers: broadcast from
    rank 0 at init"): every parameter AND every buffer of ``model`` is overwritten with rank ``src``'s values --
    BatchNorm ``running_mean / running_var / num_batches_tracked`` and, above all, the Performer's
    ``fast_attention.projection_matrix``, a RANDOM buffer drawn at construction
    (graphgps/layer/performer_layer.py:272-273): ranks seeded differently would otherwise train silently different
    models whose gradients are then averaged.  Tensors travel in flat per-dtype buckets of <= ``bucket_bytes`` (one
    broadcast each: a handful of collectives for 19.4 M parameters instead of ~700); in-place copies, so views held by
    an optimizer arena or a LinearGroup stack stay valid.  Returns the number of bytes broadcast (0 when no process
    group is initialised or the world has one rank).  Call once after construction / checkpoint loading, before the
    first step (``train.train_epoch`` and ``bench.py`` do)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) < 2:
        return 0
    return _broadcast_tensors(list(model.parameters()) + list(model.buffers()), src, process_group, bucket_bytes)


def broadcast_buffers(model: nn.Module, src: int = 0, process_group=None, bucket_bytes: int = 64 << 20) -> int:
    """Buffers only (BatchNorm running statistics, ``num_batches_tracked``, the Performer projections), from rank ``src``.
    torch's DDP re-broadcasts buffers on EVERY forward (``broadcast_buffers=True``); here the training step keeps them per
    rank (the step is a replayed hipGraph: no collective inside it) and they are brought back in line where a reader
    outside the step looks at them -- before an evaluation pass and before a checkpoint is written
    (``train.eval_epoch`` calls this).  Returns the bytes broadcast (0 without a process group)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) < 2:
        return 0
    return _broadcast_tensors(list(model.buffers()), src, process_group, bucket_bytes)


def _broadcast_tensors(all_tensors, src, process_group, bucket_bytes) -> int:
    tensors, seen = [], set()
    for t in all_tensors:
        key = (t.data_ptr(), t.numel(), t.dtype)
        if t.numel() == 0 or key in seen:           # tied / aliased storage travels once
            continue
        seen.add(key)
        tensors.append(t.data)
    total = 0
    by_dtype: Dict[tuple, List[torch.Tensor]] = {}      # one flat bucket per (dtype, device): a CPU buffer next to
    for t in tensors:                                    # device parameters must not meet them in one torch.cat
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for _, ts in by_dtype.items():
            i = 0
            while i < len(ts):
                chunk, nbytes = [], 0
                while i < len(ts) and (not chunk or nbytes + ts[i].numel() * ts[i].element_size() <= bucket_bytes):
                    chunk.append(ts[i])
                    nbytes += ts[i].numel() * ts[i].element_size()
                    i += 1
                flat = torch.cat([t.reshape(-1) for t in chunk])
                dist.broadcast(flat, src=src, group=process_group)
                off = 0
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
                total += nbytes
    return total


class _Bucket:
    __slots__ = ("name", "params", "flat", "views", "pending", "work")

    def __init__(self, name: str, params: List[nn.Parameter]):
        self.name, self.params = name, params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.work = None


def default_buckets(model: nn.Module) -> Dict[str, List[nn.Parameter]]:
    """One bucket per ``layers.<i>`` GPS layer, one for everything else (encoder + head)."""
    groups: Dict[str, List[nn.Parameter]] = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        parts = name.split(".")
        key = ".".join(parts[:2]) if parts[0] == "layers" and len(parts) > 2 else "_rest"
        groups.setdefault(key, []).append(p)
    return groups


class GradBucketReducer:
    """Averages gradients across ranks with per-bucket async all-reduce overlapped with backward.

    Usage per step::

        reducer.zero_grad()          # p.grad = None (no memsets)
        loss.backward()              # hooks pack + launch all-reduces as buckets complete
        reducer.finish()             # wait; p.grad now views the rank-averaged flat buffers
    """

    def __init__(self, model: nn.Module, process_group=None,
                 buckets: Optional[Dict[str, List[nn.Parameter]]] = None,
                 force_collective: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collective: issue the all-reduces even on a single rank (plumbing validation)
        self.active = self.world > 1 or (force_collective and dist.is_initialized())
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self.native_avg = backend == "nccl"
        self.buckets = [_Bucket(k, v) for k, v in (buckets or default_buckets(model)).items()]
        self._owner = {}
        for b in self.buckets:
            for p in b.params:
                self._owner[p] = b
                p.register_post_accumulate_grad_hook(self._hook)

    @property
    def num_bytes(self) -> int:
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)

    def _launch(self, b: _Bucket) -> None:
        if b.work is not None:
            return
        # pack: one multi-tensor copy (parameters that got no gradient contribute zeros)
        if b.flat.is_cuda:   # weight gradients are produced on the side stream (fused.py)
            from .fused import join_side_stream
            join_side_stream(b.flat.device)
        srcs = [p.grad if p.grad is not None else torch.zeros_like(v)
                for p, v in zip(b.params, b.views)]
        torch._foreach_copy_(b.views, srcs)
        if self.active:
            op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
            b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
        else:
            b.work = True

    def _hook(self, p: nn.Parameter) -> None:
        b = self._owner[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def zero_grad(self) -> None:
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            for p in b.params:
                p.grad = None

    def finish(self) -> None:
        for b in self.buckets:
            self._launch(b)      # buckets with parameters that received no gradient this step
        for b in self.buckets:
            if self.active:
                b.work.wait()
                if not self.native_avg and self.world > 1:
                    b.flat.mul_(1.0 / self.world)
            for p, v in zip(b.params, b.views):
                p.grad = v


class FlatGradExchange:
    """The data-parallel exchange on a flat gradient arena (optim.ParamArena): averaging all-reduces of ``arena.flat_g``,
    issued OUTSIDE autograd so that the compute either side of them replays from hipGraphs (the
    reference has no distributed code -- graphgps/train/custom_train.py:16-47 is the single-process step this wraps).

    Two forms, both driven by ``train.TrainStep``:
      * ``all_reduce()``: ONE collective over the whole arena (77.7 MB for GPS-medium) between [fwd + bwd + pack] and
        [clip + AdamW];
      * ``start(lo, hi)`` / ``finish(handles)``: the arena in RANGES, each started as soon as its half of the backward has
        produced it (``TrainStep(backward_split=...)``: the upper layers' range is in flight -- on the process group's own
        stream for RCCL -- while the lower layers are still being differentiated; only the last range is exposed).
    Arithmetic of the exposed time at N = 8 for the 8.3 ms PCQM4M step: DESIGN.md section 6.
    ``GradBucketReducer`` (above) remains the hook-driven variant for eager steps of arbitrary models."""

    def __init__(self, arena, process_group=None, force_collective: bool = False):
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collective and dist.is_initialized())
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self.native_avg = backend == "nccl"

    @property
    def num_bytes(self) -> int:
        return self.arena.num_bytes if self.active else 0

    def all_reduce(self) -> None:
        """Average ``arena.flat_g`` over the ranks, in place, on the current stream."""
        if not self.active:
            return
        self.finish([self.start(0, None)])

    def start(self, lo: int = 0, hi: Optional[int] = None):
        """Begin averaging ``arena.flat_g[lo:hi]`` (element offsets) over the ranks; returns a handle for ``finish``.
        The collective is ordered behind the work already enqueued on the current stream and does not block it."""
        if not self.active:
            return None
        flat = self.arena.flat_g[lo:hi]
        if flat.numel() == 0:
            return None
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        return (dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat)

    def finish(self, handles) -> None:
        """The current stream waits for the started ranges (and scales them where the backend has no averaging op)."""
        for h in handles:
            if h is None:
                continue
            work, flat = h
            work.wait()
            if not self.native_avg and self.world > 1:
                flat.mul_(1.0 / self.world)
