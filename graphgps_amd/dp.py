"""Data-parallel gradient exchange for the GPS step: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; graphs are independent
units, so the batch shards across ranks with ONE exchange per step: a sum-all-reduce of the
gradients (19.4 M fp32 = 77.7 MB for GPS-medium), here bucketed per GPS layer (~7.7 MB at
d = 384) and launched from autograd hooks as soon as a bucket's last gradient has been
accumulated, so the collective of layer l overlaps the backward of layers < l.

Design points (MI355X: 8 GPUs, xGMI point-to-point, 7 links x ~153 GB/s per GPU):
  * gradients live IN the flat bucket buffers (each ``p.grad`` is a view), so no pack/unpack
    copies and exactly one collective per bucket;
  * buckets are per layer, not fixed-size chunks: 7.7 MB is already past the latency knee of a
    direct reduce-scatter + all-gather over 7 links, and layer granularity is what gives
    backward overlap;
  * BatchNorm statistics stay per rank (standard DDP semantics; the reference has no SyncBN).
Works with ``backend='nccl'`` (= RCCL on ROCm) and with ``gloo`` (CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class _Bucket:
    __slots__ = ("name", "params", "flat", "pending", "work")

    def __init__(self, name: str, params: List[nn.Parameter]):
        self.name, self.params = name, params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
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

        reducer.zero_grad()          # memset of the flat buffers
        loss.backward()              # hooks launch all-reduces as buckets complete
        reducer.finish()             # wait + 1/world scaling; grads are now rank-averaged
    """

    def __init__(self, model: nn.Module, process_group=None,
                 buckets: Optional[Dict[str, List[nn.Parameter]]] = None,
                 force_collective: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collective: issue the all-reduces even on a single rank (plumbing validation)
        self.active = self.world > 1 or (force_collective and dist.is_initialized())
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
        if self.active and b.work is None:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p: nn.Parameter) -> None:
        b = self._owner[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def zero_grad(self) -> None:
        for b in self.buckets:
            b.flat.zero_()
            b.pending = len(b.params)
            b.work = None
            off = 0
            for p in b.params:   # re-point in case an optimizer replaced .grad
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + off * b.flat.element_size():
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
                off += p.numel()

    def finish(self) -> None:
        if not self.active:
            return
        for b in self.buckets:
            self._launch(b)      # buckets with parameters that received no gradient this step
        inv = 1.0 / self.world
        for b in self.buckets:
            b.work.wait()
            b.flat.mul_(inv)
