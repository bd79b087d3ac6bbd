"""Flat parameter arena + fused clip/AdamW for the GPS training step.

Caller-side mirror of ``graphgps/train/custom_train.py:33-37`` (``clip_grad_norm_`` ->
``optimizer.step()`` -> ``optimizer.zero_grad()``) for the optimizer the reference registers as
``'adamW'`` (``graphgps/optimizer/extra_optimizers.py:21-24`` = ``torch.optim.AdamW(params,
lr=base_lr, weight_decay=weight_decay)``).

MI355X-first layout: every parameter is re-pointed into ONE contiguous fp32 buffer (19.4 M floats
= 78 MB for GPS-medium; HBM holds 288 GB), gradients are packed into a second buffer of the same
layout, and the whole update is two HIP launches (csrc/optim.hip) instead of ~10 multi-tensor
launches + a norm reduction tree.  The same flat gradient buffer is what the data-parallel step
hands to RCCL as ONE 78 MB all-reduce (dp.py: ``FlatGradExchange``).

``FlatAdamW`` is a ``torch.optim.Optimizer`` (one param group; ``param_groups[0]['lr']`` is what
the reference's LR schedulers write), hyper-parameters and the step counter live on the device so
a captured hipGraph replays with a new learning rate after ``sync_hyper()``.
There is no CPU path: ``step()`` on CPU parameters raises.
"""
from __future__ import annotations

import bisect

from typing import Iterable, List, Optional

import weakref

import torch
import torch.nn as nn

from . import lib as _lib
from .lib import check, current_stream, ptr

_ALIGN = 64          # floats (256 B): start of every storage block in the arena


_ARENAS = {}      # data_ptr of an arena's parameter storage -> weakref(arena): lets the backward kernels find a gradient slot


def grad_slot(t: torch.Tensor):
    """The view of the gradient arena that mirrors ``t`` -- a parameter, or a zero-copy stack of adjacent parameters
    (fused.LinearGroup) -- when ``t`` lives in a ParamArena; else None.  A weight-gradient kernel that writes there has
    produced ``p.grad`` in place: autograd adopts the returned view and ``pack_grads`` finds nothing left to copy (the
    per-step multi-tensor copy of all 77.7 MB of gradients was 0.16 ms of the PCQM4M step)."""
    ref = _ARENAS.get(t.untyped_storage().data_ptr())
    arena = ref() if ref is not None else None
    if arena is None or not t.is_contiguous() or arena.flat_p.untyped_storage().data_ptr() != t.untyped_storage().data_ptr():
        return None
    off = t.storage_offset()
    if off < 0 or off + t.numel() > arena.flat_g.numel():
        return None
    # ONE direct writer per range and step: a weight applied twice in one backward (a Linear called twice,
    # tied weights held as separate Parameters) reaches here once per use, each time with ``p.grad is None`` --
    # AccumulateGrad only runs after ALL contributions arrived -- and two kernels writing the same slot would leave
    # 2 * g_last where g_1 + g_2 belongs.  The first use of a range claims it (until FlatAdamW.zero_grad); later uses get
    # None, i.e. a fresh tensor that autograd sums with the first.
    claimed = arena.__dict__.setdefault("_claimed", [])
    lo, hi = off, off + t.numel()
    i = bisect.bisect_left(claimed, (lo, lo))
    if (i < len(claimed) and claimed[i][0] < hi) or (i > 0 and claimed[i - 1][1] > lo):
        return None
    claimed.insert(i, (lo, hi))
    return arena.flat_g[off:off + t.numel()].view(t.shape)


class ParamArena:
    """Owns the flat parameter / gradient buffers and the chunk table the kernels walk.

    Parameters that already share one storage (a ``fused.LinearGroup`` stack: A|B|D|E|in_proj of a
    GPS layer) are moved as ONE block, keeping their relative offsets, so the stacked-weight views
    stay valid."""

    def __init__(self, params: Iterable[nn.Parameter]):
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("ParamArena: no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("ParamArena: parameters must be fp32 on one device")
        self.device = dev
        self.flat_p = self.flat_g = None
        self.adopt()

    # ------------------------------------------------------------------ layout
    def adopt(self) -> Optional[torch.Tensor]:
        """(Re)build the arena from wherever the parameters currently live.  Returns, for a
        rebuild, the int64 map old-offset-per-parameter (so optimizer state can follow), else
        None."""
        old_offsets = getattr(self, "offsets", None)
        # parameters that still live in the PREVIOUS arena are not a "shared storage" group: keeping their relative
        # layout would carry the holes of everything that has since moved out (a LinearGroup stacked after the arena
        # was built: 77.7 MB of parameters became a 119 MB arena, and the flat all-reduce moved the holes too)
        prev = self.flat_p.untyped_storage().data_ptr() if self.flat_p is not None else None
        blocks, in_prev = {}, []
        for i, p in enumerate(self.params):
            key = p.data.untyped_storage().data_ptr()
            if key == prev:
                in_prev.append(i)
            else:
                blocks.setdefault(key, []).append(i)
        # ... they are re-packed as runs of parameters that are adjacent there (a stack that was adopted earlier stays
        # one block, so its stacked view stays a view)
        # Parameters that alias or overlap one range (tied weights held as separate Parameter objects) stay in one run
        # as well: copied separately they would silently come untied.
        in_prev.sort(key=lambda i: self.params[i].data.storage_offset())
        run, run_end = [], 0
        for i in in_prev:
            off = self.params[i].data.storage_offset()
            if run and off > run_end:           # a hole in front of this one: the run ends
                blocks[("run", run[0])] = run
                run = []
            run_end = max(run_end, off + self.params[i].numel()) if run else off + self.params[i].numel()
            run.append(i)
        if run:
            blocks[("run", run[0])] = run
        offsets = [0] * len(self.params)
        plan, total = [], 0
        for key, idxs in blocks.items():
            if len(idxs) == 1:
                i = idxs[0]
                offsets[i] = total
                plan.append(("one", i, total))
                total += -(-self.params[i].numel() // _ALIGN) * _ALIGN
                continue
            # shared storage: keep the members' relative layout
            lo = min(self.params[i].data.storage_offset() for i in idxs)
            hi = max(self.params[i].data.storage_offset() + self.params[i].numel() for i in idxs)
            for i in idxs:
                if not self.params[i].data.is_contiguous():
                    raise ValueError("ParamArena: non-contiguous parameter in a shared storage")
                offsets[i] = total + self.params[i].data.storage_offset() - lo
            plan.append(("block", idxs, total, lo, hi))
            total += -(-(hi - lo) // _ALIGN) * _ALIGN
        flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for item in plan:
                if item[0] == "one":
                    _, i, off = item
                    p = self.params[i]
                    flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                else:
                    _, idxs, off, lo, hi = item
                    base = self.params[idxs[0]].data
                    src = base.new_empty(0).set_(base.untyped_storage(), lo, (hi - lo,))
                    flat_p[off:off + hi - lo].copy_(src)
            for i, p in enumerate(self.params):
                p.data = flat_p[offsets[i]:offsets[i] + p.numel()].view(p.shape)
        self.flat_p = flat_p
        self.flat_g = torch.zeros_like(flat_p)
        _ARENAS[flat_p.untyped_storage().data_ptr()] = weakref.ref(self)
        self.offsets = offsets
        self.grad_views = [self.flat_g[o:o + p.numel()].view(p.shape)
                           for o, p in zip(offsets, self.params)]
        self._ptrs = [p.data_ptr() for p in self.params]
        self._build_chunks()
        self._active_host = None
        self.active = torch.ones(len(self.params), dtype=torch.uint8, device=self.device)
        return old_offsets

    def _build_chunks(self) -> None:
        ch = _lib.load().gps_optim_chunk() if self.device.type == "cuda" else 4096
        offs, lens, owner = [], [], []
        for i, (o, p) in enumerate(zip(self.offsets, self.params)):
            n = p.numel()
            for s in range(0, n, ch):
                offs.append(o + s)
                lens.append(min(ch, n - s))
                owner.append(i)
        self.n_chunks = len(offs)
        self.chunk_off = torch.tensor(offs, dtype=torch.int64, device=self.device)
        self.chunk_len = torch.tensor(lens, dtype=torch.int32, device=self.device)
        self.chunk_param = torch.tensor(owner, dtype=torch.int32, device=self.device)

    def intact(self) -> bool:
        """Every parameter still lives where the arena put it (``.to()``, ``load_state_dict`` with
        ``assign=True`` or a late ``LinearGroup`` re-stack move them)."""
        return self._ptrs == [p.data_ptr() for p in self.params]

    @property
    def num_bytes(self) -> int:
        return self.flat_g.numel() * 4

    # ------------------------------------------------------------------ gradients
    def pack_grads(self) -> None:
        """``flat_g`` <- the gradients autograd produced (one multi-tensor copy); parameters
        without a gradient are flagged inactive.  Afterwards every ``p.grad`` IS its arena view."""
        from .fused import join_side_stream
        if self.device.type == "cuda":
            join_side_stream(self.device)      # weight gradients are produced on the side stream
        dst, src, act = [], [], []
        for p, v in zip(self.params, self.grad_views):
            g = p.grad
            act.append(g is not None)
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        if act != self._active_host:
            capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if capturing:
                raise RuntimeError("ParamArena: the set of parameters with gradients changed "
                                   "inside a hipGraph capture; run one eager step first")
            self._active_host = act
            self.active.copy_(torch.tensor(act, dtype=torch.uint8))
            if not all(act):                   # inactive slots must read as zero for all-reduce
                for a, v in zip(act, self.grad_views):
                    if not a:
                        v.zero_()
        for p, v, a in zip(self.params, self.grad_views, act):
            if a:
                p.grad = v


    def pack_range(self, i0: int, i1: Optional[int] = None) -> None:
        """``pack_grads`` for the parameters ``i0 .. i1-1`` only, leaving the active flags alone: what a backward split in
        two (train.TrainStep ``backward_split``) runs after its first half, so that the gradients of the upper half of the
        network sit in ``flat_g`` -- ready for their all-reduce -- while the lower half is still being differentiated.
        The full ``pack_grads`` at the end of the backward finds these parameters already adopted."""
        from .fused import join_side_stream
        if self.device.type == "cuda":
            join_side_stream(self.device)
        i1 = len(self.params) if i1 is None else i1
        dst, src = [], []
        for p, v in zip(self.params[i0:i1], self.grad_views[i0:i1]):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params[i0:i1], self.grad_views[i0:i1]):
            if p.grad is not None:
                p.grad = v


class FlatAdamW(torch.optim.Optimizer):
    """AdamW (+ optional ``clip_grad_norm_``) over a :class:`ParamArena`, two HIP launches."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      max_grad_norm=max_grad_norm))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW supports a single parameter group")
        self.arena = ParamArena(params)
        dev = self.arena.device
        n = self.arena.flat_p.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(8, dtype=torch.float64, device=dev)
        self.param_step = torch.zeros(len(self.arena.params), dtype=torch.float32, device=dev)
        self._ws = torch.empty(max(self.arena.n_chunks, 1), dtype=torch.float32, device=dev)
        self._hyper_host = None
        self._packed = False
        self.sync_hyper()

    # ------------------------------------------------------------------ hyper-parameters
    def _hyper_tuple(self):
        g = self.param_groups[0]
        mx = g.get("max_grad_norm")
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                float(g["weight_decay"]), float(mx) if mx else 0.0)

    def sync_hyper(self) -> None:
        """Upload lr / betas / eps / weight_decay / max_norm if the param group changed (what an LR
        scheduler does once per epoch).  Call it OUTSIDE a captured region."""
        h = self._hyper_tuple()
        if h != self._hyper_host:
            self.hyper[:6].copy_(torch.tensor(h, dtype=torch.float64))
            self._hyper_host = h

    @property
    def total_norm(self) -> torch.Tensor:
        """Gradient norm of the last step (device scalar, what clip_grad_norm_ returns)."""
        return self.hyper[7]

    @property
    def step_count(self) -> torch.Tensor:
        return self.hyper[6]

    # ------------------------------------------------------------------ the step
    def _readopt(self) -> None:
        old = self.arena.adopt()
        n = self.arena.flat_p.numel()
        m, v = self.exp_avg, self.exp_avg_sq
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.arena.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        for o_old, o_new, p in zip(old, self.arena.offsets, self.arena.params):
            k = p.numel()
            self.exp_avg[o_new:o_new + k].copy_(m[o_old:o_old + k])
            self.exp_avg_sq[o_new:o_new + k].copy_(v[o_old:o_old + k])
        self._ws = torch.empty(max(self.arena.n_chunks, 1), dtype=torch.float32,
                               device=self.arena.device)

    def pack_grads(self) -> None:
        """Gather the step's gradients into ``arena.flat_g`` (the buffer a data-parallel step
        all-reduces before ``step()``)."""
        if not self.arena.intact():
            if self.arena.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamW: parameters moved out of the arena inside a capture")
            self._readopt()
        self.arena.pack_grads()
        self._packed = True

    def pack_range(self, i0: int, i1: Optional[int] = None) -> bool:
        """``arena.pack_range`` (the gradients of parameters ``i0 .. i1-1`` into ``flat_g``, mid-backward) when the arena
        is intact; False -- nothing packed, the caller must not let that range travel early -- when parameters have moved
        (the full ``pack_grads`` at the end of the backward re-adopts them)."""
        if not self.arena.intact():
            return False
        self.arena.pack_range(i0, i1)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a = self.arena
        if a.device.type != "cuda":
            raise _lib.GpsHipError("FlatAdamW.step: parameters are on the CPU; the optimizer "
                                   "step is a HIP kernel (csrc/optim.hip), there is no CPU path")
        if not self._packed:
            self.pack_grads()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        check(_lib.load().gps_adamw_step(ptr(a.flat_p), ptr(a.flat_g), ptr(self.exp_avg),
                                         ptr(self.exp_avg_sq), ptr(a.chunk_off), ptr(a.chunk_len),
                                         ptr(a.chunk_param), ptr(a.active), a.n_chunks,
                                         ptr(self.hyper), ptr(self.param_step), ptr(self._ws),
                                         current_stream(a.device)),
              "gps_adamw_step")
        self._packed = False
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        # gradients are re-packed every step, so dropping the references is all there is to do
        for p in self.arena.params:
            p.grad = None
        self.arena.__dict__["_claimed"] = []       # direct-write claims of the step that ended (grad_slot)
        self._packed = False

    # ------------------------------------------------------------------ checkpointing
    def state_dict(self):
        """torch.optim.AdamW layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so
        checkpoints interchange with the reference's optimizer (custom_train.py:130-131 saves
        ``optimizer.state_dict()`` through GraphGym's save_ckpt)."""
        a = self.arena
        steps = self.param_step.detach().cpu()
        state = {}
        for i, (o, p) in enumerate(zip(a.offsets, a.params)):
            k = p.numel()
            if float(steps[i]) == 0.0:        # torch creates state lazily, at the first gradient
                continue
            state[i] = dict(step=steps[i].clone(),
                            exp_avg=self.exp_avg[o:o + k].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[o:o + k].view(p.shape).clone())
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(a.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd) -> None:
        a = self.arena
        g = sd["param_groups"][0]
        for k, v in g.items():
            if k != "params":
                self.param_groups[0][k] = v
        self.param_groups[0].setdefault("max_grad_norm", None)
        steps = [0.0] * len(a.params)
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for i, (o, p) in enumerate(zip(a.offsets, a.params)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                k = p.numel()
                self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps[i] = float(st["step"])
            self.param_step.copy_(torch.tensor(steps, dtype=torch.float32))
            self.hyper[6] = max(steps)
        self._hyper_host = None
        self.sync_hyper()


def adamW_optimizer(params, base_lr: float, weight_decay: float,
                    max_grad_norm: Optional[float] = None) -> FlatAdamW:
    """``register_optimizer('adamW')`` (extra_optimizers.py:21-24), same signature."""
    return FlatAdamW(params, lr=base_lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)


from .graphgym import register as _register  # noqa: E402

_register.register_optimizer('adamW', adamW_optimizer, overwrite=True)
