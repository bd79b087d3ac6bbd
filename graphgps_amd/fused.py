"""Operators of the modular (operator-by-operator) path around the sparse / attention kernels.

``bn_act``       y = res + dropout(relu(BatchNorm1d(z)))   -- every stage optional   (csrc/bn_fused.hip)
``add_dropout``  out = a + dropout(b)
``relu_dropout`` out = dropout(relu(x))
``linear``       y = x W^T + b with the weight + bias gradient from the split-K MFMA kernel (csrc/wgrad.hip)
                 on a side HIP stream that joins at the end of backward
``LinearGroup``  several nn.Linear that share an input evaluated as ONE GEMM over a zero-copy stack of
                 their weights (the stack is a view of the parameters' own storage)

They stand in for the module chains of the reference (``gatedgcn_layer.py:72-83``,
``gps_layer.py:191-194,212-217,225-229,253-257``) in TRAINING mode on the GPU; semantics are
``torch.nn.BatchNorm1d`` / ``F.dropout`` / ``relu`` exactly, except that dropout masks come from
the library's counter hash (seeded from torch's CPU generator) instead of ATen's Philox stream.
Evaluation mode, non-ReLU activations and BatchNorm configurations the kernels do not cover
(``affine=False``, ``momentum=None``) take the equivalent torch ops.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from .lib import check, current_stream, ptr
from .ops import _f32c, _require_cuda, draw_dropout_seed


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, res, running_mean, running_var, eps, momentum, relu, p_drop,
                seed):
        L = _lib.load()
        dev = _require_cuda(z, gamma, beta, res)
        z = _f32c(z, "z")
        R, d = z.shape
        res_c = _f32c(res, "res") if res is not None else None
        f32 = dict(dtype=torch.float32, device=dev)
        mean, rstd = torch.empty(d, **f32), torch.empty(d, **f32)
        ws = torch.empty(max(L.gps_bn_workspace_floats(R, d), 1), **f32)
        st = current_stream(dev)
        check(L.gps_bn_stats(ptr(z), R, d, eps, momentum, ptr(mean), ptr(rstd), ptr(running_mean),
                             ptr(running_var), ptr(ws), st), "gps_bn_stats")
        y = torch.empty_like(z)
        check(L.gps_bn_apply(ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(res_c), R, d,
                             int(relu), p_drop, seed, ptr(y), st), "gps_bn_apply")
        ctx.save_for_backward(z, gamma, beta, mean, rstd)
        ctx.cfg = (bool(relu), float(p_drop), int(seed), res is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        L = _lib.load()
        z, gamma, beta, mean, rstd = ctx.saved_tensors
        relu, p_drop, seed, has_res = ctx.cfg
        g_y = _f32c(g_y, "g_y")
        R, d = z.shape
        dev = z.device
        f32 = dict(dtype=torch.float32, device=dev)
        g_z = torch.empty_like(z)
        g_gamma, g_beta = torch.empty(d, **f32), torch.empty(d, **f32)
        ws = torch.empty(max(L.gps_bn_workspace_floats(R, d), 1), **f32)
        check(L.gps_bn_bwd(ptr(z), ptr(g_y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), R, d,
                           int(relu), p_drop, seed, ptr(g_z), ptr(g_gamma), ptr(g_beta), ptr(ws),
                           current_stream(dev)), "gps_bn_bwd")
        return (g_z, g_gamma, g_beta, g_y if has_res else None, None, None, None, None, None, None,
                None)


def _fusable(bn: nn.BatchNorm1d, z: torch.Tensor) -> bool:
    return (bn.training and z.is_cuda and bn.affine and bn.track_running_stats
            and bn.momentum is not None and z.dim() == 2 and z.shape[0] >= 2
            and z.dtype == torch.float32)


def bn_act(z: torch.Tensor, bn: nn.BatchNorm1d, relu: bool = False, p_drop: float = 0.0,
           res: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> torch.Tensor:
    """``res + dropout(relu(bn(z)))`` with train-mode batch statistics, one stats pass + one
    apply pass; updates ``running_mean/var`` and ``num_batches_tracked`` like the module."""
    if not _fusable(bn, z):
        y = bn(z)
        if relu:
            y = F.relu(y)
        y = F.dropout(y, p_drop, training=bn.training)
        return y if res is None else res + y
    if p_drop > 0.0 and seed is None:
        seed = draw_dropout_seed()
    bn.num_batches_tracked.add_(1)
    return _BNAct.apply(z, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, float(bn.eps),
                        float(bn.momentum), bool(relu), float(p_drop), int(seed or 0))


class _ActDropAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, relu, p_drop, seed):
        L = _lib.load()
        dev = _require_cuda(a, b)
        b = _f32c(b, "b")
        a_c = _f32c(a, "a") if a is not None else None
        R, d = b.shape
        out = torch.empty_like(b)
        check(L.gps_act_drop_add(ptr(a_c), ptr(b), R, d, int(relu), p_drop, seed, ptr(out),
                                 current_stream(dev)), "gps_act_drop_add")
        if relu:
            ctx.save_for_backward(b)
        ctx.cfg = (bool(relu), float(p_drop), int(seed), a is not None, R, d)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.load()
        relu, p_drop, seed, has_a, R, d = ctx.cfg
        g = _f32c(g, "g")
        pre = ctx.saved_tensors[0] if relu else None
        g_b = torch.empty_like(g)
        check(L.gps_act_drop_bwd(ptr(g), ptr(pre), R, d, int(relu), p_drop, seed, ptr(g_b),
                                 current_stream(g.device)), "gps_act_drop_bwd")
        return (g if has_a else None), g_b, None, None, None


def add_dropout(a: torch.Tensor, b: torch.Tensor, p_drop: float, training: bool,
                seed: Optional[int] = None) -> torch.Tensor:
    """``a + dropout(b)`` (residual connections of gps_layer.py:188-189,212-213,225)."""
    if not (training and p_drop > 0.0 and b.is_cuda and b.dim() == 2 and b.dtype == torch.float32):
        return a + F.dropout(b, p_drop, training=training)
    return _ActDropAdd.apply(a, b, False, float(p_drop), int(seed or draw_dropout_seed()))


# -------------------------------------------------------------------------------------------
# weight / bias gradients on a second HIP stream
# -------------------------------------------------------------------------------------------
# The weight-gradient GEMMs (long K = N or E rows, small [out, in] result) are off the critical
# path of the backward chain: nothing needs them before the optimizer.  They are issued on a side
# stream so they overlap the small latency-bound kernels of the chain (BN passes, attention,
# sparse kernels) instead of serialising with them.  The main stream re-joins the side stream
# (a) at the end of every backward pass (autograd engine callback) and (b) wherever dp.py packs
# gradients.  GPS_WGRAD_SIDE_STREAM=0 disables it for the per-module path below.
#
# Two switches since round 2:
#   * this file (nn.Linear-shaped modules outside the fused block: encoders, heads, the Performer / GINE / SAN
#     layers): side stream ON by default, as validated in round 1.  (Turned off, the captured code2 step used to die at
#     replay with "Write access to a read-only page": a one-stream capture is a purely linear hipGraph, which this
#     runtime mishandles -- train.py:TrainStep.capture now adds a trivial forked node to every capture; DESIGN.md
#     section 7, tools/runs/gpu_r2w.sh.)
#   * layer/gps_block.py (the fused CustomGatedGCN+Transformer block): GPS_BLOCK_WGRAD_SIDE_STREAM, default OFF.
#     It paid while the projection GEMMs left half of every CU's LDS free (12.2 vs 12.6 ms per step); the ring
#     GEMM (csrc/gemm_panel.hip) owns a CU's whole LDS, so a co-scheduled weight-gradient workgroup and a ring
#     workgroup exclude each other from the CU, and one stream replayed as a hipGraph is faster: 11.39 ms
#     (graph, one stream) vs 11.60 ms (eager, two streams), PCQM4M GPS-medium step before the streaming
#     weight-gradient kernel (tools/runs/gpu_r2q.sh).
import os as _os

_SIDE_ENABLED = _os.environ.get("GPS_WGRAD_SIDE_STREAM", "1") != "0"
_WGRAD_F16 = _os.environ.get("GPS_WGRAD_F16", "1") != "0"
_BLOCK_SIDE_ENABLED = _os.environ.get("GPS_BLOCK_WGRAD_SIDE_STREAM", "0") != "0"
_side_streams = {}
_join_pending = set()


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    st = _side_streams.get(dev.index)
    if st is None:
        st = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return st


def join_side_stream(dev: Optional[torch.device] = None) -> None:
    """Make the current stream wait for the weight-gradient stream(s)."""
    for idx, st in _side_streams.items():
        if dev is None or dev.index == idx:
            torch.cuda.current_stream(torch.device("cuda", idx)).wait_stream(st)
    _join_pending.clear()


def _queue_join(dev: torch.device) -> None:
    if dev.index in _join_pending:
        return
    _join_pending.add(dev.index)
    try:   # runs once, when the backward pass that is executing right now has finished
        torch.autograd.Variable._execution_engine.queue_callback(lambda: join_side_stream(dev))
    except RuntimeError:   # not inside a backward pass (e.g. torch.autograd.grad on a sub-graph)
        join_side_stream(dev)


def _param_grads(g: torch.Tensor, x: torch.Tensor, need_w: bool, need_b: bool, params=(), targets=None):
    """(g^T x, column sums of g): rocBLAS GEMM + the library's two-stage colsum.  ``params`` are
    the leaf parameters the results go to: if any already holds a ``.grad`` autograd will
    accumulate into it on the main stream right after this function returns, so the side stream
    is only used when they are all empty (the zero_grad(set_to_none=True) regime).  ``targets`` = (weight, bias) the
    results belong to (possibly zero-copy stacks): in that regime, and when they live in an optimizer arena, the kernel
    writes straight into their gradient slots (optim.grad_slot) and the per-step gradient packing has nothing to copy."""
    dev = g.device
    accumulating = any(p is not None and p.grad is not None for p in params)

    def compute():
        L = _lib.load()
        R, d = g.shape
        k = x.shape[1]
        st = current_stream(dev)
        if need_w and d % 4 == 0 and k % 4 == 0 and g.stride(0) % 4 == 0 and x.stride(0) % 4 == 0:
            # split-K MFMA kernel: weight AND bias gradient in one pass (csrc/wgrad.hip)
            g_w = g_b = None
            if targets is not None and not accumulating:
                from .optim import grad_slot
                g_w = grad_slot(targets[0]) if targets[0] is not None else None
                g_b = grad_slot(targets[1]) if need_b and targets[1] is not None else None
            if g_w is None or g_w.shape != (d, k):
                g_w = torch.empty(d, k, dtype=torch.float32, device=dev)
            if need_b and (g_b is None or g_b.shape != (d,)):
                g_b = torch.empty(d, dtype=torch.float32, device=dev)
            ws = torch.empty(max(L.gps_wgrad_workspace_floats(R, d, k), 4), dtype=torch.float32,
                             device=dev)
            from . import gemm as _gemm
            if _gemm.F16 and _WGRAD_F16 and d % 128 == 0 and k % 128 == 0 and R >= _RING_MIN_ROWS:
                # fp16 form of the streaming kernel (csrc/wgrad.hip): the operands' max|.| words from one pre-pass
                words = _gemm.absmax([g, x])
                check(L.gps_wgrad16(ptr(g), g.stride(0), ptr(x), x.stride(0), R, d, k, ptr(words[0]), ptr(words[1]),
                                    ptr(g_w), ptr(g_b), ptr(ws), st), "gps_wgrad16")
                return g_w, g_b
            check(L.gps_wgrad(ptr(g), g.stride(0), ptr(x), x.stride(0), R, d, k, ptr(g_w), ptr(g_b),
                              ptr(ws), st), "gps_wgrad")
            return g_w, g_b
        g_w = g.t().mm(x) if need_w else None
        g_b = None
        if need_b:
            g_b = torch.empty(d, dtype=torch.float32, device=dev)
            ws = torch.empty(max(L.gps_bn_workspace_floats(R, d), 1), dtype=torch.float32, device=dev)
            check(L.gps_colsum(ptr(g), R, d, ptr(g_b), ptr(ws), st), "gps_colsum")
        return g_w, g_b

    if not _SIDE_ENABLED or accumulating:
        return compute()
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(cur)                 # g and x are complete
    with torch.cuda.stream(side):
        g_w, g_b = compute()
    g.record_stream(side)                 # keep the caching allocator from recycling them early
    x.record_stream(side)
    _queue_join(dev)
    return g_w, g_b


# Rows below which a projection stays on the library GEMM (the ring kernel's split launch + 64-row panels do not pay on a
# handful of rows); GPS_GEMM_PANEL=0 (gemm.ENABLED) switches the ring GEMM off everywhere.
_RING_MIN_ROWS = 256


def _ring_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    from . import gemm as _gemm
    N, K = weight.shape
    return (_gemm.supported(N, K) and x.shape[0] >= _RING_MIN_ROWS and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and weight.stride(1) == 1 and weight.stride(0) % 4 == 0
            and weight.data_ptr() % 16 == 0)


def _ring_forward(ctx, x, weight, bias, need_gx):
    """y = x W^T + b on the ring GEMM (csrc/gemm_panel.hip: fp32-exact products on the bf16 MFMA pipe); the W^T image for
    the input gradient is made in the same split launch and kept on ``ctx``."""
    from . import gemm as _gemm
    N, K = weight.shape
    tn = need_gx and _gemm.supported(K, N)
    (img_nt, img_tn), = _gemm.split_weights([weight], nt=True, tn=tn)
    ctx.img_tn = img_tn
    return _gemm.gemm_panel(x, img_nt, N, bias=bias)


def _ring_input_grad(ctx, g, weight):
    from . import gemm as _gemm
    img = getattr(ctx, "img_tn", None)
    if img is not None and g.shape[0] >= _RING_MIN_ROWS and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0:
        return _gemm.gemm_panel(g, img, weight.shape[1])
    return g.mm(weight)


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  Forward and input gradient on the ring GEMM where the shape qualifies (N % 64 == 0, K % 32 == 0,
    >= 256 rows: every projection of a d = 256 / 384 layer -- ogbg-code2-GPS.yaml, pcqm4m-GPS*.yaml -- outside the fused
    blocks), the library GEMMs otherwise; weight + bias gradient by the streaming split-K kernel (csrc/wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.img_tn = None
        if _ring_ok(x, weight):
            return _ring_forward(ctx, x, weight, bias, ctx.needs_input_grad[0])
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _f32c(g, "g")
        g_w, g_b = _param_grads(g, x, ctx.needs_input_grad[1],
                                ctx.has_bias and ctx.needs_input_grad[2], ctx.params, targets=ctx.params)
        g_x = _ring_input_grad(ctx, g, weight) if ctx.needs_input_grad[0] else None
        return g_x, g_w, g_b


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.linear`` for the [N,d] / [E,d] projections of the layer (A..E, in/out-proj, FFN)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 1
            and torch.is_grad_enabled()):
        return F.linear(x, weight, bias)
    return _Linear.apply(x, weight, bias)


# Rows up to which a Linear (+ ReLU) runs on csrc/small_gemm.hip: the graph-level heads' stages on one row per graph.
_SMALL_MAX_ROWS = 2048
_SMALL_LINEAR = _os.environ.get("GPS_SMALL_LINEAR", "1") != "0"      # 0: the library GEMMs (A/B)


class _SmallLinear(torch.autograd.Function):
    """``act(x W^T + b)`` for a few hundred rows (csrc/small_gemm.hip: exact fp32 MFMA products, one launch forward, two
    backward -- the ReLU mask of the layer's own output is applied while its output gradient is loaded).  Reference:
    graphgps/head/san_graph.py:36-41.  The library GEMMs pick macro tiles for large problems on these shapes
    ([256 x 384] x [384 x 192] as ONE workgroup, 74 us)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        L = _lib.load()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        check(L.gps_small_linear_fwd(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(bias), M, N, K, int(relu),
                                     ptr(y), N, current_stream(x.device)), "gps_small_linear_fwd")
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.params = bool(relu), bias is not None, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        L = _lib.load()
        x, weight, y = ctx.saved_tensors
        g = _f32c(g, "g")
        M, K = x.shape
        N = weight.shape[0]
        dev = g.device
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        g_x = torch.empty(M, K, dtype=torch.float32, device=dev) if need_x else None
        g_w = g_b = None
        if need_w:
            if not any(p is not None and p.grad is not None for p in ctx.params):
                from .optim import grad_slot          # straight into the optimizer's gradient arena when there is one
                g_w = grad_slot(weight)
                g_b = grad_slot(ctx.params[1]) if need_b else None
            if g_w is None:
                g_w = torch.empty(N, K, dtype=torch.float32, device=dev)
            if need_b and g_b is None:
                g_b = torch.empty(N, dtype=torch.float32, device=dev)
        check(L.gps_small_linear_bwd(ptr(g), g.stride(0), ptr(y), N if y is not None else 0, ptr(x), x.stride(0),
                                     ptr(weight), weight.stride(0), M, N, K, ptr(g_x), K, ptr(g_w), K, ptr(g_b),
                                     current_stream(dev)), "gps_small_linear_bwd")
        return g_x, g_w, g_b, None


def small_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    """``relu(F.linear(x, weight, bias))`` / ``F.linear`` for inputs of a few hundred rows (one per graph)."""
    if (_SMALL_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and 1 <= x.shape[0] <= _SMALL_MAX_ROWS and x.stride(1) == 1 and weight.stride(1) == 1
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))):
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
            return _SmallLinear.apply(x, weight, bias, relu)
        L = _lib.load()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        check(L.gps_small_linear_fwd(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(bias), M, N, K, int(relu),
                                     ptr(y), N, current_stream(x.device)), "gps_small_linear_fwd")
        return y
    y = F.linear(x, weight, bias)
    return torch.relu(y) if relu else y


class _GroupLinear(torch.autograd.Function):
    """y = x [W_1; ...; W_k]^T + [b_1; ...; b_k] where the stacked weight is a zero-copy view of
    the k parameters' shared storage; the backward hands each parameter its row-slice of the
    stacked weight gradient (views, no split kernels)."""

    @staticmethod
    def forward(ctx, x, wcat, bcat, sizes, *params):
        ctx.save_for_backward(x, wcat)
        ctx.sizes, ctx.has_bias = sizes, bcat is not None
        ctx.params = params
        ctx.bcat = bcat
        ctx.img_tn = None
        if _ring_ok(x, wcat):
            return _ring_forward(ctx, x, wcat, bcat, ctx.needs_input_grad[0])
        return F.linear(x, wcat, bcat)

    @staticmethod
    def backward(ctx, g):
        x, wcat = ctx.saved_tensors
        g = _f32c(g, "g")
        g_w, g_b = _param_grads(g, x, True, ctx.has_bias, ctx.params, targets=(wcat, ctx.bcat))
        g_x = _ring_input_grad(ctx, g, wcat) if ctx.needs_input_grad[0] else None
        outs, off = [], 0
        for n in ctx.sizes:
            outs.append(g_w[off:off + n])
            off += n
        if ctx.has_bias:
            off = 0
            for n in ctx.sizes:
                outs.append(g_b[off:off + n])
                off += n
        return (g_x, None, None, None, *outs)


class LinearGroup:
    """Several ``nn.Linear`` modules that consume the same input (A, B, D, E of GatedGCN --
    gatedgcn_layer.py:57-61; to_q/to_k/to_v of the Performer) evaluated as ONE GEMM.

    The parameters stay separate leaves with their reference names (state_dict contract) but
    their storage is re-pointed into one flat buffer, so the stacked [sum(out), in] weight
    exists without a per-step ``torch.cat`` and its gradient needs no per-step split."""

    def __init__(self, linears=None, weights=None, biases=None):
        """Either ``linears`` (nn.Linear modules) or explicit ``weights`` / ``biases`` parameter
        lists (e.g. A..E plus MultiheadAttention.in_proj_weight / in_proj_bias)."""
        if linears is not None:
            linears = list(linears)
            weights = [l.weight for l in linears]
            biases = [l.bias for l in linears] if linears[0].bias is not None else None
        self.ws = list(weights)
        self.bs = list(biases) if biases is not None else None
        self._w = self._b = self._key = None

    @property
    def sizes(self):
        return tuple(w.shape[0] for w in self.ws)

    @staticmethod
    def _adjacent(ts):
        """The tensors sit back to back, in order, in one storage (so their row-stack is a view)."""
        st = ts[0].data.untyped_storage().data_ptr()
        nxt = ts[0].data_ptr()
        for t in ts:
            if not t.data.is_contiguous() or t.data_ptr() != nxt \
                    or t.data.untyped_storage().data_ptr() != st:
                return False
            nxt += t.numel() * t.element_size()
        return True

    @staticmethod
    def _view_over(ts, shape):
        base = ts[0].data
        return base.new_empty(0).set_(base.untyped_storage(), base.storage_offset(), shape)

    def _stacked(self):
        """(stacked weight [sum(out), in], stacked bias) as VIEWS of the parameters' storage.
        Whoever owns that storage may move it (``.to()``, a parameter arena -- optim.py) as long as
        the members stay adjacent; otherwise they are re-stacked here once."""
        ws, bs = self.ws, self.bs
        key = tuple(t.data_ptr() for t in (ws + bs if bs is not None else ws))
        if self._w is not None and self._key == key:
            return self._w, self._b
        with torch.no_grad():
            if not self._adjacent(ws):
                flat = torch.cat([w.data for w in ws], dim=0).contiguous()
                off = 0
                for w in ws:
                    w.data = flat[off:off + w.shape[0]]
                    off += w.shape[0]
            self._w = self._view_over(ws, (sum(w.shape[0] for w in ws), ws[0].shape[1]))
            if bs is not None:
                if not self._adjacent(bs):
                    flat = torch.cat([b.data for b in bs], dim=0).contiguous()
                    off = 0
                    for b in bs:
                        b.data = flat[off:off + b.shape[0]]
                        off += b.shape[0]
                self._b = self._view_over(bs, (sum(b.shape[0] for b in bs),))
            else:
                self._b = None
        self._key = tuple(t.data_ptr() for t in (ws + bs if bs is not None else ws))
        return self._w, self._b

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_grad_enabled()):
            w = torch.cat(self.ws, dim=0)
            b = torch.cat(self.bs, dim=0) if self.bs is not None else None
            return F.linear(x, w, b)
        w, b = self._stacked()
        params = self.ws + (self.bs if b is not None else [])
        return _GroupLinear.apply(x, w, b, self.sizes, *params)


def relu_dropout(x: torch.Tensor, p_drop: float, training: bool,
                 seed: Optional[int] = None) -> torch.Tensor:
    """``dropout(relu(x))`` (FFN, gps_layer.py:256)."""
    if not (training and p_drop > 0.0 and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
        return F.dropout(F.relu(x), p_drop, training=training)
    return _ActDropAdd.apply(None, x, True, float(p_drop), int(seed or draw_dropout_seed()))
