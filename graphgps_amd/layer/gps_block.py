"""One autograd node per GPS block (training mode): ``CustomGatedGCN+Transformer`` (``_GPSBlock``) and
``GINE+Transformer`` (``_GPSBlockGINE``).

Same arithmetic as the operator-by-operator path in ``gps_layer.py`` / ``gatedgcn_layer.py`` /
``gine_conv_layer.py`` (which stays the general path: evaluation, Performer, EquivStableLapPE, non-ReLU
activations, ``batch_norm=False``).  A layer's forward and backward are two straight-line Python functions
that call the C ABI and rocBLAS directly, with hand-written backward formulas (the ones SURVEY.md section 8a
lists and the per-operator tests pin).  What that buys:
  * host time: the modular path spends ~18 us of Python/autograd bookkeeping per launch;
  * merged GEMMs: A|B|D|E and the attention in-projection as ONE ``[N,d] x [d,7d]`` GEMM (forward, dgrad
    and weight gradient);
  * the norm / residual / dropout stages as task lists (csrc/block_norm.hip) whose column reductions finish
    inside the producing launch (csrc/col_tree.hpp); the batch statistics of za / z2 come out of the ring GEMM
    epilogues, those of x~ / e^ from one statistics-only launch (or, GPS_GG_STATS=1, out of the GatedGCN forward):
    9 norm launches per layer (8 with GPS_GG_STATS=1; round 2: 17, the operator path: 37);
  * all weight gradients of the block as ONE grouped split-K launch (csrc/wgrad.hip) that writes straight into the
    optimizer's gradient arena;
  * one stream (round 3 default; GPS_BRANCH_STREAM=1|2 forks the attention half while capturing / always: the ring
    GEMM owns a CU's whole LDS, so the forked half cannot co-run with it and each fork / join costs ~10 us);
  * per-step set-up shared by the whole layer stack (stack_begin / stack_end): one weight-image launch, one
    BatchNorm-counter launch, and the block's submodule / parameter references cached per layer (_Refs).

Reference lines: graphgps/layer/gps_layer.py:155-232, graphgps/layer/gatedgcn_layer.py:45-88.
"""
from __future__ import annotations

import ctypes as _ctypes
import os as _os

import torch

from .. import gemm as _gemm
from .. import lib as _lib
from .. import norm as _norm
from ..fused import _BLOCK_SIDE_ENABLED as _SIDE_ENABLED, _queue_join, _side_stream
from ..lib import check, current_stream, ptr
from ..ops import GraphIndex, _nmax_dev, draw_dropout_seed, favor_workspace_floats

_E = torch.empty
_BY_REF = _ctypes.byref

# A/B switches.  GPS_GG_STATS (default 1 since round 6): the batch statistics of x~ / e^ out of the GatedGCN forward itself
# -- per-node-block records + a 96-workgroup combine launch inside the same C call -- instead of a statistics-only row pass
# over both tensors (=0: 20 us and 35 MB per layer).  Rounds 3 - 5 had this off: the records were then combined IN the
# GatedGCN launch through the arrival tree, whose tail cost the kernel as much as the pass (26 -> 47 us).
# GPS_GEMM_STATS=0: za / z2 and their statistics by row tasks instead of the ring GEMM epilogue (default 1: -0.3 ms per step).
_GG_STATS = _os.environ.get("GPS_GG_STATS", "1") != "0"
_GEMM_STATS = _os.environ.get("GPS_GEMM_STATS", "1") != "0"
# GPS_GEMM_PAIR=0: the edge projection C(e) and the merged node projection (forward), and their two input-gradient GEMMs
# (backward), as two dispatches each instead of one (csrc/gemm_panel.hip k_gemm_ring16_pair) -- A/B
_GEMM_PAIR = _os.environ.get("GPS_GEMM_PAIR", "1") != "0"
# GPS_GG_BN_FOLD=0: bn_node_x / bn_edge_e backward applies as task-list launches in front of the GatedGCN backward (A/B)
# (a mask: 1 = the node fold, 2 = the edge fold, 3 = both)
_GG_BN_FOLD = int(_os.environ.get("GPS_GG_BN_FOLD", "3"))
_STACK_PREP = _os.environ.get("GPS_STACK_PREP", "1") != "0"
# GPS_GEMM_BWDSUMS=0: the column sums of norm1_local + norm1_attn's backward by a gps_norm_bwd_partial launch instead of the
# epilogue of the GEMM that produces their output gradient, g_h = g_z2 + g_f1 W1 (csrc/gemm_panel.hip epilogue 4) -- A/B
_GEMM_BWDSUMS = _os.environ.get("GPS_GEMM_BWDSUMS", "1") != "0"

# Work that is per layer only by accident, hoisted to the layer STACK when a network drives the blocks (network/base.py
# brackets its layer stack with stack_begin / stack_end; a block called on its own behaves as before):
#   * the weight images of every block's five projections come out of ONE split launch at the head of the stack instead
#     of one 12 us launch per layer (the weights only change in the optimizer step);
#   * the BatchNorm ``num_batches_tracked`` counters of all blocks take ONE multi-tensor add at the end of the stack
#     instead of one 5 us launch per layer.
_STACK = {"active": False, "nbt": []}


def _count_batches(counters):
    if _STACK["active"]:
        _STACK["nbt"].extend(counters)
    else:
        torch._foreach_add_(counters, 1)


class _Refs:
    """The submodules and parameters of one CustomGatedGCN + Transformer layer, looked up once: ``nn.Module.__getattr__``
    costs ~0.3 us per access and the block touched ~140 of them per layer and step (1.4k per step, 0.7 ms of host time
    forward + backward).  Parameters and modules are cached by identity (``p.data`` may be re-pointed by an arena, the
    Parameter object stays); ``valid`` notices a replaced submodule."""
    __slots__ = ("lm", "sa", "A", "B", "D", "E", "C", "out_proj", "ff1", "ff2", "bnx", "bne", "bnl", "bna", "bn2",
                 "drop_attn", "ff_drop1", "ff_drop2", "params", "_ids", "perf")

    def __init__(self, layer):
        m = layer._modules
        self.lm, self.sa = m["local_model"], m["self_attn"]
        lmm = self.lm._modules
        self.A, self.B, self.D, self.E, self.C = lmm["A"], lmm["B"], lmm["D"], lmm["E"], lmm["C"]
        self.perf = layer.global_model_type == 'Performer'
        # (Performer: the output projection is ``to_out``, performer_layer.py:461-464)
        self.out_proj = self.sa._modules["to_out" if self.perf else "out_proj"]
        self.ff1, self.ff2 = m["ff_linear1"], m["ff_linear2"]
        self.bnx, self.bne = lmm["bn_node_x"], lmm["bn_edge_e"]
        self.bnl, self.bna, self.bn2 = m["norm1_local"], m["norm1_attn"], m["norm2"]
        self.drop_attn, self.ff_drop1, self.ff_drop2 = m["dropout_attn"], m["ff_dropout1"], m["ff_dropout2"]
        W, Bv = (lambda mod: mod._parameters["weight"]), (lambda mod: mod._parameters["bias"])
        sam = self.sa._modules
        inproj = ([W(sam["to_q"]), W(sam["to_k"]), W(sam["to_v"])] if self.perf
                  else [self.sa._parameters["in_proj_weight"], self.sa._parameters["in_proj_bias"]])
        self.params = [W(self.A), W(self.B), W(self.D), W(self.E), Bv(self.A), Bv(self.B), Bv(self.D), Bv(self.E),
                       W(self.C), Bv(self.C), W(self.bnx), Bv(self.bnx), W(self.bne), Bv(self.bne),
                       W(self.bnl), Bv(self.bnl), *inproj, W(self.out_proj), Bv(self.out_proj),
                       W(self.bna), Bv(self.bna), W(self.ff1), Bv(self.ff1), W(self.ff2), Bv(self.ff2),
                       W(self.bn2), Bv(self.bn2)]
        self._ids = self._snapshot(layer)

    @staticmethod
    def _snapshot(layer):
        m = layer._modules
        lm, sa = m["local_model"], m["self_attn"]
        mods = (lm, sa, m["ff_linear1"], m["ff_linear2"], m["norm1_local"], m["norm1_attn"], m["norm2"],
                lm._modules["C"], lm._modules["bn_node_x"], lm._modules["bn_edge_e"],
                sa._modules.get("out_proj") or sa._modules["to_out"])
        return tuple(id(x) for x in mods) + tuple(id(x._parameters.get("weight")) for x in mods[2:])

    def valid(self, layer) -> bool:
        return self._ids == self._snapshot(layer)


def _refs(layer) -> _Refs:
    r = layer.__dict__.get("_blk_refs")
    if r is None or not r.valid(layer):
        r = layer.__dict__["_blk_refs"] = _Refs(layer)
    return r


def _W(mod):
    return mod._parameters["weight"]


def _B(mod):
    return mod._parameters["bias"]


def _ensure_xgroup(layer):
    from ..fused import LinearGroup
    lm, sa = layer.local_model, layer.self_attn
    if getattr(layer, "_xgroup", None) is None:
        # zero-copy stack of every weight that multiplies the layer input: A, B, D, E, in_proj (Performer: to_q, to_k, to_v,
        # which carry no bias -- the merged GEMM's bias row is padded with zeros, _merged_bias)
        if layer.global_model_type == 'Performer':
            layer._xgroup = LinearGroup(weights=[lm.A.weight, lm.B.weight, lm.D.weight, lm.E.weight,
                                                 sa.to_q.weight, sa.to_k.weight, sa.to_v.weight],
                                        biases=[lm.A.bias, lm.B.bias, lm.D.bias, lm.E.bias])
        else:
            layer._xgroup = LinearGroup(weights=[lm.A.weight, lm.B.weight, lm.D.weight, lm.E.weight,
                                                 sa.in_proj_weight],
                                        biases=[lm.A.bias, lm.B.bias, lm.D.bias, lm.E.bias,
                                                sa.in_proj_bias])
    return layer._xgroup._stacked()


def _merged_bias(layer, wcat, bcat):
    """Bias row of the merged projection.  Transformer: the stacked biases themselves.  Performer: A..E's biases followed
    by zeros for q | k | v (performer_layer.py:436: qkv_bias=False), a per-layer buffer whose head is refreshed from the
    parameters once per step (the optimizer moves them)."""
    if bcat.shape[0] == wcat.shape[0]:
        return bcat
    buf = layer.__dict__.get("_pbias")
    if buf is None or buf.shape[0] != wcat.shape[0] or buf.device != wcat.device:
        buf = layer.__dict__["_pbias"] = torch.zeros(wcat.shape[0], dtype=torch.float32, device=wcat.device)
    buf[:bcat.shape[0]].copy_(bcat)
    return buf


def stack_begin(layers, batch) -> bool:
    """Called by the network in front of its layer stack (training, gradients on, CUDA).  Returns whether stack_end is
    due."""
    x = getattr(batch, "x", None)
    if not (_STACK_PREP and torch.is_tensor(x) and x.is_cuda and torch.is_grad_enabled()) or _STACK["active"]:
        return False
    _STACK["active"], _STACK["nbt"] = True, []
    _STACK.pop("words", None)
    _STACK.pop("pool", None)
    try:
        _STACK["owners"] = _presplit_stack(layers, x)
        if _STACK["owners"] and _gemm.F16:
            # the max|.| records of every block of the stack (fp16-form GEMMs): ONE zero-fill for all layers (a fill per
            # layer and pass was 20 tiny launches, ~0.1 ms, per step)
            _STACK["pool"] = [_gemm.amax_records(_N_REC * len(_STACK["owners"]), x.device), 0]
    except BaseException:
        _STACK["active"] = False            # never leave the bracket half open: blocks would defer their counters forever
        raise
    return True


def _presplit_stack(layers, x):
    weights, owners = [], []
    for layer in layers:
        if not (getattr(layer, "training", False) and getattr(layer, "local_gnn_type", None) == 'CustomGatedGCN'
                and getattr(layer, "global_model_type", None) in ('Transformer', 'Performer') and block_supported(layer, x)):
            continue
        d = layer.dim_h
        if not _panel_ok(layer, d):
            continue
        wcat, _ = _ensure_xgroup(layer)
        R = _refs(layer)
        weights += [wcat, _W(R.C), _W(R.out_proj), _W(R.ff1), _W(R.ff2)]
        owners.append(layer)
    if owners:
        imgs = _gemm.split_weights(weights)
        for i, layer in enumerate(owners):
            layer.__dict__["_presplit"] = imgs[5 * i:5 * i + 5]
    return owners


def _inner(layer, d) -> int:
    """Width of q, k and v each: d for nn.MultiheadAttention, 64 * heads for the Performer (dim_head defaults to 64
    whatever dim_h / heads is, performer_layer.py:427,441-442)."""
    return layer.self_attn.to_q.weight.shape[0] if layer.global_model_type == 'Performer' else d


def _panel_ok(layer, d) -> bool:
    inner = _inner(layer, d)
    return (_gemm.supported(d, d) and _gemm.supported(2 * d, d) and _gemm.supported(4 * d + 3 * inner, d)
            and _gemm.supported(d, inner) and _gemm.supported(inner, d) and _gemm.supported(d, 4 * d + 3 * inner))


_CHECK_TICKS = _os.environ.get("GPS_CHECK_TICKS", "0") != "0"


def stack_end() -> None:
    nbt, _STACK["nbt"] = _STACK["nbt"], []
    _STACK["active"] = False
    if _CHECK_TICKS and not torch.cuda.is_current_stream_capturing():
        # debugging aid: every arrival counter of every block is zero between launches (a host sync per stack)
        for layer in _STACK.get("owners", []):
            sa = getattr(layer, "_gps_sync", None)
            if sa is not None and sa.nonzero_words():
                raise _lib.GpsHipError("arrival counters of an in-launch reduction are non-zero between launches "
                                       "(csrc/col_tree.hpp): a previous launch did not complete")
    _STACK.pop("words", None)
    _STACK.pop("pool", None)
    for layer in _STACK.pop("owners", []):
        layer.__dict__.pop("_presplit", None)      # (a layer that did not take the block path this time)
    if nbt:
        torch._foreach_add_(nbt, 1)

# max|.| records of one block (fp16-form GEMMs): forward x, e, o, h, t, out, e1 | backward g_f2, g_f1, g_ao, g_pq, g_ce
_N_REC = 16
_R_X, _R_E, _R_O, _R_H, _R_T, _R_OUT, _R_E1, _R_GF2, _R_GF1, _R_GAO, _R_GPQ, _R_GCE = 0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12


_stats_words_cache = {}


def _stats_words(L, N: int) -> int:
    """Counter words the statistics epilogue of an N-column ring GEMM needs (checked against the site size)."""
    v = _stats_words_cache.get(N)
    if v is None:
        v = _stats_words_cache[N] = int(L.gps_gemm_stats_sync_words(N))
    return v


def _block_records(dev):
    """The 16 zeroed records of one block: a slice of the stack's pool, or an allocation of its own."""
    pool = _STACK.get("pool") if _STACK["active"] else None
    if pool is not None and pool[1] + _N_REC <= pool[0].shape[0]:
        rec = pool[0][pool[1]:pool[1] + _N_REC]
        pool[1] += _N_REC
        return rec
    return _gemm.amax_records(_N_REC, dev)


# launch sites of one layer that own arrival counters (norm.SyncArena.site): sites that may be in flight together differ
_S_GG, _S_AO, _S_XE, _S_MID, _S_Z2, _S_B1, _S_B3, _S_B4 = range(8)


class _K:
    """Raw (non-autograd) launch helpers; every call enqueues on torch's current stream."""

    @staticmethod
    def bn_apply(L, z, mean, rstd, bn, res, relu, p, seed, st):
        R, d = z.shape
        y = torch.empty_like(z)
        check(L.gps_bn_apply(ptr(z), ptr(mean), ptr(rstd), ptr(bn.weight), ptr(bn.bias), ptr(res), R, d,
                             int(relu), p, seed, ptr(y), st), "gps_bn_apply")
        return y

    @staticmethod
    def act_drop_add(L, a, b, relu, p, seed, st):
        R, d = b.shape
        out = torch.empty_like(b)
        check(L.gps_act_drop_add(ptr(a), ptr(b), R, d, int(relu), p, seed, ptr(out), st),
              "gps_act_drop_add")
        return out

    @staticmethod
    def act_drop_bwd(L, g, pre, relu, p, seed, st):
        R, d = g.shape
        out = torch.empty_like(g)
        check(L.gps_act_drop_bwd(ptr(g), ptr(pre), R, d, int(relu), p, seed, ptr(out), st),
              "gps_act_drop_bwd")
        return out

    @staticmethod
    def param_grads(L, g, x, params=()):
        """(g^T x, colsum(g)) on the side stream (see fused.py); on the main stream when a target
        parameter already holds a ``.grad`` (gradient accumulation, see ``_accumulating``)."""
        dev = g.device

        def compute():
            R, d = g.shape
            k = x.shape[1]
            g_w = _E(d, k, dtype=g.dtype, device=dev)
            g_b = _E(d, dtype=g.dtype, device=dev)
            ws = _E(max(L.gps_wgrad_workspace_floats(R, d, k), 4), dtype=g.dtype, device=dev)
            check(L.gps_wgrad(ptr(g), g.stride(0), ptr(x), x.stride(0), R, d, k, ptr(g_w), ptr(g_b),
                              ptr(ws), current_stream(dev)), "gps_wgrad")
            return g_w, g_b

        if not _SIDE_ENABLED or _accumulating(params):
            return compute()
        cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = compute()
        g.record_stream(side)
        x.record_stream(side)
        _queue_join(dev)
        return out


def _accumulating(params) -> bool:
    """True when autograd will ACCUMULATE into an existing ``.grad`` of one of ``params`` right after this
    backward node returns (``train_epoch`` with ``optim.batch_accumulation`` > 1, custom_train.py:29-38: from the
    second micro-batch on).  ``AccumulateGrad`` runs ``p.grad.add_(g)`` on the MAIN stream as soon as the node
    returns, i.e. before the end-of-backward join with the side stream, so in that regime the weight gradients
    have to be produced on the main stream (same rule as ``fused._param_grads``)."""
    for p in params:
        if p is not None and p.grad is not None:
            return True
    return False


def _grouped_param_grads(L, pairs, params=(), targets=None, words=None):
    """[(g, x), ...] -> [(g^T x, colsum(g)), ...]: all weight/bias gradients of the block in ONE
    split-K MFMA launch + one reduce launch (csrc/wgrad.hip grouped form) on the side stream
    (main stream when ``params`` already hold gradients: see ``_accumulating``).
    ``targets`` = [(weight, bias), ...] the (stacked) parameters the results belong to: when they live in an optimizer
    arena and nothing is being accumulated the kernel writes straight into their gradient slots (optim.grad_slot).
    ``words`` = [(max|g| record, max|x| record), ...] (int32 [512] tensors, gemm.absmax): the fp16 form of the contraction."""
    dev = pairs[0][0].device
    n = len(pairs)
    direct = targets is not None and not _accumulating(params)

    def compute():
        from ..optim import grad_slot
        probs = (_lib.WgradProblem * n)()
        outs = []
        for i, (q, (g, x)) in enumerate(zip(probs, pairs)):
            R, M = g.shape
            Nn = x.shape[1]
            g_w = g_b = None
            if direct:
                tw, tb = targets[i]
                g_w = grad_slot(tw) if tw is not None else None
                g_b = grad_slot(tb) if tb is not None else None
            if g_w is None or g_w.shape != (M, Nn):
                g_w = _E(M, Nn, dtype=g.dtype, device=dev)
            if g_b is None or g_b.shape != (M,):
                g_b = _E(M, dtype=g.dtype, device=dev)
            q.g, q.x, q.gw, q.gb = g.data_ptr(), x.data_ptr(), g_w.data_ptr(), g_b.data_ptr()
            q.ldg, q.ldx, q.R, q.M, q.Nn = g.stride(0), x.stride(0), R, M, Nn
            if words is not None:
                q.g_amax, q.x_amax = words[i][0].data_ptr(), words[i][1].data_ptr()
            outs.append((g_w, g_b))
        ws = _E(max(L.gps_wgrad_grouped_workspace_floats(n, probs), 4), dtype=torch.float32, device=dev)
        check(L.gps_wgrad_grouped(n, probs, ptr(ws), current_stream(dev)), "gps_wgrad_grouped")
        return outs

    if not _SIDE_ENABLED or _accumulating(params):
        return compute()
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        outs = compute()
    for g, x in pairs:
        g.record_stream(side)
        x.record_stream(side)
    _queue_join(dev)
    return outs


_GROUPED_WGRAD = _os.environ.get("GPS_WGRAD_GROUPED", "1") != "0"
# GPS_WGRAD_F16=0: the weight gradients stay on the 3 x bf16 / 6-product form while the ring GEMMs take the fp16 form (A/B)
_WGRAD_F16 = _os.environ.get("GPS_WGRAD_F16", "1") != "0"

# The attention half (attention core + out-projection GEMM) and the local half (GatedGCN core) of a block only meet at
# the norm stage, so the attention half CAN run on its own HIP stream (a fork / join inside a captured graph).  Round 1-2
# took that fork while capturing.  Round 3 measured it again on the fused step (tools/runs/gpu_r3q.sh, same box, ms per
# step): forked 9.92, serial 9.95 -- nothing, because what overlaps does not run concurrently: a ring-GEMM workgroup owns
# its CU's whole LDS, so the GatedGCN workgroups (40-140 KB of LDS) only get the CUs the GEMM has not taken and both
# kernels stretch (GatedGCN forward 26 -> 71 us, backward 52 -> 89 us, block attention backward 54 -> 70 us) while every
# fork and join costs ~10 us of queue latency in the replayed graph.  Default since round 3: ONE stream (= 0); the
# per-kernel durations inside the step are then the kernels' own.  GPS_BRANCH_STREAM=1 forks while capturing, =2 always.
_BRANCH = _os.environ.get("GPS_BRANCH_STREAM", "0")
# Round 5: a NARROWER fork (GPS_CORE_FORK=1, Transformer block, one-stream mode only): just the attention CORE kernel runs on
# the branch stream, beside the GatedGCN core on the main stream -- the two kernels of a block that are not GEMMs, need
# little or no LDS between them and bound on different things (the attention core: VALU issue and its own load -> compute
# -> store lockstep; the GatedGCN core: HBM).  Every GEMM stays on the main stream, so no ring-GEMM workgroup ever
# competes with a forked kernel for a CU's LDS (what made the round-3 fork of the whole attention half useless).
# The backward pair only co-resides when the GatedGCN backward's LDS stash leaves room for an attention workgroup
# (GPS_GG_STASH_KB <= ~80 next to the 45 KB of k_sattn_bwd).
# Default "bwd": the backward pair only (k_sattn_bwd beside k_gatedgcn_bwd: ~120 us serial -> ~85 us) -- the forward pair
# gains nothing measurable (the two ~25 us kernels overlap for ~18 us and the fork / join costs ~16 us of queue latency:
# profiles/r05_kernel_trace_stats_pcqm4m_corefork.txt) and forking it would only blur the in-step duration of the GatedGCN
# forward, the kernel bench.py's `roofline` block is quoted on.  "1" = both passes, "0" = never.
_CORE_FORK_MODE = _os.environ.get("GPS_CORE_FORK", "bwd")
_CORE_FORK = _CORE_FORK_MODE != "0"
_CORE_FORK_FWD = _CORE_FORK_MODE == "1"
_branch_streams = {}


def _branch_stream(dev):
    st = _branch_streams.get(dev.index)
    if st is None:
        st = _branch_streams[dev.index] = torch.cuda.Stream(device=dev)
    return st


class _Fork:
    """``with _Fork(dev) as f:`` runs the body on the branch stream after the current stream's work so
    far; ``f.join(*tensors)`` makes the current stream wait for it (tensors allocated inside are handed
    over to the current stream's allocator bookkeeping)."""

    def __init__(self, dev, mode):
        self.dev = dev
        self.enabled = mode == "2" or (mode == "1" and torch.cuda.is_current_stream_capturing())

    def __enter__(self):
        if self.enabled:
            self.cur = torch.cuda.current_stream(self.dev)
            self.br = _branch_stream(self.dev)
            self.br.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.br)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.enabled:
            self.cur.wait_stream(self.br)
            # Eagerly the caching allocator has to be told that the joined stream reads these blocks.  While CAPTURING they
            # come from the graph's private pool, where a block only ever returns to the free list of the stream that
            # allocated it -- the branch stream, whose next use is ordered behind the next fork -- so the hand-over is
            # implied; and record_stream inside a capture makes the allocator record pooled events at capture_end, which on
            # this runtime (ROCm 7.0.2 under torch 2.10) segfaults in hipStreamEndCapture whenever such an event was ever
            # recorded on another stream before (tools/capture_probe.py: any eager step on loader-staged tensors).
            if _os.environ.get("GPS_FORK_RECORD_STREAM", "0") == "1" or not torch.cuda.is_current_stream_capturing():
                for t in tensors:
                    t.record_stream(self.cur)


_bn_desc = _norm.bn_desc


class _GPSBlock(torch.autograd.Function):
    """Kernel sequence of one block (launch counts for d = 384): 5 GEMMs + GatedGCN + attention + 10
    norm/residual/dropout launches forward; 5 dgrad GEMMs + 1 grouped wgrad (+ reduce) + GatedGCN (2) +
    attention (3) + 10 norm launches backward.  csrc/block_norm.hip documents the merged stages."""

    @staticmethod
    def forward(ctx, x, e, layer, gi: GraphIndex, seed: int, *params):
        # ``params`` (the layer's leaf parameters, fixed order below) are inputs only so that
        # autograd routes the returned gradients to them; the values are read off the modules.
        L = _lib.load()
        dev = x.device
        st = current_stream(dev)
        R = _refs(layer)
        lm, sa = R.lm, R.sa
        N, d = x.shape
        E = e.shape[0]
        H = layer.num_heads
        perf = R.perf
        inner = _inner(layer, d)            # width of q, k, v each (Performer: 64 * heads)
        dh = inner // H
        p = float(lm.dropout)
        p_l = float(R.drop_attn.p)
        p_f1, p_f2 = float(R.ff_drop1.p), float(R.ff_drop2.p)
        p_at = float(layer.attn_dropout)
        if perf:
            # Performer: dropout(attn_dropout) on the OUTPUT of to_out (performer_layer.py:500-503), then GPSLayer's
            # dropout_attn (gps_layer.py:212): two independent Bernoulli masks with their 1/(1-p) scalings ARE one mask
            # with keep probability (1 - p_at)(1 - p_l) and its scaling -- the same distribution, one hash per element
            p_l = 1.0 - (1.0 - float(sa.dropout.p)) * (1.0 - p_l)
            p_at = 0.0
        s = [(seed + 0x9E3779B97F4A7C15 * (i + 1)) & 0xFFFFFFFFFFFFFFFF for i in range(7)]
        f32 = dict(dtype=torch.float32, device=dev)

        # -- one GEMM for everything that consumes the layer input x: Ax|Bx|Dx|Ex (gatedgcn_layer.py:
        # 57-61) and the attention in-projection q|k|v (gps_layer.py:238): [N,d] x [d,7d]
        wcat, bcat = layer._xgroup._stacked()
        bias_m = _merged_bias(layer, wcat, bcat)
        ldp = 4 * d + 3 * inner
        # dense side: the row-panel GEMM (csrc/gemm_panel.hip) or the library GEMMs
        panel = _panel_ok(layer, d) and E >= 1
        imgs = None
        # The C projection of the edges goes FIRST: it depends on nothing but e, and with it out of the way the two halves
        # of the block that fork after the merged projection are balanced -- [attention core -> out-projection] beside
        # [GatedGCN core] -- instead of [attention, out-projection] beside [C projection -> GatedGCN core].
        if panel:       # weight images of the block's five projections (W and W^T): made for the whole stack at once
            imgs = layer.__dict__.pop("_presplit", None)        # (stack_begin), else ONE launch per layer and step
            if imgs is None:
                imgs = _gemm.split_weights([wcat, _W(R.C), _W(R.out_proj), _W(R.ff1),
                                            _W(R.ff2)])
            # fp16 form of the ring GEMM (gemm.F16): the word of max|A| of every GEMM operand of this layer, made once per
            # tensor (one batched launch where two operands are ready together) and shared by the GEMMs that read it
            am = rec = None
            if imgs[0][0].amax is not None:
                # records (_R_*): x, e, o, h, t; out, e1 = this layer's outputs (the next layer's x and e); the backward's.
                # x / e arrive with their records when the previous block of the stack produced them (its norm tasks
                # tracked the maxima), otherwise one pre-pass makes them; o: a pre-pass; h, t, the outputs: by their producers
                rec = _block_records(dev)
                am = [rec[i] for i in range(7)]
                handed = _STACK.pop("words", None) if _STACK["active"] else None
                if handed is not None and handed[0] == x.data_ptr() and handed[1] == e.data_ptr():
                    am[_R_X], am[_R_E] = handed[2], handed[3]
                else:
                    _gemm.absmax([x, e], out=rec[0:2])
            aw = (lambda i: None) if am is None else (lambda i: am[i])
            if am is not None and _GEMM_PAIR:      # one dispatch for the two projections that depend on nothing but x / e
                pq, ce = _gemm.gemm_panel_pair(dict(a=x, image=imgs[0][0], N=ldp, bias=bias_m, a_amax=am[0]),
                                               dict(a=e, image=imgs[1][0], N=d, bias=_B(R.C), a_amax=am[1]))
            else:
                ce = _gemm.gemm_panel(e, imgs[1][0], d, bias=_B(R.C), a_amax=aw(1))
                pq = _gemm.gemm_panel(x, imgs[0][0], ldp, bias=bias_m, a_amax=aw(0))
        else:
            am, rec, aw = None, None, (lambda i: None)
            ce = torch.addmm(_B(R.C), e, _W(R.C).t())
            pq = torch.addmm(bias_m, x, wcat.t())               # [N, 4d + 3 inner]
        P, fs = pq.data_ptr(), d * 4
        # -- the five BatchNorms: descriptors over one [10, d] statistics buffer ----------------------------
        stats = _E(10, d, **f32)                                # (mean, rstd) x 5
        bnx = _bn_desc(R.bnx, stats[0], stats[1])
        bne = _bn_desc(R.bne, stats[2], stats[3])
        bnl = _bn_desc(R.bnl, stats[4], stats[5])
        bna = _bn_desc(R.bna, stats[6], stats[7])
        bn2 = _bn_desc(R.bn2, stats[8], stats[9])
        ref = _BY_REF
        sync = _norm.sync_arena(layer, dev)
        rn, re_ = gi.n_real, gi.e_real       # padded batches: device words with the real row counts (else None)
        if rn is not None and (not panel or imgs[0][0].amax is None):
            raise _lib.GpsHipError("padded batches need the default block path (fp16-form ring GEMMs)")
        # (round 5: the Performer block takes padded batches too -- FAVOR+ is per graph, its Nmax is taken over the REAL
        # graphs (ops._nmax_dev / gi.b_real), every BatchNorm task and statistics epilogue below counts real rows)
        gemm_stats = panel and _GEMM_STATS and _gemm.stats_supported(N, d, inner) and _gemm.stats_supported(N, d, 2 * d)
        # -- local branch: GatedGCN core ---------------------------------------------------------
        def local_half():
            xt, eh = _E(N, d, **f32), _E(E, d, **f32)
            if _GG_STATS and d % 8 == 0 and N >= 2 and E >= 2:
                # the statistics of x~ (bn_node_x) and e^ (bn_edge_e) fall out of the same call: records per node block
                # from the forward, combined by a second small launch (csrc/gatedgcn.hip k_gg_stats_finalize)
                wsf = L.gps_gatedgcn_stats_floats(N, d)
                gws = _E(wsf, **f32)
                check(L.gps_gatedgcn_fwd_stats(P, P + fs, P + 2 * fs, P + 3 * fs, ldp, ptr(ce), ptr(gi.rowptr_dst),
                                               ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh),
                                               None, ref(bnx), ref(bne), ptr(gws), wsf, ptr(rn), st),
                      "gps_gatedgcn_fwd_stats")
            else:
                check(L.gps_gatedgcn_fwd(P, P + fs, P + 2 * fs, P + 3 * fs, ldp, ptr(ce), ptr(gi.rowptr_dst),
                                         ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh),
                                         None, st), "gps_gatedgcn_fwd")
                _norm.fwd([_norm.fwd_task(_norm.LOAD, xt, N, stats=bnx, rdev=rn),
                           _norm.fwd_task(_norm.LOAD, eh, E, stats=bne, rdev=re_)], d, dev, sync.site(_S_XE))
            return xt, eh

        core_fork = _CORE_FORK_FWD and _BRANCH == "0" and not perf
        # -- global branch (forked): varlen attention over the PRE-layer x (gps_layer.py:199-201,234-241)
        with _Fork(dev, "2" if core_fork else _BRANCH) as fork:
            sb = current_stream(dev)
            o, lse = _E(N, inner, **f32), _E(H, N, **f32)       # (Performer: `lse` holds the query row maxima mq)
            scale = float(dh) ** -0.5
            fav = None
            if perf:    # FAVOR+ over the ptr segments (csrc/favor.hip; performer_layer.py:119-144,200-205)
                proj = sa.fast_attention.projection_matrix
                BH = gi.B * H
                fav = (proj, _E(BH, 272, dh, **f32), _E(BH, 272, **f32), _E(BH, dtype=torch.int64, device=dev),
                       _E(H, N, **f32))                         # projection, ctx, ksum, kmax, D
                wsf = favor_workspace_floats(N, gi.B, H)
                fws = _E(wsf, **f32) if wsf else None
                check(L.gps_favor_fwd(P + 4 * fs, ldp, ptr(proj), proj.shape[0], ptr(gi.ptr), ptr(_nmax_dev(gi)),
                                      ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, gi.B, H, dh, ptr(o),
                                      ptr(fav[1]), ptr(fav[2]), ptr(fav[3]), ptr(lse), ptr(fav[4]), ptr(fws), wsf, sb),
                      "gps_favor_fwd")
            else:
                check(L.gps_seg_attn_fwd(P + 4 * fs, ldp, ptr(gi.ptr), ptr(gi.tile_graph), ptr(gi.tile_row0),
                                         gi.max_tiles, N, H, dh, scale, p_at, s[2], ptr(o), ptr(lse),
                                         gi.B, int(gi.nmax_host), ptr(aw(_R_O)), ptr(gi.attn_order(H)), sb), "gps_seg_attn_fwd")
            if am is not None and perf:         # (the attention kernel raised o's record itself; FAVOR+ does not)
                _gemm.absmax([o], out=rec[_R_O:_R_O + 1])

            def out_projection():
                if gemm_stats:  # za = x + drop(out_proj(o)) and the statistics of za (norm1_attn) in the GEMM's epilogue
                    return _gemm.gemm_panel_stats(o, imgs[2][0], d, _B(R.out_proj), x, p_l, s[3], bna,
                                                  sync.site(_S_AO, _stats_words(L, d)), a_amax=aw(2), m_dev=rn), None
                return None, (_gemm.gemm_panel(o, imgs[2][0], d, bias=_B(R.out_proj), a_amax=aw(2)) if panel
                              else torch.addmm(_B(R.out_proj), o, _W(R.out_proj).t()))
            if not core_fork:
                za, ao = out_projection()
        if core_fork:           # the GatedGCN core beside the attention core; the out-projection GEMM behind the join
            xt, eh = local_half()
            fork.join(o, lse)
            fork.enabled = False
            za, ao = out_projection()
        else:
            xt, eh = local_half()
        # -- x1 = x + drop(relu(BN_x(xt))) [+ statistics -> norm1_local], e1 = e + drop(relu(BN_e(eh))),
        #    za = x + drop(ao) [+ statistics -> norm1_attn] unless the out-projection already produced it -- in which case
        #    this launch needs nothing from the attention half and the join moves behind it (a cross-stream dependency
        #    costs ~10 us of queue latency in a replayed graph even when it is long satisfied; here it hides under this launch)
        x1, e1 = _E(N, d, **f32), _E(E, d, **f32)
        mid = [_norm.fwd_task(_norm.BN_ACT, xt, N, res=x, bn1=bnx, relu=True, p=p, seed=s[0], out=x1, stats=bnl, rdev=rn),
               _norm.fwd_task(_norm.BN_ACT, eh, E, res=e, bn1=bne, relu=True, p=p, seed=s[1], out=e1, amax=aw(6))]
        if za is None:
            fork.join(o, lse, ao)
            za = _E(N, d, **f32)
            mid.append(_norm.fwd_task(_norm.ADD_DROP, x, N, b=ao, p=p_l, seed=s[3], out=za, stats=bna, rdev=rn))
        _norm.fwd(mid, d, dev, sync.site(_S_MID))
        if ao is None:
            fork.join(o, lse, za)
        h = _E(N, d, **f32)                                     # BN_l(x1) + BN_a(za)  (gps_layer.py:222)
        _norm.fwd([_norm.fwd_task(_norm.BN_DUAL, x1, N, b=za, bn1=bnl, bn2=bna, out=h, amax=aw(3))], d, dev, None)

        # -- FFN + norm2 (gps_layer.py:225-229,253-257) ----------------------------------------
        if panel:       # t = drop(relu(ff1(h))) in the GEMM's epilogue: f1 is never materialised
            f1 = None
            t = _gemm.gemm_panel(h, imgs[3][0], 2 * d, bias=_B(R.ff1), epilogue=1, p_drop=p_f1, seed=s[4], a_amax=aw(3),
                                 c_amax=aw(4))
        else:
            f1 = torch.addmm(_B(R.ff1), h, _W(R.ff1).t())
            t = _K.act_drop_add(L, None, f1, True, p_f1, s[4], st)
        if gemm_stats:  # z2 = h + drop(ff2(t)) and the statistics of z2 (norm2) in the GEMM's epilogue
            z2 = _gemm.gemm_panel_stats(t, imgs[4][0], d, _B(R.ff2), h, p_f2, s[5], bn2, sync.site(_S_Z2, _stats_words(L, d)),
                                        a_amax=aw(4), m_dev=rn)
        else:
            f2 = (_gemm.gemm_panel(t, imgs[4][0], d, bias=_B(R.ff2), a_amax=aw(4)) if panel
                  else torch.addmm(_B(R.ff2), t, _W(R.ff2).t()))
            z2 = _E(N, d, **f32)                                # h + drop(f2) and its statistics
            _norm.fwd([_norm.fwd_task(_norm.ADD_DROP, h, N, b=f2, p=p_f2, seed=s[5], out=z2, stats=bn2, rdev=rn)], d, dev,
                      sync.site(_S_Z2))
        out = _E(N, d, **f32)
        _norm.fwd([_norm.fwd_task(_norm.BN_ACT, z2, N, bn1=bn2, out=out, amax=aw(5))], d, dev, None)
        if am is not None and _STACK["active"]:        # hand the outputs' words to the next block of the stack
            _STACK["words"] = (out.data_ptr(), e1.data_ptr(), am[5], am[6])
        _count_batches([R.bnx._buffers["num_batches_tracked"], R.bne._buffers["num_batches_tracked"],
                        R.bnl._buffers["num_batches_tracked"], R.bna._buffers["num_batches_tracked"],
                        R.bn2._buffers["num_batches_tracked"]])

        ctx.save_for_backward(x, e, pq, eh, xt, x1, o, lse, za, h, f1 if f1 is not None else t, t, z2, stats,
                              *(fav if fav is not None else ()))
        ctx.layer, ctx.gi, ctx.seeds = layer, gi, s
        ctx.imgs = imgs     # W^T images for the input-gradient GEMMs (None: library GEMMs)
        ctx.am = am         # fp16 form: the max|.| records of x, e, o, h, t (the weight gradients' second operands)
        ctx.bm = None if rec is None else rec[_R_GF2:_R_GF2 + 5]        # ... and the (zeroed) records of the backward's
        ctx.cfg = (p, p_l, p_f1, p_f2, p_at, H, dh, scale)
        return out, e1

    @staticmethod
    def backward(ctx, g_out, g_e1):
        L = _lib.load()
        x, e, pq, eh, xt, x1, o, lse, za, h, f1, t, z2, stats, *fav = ctx.saved_tensors
        layer, gi, s = ctx.layer, ctx.gi, ctx.seeds
        p, p_l, p_f1, p_f2, p_at, H, dh, scale = ctx.cfg
        R = _refs(layer)
        lm, sa = R.lm, R.sa
        dev = x.device
        st = current_stream(dev)
        N, d = x.shape
        E = e.shape[0]
        perf = R.perf
        inner = o.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        g_out = g_out.contiguous()
        g_e1 = g_e1.contiguous() if g_e1 is not None else torch.zeros(E, d, **f32)
        bnx = _bn_desc(R.bnx, stats[0], stats[1])
        bne = _bn_desc(R.bne, stats[2], stats[3])
        bnl = _bn_desc(R.bnl, stats[4], stats[5])
        bna = _bn_desc(R.bna, stats[6], stats[7])
        bn2 = _bn_desc(R.bn2, stats[8], stats[9])
        sync = _norm.sync_arena(layer, dev)
        rn, re_ = gi.n_real, gi.e_real       # padded batches: 1/R_real and a zero gate on the padding rows in every apply
        gpar = _E(10, d, **f32)              # (g_gamma, g_beta) of the five norms ...
        g_bxw, g_bxb, g_bew, g_beb, g_nlw, g_nlb, g_naw, g_nab, g_n2w, g_n2b = gpar.unbind(0)
        if not _accumulating(R.params):      # ... or their slots in the optimizer's gradient arena
            from ..optim import grad_slot
            bns = (R.bnx, R.bne, R.bnl, R.bna, R.bn2)
            slots = [grad_slot(q) for bn in bns for q in (bn.weight, bn.bias)]
            if all(q is not None for q in slots):
                g_bxw, g_bxb, g_bew, g_beb, g_nlw, g_nlb, g_naw, g_nab, g_n2w, g_n2b = slots

        # norm2 <- z2 = h + drop(f2):  g_z2 and g_f2 = dropmask(g_z2);  bn_edge_e <- e^ (its output gradient g_e1 is an
        # input of this node, so its column sums and its apply ride along): ONE partial launch, ONE apply launch
        imgs = ctx.imgs
        bm = ctx.bm if imgs is not None else None       # fp16 form: records of g_f2, g_f1, g_ao (by their producers), g_pq, g_ce
        g_z2, g_f2, g_eh = _E(N, d, **f32), _E(N, d, **f32), _E(E, d, **f32)
        b1 = [_norm.bwd_task(z2, g_out, bn2, N, g_n2w, g_n2b, g_z=g_z2, g_drop=g_f2, p2=p_f2, seed2=s[5],
                             amax_drop=None if bm is None else bm[0], rdev=rn),
              _norm.bwd_task(eh, g_e1, bne, E, g_bew, g_beb, relu=True, p=p, seed=s[1], g_z=g_eh, rdev=re_)]
        _norm.bwd_partial(b1, d, dev, sync.site(_S_B1))
        # GPS_GG_BN_FOLD (default): bn_edge_e's apply (g_e^ from g_e1) and bn_node_x's (g_x~ from g_x1) are evaluated by the
        # GatedGCN backward while it loads those gradients -- it is their only reader and reads e^ / x~ anyway
        # (gps_gatedgcn_bwd_bn): no g_e^ / g_x~ tensors, the edge task leaves this apply launch, bn_node_x's goes away
        fold = int(_GG_BN_FOLD) if E >= 1 else 0
        _norm.bwd_apply(b1[:1] if fold & 2 else b1, d, dev, None)
        # f2 = ff2(t);  t = drop(relu(f1));  f1 = ff1(h)
        bw = (lambda i: None) if bm is None else (lambda i: bm[i])
        if imgs is not None:
            # g_f1 = relu/dropout mask of t applied to g_f2 W2 (the mask of t is the mask of f1 wherever it matters:
            # a kept element has t > 0 iff f1 > 0, a dropped one has gradient 0 either way), in the GEMM's epilogue
            g_f1 = _gemm.gemm_panel(g_f2, imgs[4][1], 2 * d, epilogue=2, mask_src=t, p_drop=p_f1, seed=s[4], a_amax=bw(0),
                                    c_amax=bw(1))
            # residual + FFN input; g_h is the output gradient of norm1_local(x1) + norm1_attn(za): their column sums leave
            # with it (csrc/gemm_panel.hip epilogue 4) instead of the partial launch below
            dual_sums = _GEMM_BWDSUMS and bm is not None and _gemm.colsums_supported(N, d, 2 * d)
            if dual_sums:
                g_h = _gemm.gemm_panel_sums(dict(a=g_f1, image=imgs[3][1], N=d, addend=g_z2, out=g_z2, a_amax=bw(1)),
                                            dict(z=x1, bn=bnl, sum_g=g_nlb, sum_gz=g_nlw, z2=za, bn2=bna, sum_g2=g_nab,
                                                 sum_gz2=g_naw), sync.site(_S_B3, _stats_words(L, d)))
            else:
                g_h = _gemm.gemm_panel(g_f1, imgs[3][1], d, addend=g_z2, out=g_z2, a_amax=bw(1))
        else:
            dual_sums = False
            g_t = g_f2.mm(_W(R.ff2))
            g_f1 = _K.act_drop_bwd(L, g_t, f1, True, p_f1, s[4], st)
            g_h = g_z2.addmm_(g_f1, _W(R.ff1))              # residual + FFN input

        # h = BN_l(x1) + BN_a(za);  za = x + drop(ao):  g_x1, g_x1 + g_za, g_ao = dropmask(g_za).  The apply CHAINS into
        # bn_node_x (x1 = x + drop(relu(BN_x(xt))): g_x1 is that BatchNorm's output gradient): its column sums come out of
        # the same pass, so bn_node_x needs no partial launch of its own
        g_x1, g_xres, g_ao = _E(N, d, **f32), _E(N, d, **f32), _E(N, d, **f32)
        b3 = [_norm.bwd_task(x1, g_h, bnl, N, g_nlw, g_nlb, z2=za, bn2=bna, g_gamma2=g_naw, g_beta2=g_nab,
                             g_z=g_x1, g_sum=g_xres, g_drop=g_ao, p2=p_l, seed2=s[3],
                             cz=xt, cbn=bnx, crelu=True, cp=p, cseed=s[0], cg_gamma=g_bxw, cg_beta=g_bxb,
                             amax_drop=bw(2), rdev=rn)]
        if not dual_sums:
            _norm.bwd_partial(b3, d, dev, sync.site(_S_B3))
        _norm.bwd_apply(b3, d, dev, sync.site(_S_B4))
        # gradient of the merged projection: attention writes dq|dk|dv into columns 4d.., GatedGCN
        # writes g_Ax|g_Bx|g_Dx|g_Ex into columns 0..4d of ONE [N,7d] buffer -> one dgrad, one wgrad
        ldp = 4 * d + 3 * inner
        fs = d * 4
        g_pq = _E(N, ldp, **f32)
        G, P = g_pq.data_ptr(), pq.data_ptr()
        core_fork = _CORE_FORK and _BRANCH == "0" and not perf

        def bnx_apply():
            # x1 = x + drop(relu(BN_x(xt))):  bn_node_x's apply (its sums came from the chain above)
            g = _E(N, d, **f32)
            _norm.bwd_apply([_norm.bwd_task(xt, g_x1, bnx, N, g_bxw, g_bxb, relu=True, p=p, seed=s[0], g_z=g, rdev=rn)], d,
                            dev, None)
            return g
        g_xt = None
        if core_fork:
            # everything the forked pair needs is made BEFORE the fork, alone on the chip: the 11 us bn_node_x apply ran 33 us
            # beside the attention backward and held the GatedGCN backward back behind it (profiles/r05_timeline_pcqm4m.txt);
            # the out-projection's input gradient stays on the main stream like every GEMM
            if not fold & 1:
                g_xt = bnx_apply()
            g_o = (_gemm.gemm_panel(g_ao, imgs[2][1], inner, a_amax=bw(2)) if imgs is not None
                   else g_ao.mm(_W(R.out_proj)))
        with _Fork(dev, "2" if core_fork else _BRANCH) as fork:            # attention half of the backward
            sb = current_stream(dev)
            if not core_fork:
                g_o = (_gemm.gemm_panel(g_ao, imgs[2][1], inner, a_amax=bw(2)) if imgs is not None
                       else g_ao.mm(_W(R.out_proj)))
            if perf:
                proj, cbuf, ksum, kmax, Dn = fav
                gD, g_ctx, g_ksum = _E(H, N, **f32), torch.empty_like(cbuf), torch.empty_like(ksum)
                gm_part = _E(max(gi.max_tiles * H, 1), **f32)
                wsf = favor_workspace_floats(N, gi.B, H)
                fws = _E(wsf, **f32) if wsf else None
                check(L.gps_favor_bwd(ptr(g_o), P + 4 * fs, ldp, ptr(proj), proj.shape[0], ptr(o), ptr(gi.ptr),
                                      ptr(_nmax_dev(gi)), ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, gi.B,
                                      H, dh, ptr(cbuf), ptr(ksum), ptr(kmax), ptr(lse), ptr(Dn), ptr(gD), ptr(g_ctx),
                                      ptr(g_ksum), ptr(gm_part), G + 4 * fs, ldp, ptr(fws), wsf, sb), "gps_favor_bwd")
            else:
                delta = _E(H, N, **f32)
                check(L.gps_seg_attn_bwd(ptr(g_o), P + 4 * fs, ldp, ptr(o), ptr(lse), ptr(gi.ptr),
                                         ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh, scale,
                                         p_at, s[2], ptr(delta), G + 4 * fs, ldp, gi.B, int(gi.nmax_host), ptr(bw(3)),
                                         ptr(gi.attn_order(H)), sb), "gps_seg_attn_bwd")

        g_ce = _E(E, d, **f32)
        if fold:
            fx = fe = None
            if fold & 1:
                fx = _BY_REF(_lib.BnBwdFold(_ctypes.addressof(bnx), g_bxb.data_ptr(), g_bxw.data_ptr(), p, s[0], 1, ptr(rn)))
            elif g_xt is None:
                g_xt = bnx_apply()
            if fold & 2:
                fe = _BY_REF(_lib.BnBwdFold(_ctypes.addressof(bne), g_beb.data_ptr(), g_bew.data_ptr(), p, s[1], 1, ptr(re_)))
            check(L.gps_gatedgcn_bwd_bn(ptr(g_x1 if fold & 1 else g_xt), d, ptr(g_e1 if fold & 2 else g_eh), ptr(eh), P, P + fs,
                                        ldp, ptr(xt), ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                                        ptr(gi.rowptr_src), ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, d,
                                        ptr(g_ce), G, G + fs, G + 2 * fs, G + 3 * fs, ldp, None, ptr(bw(3)), ptr(bw(4)),
                                        fx, fe, st), "gps_gatedgcn_bwd_bn")
        else:
            if g_xt is None:
                g_xt = bnx_apply()
            check(L.gps_gatedgcn_bwd(ptr(g_xt), d, ptr(g_eh), ptr(eh), P, P + fs, ldp, ptr(xt),
                                     ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                                     ptr(gi.rowptr_src), ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, d,
                                     ptr(g_ce), G, G + fs, G + 2 * fs, G + 3 * fs, ldp, None, ptr(bw(3)), ptr(bw(4)), st),
                  "gps_gatedgcn_bwd")
        fork.join()
        wcat, bcat = layer._xgroup._stacked()
        pairs = [(g_pq, x), (g_ce, e), (g_ao, o), (g_f1, h), (g_f2, t)]
        leaves = R.params
        words = None
        if bm is not None:
            # g_pq's record: the GatedGCN backward raised it over its four column blocks (and made g_ce's), the attention
            # backward over dq | dk | dv; FAVOR+ does not track: its columns take one strided pre-pass
            if perf:
                _gemm.absmax([g_pq[:, 4 * d:]], out=bm[3:4])
            am = ctx.am
            if am is not None and _WGRAD_F16:
                words = [(bm[3], am[_R_X]), (bm[4], am[_R_E]), (bm[2], am[_R_O]), (bm[1], am[_R_H]), (bm[0], am[_R_T])]
        if _GROUPED_WGRAD:
            targets = [(wcat, bcat), (_W(R.C), _B(R.C)), (_W(R.out_proj), _B(R.out_proj)),
                       (_W(R.ff1), _B(R.ff1)), (_W(R.ff2), _B(R.ff2))]
            ((g_wcat, g_bcat), (g_wc, g_bc), (g_wo, g_bo), (g_w1, g_b1), (g_w2, g_b2)) = \
                _grouped_param_grads(L, pairs, leaves, targets, words)
        else:
            ((g_wcat, g_bcat), (g_wc, g_bc), (g_wo, g_bo), (g_w1, g_b1), (g_w2, g_b2)) = \
                [_K.param_grads(L, g, a, leaves) for g, a in pairs]
        if imgs is not None and bm is not None and _GEMM_PAIR:
            # the two input gradients of the block in one dispatch, the K = 4d + 3 inner one first
            g_x, g_e = _gemm.gemm_panel_pair(dict(a=g_pq, image=imgs[0][1], N=d, addend=g_xres, out=g_xres, a_amax=bm[3]),
                                             dict(a=g_ce, image=imgs[1][1], N=d, addend=g_e1, a_amax=bm[4]))
        elif imgs is not None:
            g_x = _gemm.gemm_panel(g_pq, imgs[0][1], d, addend=g_xres, out=g_xres, a_amax=bw(3))
            g_e = _gemm.gemm_panel(g_ce, imgs[1][1], d, addend=g_e1, a_amax=bw(4))
        else:
            g_x = g_xres.addmm_(g_pq, wcat)          # residuals of za and x1 + A..E + in-proj inputs
            g_e = torch.addmm(g_e1, g_ce, _W(R.C))   # residual of e1 + C input
        if perf:        # to_q | to_k | to_v rows of the stacked gradient (no bias: the zero columns' sums are dropped)
            inproj = [g_wcat[4 * d + i * inner:4 * d + (i + 1) * inner] for i in range(3)]
        else:
            inproj = [g_wcat[4 * d:], g_bcat[4 * d:]]

        abde = [g_wcat[i * d:(i + 1) * d] for i in range(4)] + [g_bcat[i * d:(i + 1) * d] for i in range(4)]
        # order must match _Refs.params
        return (g_x, g_e, None, None, None,
                *abde, g_wc, g_bc, g_bxw, g_bxb, g_bew, g_beb, g_nlw, g_nlb,
                *inproj, g_wo, g_bo, g_naw, g_nab, g_w1, g_b1, g_w2, g_b2, g_n2w, g_n2b)


class _GPSBlockGINE(torch.autograd.Function):
    """The same single-node treatment for ``GINE+Transformer`` blocks (zinc-GPS+RWSE.yaml and most
    configs/GPS/*.yaml): GINE core -> its 2-layer MLP, attention, both residual+dropout+norm stages as
    one task list, dual apply, FFN + norm2.  GINE does not update ``edge_attr``
    (graphgps/layer/gps_layer.py:176-189)."""

    @staticmethod
    def forward(ctx, x, e, layer, gi: GraphIndex, seed: int, *params):
        L = _lib.load()
        dev = x.device
        st = current_stream(dev)
        lm, sa = layer.local_model, layer.self_attn
        lin1, lin2 = lm.nn[0], lm.nn[2]
        N, d = x.shape
        E = e.shape[0]
        H = layer.num_heads
        dh = d // H
        p_loc, p_l = float(layer.dropout_local.p), float(layer.dropout_attn.p)
        p_f1, p_f2 = float(layer.ff_dropout1.p), float(layer.ff_dropout2.p)
        p_at = float(layer.attn_dropout)
        s = [(seed + 0x9E3779B97F4A7C15 * (i + 1)) & 0xFFFFFFFFFFFFFFFF for i in range(7)]
        f32 = dict(dtype=torch.float32, device=dev)
        ref = _BY_REF

        # -- attention half (forked while capturing) ------------------------------------------------
        with _Fork(dev, _BRANCH) as fork:
            sb = current_stream(dev)
            qkv = torch.addmm(sa.in_proj_bias, x, sa.in_proj_weight.t())
            o, lse = _E(N, d, **f32), _E(H, N, **f32)
            scale = float(dh) ** -0.5
            check(L.gps_seg_attn_fwd(ptr(qkv), 3 * d, ptr(gi.ptr), ptr(gi.tile_graph), ptr(gi.tile_row0),
                                     gi.max_tiles, N, H, dh, scale, p_at, s[2], ptr(o), ptr(lse),
                                     gi.B, int(gi.nmax_host), None, ptr(gi.attn_order(H)), sb), "gps_seg_attn_fwd")
            ao = torch.addmm(sa.out_proj.bias, o, sa.out_proj.weight.t())
        # -- local half: GINE core + MLP (gps_layer.py:62-69,183-185) -------------------------------
        agg = _E(N, d, **f32)
        check(L.gps_gine_fwd(ptr(x), ptr(e), ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                             N, E, d, float(lm.initial_eps), ptr(agg), None, st), "gps_gine_fwd")
        g1 = torch.addmm(lin1.bias, agg, lin1.weight.t())
        g1r = _K.act_drop_add(L, None, g1, True, 0.0, 0, st)
        g2 = torch.addmm(lin2.bias, g1r, lin2.weight.t())
        fork.join(qkv, o, lse, ao)

        # -- zl = x + drop(g2), za = x + drop(ao) with their statistics; h = BN_l(zl) + BN_a(za) -----
        stats = _E(6, d, **f32)
        bnl = _bn_desc(layer.norm1_local, stats[0], stats[1])
        bna = _bn_desc(layer.norm1_attn, stats[2], stats[3])
        bn2 = _bn_desc(layer.norm2, stats[4], stats[5])
        sync = _norm.sync_arena(layer, dev)
        zl, za = _E(N, d, **f32), _E(N, d, **f32)
        rn = gi.n_real                       # padded batches (round 5): statistics over the real rows (device word)
        _norm.fwd([_norm.fwd_task(_norm.ADD_DROP, x, N, b=g2, p=p_loc, seed=s[0], out=zl, stats=bnl, rdev=rn),
                   _norm.fwd_task(_norm.ADD_DROP, x, N, b=ao, p=p_l, seed=s[3], out=za, stats=bna, rdev=rn)],
                  d, dev, sync.site(_S_MID))
        h = _E(N, d, **f32)
        _norm.fwd([_norm.fwd_task(_norm.BN_DUAL, zl, N, b=za, bn1=bnl, bn2=bna, out=h)], d, dev, None)
        # -- FFN + norm2 --------------------------------------------------------------------------------
        f1 = torch.addmm(layer.ff_linear1.bias, h, layer.ff_linear1.weight.t())
        t = _K.act_drop_add(L, None, f1, True, p_f1, s[4], st)
        f2 = torch.addmm(layer.ff_linear2.bias, t, layer.ff_linear2.weight.t())
        z2 = _E(N, d, **f32)
        _norm.fwd([_norm.fwd_task(_norm.ADD_DROP, h, N, b=f2, p=p_f2, seed=s[5], out=z2, stats=bn2, rdev=rn)], d, dev,
                  sync.site(_S_Z2))
        out = _E(N, d, **f32)
        _norm.fwd([_norm.fwd_task(_norm.BN_ACT, z2, N, bn1=bn2, out=out)], d, dev, None)
        _count_batches([layer.norm1_local.num_batches_tracked, layer.norm1_attn.num_batches_tracked,
                        layer.norm2.num_batches_tracked])
        ctx.save_for_backward(x, e, agg, g1, g1r, qkv, o, lse, zl, za, h, f1, t, z2, stats)
        ctx.layer, ctx.gi, ctx.seeds = layer, gi, s
        ctx.cfg = (p_loc, p_l, p_f1, p_f2, p_at, H, dh, scale)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.load()
        x, e, agg, g1, g1r, qkv, o, lse, zl, za, h, f1, t, z2, stats = ctx.saved_tensors
        layer, gi, s = ctx.layer, ctx.gi, ctx.seeds
        p_loc, p_l, p_f1, p_f2, p_at, H, dh, scale = ctx.cfg
        lm, sa = layer.local_model, layer.self_attn
        lin1, lin2 = lm.nn[0], lm.nn[2]
        dev = x.device
        st = current_stream(dev)
        N, d = x.shape
        E = e.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        ref = _BY_REF
        g_out = g_out.contiguous()
        bnl = _bn_desc(layer.norm1_local, stats[0], stats[1])
        bna = _bn_desc(layer.norm1_attn, stats[2], stats[3])
        bn2 = _bn_desc(layer.norm2, stats[4], stats[5])
        sync = _norm.sync_arena(layer, dev)
        gpar = _E(6, d, **f32)
        g_nlw, g_nlb, g_naw, g_nab, g_n2w, g_n2b = gpar.unbind(0)

        g_z2, g_f2 = _E(N, d, **f32), _E(N, d, **f32)
        rn = gi.n_real                       # padded batches: 1 / R_real, exact zeros on the padding rows of every apply
        b1 = [_norm.bwd_task(z2, g_out, bn2, N, g_n2w, g_n2b, g_z=g_z2, g_drop=g_f2, p2=p_f2, seed2=s[5], rdev=rn)]
        _norm.bwd_partial(b1, d, dev, sync.site(_S_B1))
        _norm.bwd_apply(b1, d, dev, None)
        g_t = g_f2.mm(layer.ff_linear2.weight)
        g_f1 = _K.act_drop_bwd(L, g_t, f1, True, p_f1, s[4], st)
        g_h = g_z2.addmm_(g_f1, layer.ff_linear1.weight)
        # g_g2 = dropmask_local(g_zl), g_xres = g_zl + g_za, g_ao = dropmask_attn(g_za)
        g_g2, g_xres, g_ao = _E(N, d, **f32), _E(N, d, **f32), _E(N, d, **f32)
        b3 = [_norm.bwd_task(zl, g_h, bnl, N, g_nlw, g_nlb, z2=za, bn2=bna, g_gamma2=g_naw, g_beta2=g_nab,
                             g_z=g_g2, p1x=p_loc, seed1x=s[0], g_sum=g_xres, g_drop=g_ao, p2=p_l, seed2=s[3], rdev=rn)]
        _norm.bwd_partial(b3, d, dev, sync.site(_S_B3))
        _norm.bwd_apply(b3, d, dev, None)
        with _Fork(dev, _BRANCH) as fork:            # attention half
            sb = current_stream(dev)
            g_o = g_ao.mm(sa.out_proj.weight)
            g_qkv, delta = _E(N, 3 * d, **f32), _E(H, N, **f32)
            check(L.gps_seg_attn_bwd(ptr(g_o), ptr(qkv), 3 * d, ptr(o), ptr(lse), ptr(gi.ptr),
                                     ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh, scale,
                                     p_at, s[2], ptr(delta), ptr(g_qkv), 3 * d, gi.B, int(gi.nmax_host), None,
                                     ptr(gi.attn_order(H)), sb), "gps_seg_attn_bwd")
        # local half: MLP backward, GINE core backward
        g_g1r = g_g2.mm(lin2.weight)
        g_g1 = _K.act_drop_bwd(L, g_g1r, g1, True, 0.0, 0, st)
        g_agg = g_g1.mm(lin1.weight)
        g_xg, g_e = _E(N, d, **f32), _E(E, d, **f32)
        check(L.gps_gine_bwd(ptr(g_agg), ptr(x), ptr(e), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                             ptr(gi.eid_by_dst), ptr(gi.rowptr_src), ptr(gi.eid_by_src), N, E, d,
                             float(lm.initial_eps), ptr(g_xg), ptr(g_e), None, st), "gps_gine_bwd")
        fork.join(g_qkv)
        pairs = [(g_qkv, x), (g_ao, o), (g_g2, g1r), (g_g1, agg), (g_f1, h), (g_f2, t)]
        leaves = block_params_gine(layer)
        if _GROUPED_WGRAD:
            grads = _grouped_param_grads(L, pairs, leaves)
        else:
            grads = [_K.param_grads(L, g, a, leaves) for g, a in pairs]
        (g_wi, g_bi), (g_wo, g_bo), (g_wl2, g_bl2), (g_wl1, g_bl1), (g_w1, g_b1), (g_w2, g_b2) = grads
        g_x = g_xres.addmm_(g_qkv, sa.in_proj_weight)
        g_x.add_(g_xg)
        # order must match block_params_gine()
        return (g_x, g_e, None, None, None,
                g_wl1, g_bl1, g_wl2, g_bl2, g_nlw, g_nlb, g_wi, g_bi, g_wo, g_bo, g_naw, g_nab,
                g_w1, g_b1, g_w2, g_b2, g_n2w, g_n2b)


def block_params_gine(layer):
    lm, sa = layer.local_model, layer.self_attn
    return [lm.nn[0].weight, lm.nn[0].bias, lm.nn[2].weight, lm.nn[2].bias,
            layer.norm1_local.weight, layer.norm1_local.bias,
            sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
            layer.norm1_attn.weight, layer.norm1_attn.bias,
            layer.ff_linear1.weight, layer.ff_linear1.bias,
            layer.ff_linear2.weight, layer.ff_linear2.bias,
            layer.norm2.weight, layer.norm2.bias]


def gine_block_static_ok(layer) -> bool:
    """The part of gine_block_supported that only depends on how the layer was built."""
    import torch.nn as nn
    lm = layer.local_model
    if layer.local_gnn_type != 'GINE' or layer.global_model_type != 'Transformer' or layer.equivstable_pe:
        return False
    if not layer.batch_norm or not isinstance(layer.act_fn_ff, nn.ReLU):
        return False
    seq = getattr(lm, "nn", None)
    if not (isinstance(seq, nn.Sequential) and len(seq) == 3 and isinstance(seq[0], nn.Linear)
            and isinstance(seq[1], nn.ReLU) and isinstance(seq[2], nn.Linear)):
        return False
    for bn in (layer.norm1_local, layer.norm1_attn, layer.norm2):
        if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
            return False
    return True


def gine_block_supported(layer, x, e=None) -> bool:
    if not (layer.training and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32):
        return False
    if not gine_block_static_ok(layer):
        return False
    d = x.shape[1]
    if d % 4 != 0 or d > 1024 or x.shape[0] < 2 or not x.is_contiguous():
        return False
    return e is None or (e.dim() == 2 and e.shape[1] == d and e.is_contiguous())


def gps_block_gine(layer, x, e, gi):
    return _GPSBlockGINE.apply(x, e, layer, gi, draw_dropout_seed(), *block_params_gine(layer))


def block_params(layer):
    """Leaf parameters of a CustomGatedGCN + Transformer / Performer block in the order ``_GPSBlock.backward`` returns
    their gradients (``_Refs.params``)."""
    return list(_refs(layer).params)


def _block_static_ok(layer) -> bool:
    """The part of block_supported that only depends on how the layer was built (cached on the layer)."""
    import torch.nn as nn
    ok = layer.__dict__.get("_blk_static")
    if ok is None:
        lm = layer.local_model
        glob = layer.global_model_type
        sa = layer.self_attn
        glob_ok = glob == 'Transformer' or (
            glob == 'Performer' and all(hasattr(sa, k) for k in ("to_q", "to_k", "to_v", "to_out", "fast_attention"))
            and sa.to_q.bias is None and sa.to_out.bias is not None
            and sa.to_q.weight.shape[0] == 64 * layer.num_heads                     # csrc/favor.hip: dim_head 64, m <= 272
            and sa.fast_attention.projection_matrix.shape[0] <= 272 and sa.fast_attention.projection_matrix.shape[1] == 64)
        ok = (layer.local_gnn_type == 'CustomGatedGCN' and glob_ok
              and bool(layer.batch_norm) and bool(lm.residual) and not getattr(lm, "EquivStablePE", False)
              and isinstance(lm.act_fn_x, nn.ReLU) and isinstance(lm.act_fn_e, nn.ReLU)
              and isinstance(layer.act_fn_ff, nn.ReLU)
              and all(bn.affine and bn.track_running_stats and bn.momentum is not None
                      for bn in (lm.bn_node_x, lm.bn_edge_e, layer.norm1_local, layer.norm1_attn, layer.norm2)))
        layer.__dict__["_blk_static"] = ok
    return ok


def block_supported(layer, x, e=None) -> bool:
    """The single-node path covers the measured configuration: CustomGatedGCN + Transformer,
    BatchNorm, ReLU, training mode with gradients, fp32 on the GPU."""
    if not (layer.training and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32):
        return False
    if not _block_static_ok(layer):
        return False
    d = x.shape[1]
    if d % 4 != 0 or d > 1024 or x.shape[0] < 2:          # csrc/block_norm.hip row mapping
        return False
    if e is not None and (e.dim() != 2 or e.shape[0] < 2 or e.shape[1] != d or not e.is_contiguous()):
        return False
    return x.is_contiguous()


def gps_block(layer, x, e, gi):
    _ensure_xgroup(layer)
    return _GPSBlock.apply(x, e, layer, gi, draw_dropout_seed(), *_refs(layer).params)


# ---------------------------------------------------------------------------------------------------------------------
# Inference form (graphgps/train/custom_train.py:50-77 eval_epoch, the `PCQM4Mv2-inference` mode: model.eval() under
# no_grad): the same kernels minus everything training needs -- BatchNorms read their running statistics, no dropout, no
# statistics tasks or in-launch trees, nothing saved, no input-gradient images.  13 launches per layer (+ the weight
# images and the two operand maxima).
# ---------------------------------------------------------------------------------------------------------------------
def block_eval_supported(layer, x, e=None) -> bool:
    """CustomGatedGCN + Transformer, BatchNorm with running statistics, ReLU, ``layer.eval()`` with gradients off, fp32 on
    the GPU, widths the ring GEMM tiles."""
    if layer.training or torch.is_grad_enabled() or not (x.is_cuda and x.dtype == torch.float32):
        return False
    if not _block_static_ok(layer) or layer.global_model_type != 'Transformer':
        return False
    d = x.shape[1]
    if d % 4 != 0 or d > 1024 or x.shape[0] < 2 or not x.is_contiguous():
        return False
    if e is None or e.dim() != 2 or e.shape[0] < 2 or e.shape[1] != d or not e.is_contiguous() or e.dtype != torch.float32:
        return False
    return _panel_ok(layer, d)


def _eval_stats(R: _Refs, d: int, dev) -> torch.Tensor:
    """[10, d]: (running_mean, 1 / sqrt(running_var + eps)) of the block's five BatchNorms -- what F.batch_norm uses in
    eval mode (gatedgcn_layer.py:72-73, gps_layer.py:191-194,212-229 under model.eval())."""
    bns = (R.bnx, R.bne, R.bnl, R.bna, R.bn2)
    out = _E(10, d, dtype=torch.float32, device=dev)
    means, rstds = out[0::2], out[1::2]
    means.copy_(torch.stack([bn._buffers["running_mean"] for bn in bns]))
    var = torch.stack([bn._buffers["running_var"] for bn in bns])
    eps = {float(bn.eps) for bn in bns}
    if len(eps) == 1:
        rstds.copy_(torch.rsqrt(var + eps.pop()))
    else:
        for i, bn in enumerate(bns):
            rstds[i].copy_(torch.rsqrt(var[i] + float(bn.eps)))
    return out


@torch.no_grad()
def gps_block_eval(layer, x, e, gi):
    """``(x, e) -> (h, e_new)`` of one layer in eval mode: merged A|B|D|E|q|k|v projection and C projection on the ring
    GEMM, GatedGCN core, attention core (no dropout), out-projection with the residual in its epilogue, the two
    residual + ReLU(BatchNorm) streams, the dual BatchNorm sum, FFN (ReLU in the first GEMM's epilogue, residual in the
    second's) and norm2 -- gps_layer.py:155-232 with every BatchNorm on its running statistics."""
    L = _lib.load()
    dev = x.device
    st = current_stream(dev)
    wcat, bcat = _ensure_xgroup(layer)
    R = _refs(layer)
    N, d = x.shape
    E = e.shape[0]
    H = layer.num_heads
    dh = d // H
    ldp = 7 * d
    fs = d * 4
    f32 = dict(dtype=torch.float32, device=dev)
    imgs = _gemm.split_weights([wcat, _W(R.C), _W(R.out_proj), _W(R.ff1), _W(R.ff2)], tn=False)
    am = None
    if imgs[0][0].amax is not None:             # fp16-form GEMMs: the records of max|operand|, made by the producers
        rec = _gemm.amax_records(_N_REC, dev)
        am = [rec[i] for i in range(7)]
        _gemm.absmax([x, e], out=rec[0:2])
    aw = (lambda i: None) if am is None else (lambda i: am[i])
    ce = _gemm.gemm_panel(e, imgs[1][0], d, bias=_B(R.C), a_amax=aw(_R_E))
    pq = _gemm.gemm_panel(x, imgs[0][0], ldp, bias=_merged_bias(layer, wcat, bcat), a_amax=aw(_R_X))
    P = pq.data_ptr()
    stats = _eval_stats(R, d, dev)
    bnx, bne, bnl, bna, bn2 = (_bn_desc(bn, stats[2 * i], stats[2 * i + 1])
                               for i, bn in enumerate((R.bnx, R.bne, R.bnl, R.bna, R.bn2)))
    xt, eh = _E(N, d, **f32), _E(E, d, **f32)
    check(L.gps_gatedgcn_fwd(P, P + fs, P + 2 * fs, P + 3 * fs, ldp, ptr(ce), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                             ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh), None, st), "gps_gatedgcn_fwd")
    o, lse = _E(N, d, **f32), _E(H, N, **f32)
    check(L.gps_seg_attn_fwd(P + 4 * fs, ldp, ptr(gi.ptr), ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh,
                             float(dh) ** -0.5, 0.0, 0, ptr(o), ptr(lse), gi.B, int(gi.nmax_host), ptr(aw(_R_O)),
                             ptr(gi.attn_order(H)), st), "gps_seg_attn_fwd")
    za = _gemm.gemm_panel(o, imgs[2][0], d, bias=_B(R.out_proj), addend=x, a_amax=aw(_R_O))     # x + out_proj(o)
    x1, e1, h, out = _E(N, d, **f32), _E(E, d, **f32), _E(N, d, **f32), _E(N, d, **f32)
    _norm.fwd([_norm.fwd_task(_norm.BN_ACT, xt, N, res=x, bn1=bnx, relu=True, out=x1),
               _norm.fwd_task(_norm.BN_ACT, eh, E, res=e, bn1=bne, relu=True, out=e1, amax=aw(6))], d, dev, None)
    _norm.fwd([_norm.fwd_task(_norm.BN_DUAL, x1, N, b=za, bn1=bnl, bn2=bna, out=h, amax=aw(_R_H))], d, dev, None)
    t = _gemm.gemm_panel(h, imgs[3][0], 2 * d, bias=_B(R.ff1), epilogue=1, p_drop=0.0, seed=0, a_amax=aw(_R_H),
                         c_amax=aw(_R_T))
    z2 = _gemm.gemm_panel(t, imgs[4][0], d, bias=_B(R.ff2), addend=h, a_amax=aw(_R_T))
    _norm.fwd([_norm.fwd_task(_norm.BN_ACT, z2, N, bn1=bn2, out=out, amax=aw(_R_OUT))], d, dev, None)
    return out, e1
