"""Autograd front-ends of the HIP kernels (the operators ``GPSLayer`` is written in).

Every function here enqueues kernels from ``libgps_hip.so`` on PyTorch's current stream and
does nothing else: no device synchronisation, no host read-back, no fallback.  Tensors are
allocated by torch's caching allocator and kept alive by the autograd context.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import os

import torch

from . import lib as _lib
from .lib import check, current_stream, ptr


def _require_cuda(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.GpsHipError(
                "graphgps_amd ops run on an MI355X only (got a CPU tensor); there is no CPU "
                "fallback in the product path -- the CPU oracle lives in oracle/ and is test-only")
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise _lib.GpsHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GpsHipError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# -------------------------------------------------------------------------------------------
# per-batch graph index
# -------------------------------------------------------------------------------------------
_ATTN_ORDER = os.environ.get("GPS_ATTN_ORDER", "1") != "0"


@dataclass
class GraphIndex:
    """Device-side index of one batch; built once, shared by all layers (fwd and bwd)."""
    N: int
    E: int
    B: int
    rowptr_dst: torch.Tensor
    src_by_dst: torch.Tensor
    eid_by_dst: torch.Tensor
    rowptr_src: torch.Tensor
    dst_by_src: torch.Tensor
    eid_by_src: torch.Tensor
    ptr: torch.Tensor          # int32 [B+1]
    tile_graph: torch.Tensor   # int32 [max_tiles]
    tile_row0: torch.Tensor    # int32 [max_tiles]
    max_tiles: int
    key: tuple = ()
    edge_src: Optional[torch.Tensor] = None   # int64 [E] = edge_index[0] (edge order)
    edge_dst: Optional[torch.Tensor] = None   # int64 [E] = edge_index[1]
    nmax_host: int = 0         # longest graph of the batch as the HOST knows it (0 = unknown; a kernel-selection hint)
    # padded batches (loader.BucketPadding, round 4; batch.gps_counts): int32 [1] device words with the number of REAL nodes / edges -- rows
    # past them are padding, kept out of every BatchNorm statistic and forced to zero gradient; None = no padding
    n_real: Optional[torch.Tensor] = None
    e_real: Optional[torch.Tensor] = None
    b_real: Optional[torch.Tensor] = None     # ... and of REAL graphs (round 5: the Performer's Nmax)
    orders: Optional[dict] = None             # head count -> int32 [B]: attn_order()

    def attn_order(self, num_heads: int) -> Optional[torch.Tensor]:
        """Balanced dispatch order of the graphs for the block-form attention kernels (``gps_attn_graph_order``: long and
        short graphs dealt to the CUs in a snake; scheduling only, results do not depend on it), made on first use per head
        count on the current stream -- inside the captured step when the index is.  None where it cannot matter (one
        graph, graphs beyond the block form) or with GPS_ATTN_ORDER=0 (A/B)."""
        if not _ATTN_ORDER or self.B < 2 or not 0 < int(self.nmax_host) <= 64:
            return None
        if self.orders is None:
            self.orders = {}
        if num_heads in self.orders:
            return self.orders[num_heads]
        t = None
        if t is None:
            t = torch.empty(self.B, dtype=torch.int32, device=self.ptr.device)
            check(_lib.load().gps_attn_graph_order(ptr(self.ptr), self.B, int(num_heads), ptr(t),
                                                   current_stream(self.ptr.device)), "gps_attn_graph_order")
            self.orders[num_heads] = t
        return t


def build_graph_index(edge_index: torch.Tensor, num_nodes: int, num_graphs: int,
                      batch_vec: Optional[torch.Tensor] = None,
                      ptr_vec: Optional[torch.Tensor] = None) -> GraphIndex:
    """CSR-by-target + CSC-by-source + int32 ``ptr`` + attention tile map for one batch.

    ``edge_index`` is the reference's int64 ``[2, E]`` (row 0 = source, row 1 = target;
    graphgps/layer/gps_layer.py:169).  ``ptr_vec`` (PyG ``batch.ptr``) is used when present,
    otherwise ``ptr`` is derived on the device from the sorted ``batch_vec``
    (graphgps/layer/gps_layer.py:199 reads only ``batch.batch``)."""
    L = _lib.load()
    dev = _require_cuda(edge_index, batch_vec, ptr_vec)
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise _lib.GpsHipError(f"edge_index must be int64 [2, E], got {edge_index.dtype} "
                               f"{tuple(edge_index.shape)}")
    edge_index = edge_index.contiguous()
    N, E, B = int(num_nodes), int(edge_index.shape[1]), int(num_graphs)
    i32 = dict(dtype=torch.int32, device=dev)
    stream = current_stream(dev)
    # rowptr_dst | rowptr_src | workspace in ONE allocation: the library then zero-fills what it needs zeroed with one fill
    # node instead of three (a fill node costs ~5 us in a replayed step); same for the two tile-map vectors below
    ws_bytes = L.gps_graph_index_workspace_bytes(N, E)
    ws_words = (max(ws_bytes, 4) + 3) // 4
    pack = torch.empty(2 * (N + 1) + ws_words, **i32)
    rowptr_dst, rowptr_src, ws = pack[:N + 1], pack[N + 1:2 * (N + 1)], pack[2 * (N + 1):]
    ws_bytes = ws_words * 4
    src_by_dst = torch.empty(E, **i32)
    eid_by_dst = torch.empty(E, **i32)
    dst_by_src = torch.empty(E, **i32)
    eid_by_src = torch.empty(E, **i32)
    check(L.gps_graph_index_build(ptr(edge_index), N, E, ptr(rowptr_dst), ptr(src_by_dst),
                                  ptr(eid_by_dst), ptr(rowptr_src), ptr(dst_by_src),
                                  ptr(eid_by_src), ptr(ws), ws_bytes, stream),
          "gps_graph_index_build")
    if ptr_vec is not None:
        p32 = ptr_vec.to(torch.int32).contiguous()
        if p32.numel() != B + 1:
            raise _lib.GpsHipError(f"ptr has {p32.numel()} entries, expected {B + 1}")
    else:
        if batch_vec is None:
            raise _lib.GpsHipError("need batch.batch or batch.ptr")
        if batch_vec.dtype != torch.int64:
            batch_vec = batch_vec.long()
        batch_vec = batch_vec.contiguous()
        p32 = torch.empty(B + 1, **i32)
        check(L.gps_segment_ptr_from_batch(ptr(batch_vec), N, B, ptr(p32), stream),
              "gps_segment_ptr_from_batch")
    max_tiles = N // 16 + B
    tiles = torch.empty(2 * max(max_tiles, 1), **i32)
    tile_graph, tile_row0 = tiles[:max(max_tiles, 1)], tiles[max(max_tiles, 1):]
    check(L.gps_attn_tile_map(ptr(p32), B, max_tiles, ptr(tile_graph), ptr(tile_row0), stream),
          "gps_attn_tile_map")
    return GraphIndex(N, E, B, rowptr_dst, src_by_dst, eid_by_dst, rowptr_src, dst_by_src,
                      eid_by_src, p32, tile_graph, tile_row0, max_tiles,
                      edge_src=edge_index[0], edge_dst=edge_index[1])


def _host_max_graph_nodes(batch) -> int:
    """Longest graph of the batch WITHOUT stalling the step: from the shared per-batch record
    (``batch._gps_meta['nmax']``: written by ``loader.DeviceLoader`` from the host-side ``ptr`` before the H2D copy,
    or by an earlier call here), from a ``ptr`` / ``batch`` vector that still lives on the host, or -- once per
    source batch, never inside a hipGraph capture -- by one device read.  0 = unknown (the kernels that want the
    bound then take their general form).  It is a kernel-selection hint, never an array bound."""
    d = getattr(batch, "__dict__", None)
    if d is None:
        return 0
    meta = d.get("_gps_meta")
    if meta is None and hasattr(batch, "shallow_copy"):        # graphgps_amd.data.Batch: give it its record
        meta = d["_gps_meta"] = {}
    if meta is not None and "nmax" in meta:
        return int(meta["nmax"])
    p, bv = getattr(batch, "ptr", None), getattr(batch, "batch", None)
    val = None
    if torch.is_tensor(p) and p.numel() > 1 and not p.is_cuda:
        val = int((p[1:] - p[:-1]).max())
    elif p is None and torch.is_tensor(bv) and bv.numel() and not bv.is_cuda:
        val = int(torch.bincount(bv).max())
    elif meta is not None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        # a device-resident batch that carries the shareable record (graphgps_amd.data.Batch: shallow copies of
        # one batch share it): one synchronising read, paid once for all of its copies
        if torch.is_tensor(p) and p.numel() > 1:
            val = int((p[1:] - p[:-1]).max().item())
        elif torch.is_tensor(bv) and bv.numel():
            val = int(torch.bincount(bv).max().item())
    if val is None:
        return 0
    if meta is not None:
        meta["nmax"] = val
    return val


def graph_index_of(batch) -> GraphIndex:
    """Index cached on the batch object (one build per batch, not per layer)."""
    ei = batch.edge_index
    N = int(batch.x.shape[0])
    key = (ei.data_ptr(), int(ei.shape[1]), N, str(ei.device))
    cached = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
    if cached is not None and cached.key == key:
        return cached
    B = int(batch.num_graphs)
    gi = build_graph_index(ei, N, B, batch_vec=getattr(batch, "batch", None),
                           ptr_vec=getattr(batch, "ptr", None))
    gi.key = key
    gi.nmax_host = _host_max_graph_nodes(batch)
    counts = getattr(batch, "gps_counts", None)
    if torch.is_tensor(counts):          # int32 [real nodes, real edges, real graphs] on the device: a padded batch
        if counts.dtype != torch.int32 or counts.numel() < 2 or counts.device != ei.device:
            raise _lib.GpsHipError(f"batch.gps_counts must be int32 [>= 2] on {ei.device}, got {counts.dtype} "
                                   f"{tuple(counts.shape)} on {counts.device}")
        gi.n_real, gi.e_real = counts[0:1], counts[1:2]
        gi.b_real = counts[2:3] if counts.numel() >= 3 else None
    try:
        batch.__dict__["_gps_index"] = gi
    except Exception:
        pass
    return gi


# -------------------------------------------------------------------------------------------
# GatedGCN sparse core
# -------------------------------------------------------------------------------------------
class _GatedGCNAggregate(torch.autograd.Function):
    """(proj [N,4d] = Ax|Bx|Dx|Ex, Ce [E,d][, r [E]]) -> (x_tilde [N,d], e_hat [E,d]).

    graphgps/layer/gatedgcn_layer.py:67-70,90-136.  ``r`` is the EquivStableLapPE gate r_ij (:101-104):
    sigma_ij * r_ij replaces sigma_ij in both sums."""

    @staticmethod
    def forward(ctx, proj: torch.Tensor, ce: torch.Tensor, gi: GraphIndex, r=None):
        L = _lib.load()
        dev = _require_cuda(proj, ce)
        proj, ce = _f32c(proj, "proj"), _f32c(ce, "Ce")
        N, E = gi.N, gi.E
        d = proj.shape[1] // 4
        if proj.shape != (N, 4 * d) or ce.shape != (E, d):
            raise _lib.GpsHipError(f"gatedgcn: proj {tuple(proj.shape)} / Ce {tuple(ce.shape)} do "
                                   f"not match N={N} E={E}")
        if r is not None:
            r = _f32c(r.reshape(-1), "r_ij")
            if r.numel() != E:
                raise _lib.GpsHipError(f"gatedgcn: r_ij has {r.numel()} entries for E={E} edges")
        need_grad = any(ctx.needs_input_grad)
        x_tilde = torch.empty(N, d, dtype=torch.float32, device=dev)
        e_hat = torch.empty(E, d, dtype=torch.float32, device=dev)
        # nothing extra is saved: the backward recomputes num_i and den_i from e_hat
        base, fs = proj.data_ptr(), d * 4  # fs = byte offset between the Ax|Bx|Dx|Ex column blocks
        check(L.gps_gatedgcn_fwd(base, base + fs, base + 2 * fs, base + 3 * fs, 4 * d, ptr(ce),
                                 ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E,
                                 d, ptr(x_tilde), ptr(e_hat), ptr(r),
                                 current_stream(dev)), "gps_gatedgcn_fwd")
        if need_grad:
            ctx.save_for_backward(proj, e_hat, x_tilde, r)
            ctx.gi = gi
        return x_tilde, e_hat

    @staticmethod
    def backward(ctx, g_x: torch.Tensor, g_e: torch.Tensor):
        L = _lib.load()
        proj, e_hat, x_tilde, r = ctx.saved_tensors
        gi: GraphIndex = ctx.gi
        dev = proj.device
        N, E = gi.N, gi.E
        d = proj.shape[1] // 4
        g_x = _f32c(g_x, "g_x") if g_x is not None else torch.zeros(N, d, device=dev)
        g_e = _f32c(g_e, "g_e") if g_e is not None else torch.zeros(E, d, device=dev)
        g_proj = torch.empty(N, 4 * d, dtype=torch.float32, device=dev)
        g_ce = torch.empty(E, d, dtype=torch.float32, device=dev)
        gb, fs = g_proj.data_ptr(), d * 4
        check(L.gps_gatedgcn_bwd(ptr(g_x), d, ptr(g_e), ptr(e_hat), proj.data_ptr(), proj.data_ptr() + fs, 4 * d,
                                 ptr(x_tilde), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                                 ptr(gi.eid_by_dst), ptr(gi.rowptr_src), ptr(gi.dst_by_src),
                                 ptr(gi.eid_by_src), N, E, d, ptr(g_ce), gb, gb + fs, gb + 2 * fs,
                                 gb + 3 * fs, 4 * d, ptr(r), None, None, current_stream(dev)), "gps_gatedgcn_bwd")
        g_r = None
        if r is not None and ctx.needs_input_grad[3]:
            # g_r[e] = sum_c g_sigma'[e,c] * sigma[e,c],  g_sigma' = a_i * Bx_j + b_i  (a, b as in the kernel).
            # A per-edge channel reduction: plain gathers + a row sum (rare option, [E,d] temporaries)
            # (fp64: a * Bx_j + b cancels, and this scalar feeds a 1 -> d -> 1 MLP's gradients)
            src, dst = gi.edge_src, gi.edge_dst
            den = torch.zeros(N, d, dtype=torch.float64, device=dev).index_add_(
                0, dst, torch.sigmoid(e_hat.double()) * r.double().view(-1, 1))     # sum_j sigma_ij r_ij
            a = g_x.double() / (den + 1e-6)
            bterm = -a * (x_tilde.double() - proj[:, :d].double())     # aggr_i = x_tilde_i - Ax_i
            gs = a.index_select(0, dst) * proj[:, d:2 * d].double().index_select(0, src) \
                + bterm.index_select(0, dst)
            g_r = (gs * torch.sigmoid(e_hat.double())).sum(-1).float()
        return g_proj, g_ce, None, g_r


def gatedgcn_aggregate(proj: torch.Tensor, ce: torch.Tensor, gi: GraphIndex, r=None):
    return _GatedGCNAggregate.apply(proj, ce, gi, r)


# -------------------------------------------------------------------------------------------
# GINE sparse core
# -------------------------------------------------------------------------------------------
class _GINEAggregate(torch.autograd.Function):
    """out_i = (1+eps) x_i + sum_{j->i} relu(x_j + e_ji) [* r_ji]  (PyG GINEConv, pre-MLP; with ``r``:
    GINEConvESLapPE, graphgps/layer/gine_conv_layer.py:70-84)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, e: torch.Tensor, gi: GraphIndex, eps: float, r=None):
        L = _lib.load()
        dev = _require_cuda(x, e)
        x, e = _f32c(x, "x"), _f32c(e, "edge_attr")
        N, E, d = gi.N, gi.E, x.shape[1]
        if x.shape[0] != N or e.shape != (E, d):
            raise ValueError("Node and edge feature dimensionalities do not match. Consider "
                             "setting the 'edge_dim' attribute of 'GINEConv'")
        if r is not None:
            r = _f32c(r.reshape(-1), "r_ij")
            if r.numel() != E:
                raise _lib.GpsHipError(f"gine: r_ij has {r.numel()} entries for E={E} edges")
        out = torch.empty_like(x)
        check(L.gps_gine_fwd(ptr(x), ptr(e), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                             ptr(gi.eid_by_dst), N, E, d, float(eps), ptr(out), ptr(r),
                             current_stream(dev)), "gps_gine_fwd")
        ctx.save_for_backward(x, e, r)
        ctx.gi, ctx.eps = gi, float(eps)
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        L = _lib.load()
        x, e, r = ctx.saved_tensors
        gi: GraphIndex = ctx.gi
        g_out = _f32c(g_out, "g_out")
        g_x, g_e = torch.empty_like(x), torch.empty_like(e)
        check(L.gps_gine_bwd(ptr(g_out), ptr(x), ptr(e), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                             ptr(gi.eid_by_dst), ptr(gi.rowptr_src), ptr(gi.eid_by_src), gi.N,
                             gi.E, x.shape[1], ctx.eps, ptr(g_x), ptr(g_e), ptr(r),
                             current_stream(x.device)), "gps_gine_bwd")
        g_r = None
        if r is not None and ctx.needs_input_grad[4]:
            src, dst = gi.edge_src, gi.edge_dst
            g_r = (g_out.double().index_select(0, dst)
                   * (x.index_select(0, src) + e).relu().double()).sum(-1).float()
        return g_x, g_e, None, None, g_r


def gine_aggregate(x: torch.Tensor, e: torch.Tensor, gi: GraphIndex, eps: float = 0.0, r=None):
    return _GINEAggregate.apply(x, e, gi, eps, r)


# -------------------------------------------------------------------------------------------
# GCN sparse core
# -------------------------------------------------------------------------------------------
def gcn_dinv(gi: GraphIndex) -> torch.Tensor:
    """deg^-1/2 of gcn_norm (one unit self loop per node + every non-loop incoming edge), cached on the
    index: 10 layers and their backward passes share it."""
    dinv = gi.__dict__.get("_gcn_dinv")
    if dinv is None:
        dev = gi.rowptr_dst.device
        dinv = torch.empty(gi.N, dtype=torch.float32, device=dev)
        check(_lib.load().gps_gcn_dinv(ptr(gi.rowptr_dst), ptr(gi.src_by_dst), gi.N, gi.E, ptr(dinv),
                                       current_stream(dev)), "gps_gcn_dinv")
        gi.__dict__["_gcn_dinv"] = dinv
    return dinv


class _GCNAggregate(torch.autograd.Function):
    """out = D^-1/2 (A + I) D^-1/2 x over the batch's edges (PyG GCNConv.propagate after gcn_norm); the
    backward is the transposed product = the same kernel on the source-keyed half of the index."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, gi: GraphIndex):
        L = _lib.load()
        dev = _require_cuda(x)
        x = _f32c(x, "x")
        if x.dim() != 2 or x.shape[0] != gi.N:
            raise _lib.GpsHipError(f"gcn_aggregate: x {tuple(x.shape)} vs N={gi.N}")
        dinv = gcn_dinv(gi)
        out = torch.empty_like(x)
        check(L.gps_gcn_spmm(ptr(x), x.shape[1], ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(dinv), gi.N,
                             gi.E, x.shape[1], ptr(out), current_stream(dev)), "gps_gcn_spmm")
        ctx.gi = gi
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        L = _lib.load()
        gi: GraphIndex = ctx.gi
        g_out = _f32c(g_out, "g_out")
        g_x = torch.empty_like(g_out)
        check(L.gps_gcn_spmm(ptr(g_out), g_out.shape[1], ptr(gi.rowptr_src), ptr(gi.dst_by_src),
                             ptr(gcn_dinv(gi)), gi.N, gi.E, g_out.shape[1], ptr(g_x),
                             current_stream(g_out.device)), "gps_gcn_spmm")
        return g_x, None


def gcn_aggregate(x: torch.Tensor, gi: GraphIndex) -> torch.Tensor:
    return _GCNAggregate.apply(x, gi)


class _GINAggregate(torch.autograd.Function):
    """out_i = (1 + eps) x_i + sum_{j->i} x_j over every stored edge (PyG GINConv before its MLP)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, gi: GraphIndex, eps: float):
        dev = _require_cuda(x)
        x = _f32c(x, "x")
        if x.dim() != 2 or x.shape[0] != gi.N:
            raise _lib.GpsHipError(f"gin_aggregate: x {tuple(x.shape)} vs N={gi.N}")
        out = torch.empty_like(x)
        check(_lib.load().gps_adj_sum(ptr(x), x.shape[1], ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                                      1.0 + float(eps), gi.N, gi.E, x.shape[1], ptr(out),
                                      current_stream(dev)), "gps_adj_sum")
        ctx.gi, ctx.eps = gi, float(eps)
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        gi: GraphIndex = ctx.gi
        g_out = _f32c(g_out, "g_out")
        g_x = torch.empty_like(g_out)
        check(_lib.load().gps_adj_sum(ptr(g_out), g_out.shape[1], ptr(gi.rowptr_src), ptr(gi.dst_by_src),
                                      1.0 + ctx.eps, gi.N, gi.E, g_out.shape[1], ptr(g_x),
                                      current_stream(g_out.device)), "gps_adj_sum")
        return g_x, None, None


def gin_aggregate(x: torch.Tensor, gi: GraphIndex, eps: float = 0.0) -> torch.Tensor:
    return _GINAggregate.apply(x, gi, eps)


# -------------------------------------------------------------------------------------------
# segment attention
# -------------------------------------------------------------------------------------------
def draw_dropout_seed() -> int:
    """64-bit seed from torch's CPU generator: reproducible under torch.manual_seed, no
    device sync."""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF


_dropout_salt = None


def enable_dropout_salt(device) -> torch.Tensor:
    """Install (once) the device-resident dropout salt and return it (int64[1]).  A training step
    that is going to be captured in a hipGraph must do ``salt.add_(1)`` inside the step: replays
    reuse the captured by-value seeds, the salt is what makes their masks differ (the kernels add
    ``salt * 0x9E3779B97F4A7C15`` to every seed; include/gps_hip.h: gps_set_dropout_salt)."""
    global _dropout_salt
    dev = torch.device(device)
    if _dropout_salt is None or _dropout_salt.device != dev:
        _dropout_salt = torch.zeros(1, dtype=torch.int64, device=dev)
        check(_lib.load().gps_set_dropout_salt(ptr(_dropout_salt)), "gps_set_dropout_salt")
    return _dropout_salt


class _SegmentAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, gi: GraphIndex, num_heads: int, p_drop: float, seed: int,
                bias: Optional[torch.Tensor] = None):
        L = _lib.load()
        dev = _require_cuda(qkv, bias)
        qkv = _f32c(qkv, "qkv")
        N, H = gi.N, int(num_heads)
        d = qkv.shape[1] // 3
        dh = d // H
        if qkv.shape != (N, 3 * d) or dh * H != d:
            raise _lib.GpsHipError(f"segment_attention: qkv {tuple(qkv.shape)} vs N={N} H={H}")
        out = torch.empty(N, d, dtype=torch.float32, device=dev)
        lse = torch.empty(H, N, dtype=torch.float32, device=dev)
        scale = float(dh) ** -0.5
        if bias is None:
            check(L.gps_seg_attn_fwd(ptr(qkv), 3 * d, ptr(gi.ptr), ptr(gi.tile_graph),
                                     ptr(gi.tile_row0), gi.max_tiles, N, H, dh, scale, float(p_drop),
                                     seed, ptr(out), ptr(lse), gi.B, int(gi.nmax_host), None, ptr(gi.attn_order(H)),
                                     current_stream(dev)), "gps_seg_attn_fwd")
            ctx.save_for_backward(qkv, out, lse)
        else:
            bias = _f32c(bias, "attn_bias")
            nmax = _check_attn_bias(bias, gi, H)
            check(L.gps_seg_attn_bias_fwd(ptr(qkv), 3 * d, ptr(bias), nmax, ptr(gi.ptr),
                                          ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh,
                                          scale, float(p_drop), seed, ptr(out), ptr(lse),
                                          current_stream(dev)), "gps_seg_attn_bias_fwd")
            ctx.save_for_backward(qkv, out, lse, bias)
        ctx.biased = bias is not None
        ctx.gi, ctx.H, ctx.dh, ctx.scale, ctx.p_drop, ctx.seed = gi, H, dh, scale, float(p_drop), seed
        return out

    @staticmethod
    def backward(ctx, d_out: torch.Tensor):
        L = _lib.load()
        qkv, out, lse = ctx.saved_tensors[:3]
        gi: GraphIndex = ctx.gi
        dev = qkv.device
        d_out = _f32c(d_out, "d_out")
        N, H, dh = gi.N, ctx.H, ctx.dh
        d_qkv = torch.empty_like(qkv)
        delta = torch.empty(H, N, dtype=torch.float32, device=dev)
        if not ctx.biased:
            check(L.gps_seg_attn_bwd(ptr(d_out), ptr(qkv), qkv.shape[1], ptr(out), ptr(lse), ptr(gi.ptr),
                                     ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh,
                                     ctx.scale, ctx.p_drop, ctx.seed, ptr(delta), ptr(d_qkv),
                                     d_qkv.shape[1], gi.B, int(gi.nmax_host), None, ptr(gi.attn_order(H)),
                                     current_stream(dev)), "gps_seg_attn_bwd")
            return d_qkv, None, None, None, None, None
        bias = ctx.saved_tensors[3]
        d_bias = torch.zeros_like(bias)       # padded region: zero gradient, as under the reference's mask
        check(L.gps_seg_attn_bias_bwd(ptr(d_out), ptr(qkv), qkv.shape[1], ptr(bias), bias.shape[1],
                                      ptr(out), ptr(lse), ptr(gi.ptr), ptr(gi.tile_graph),
                                      ptr(gi.tile_row0), gi.max_tiles, N, H, dh, ctx.scale, ctx.p_drop,
                                      ctx.seed, ptr(delta), ptr(d_qkv), d_qkv.shape[1], ptr(d_bias),
                                      current_stream(dev)), "gps_seg_attn_bias_bwd")
        return d_qkv, None, None, None, None, d_bias


def max_graph_size(gi: GraphIndex) -> int:
    """Largest graph of the batch (what ``to_dense_batch`` pads to).  One device->host read per batch,
    cached on the index; only the biased-attention path needs it on the host (to validate the dense
    ``attn_bias`` it is handed), the plain path never asks."""
    n = gi.__dict__.get("_nmax_host")
    if n is None:
        n = int((gi.ptr[1:] - gi.ptr[:-1]).max().item()) if gi.B > 0 else 0
        gi.__dict__["_nmax_host"] = n
    return n


def _check_attn_bias(bias: torch.Tensor, gi: GraphIndex, H: int) -> int:
    if bias.dim() != 3 or bias.shape[0] != gi.B * H or bias.shape[1] != bias.shape[2]:
        raise _lib.GpsHipError(f"attn_bias must be [B*H, nmax, nmax] = [{gi.B * H}, n, n], got "
                               f"{tuple(bias.shape)}")
    nmax = int(bias.shape[1])
    need = max_graph_size(gi)
    if nmax < need:
        raise _lib.GpsHipError(f"attn_bias is padded to {nmax} nodes but the largest graph has {need}")
    return nmax


def segment_attention(qkv: torch.Tensor, gi: GraphIndex, num_heads: int, p_drop: float = 0.0,
                      seed: Optional[int] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh) [+ bias]) (dropout) v per graph and head, straight off ``ptr``.

    ``qkv`` is the packed in-projection output [N, 3d] (``torch.nn.MultiheadAttention``'s
    ``in_proj_weight`` layout).  Replaces graphgps/layer/gps_layer.py:199-201 + the core of
    ``nn.MultiheadAttention``.  ``bias`` is the reference's dense ``batch.attn_bias``
    ``[B*H, nmax, nmax]`` (gps_layer.py:201-203, graphormer_layer.py:43-44): its gradient comes back in
    the same dense layout."""
    if p_drop > 0.0 and seed is None:
        seed = draw_dropout_seed()
    return _SegmentAttention.apply(qkv, gi, num_heads, float(p_drop), int(seed or 0), bias)


def attn_dropout_keep_mask(seed: int, q_global: torch.Tensor, head: int, num_heads: int,
                           key_local: torch.Tensor, p_drop: float, paired: bool = False) -> torch.Tensor:
    """Bit-exact host model of the kernels' counter-based dropout masks; ``q_global`` [Q] and ``key_local`` [K]
    are integer tensors; returns bool [Q, K].  Used by the parity tests to inject the SAME mask into the oracle.

    ``paired=False``: the row kernels (csrc/bn_fused.hip, block_norm.hip), one hash per element keyed (row,
    column), 24 bits against p.  ``paired=True``: the attention kernels (csrc/attn_common.hpp: row_hash /
    pair_hash / keep_elem), one hash per PAIR of adjacent keys, 16 bits each against round(p * 65536) -- the
    drop probability is then ``attn_dropout_effective_p(p)`` and survivors are scaled by 1 / (1 - that)."""
    M = 0xFFFFFFFF

    def mix(x):
        x = x & M
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & M
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & M
        x = x ^ (x >> 16)
        return x

    q = q_global.to(torch.int64).cpu()
    k = key_local.to(torch.int64).cpu()
    rowid = (q * num_heads + head) & M
    rh = (mix(rowid ^ (seed & M)) + ((seed >> 32) & M)) & M
    if paired:
        h = mix((rh[:, None] + ((k[None, :] >> 1) * 0x9E3779B9)) & M)
        bits = torch.where((k[None, :] & 1) == 1, h >> 16, h & 0xFFFF)
        return bits >= int(p_drop * 65536.0 + 0.5)
    r = mix((rh[:, None] + (k[None, :] * 0x9E3779B9)) & M)
    u = (r >> 8).to(torch.float32) * (1.0 / 16777216.0)
    return u >= torch.tensor(p_drop, dtype=torch.float32)


def attn_dropout_effective_p(p_drop: float) -> float:
    """Drop probability the attention kernels realise for a nominal ``p_drop``: round(p * 2^16) / 2^16."""
    return int(p_drop * 65536.0 + 0.5) / 65536.0


# -------------------------------------------------------------------------------------------
# SAN edge attention over the real edges
# -------------------------------------------------------------------------------------------
class _EdgeAttention(torch.autograd.Function):
    """(Q, K, V [N, H*D], E [E, H*D]) -> (wv [N, H*D], z [N, H]); see include/gps_hip.h: gps_edge_attn_fwd."""

    @staticmethod
    def forward(ctx, q, k, v, e, gi: GraphIndex, H: int, softmax: bool):
        L = _lib.load()
        dev = _require_cuda(q, k, v, e)
        q, k, v, e = _f32c(q, "Q"), _f32c(k, "K"), _f32c(v, "V"), _f32c(e, "E")
        N, E = gi.N, gi.E
        HD = q.shape[1]
        D = HD // H
        if q.shape != (N, HD) or k.shape != q.shape or v.shape != q.shape or e.shape != (E, HD) or D * H != HD:
            raise _lib.GpsHipError(f"edge_attention: Q {tuple(q.shape)} / E {tuple(e.shape)} vs N={N} E={E} H={H}")
        if not L.gps_edge_attn_supported(H, D):
            raise _lib.GpsHipError(f"edge_attention: head dim {D} has no kernel (4, 8, 16, 32, 64)")
        f32 = dict(dtype=torch.float32, device=dev)
        wv = torch.empty(N, HD, **f32)
        z = torch.zeros(N, H, **f32)
        mx = torch.empty(N, H, **f32) if softmax else None
        ls = torch.empty(N, H, **f32) if softmax else None
        scale = float(D) ** -0.5
        check(L.gps_edge_attn_fwd(ptr(q), ptr(k), ptr(v), HD, ptr(e), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                                  ptr(gi.eid_by_dst), N, E, H, D, scale, int(softmax), ptr(wv), ptr(z), ptr(mx),
                                  ptr(ls), current_stream(dev)), "gps_edge_attn_fwd")
        ctx.save_for_backward(q, k, v, e, wv, mx, ls)
        ctx.gi, ctx.cfg = gi, (H, D, scale, bool(softmax))
        ctx.mark_non_differentiable(z) if softmax else None
        return wv, z

    @staticmethod
    def backward(ctx, g_wv, g_z):
        L = _lib.load()
        q, k, v, e, wv, mx, ls = ctx.saved_tensors
        gi: GraphIndex = ctx.gi
        H, D, scale, softmax = ctx.cfg
        dev = q.device
        N, E, HD = gi.N, gi.E, q.shape[1]
        g_wv = _f32c(g_wv, "g_wv") if g_wv is not None else torch.zeros_like(wv)
        g_z = _f32c(g_z, "g_z") if (g_z is not None and not softmax) else None
        g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        g_e = torch.empty_like(e)
        ws = torch.empty(max(2 * E * H, 1), dtype=torch.float32, device=dev)
        check(L.gps_edge_attn_bwd(ptr(g_wv), ptr(g_z), ptr(q), ptr(k), ptr(v), HD, ptr(e), ptr(wv), ptr(mx), ptr(ls),
                                  ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst), ptr(gi.rowptr_src),
                                  ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, H, D, scale, int(softmax), ptr(g_q),
                                  ptr(g_k), ptr(g_v), ptr(g_e), ptr(ws), current_stream(dev)), "gps_edge_attn_bwd")
        return g_q, g_k, g_v, g_e, None, None, None


def edge_attention_supported(num_heads: int, head_dim: int) -> bool:
    return head_dim in (4, 8, 16, 32, 64) and num_heads > 0


def edge_attention(q, k, v, e, gi: GraphIndex, num_heads: int, softmax: bool):
    """SAN attention over the real edges (san_layer.py:44-92 / san2_layer.py:65-105): returns
    ``(sum_e w_e V[src], sum_e w_e)`` per target node with ``w`` the clamp-exp (``softmax=False``) or the per-target
    softmax (``softmax=True``; the second output is then unused zeros) of the per-head K.Q.E score."""
    return _EdgeAttention.apply(q, k, v, e, gi, int(num_heads), bool(softmax))


# -------------------------------------------------------------------------------------------
# graph pooling over ptr segments
# -------------------------------------------------------------------------------------------
def _node_graph(gi: GraphIndex) -> torch.Tensor:
    ng = getattr(gi, "_node_graph", None)
    if ng is None:
        L = _lib.load()
        dev = gi.ptr.device
        ng = torch.empty(max(gi.N, 1), dtype=torch.int32, device=dev)
        check(L.gps_node_graph_from_ptr(ptr(gi.ptr), gi.B, ptr(ng), current_stream(dev)),
              "gps_node_graph_from_ptr")
        gi._node_graph = ng
    return ng


_POOL_SLICED = os.environ.get("GPS_POOL_SLICED", "1") != "0"


class _SegmentPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, gi: GraphIndex, mean: bool):
        L = _lib.load()
        dev = _require_cuda(x)
        x = _f32c(x, "x")
        d = x.shape[1]
        out = torch.empty(gi.B, d, dtype=torch.float32, device=dev)
        if _POOL_SLICED:                     # round 5: one workgroup per (graph, 32-row slice) + a merge in slice order
            nb = L.gps_segment_pool_workspace_bytes(gi.N, gi.B, d)
            ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
            check(L.gps_segment_pool_fwd_sliced(ptr(x), ptr(gi.ptr), gi.N, gi.B, d, int(mean), ptr(out), ptr(ws), nb,
                                                current_stream(dev)), "gps_segment_pool_fwd_sliced")
        else:                                # GPS_POOL_SLICED=0: one lane group per graph (rounds 1-4)
            check(L.gps_segment_pool_fwd(ptr(x), ptr(gi.ptr), gi.B, d, int(mean), ptr(out),
                                         current_stream(dev)), "gps_segment_pool_fwd")
        ctx.gi, ctx.mean, ctx.d = gi, bool(mean), d
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        L = _lib.load()
        gi: GraphIndex = ctx.gi
        g_out = _f32c(g_out, "g_out")
        dev = g_out.device
        g_x = torch.empty(gi.N, ctx.d, dtype=torch.float32, device=dev)
        check(L.gps_segment_pool_bwd(ptr(g_out), ptr(gi.ptr), ptr(_node_graph(gi)), gi.N, ctx.d,
                                     int(ctx.mean), ptr(g_x), current_stream(dev)),
              "gps_segment_pool_bwd")
        return g_x, None, None


def segment_pool(x: torch.Tensor, gi: GraphIndex, mode: str = "mean") -> torch.Tensor:
    """``global_add_pool`` / ``global_mean_pool`` over ``ptr`` segments (deterministic)."""
    if mode not in ("add", "sum", "mean"):
        raise ValueError(f"segment_pool: unsupported mode {mode!r}")
    return _SegmentPool.apply(x, gi, mode == "mean")


# -------------------------------------------------------------------------------------------
# FAVOR+ (Performer) linear attention over ptr segments
# -------------------------------------------------------------------------------------------
def _nmax_dev(gi: GraphIndex) -> torch.Tensor:
    """Longest graph of the batch as a device scalar (the reference's to_dense_batch Nmax)."""
    nm = getattr(gi, "_nmax", None)
    if nm is None:
        L = _lib.load()
        dev = gi.ptr.device
        nm = torch.zeros(1, dtype=torch.int32, device=dev)
        # (a padded batch: over the real graphs only -- gi.b_real, the third word of batch.gps_counts)
        check(L.gps_segment_max_len_real(ptr(gi.ptr), gi.B, ptr(gi.b_real), ptr(nm), current_stream(dev)),
              "gps_segment_max_len_real")
        gi._nmax = nm
    return nm


def favor_workspace_floats(N: int, B: int, H: int) -> int:
    """Floats of scratch the FAVOR+ context kernels take for the partial records of their row slices (0: no slicing)."""
    return int(_lib.load().gps_favor_workspace_floats(N, B, H))


class _FavorAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, proj: torch.Tensor, gi: GraphIndex, num_heads: int):
        L = _lib.load()
        dev = _require_cuda(qkv, proj)
        qkv, proj = _f32c(qkv, "qkv"), _f32c(proj, "projection_matrix")
        N, B, H = gi.N, gi.B, int(num_heads)
        inner = qkv.shape[1] // 3
        dh = inner // H
        m = proj.shape[0]
        if qkv.shape != (N, 3 * inner) or dh * H != inner or proj.shape[1] != dh:
            raise _lib.GpsHipError(f"favor_attention: qkv {tuple(qkv.shape)} / proj "
                                   f"{tuple(proj.shape)} vs N={N} H={H}")
        f32 = dict(dtype=torch.float32, device=dev)
        out = torch.empty(N, inner, **f32)
        cbuf = torch.empty(B * H, 272, dh, **f32)
        ksum = torch.empty(B * H, 272, **f32)
        kmax = torch.empty(B * H, dtype=torch.int64, device=dev)
        mq = torch.empty(H, N, **f32)
        D = torch.empty(H, N, **f32)
        nmax = _nmax_dev(gi)
        wsf = favor_workspace_floats(N, B, H)
        ws = torch.empty(wsf, **f32) if wsf else None
        check(L.gps_favor_fwd(ptr(qkv), qkv.shape[1], ptr(proj), m, ptr(gi.ptr), ptr(nmax),
                              ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, B, H, dh,
                              ptr(out), ptr(cbuf), ptr(ksum), ptr(kmax), ptr(mq), ptr(D), ptr(ws), wsf,
                              current_stream(dev)), "gps_favor_fwd")
        ctx.save_for_backward(qkv, proj, out, cbuf, ksum, kmax, mq, D)
        ctx.gi, ctx.H, ctx.dh = gi, H, dh
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        L = _lib.load()
        qkv, proj, out, cbuf, ksum, kmax, mq, D = ctx.saved_tensors
        gi: GraphIndex = ctx.gi
        dev = qkv.device
        g_out = _f32c(g_out, "g_out")
        N, B, H, dh = gi.N, gi.B, ctx.H, ctx.dh
        f32 = dict(dtype=torch.float32, device=dev)
        gD = torch.empty(H, N, **f32)
        g_ctx = torch.empty_like(cbuf)
        g_ksum = torch.empty_like(ksum)
        gm_part = torch.empty(max(gi.max_tiles * H, 1), **f32)
        d_qkv = torch.empty_like(qkv)
        wsf = favor_workspace_floats(N, B, H)
        ws = torch.empty(wsf, **f32) if wsf else None
        check(L.gps_favor_bwd(ptr(g_out), ptr(qkv), qkv.shape[1], ptr(proj), proj.shape[0], ptr(out),
                              ptr(gi.ptr), ptr(_nmax_dev(gi)), ptr(gi.tile_graph), ptr(gi.tile_row0),
                              gi.max_tiles, N, B, H, dh, ptr(cbuf), ptr(ksum), ptr(kmax), ptr(mq),
                              ptr(D), ptr(gD), ptr(g_ctx), ptr(g_ksum), ptr(gm_part), ptr(d_qkv),
                              d_qkv.shape[1], ptr(ws), wsf, current_stream(dev)), "gps_favor_bwd")
        return d_qkv, None, None, None


def favor_attention(qkv: torch.Tensor, proj: torch.Tensor, gi: GraphIndex,
                    num_heads: int) -> torch.Tensor:
    """Performer FAVOR+ attention per graph and head, straight off ``ptr`` (no padding).

    ``qkv`` = [N, 3*64H] (to_q | to_k | to_v outputs), ``proj`` = the fixed
    ``fast_attention.projection_matrix`` buffer [m, 64].  Reproduces the reference's result ON
    THE PADDED BATCH, including the padded-key contribution to the normaliser
    (graphgps/layer/performer_layer.py:485-487)."""
    return _FavorAttention.apply(qkv, proj, gi, num_heads)


# -------------------------------------------------------------------------------------------
# embedding lookup with a deterministic weight gradient (large vocabularies)
# -------------------------------------------------------------------------------------------
class _Embedding(torch.autograd.Function):
    """``weight[idx]`` (nn.Embedding without padding_idx / max_norm).  Backward: a stable sort of the token ids groups
    the lookups, csrc/segment_pool.hip sums each group's gradient rows over fixed 64-entry units (ATen: radix sort +
    sum_and_scatter, 0.4 - 1.2 ms per [25k, 256] lookup on MI355X and not reproducible bit for bit)."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.vocab = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        L = _lib.load()
        dev = g.device
        g = _f32c(g, "g")
        n, d = g.shape
        tok, perm = torch.sort(idx.to(torch.int64), stable=True)
        g_w = torch.zeros(ctx.vocab, d, dtype=torch.float32, device=dev)
        wsb = L.gps_embedding_grad_workspace_bytes(n, d)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        check(L.gps_embedding_grad(ptr(g), ptr(tok), ptr(perm), n, ctx.vocab, d, ptr(g_w), ptr(ws), wsb,
                                   current_stream(dev)), "gps_embedding_grad")
        return None, g_w


def embedding(idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``nn.Embedding`` lookup; on the GPU with gradients enabled (and d in {64, 128, 256}) the weight gradient is the
    deterministic HIP path, anything else is ``F.embedding``."""
    if weight.is_cuda and weight.dtype == torch.float32 and idx.dim() == 1 and idx.numel() > 0 \
            and torch.is_grad_enabled() and weight.requires_grad and weight.shape[1] in (64, 128, 256) \
            and weight.is_contiguous():
        return _Embedding.apply(idx.contiguous(), weight)
    return torch.nn.functional.embedding(idx, weight)
