"""Row-panel GEMM (csrc/gemm_panel.hip) behind the nn.Linear modules of the GPS block.

``C = A W^T (+ bias) (+ addend)`` with optional ReLU/dropout epilogues, fp32 in / fp32 out, products formed from fp16
pieces (round 4: two per value, 3 products) or bf16 pieces (round 2: three per value, 6 products, exact) on the MFMA pipe.
The weight operand is a pre-split IMAGE (pieces k-stage-major) made from the fp32
weight by ``split_weights`` -- once per layer and forward (the weights change under the optimizer's HIP kernel, which
torch's version counters do not see, so nothing is cached across calls: ~10 us per layer for the five weights of a
block, both images; callers that never run an input gradient pass ``tn=False``).  Replaces ``torch.addmm`` / ``mm`` (rocBLAS / hipBLASLt) for the projections of
``/root/reference/graphgps/layer/gatedgcn_layer.py:57-61`` and ``graphgps/layer/gps_layer.py:104-106,143-144,253-257``
where the shape qualifies (N % 4 == 0, K % 4 == 0: every ``dim_hidden`` the reference's configs use -- 384, 304, 256,
96, 72, 64, 52, 48; widths that are not whole 64-column panels / 32-wide k-stages run the kernel's EDGE variants over a
padded image); everything else stays on the libraries.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib as _lib
from .lib import check, current_stream, ptr

ENABLED = os.environ.get("GPS_GEMM_PANEL", "1") != "0"
MAX_SPLIT = 56      # csrc/gemm_panel.hip kMaxSplit
# Arithmetic form of the ring GEMM (round 4).  F16 (default): two fp16 pieces per operand value under a per-tensor
# power-of-two scale, 3 piece products on v_mfma_f32_32x32x16_f16 -- half the matrix-pipe work of the round-2 form (three
# bf16 pieces, 6 products), error against fp64 at or below it.  The scale comes from max|operand|, a device word
# (``absmax``): the weights' words are made with the images, an activation's word is computed once and shared by every
# GEMM that consumes the tensor.  GPS_GEMM_F16=0 keeps the 6-product form everywhere (A/B runs).
F16 = os.environ.get("GPS_GEMM_F16", "1") != "0"
MAX_SPLIT16 = 48    # csrc/gemm_panel.hip kMaxSplit16
MAX_ABS = 56        # csrc/gemm_panel.hip kMaxAbs
# include/gps_hip.h GPS_AMAX_RECORD_WORDS: a max|.| record occupies 512 int32 words -- 8 live words (fp32 bit patterns)
# 64 words apart, the rest zero; the maximum is the max over the row
AMAX_WORDS = 512


def amax_records(n: int, device) -> torch.Tensor:
    """``n`` zeroed max|.| records: int32 ``[n, 512]``; row i is the record of tensor i."""
    return torch.zeros(n, AMAX_WORDS, dtype=torch.int32, device=device)


class WImage:
    """A weight image and, for the fp16 form, the device record of max|W| it was scaled by (``amax``: int32 [512])."""
    __slots__ = ("t", "amax")

    def __init__(self, t: torch.Tensor, amax: Optional[torch.Tensor] = None):
        self.t, self.amax = t, amax

    def data_ptr(self) -> int:
        return self.t.data_ptr()


def absmax(tensors: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 ``[len(tensors), 512]``: row i = the max|.| record of ``tensors[i]`` (fp32 bit patterns; the maximum is the max
    over the row -- ``amax_value``) by csrc/gemm_panel.hip ``gps_absmax``, one launch for up to 56 row-major fp32 matrices
    with ``cols % 4 == 0``.  ``out``: records to raise instead (zeros or an earlier maximum of the same tensors)."""
    L = _lib.load()
    n = len(tensors)
    dev = tensors[0].device
    if out is None:
        out = amax_records(n, dev)
    if out.shape[-1] != AMAX_WORDS or out.numel() != n * AMAX_WORDS or not out.is_contiguous():
        raise _lib.GpsHipError("absmax: `out` must be contiguous int32 [len(tensors), 512]")
    base = out.data_ptr()
    for i0 in range(0, n, MAX_ABS):
        chunk = tensors[i0:i0 + MAX_ABS]
        descs = (_lib.AbsmaxDesc * len(chunk))()
        for j, (q, t) in enumerate(zip(descs, chunk)):
            if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
                raise _lib.GpsHipError("absmax: fp32 [rows, cols] CUDA matrices with unit column stride")
            q.A, q.ld, q.rows, q.cols, q.slot = t.data_ptr(), t.stride(0), t.shape[0], t.shape[1], base + 4 * AMAX_WORDS * (i0 + j)
        check(L.gps_absmax(len(chunk), descs, current_stream(dev)), "gps_absmax")
    return out


def supported(N: int, K: int) -> bool:
    """Shapes the ring kernel tiles: column panels of 192 / 128 / 64 (the last one may be partial), 32-wide k-stages
    (the last one may be partly empty)."""
    return ENABLED and N > 0 and K > 0 and N % 4 == 0 and K % 4 == 0


_stats_ok = {}


def stats_supported(M: int, N: int, K: int) -> bool:
    """Whether ``gemm_panel_stats`` serves ``[M, K] x [K, N]`` (whole column panels and k-stages, <= 961 row tiles)."""
    key = (M, N, K)
    v = _stats_ok.get(key)
    if v is None:
        v = _stats_ok[key] = bool(supported(N, K) and _lib.load().gps_gemm_stats_supported(M, N, K))
    return v


def amax_value(rec: torch.Tensor) -> float:
    """The maximum a record holds (a host read: tests and tools only)."""
    return float(rec.reshape(-1).max().reshape(1).view(torch.float32))


def record_of(value: float, device) -> torch.Tensor:
    """A record holding ``value`` (tests: a deliberately larger bound)."""
    r = amax_records(1, device)[0]
    r[0] = torch.tensor([value], dtype=torch.float32).view(torch.int32)[0]
    return r


def _a_word(a: torch.Tensor, a_amax: Optional[torch.Tensor]) -> int:
    """Address of max|a|'s record: the caller's, or one made here (a pre-pass over ``a``)."""
    if a_amax is None:
        a_amax = absmax([a])
    elif a_amax.numel() != AMAX_WORDS:
        raise _lib.GpsHipError("a_amax: one max|.| record (int32 [512])")
    return a_amax.data_ptr()


def gemm_panel_stats(a: torch.Tensor, image, N: int, bias: Optional[torch.Tensor], addend: torch.Tensor,
                     p_drop: float, seed: int, bn_desc, sync_ptr: int, a_amax: Optional[torch.Tensor] = None,
                     m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = addend + dropout(a @ B^T + bias; p_drop, seed)`` and the batch statistics of ``out`` over its rows
    (-> ``bn_desc.mean / rstd`` + running statistics), complete when the launch retires: the residual + dropout +
    statistics pass of ``norm1_attn`` / ``norm2`` (graphgps/layer/gps_layer.py:212-217,225-229) in the GEMM's epilogue.
    ``sync_ptr``: arrival counters (``norm.SyncArena.site``), zero at entry and exit."""
    import ctypes
    L = _lib.load()
    M, K = a.shape
    if a.stride(1) != 1 or a.dtype != torch.float32:
        raise _lib.GpsHipError("gemm_panel_stats: fp32 A with unit column stride")
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    wsf = L.gps_gemm_stats_floats(M, N, K)
    ws = torch.empty(wsf, dtype=torch.float32, device=a.device)
    if getattr(image, "amax", None) is not None:
        check(L.gps_gemm16_panel_stats(ptr(a), a.stride(0), M, K, _a_word(a, a_amax), ptr(image), ptr(image.amax), N,
                                       ptr(bias), ptr(addend), addend.stride(0), ptr(out), out.stride(0), float(p_drop),
                                       int(seed), ctypes.byref(bn_desc), ptr(ws), wsf, sync_ptr, ptr(m_dev),
                                       current_stream(a.device)), "gps_gemm16_panel_stats")
        return out
    if m_dev is not None:
        raise _lib.GpsHipError("gemm_panel_stats: padded batches (m_dev) need the fp16-form image")
    check(L.gps_gemm_panel_stats(ptr(a), a.stride(0), M, K, ptr(image), N, ptr(bias), ptr(addend), addend.stride(0),
                                 ptr(out), out.stride(0), float(p_drop), int(seed), ctypes.byref(bn_desc), ptr(ws), wsf,
                                 sync_ptr, current_stream(a.device)), "gps_gemm_panel_stats")
    return out


def _split_weights16(weights, nt, tn):
    """fp16 images: one ``gps_absmax`` launch over all weights, then one split launch per 48 of them."""
    L = _lib.load()
    dev = weights[0].device
    for w in weights:
        if w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1 or not w.is_cuda:
            raise _lib.GpsHipError("split_weights: fp32 [rows, cols] CUDA weights with unit column stride")
    words = absmax(list(weights))
    out = []
    for i0 in range(0, len(weights), MAX_SPLIT16):
        chunk = weights[i0:i0 + MAX_SPLIT16]
        descs = (_lib.GemmSplit16 * len(chunk))()
        for j, (q, w) in enumerate(zip(descs, chunk)):
            rows, cols = w.shape
            i_nt = torch.empty(L.gps_gemm16_image_elems(rows, cols), dtype=torch.int16, device=dev) if nt else None
            i_tn = torch.empty(L.gps_gemm16_image_elems(cols, rows), dtype=torch.int16, device=dev) if tn else None
            word = words[i0 + j]
            q.W, q.ldw, q.rows, q.cols = w.data_ptr(), w.stride(0), rows, cols
            q.image_nt = i_nt.data_ptr() if nt else None
            q.image_tn = i_tn.data_ptr() if tn else None
            q.amax = word.data_ptr()
            out.append((WImage(i_nt, word) if nt else None, WImage(i_tn, word) if tn else None))
        check(L.gps_gemm16_split_weights(len(chunk), descs, current_stream(dev)), "gps_gemm16_split_weights")
    return out


def split_weights(weights: Sequence[torch.Tensor], nt: bool = True, tn: bool = True, f16: Optional[bool] = None
                  ) -> List[Tuple[Optional["WImage"], Optional["WImage"]]]:
    """[(image of W, image of W^T), ...] for fp32 weights ``[rows, cols]``, up to 56 per launch (the five projections
    of every block of a 10-layer stack: ONE launch per step, layer/gps_block.py stack_begin).
    ``image of W`` serves ``x @ W.T`` (forward), ``image of W^T`` serves ``g @ W`` (input gradient)."""
    if F16 if f16 is None else f16:
        return _split_weights16(list(weights), nt, tn)
    if len(weights) > MAX_SPLIT:
        out = []
        for i in range(0, len(weights), MAX_SPLIT):
            out += split_weights(weights[i:i + MAX_SPLIT], nt, tn, False)
        return out
    L = _lib.load()
    n = len(weights)
    descs = (_lib.GemmSplit * n)()
    out = []
    dev = weights[0].device
    for q, w in zip(descs, weights):
        if w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1 or not w.is_cuda:
            raise _lib.GpsHipError("split_weights: fp32 [rows, cols] CUDA weights with unit column stride")
        rows, cols = w.shape
        # padded to whole column panels / k-stages by the library's own geometry; the split kernel writes the zeros of
        # the k padding, the surplus rows of the last panel feed columns that are never stored
        i_nt = torch.empty(L.gps_gemm_image_elems(rows, cols), dtype=torch.int16, device=dev) if nt else None
        i_tn = torch.empty(L.gps_gemm_image_elems(cols, rows), dtype=torch.int16, device=dev) if tn else None
        q.W, q.ldw, q.rows, q.cols = w.data_ptr(), w.stride(0), rows, cols
        q.image_nt = i_nt.data_ptr() if nt else None
        q.image_tn = i_tn.data_ptr() if tn else None
        out.append((WImage(i_nt) if nt else None, WImage(i_tn) if tn else None))
    check(L.gps_gemm_split_weights(n, descs, current_stream(dev)), "gps_gemm_split_weights")
    return out


def gemm_panel(a: torch.Tensor, image, N: int, bias: Optional[torch.Tensor] = None,
               addend: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, epilogue: int = 0,
               mask_src: Optional[torch.Tensor] = None, p_drop: float = 0.0, seed: int = 0,
               a_amax: Optional[torch.Tensor] = None, c_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = a @ B^T (+ bias) (+ addend)`` where ``image`` is the split image of B ``[N, K]``.
    ``a`` may be a column slice of a wider buffer (row stride >= K); ``out`` likewise (row stride >= N).
    epilogue 1: ReLU then dropout(p_drop, seed); epilogue 2: multiply by the ReLU/dropout mask of ``mask_src``.
    fp16-form images: ``a_amax`` = the word of max|a| (made by a pre-pass here when absent); ``c_amax`` (int32 [1], zero or an
    earlier maximum) is raised to max|out| by the kernel's epilogue -- the word of the next GEMM that reads ``out``."""
    L = _lib.load()
    M, K = a.shape
    if a.stride(1) != 1 or a.dtype != torch.float32:
        raise _lib.GpsHipError("gemm_panel: fp32 A with unit column stride")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if M == 0:
        return out
    if getattr(image, "amax", None) is not None:       # fp16 form: ``a_amax`` = the word of max|a| (made here when absent)
        check(L.gps_gemm16_panel(ptr(a), a.stride(0), M, K, _a_word(a, a_amax), ptr(image), ptr(image.amax), N, ptr(bias),
                                 ptr(addend), addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0),
                                 int(epilogue), ptr(mask_src), mask_src.stride(0) if mask_src is not None else 0,
                                 float(p_drop), int(seed), ptr(c_amax), current_stream(a.device)), "gps_gemm16_panel")
        return out
    if c_amax is not None:
        raise _lib.GpsHipError("gemm_panel: c_amax needs an fp16-form image")
    check(L.gps_gemm_panel(ptr(a), a.stride(0), M, K, ptr(image), N, ptr(bias), ptr(addend),
                           addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), int(epilogue),
                           ptr(mask_src), mask_src.stride(0) if mask_src is not None else 0, float(p_drop),
                           int(seed), current_stream(a.device)), "gps_gemm_panel")
    return out


def gemm_panel_pair(first, second):
    """Two independent ``gemm_panel`` products (epilogue 0) in ONE dispatch (``gps_gemm16_panel_pair``): each of ``first`` /
    ``second`` is ``dict(a=, image=, N=, bias=None, addend=None, out=None, a_amax=None, c_amax=None)``.  Workgroups of
    ``first`` are dispatched first: pass the product with the longer contraction there.  Needs fp16-form images; returns the
    two outputs."""
    import ctypes
    L = _lib.load()
    probs, outs = [], []
    for q in (first, second):
        a, image, N = q["a"], q["image"], q["N"]
        if getattr(image, "amax", None) is None:
            raise _lib.GpsHipError("gemm_panel_pair: fp16-form images only")
        M, K = a.shape
        if a.stride(1) != 1 or a.dtype != torch.float32:
            raise _lib.GpsHipError("gemm_panel_pair: fp32 A with unit column stride")
        out = q.get("out")
        if out is None:
            out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        addend, bias = q.get("addend"), q.get("bias")
        P = _lib.Gemm16Problem()
        P.A, P.lda, P.M, P.K, P.N = a.data_ptr(), a.stride(0), M, K, N
        P.a_amax = _a_word(a, q.get("a_amax"))
        P.image, P.w_amax = image.data_ptr(), image.amax.data_ptr()
        P.bias = bias.data_ptr() if bias is not None else None
        P.Cin, P.ldcin = (addend.data_ptr(), addend.stride(0)) if addend is not None else (None, 0)
        P.C, P.ldc = out.data_ptr(), out.stride(0)
        c_amax = q.get("c_amax")
        P.c_amax = c_amax.data_ptr() if c_amax is not None else None
        probs.append(P)
        outs.append(out)
    check(L.gps_gemm16_panel_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), current_stream(outs[0].device)),
          "gps_gemm16_panel_pair")
    return outs[0], outs[1]


_colsums_ok = {}


def colsums_supported(M: int, N: int, K: int) -> bool:
    """Whether ``gemm_panel_sums`` serves ``[M, K] x [K, N]`` (whole 128 / 192-column panels)."""
    key = (M, N, K)
    v = _colsums_ok.get(key)
    if v is None:
        v = _colsums_ok[key] = bool(supported(N, K) and _lib.load().gps_gemm_colsums_supported(M, N, K))
    return v


def _problem(q):
    a, image, N = q["a"], q["image"], q["N"]
    if getattr(image, "amax", None) is None:
        raise _lib.GpsHipError("gemm_panel_sums: fp16-form images only")
    M, K = a.shape
    if a.stride(1) != 1 or a.dtype != torch.float32:
        raise _lib.GpsHipError("gemm_panel_sums: fp32 A with unit column stride")
    out = q.get("out")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    addend, bias = q.get("addend"), q.get("bias")
    P = _lib.Gemm16Problem()
    P.A, P.lda, P.M, P.K, P.N = a.data_ptr(), a.stride(0), M, K, N
    P.a_amax = _a_word(a, q.get("a_amax"))
    P.image, P.w_amax = image.data_ptr(), image.amax.data_ptr()
    P.bias = bias.data_ptr() if bias is not None else None
    P.Cin, P.ldcin = (addend.data_ptr(), addend.stride(0)) if addend is not None else (None, 0)
    P.C, P.ldc = out.data_ptr(), out.stride(0)
    c_amax = q.get("c_amax")
    P.c_amax = c_amax.data_ptr() if c_amax is not None else None
    return P, out, (M, N, K)


def gemm_panel_sums(prob, sums, sync_ptr: int) -> torch.Tensor:
    """``out = addend + a @ B^T`` (``prob``: a ``gemm_panel_pair`` dict with an addend) where ``out`` is the output gradient of
    two BatchNorms -- ``sums = dict(z=, bn=, sum_g=, sum_gz=, z2=, bn2=, sum_g2=, sum_gz2=)``, ``bn`` / ``bn2``:
    ``norm.bn_desc`` structures, the sums: fp32 [N] tensors -- whose backward column sums S1 = sum g, S2 = sum g zhat leave
    with the launch (``gps_gemm16_panel_sums``) instead of a ``norm.bwd_partial`` pass over ``out`` and both BatchNorm inputs.
    ``sync_ptr``: arrival counters (``norm.SyncArena.site``), ``gps_gemm_stats_sync_words(N)`` words, zero at entry and exit."""
    import ctypes
    L = _lib.load()
    P, out, (M, N, K) = _problem(prob)
    S = _lib.GemmColsums()
    z, z2 = sums["z"], sums["z2"]
    S.z, S.ldz, S.bn = z.data_ptr(), z.stride(0), ctypes.addressof(sums["bn"])
    S.sum_g, S.sum_gz = sums["sum_g"].data_ptr(), sums["sum_gz"].data_ptr()
    S.z2, S.ldz2, S.bn2 = z2.data_ptr(), z2.stride(0), ctypes.addressof(sums["bn2"])
    S.sum_g2, S.sum_gz2 = sums["sum_g2"].data_ptr(), sums["sum_gz2"].data_ptr()
    wsf = L.gps_gemm_colsums_floats(M, N, K)
    ws = torch.empty(max(wsf, 4), dtype=torch.float32, device=z.device)
    S.ws, S.ws_floats, S.sync = ws.data_ptr(), wsf, sync_ptr
    check(L.gps_gemm16_panel_sums(ctypes.byref(P), ctypes.byref(S), current_stream(out.device)), "gps_gemm16_panel_sums")
    return out
