"""Row-panel GEMM (csrc/gemm_panel.hip) behind the nn.Linear modules of the GPS block.

``C = A W^T (+ bias) (+ addend)`` with optional ReLU/dropout epilogues, fp32 in / fp32 out, products formed exactly on
the bf16 MFMA pipe.  The weight operand is a pre-split IMAGE (three bf16 pieces, k-stage-major) made from the fp32
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


def gemm_panel_stats(a: torch.Tensor, image: torch.Tensor, N: int, bias: Optional[torch.Tensor], addend: torch.Tensor,
                     p_drop: float, seed: int, bn_desc, sync_ptr: int) -> torch.Tensor:
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
    check(L.gps_gemm_panel_stats(ptr(a), a.stride(0), M, K, ptr(image), N, ptr(bias), ptr(addend), addend.stride(0),
                                 ptr(out), out.stride(0), float(p_drop), int(seed), ctypes.byref(bn_desc), ptr(ws), wsf,
                                 sync_ptr, current_stream(a.device)), "gps_gemm_panel_stats")
    return out


def split_weights(weights: Sequence[torch.Tensor], nt: bool = True, tn: bool = True
                  ) -> List[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]]:
    """[(image of W, image of W^T), ...] for fp32 weights ``[rows, cols]``, up to 56 per launch (the five projections
    of every block of a 10-layer stack: ONE launch per step, layer/gps_block.py stack_begin).
    ``image of W`` serves ``x @ W.T`` (forward), ``image of W^T`` serves ``g @ W`` (input gradient)."""
    if len(weights) > MAX_SPLIT:
        out = []
        for i in range(0, len(weights), MAX_SPLIT):
            out += split_weights(weights[i:i + MAX_SPLIT], nt, tn)
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
        out.append((i_nt, i_tn))
    check(L.gps_gemm_split_weights(n, descs, current_stream(dev)), "gps_gemm_split_weights")
    return out


def gemm_panel(a: torch.Tensor, image: torch.Tensor, N: int, bias: Optional[torch.Tensor] = None,
               addend: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, epilogue: int = 0,
               mask_src: Optional[torch.Tensor] = None, p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    """``out = a @ B^T (+ bias) (+ addend)`` where ``image`` is the split image of B ``[N, K]``.
    ``a`` may be a column slice of a wider buffer (row stride >= K); ``out`` likewise (row stride >= N).
    epilogue 1: ReLU then dropout(p_drop, seed); epilogue 2: multiply by the ReLU/dropout mask of ``mask_src``."""
    L = _lib.load()
    M, K = a.shape
    if a.stride(1) != 1 or a.dtype != torch.float32:
        raise _lib.GpsHipError("gemm_panel: fp32 A with unit column stride")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    check(L.gps_gemm_panel(ptr(a), a.stride(0), M, K, ptr(image), N, ptr(bias), ptr(addend),
                           addend.stride(0) if addend is not None else 0, ptr(out), out.stride(0), int(epilogue),
                           ptr(mask_src), mask_src.stride(0) if mask_src is not None else 0, float(p_drop),
                           int(seed), current_stream(a.device)), "gps_gemm_panel")
    return out
