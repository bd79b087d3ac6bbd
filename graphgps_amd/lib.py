"""ctypes binding of ``libgps_hip.so`` (C ABI declared in ``include/gps_hip.h``).

The product path has NO fallback: if the shared library is missing, stale or bound to a
different HIP runtime than PyTorch's, importing/using the ops raises immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch  # noqa: F401  -- must be imported first: it maps the HIP runtime our .so binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPS_HIP_LIB: an alternative build of the same ABI (A/B timing of kernel variants in one process launch each)
LIB_PATH = os.environ.get("GPS_HIP_LIB") or os.path.join(_HERE, "csrc", "libgps_hip.so")
ABI_VERSION = 11

_lib: Optional[ctypes.CDLL] = None

_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "gps_abi_version": (c_int, []),
    "gps_last_error": (c_char_p, []),
    "gps_set_dropout_salt": (c_int, [_P]),
    "gps_graph_index_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gps_graph_index_build": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gps_segment_ptr_from_batch": (c_int, [_P, c_int64, c_int64, _P, _P]),
    "gps_attn_tile_map": (c_int, [_P, c_int64, c_int64, _P, _P, _P]),
    "gps_gatedgcn_fwd": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int,
                                 _P, _P, _P, _P]),
    "gps_gatedgcn_stats_floats": (c_size_t, [c_int64, c_int]),
    "gps_gatedgcn_fwd_stats": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int,
                                       _P, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "gps_gatedgcn_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P,
                                 c_int64, c_int64, c_int, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "gps_gatedgcn_bwd_bn": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P,
                                 c_int64, c_int64, c_int, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "gps_gine_fwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_float, _P, _P, _P]),
    "gps_gine_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_float, _P,
                             _P, _P, _P]),
    "gps_node_graph_from_ptr": (c_int, [_P, c_int64, _P, _P]),
    "gps_embedding_grad_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "gps_embedding_grad_supported": (c_int, [c_int]),
    "gps_embedding_grad": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, _P, _P, c_size_t, _P]),
    "gps_segment_pool_fwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "gps_segment_pool_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P]),
    "gps_embed_sum": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, c_int, _P, _P]),
    "gps_small_linear_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    "gps_small_linear_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, c_int64,
                                     _P, c_int64, _P, _P]),
    "gps_multihot_columns": (c_int, [c_int, _P]),
    "gps_multihot_fill": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P]),
    "gps_segment_pool_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "gps_segment_pool_fwd_sliced": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, _P, _P, c_size_t, _P]),
    "gps_bn_workspace_floats": (c_size_t, [c_int64, c_int]),
    "gps_bn_stats": (c_int, [_P, c_int64, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "gps_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, c_uint64, _P, _P]),
    "gps_bn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, c_uint64, _P, _P, _P,
                           _P, _P]),
    "gps_act_drop_add": (c_int, [_P, _P, c_int64, c_int, c_int, c_float, c_uint64, _P, _P]),
    "gps_act_drop_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_float, c_uint64, _P, _P]),
    "gps_colsum": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "gps_wgrad_workspace_floats": (c_size_t, [c_int64, c_int, c_int]),
    "gps_wgrad": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "gps_wgrad16": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "gps_optim_chunk": (c_int, []),
    "gps_adamw_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "gps_gemm_nt": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, _P, c_int64, _P, c_int64,
                            _P]),
    "gps_wgrad_grouped_workspace_floats": (c_size_t, [c_int, _P]),
    "gps_wgrad_grouped": (c_int, [c_int, _P, _P, _P]),
    "gps_norm_tree_floats": (c_size_t, [c_int64, c_int]),
    "gps_norm_sync_words": (c_int, []),
    "gps_sync_reset": (c_int, [_P, c_size_t, _P]),
    "gps_sync_nonzero": (c_int, [_P, c_size_t, _P, _P]),
    "gps_norm_fwd": (c_int, [c_int, _P, c_int, _P, c_size_t, _P, _P]),
    "gps_norm_bwd_partial": (c_int, [c_int, _P, c_int, _P, c_size_t, _P, _P]),
    "gps_norm_bwd_apply": (c_int, [c_int, _P, c_int, _P, c_size_t, _P, _P]),
    "gps_rwse_lds_nodes": (c_int, []),
    "gps_rwse": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, c_int, c_float, _P, _P, _P, _P]),
    "gps_segment_max_len": (c_int, [_P, c_int64, _P, _P]),
    "gps_segment_max_len_real": (c_int, [_P, c_int64, _P, _P, _P]),
    "gps_favor_workspace_floats": (c_size_t, [c_int64, c_int64, c_int]),
    "gps_favor_fwd": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int,
                              c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gps_favor_bwd": (c_int, [_P, _P, c_int64, _P, c_int, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64,
                              c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "gps_attn_supported_head_dim": (c_int, [c_int]),
    "gps_seg_attn_fwd": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_float,
                                 c_float, c_uint64, _P, _P, c_int64, c_int64, _P, _P, _P]),
    "gps_seg_attn_bwd": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int,
                                 c_float, c_float, c_uint64, _P, _P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "gps_attn_graph_order": (c_int, [_P, c_int64, c_int, _P, _P]),
    "gps_edge_attn_supported": (c_int, [c_int, c_int]),
    "gps_edge_attn_fwd": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_float, c_int,
                                  _P, _P, _P, _P, _P]),
    "gps_edge_attn_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64,
                                  c_int64, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    "gps_gemm_image_elems": (c_size_t, [c_int64, c_int64]),
    "gps_gemm_panel_supported": (c_int, [c_int64, c_int64]),
    "gps_gemm_split_weights": (c_int, [c_int, _P, _P]),
    "gps_gemm_panel_trace": (c_int, [_P]),
    "gps_gemm_panel": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int, _P, _P, c_int64, _P, c_int64, c_int, _P,
                               c_int64, c_float, c_uint64, _P]),
    "gps_gemm_stats_floats": (c_size_t, [c_int64, c_int, c_int]),
    "gps_gemm_stats_sync_words": (c_int, [c_int]),
    "gps_gemm_stats_supported": (c_int, [c_int64, c_int, c_int]),
    "gps_gemm_panel_stats": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int, _P, _P, c_int64, _P, c_int64, c_float,
                                     c_uint64, _P, _P, c_size_t, _P, _P]),
    "gps_absmax": (c_int, [c_int, _P, _P]),
    "gps_gemm16_image_elems": (c_size_t, [c_int64, c_int64]),
    "gps_gemm16_split_weights": (c_int, [c_int, _P, _P]),
    "gps_gemm16_panel": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, c_int, _P, _P, c_int64, _P, c_int64, c_int, _P,
                                 c_int64, c_float, c_uint64, _P, _P]),
    "gps_gemm16_panel_stats": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, c_int, _P, _P, c_int64, _P, c_int64,
                                       c_float, c_uint64, _P, _P, c_size_t, _P, _P, _P]),
    "gps_gemm16_panel_pair": (c_int, [_P, _P, _P]),
    "gps_gemm_colsums_supported": (c_int, [c_int64, c_int, c_int]),
    "gps_gemm_colsums_floats": (c_size_t, [c_int64, c_int, c_int]),
    "gps_gemm16_panel_sums": (c_int, [_P, _P, _P]),
    "gps_gcn_dinv": (c_int, [_P, _P, c_int64, c_int64, _P, _P]),
    "gps_gcn_spmm": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int64, c_int, _P, _P]),
    "gps_adj_sum": (c_int, [_P, c_int64, _P, _P, c_float, c_int64, c_int64, c_int, _P, _P]),
    "gps_seg_attn_bias_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int64, c_int, c_int,
                                      c_float, c_float, c_uint64, _P, _P, _P]),
    "gps_seg_attn_bias_bwd": (c_int, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int64,
                                      c_int, c_int, c_float, c_float, c_uint64, _P, _P, c_int64, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class BnDesc(ctypes.Structure):
    """``gps_bn`` (include/gps_hip.h): one BatchNorm1d in training mode."""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("eps", c_float),
                ("momentum", c_float)]


class WgradProblem(ctypes.Structure):
    """``gps_wgrad_problem`` (include/gps_hip.h)."""
    _fields_ = [("g", c_void_p), ("x", c_void_p), ("gw", c_void_p), ("gb", c_void_p),
                ("ldg", c_int64), ("ldx", c_int64), ("R", c_int64), ("M", ctypes.c_int32),
                ("Nn", ctypes.c_int32), ("g_amax", c_void_p), ("x_amax", c_void_p)]


class BnBwdFold(ctypes.Structure):
    """``gps_bn_bwd_fold`` (include/gps_hip.h)."""
    _fields_ = [("bn", c_void_p), ("sum_g", c_void_p), ("sum_gz", c_void_p), ("p", c_float), ("seed", c_uint64),
                ("relu", ctypes.c_int32), ("rdev", c_void_p)]


class Gemm16Problem(ctypes.Structure):
    """``gps_gemm16_problem`` (include/gps_hip.h)."""
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("M", c_int64), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
                ("a_amax", c_void_p), ("image", c_void_p), ("w_amax", c_void_p), ("bias", c_void_p), ("Cin", c_void_p),
                ("ldcin", c_int64), ("C", c_void_p), ("ldc", c_int64), ("c_amax", c_void_p)]


class GemmColsums(ctypes.Structure):
    """``gps_gemm_colsums`` (include/gps_hip.h)."""
    _fields_ = [("z", c_void_p), ("ldz", c_int64), ("bn", c_void_p), ("sum_g", c_void_p), ("sum_gz", c_void_p),
                ("z2", c_void_p), ("ldz2", c_int64), ("bn2", c_void_p), ("sum_g2", c_void_p), ("sum_gz2", c_void_p),
                ("ws", c_void_p), ("ws_floats", c_size_t), ("sync", c_void_p)]


class GemmSplit(ctypes.Structure):
    """``gps_gemm_split`` (include/gps_hip.h)."""
    _fields_ = [("W", c_void_p), ("ldw", c_int64), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("image_nt", c_void_p), ("image_tn", c_void_p)]


class AbsmaxDesc(ctypes.Structure):
    """``gps_absmax_desc`` (include/gps_hip.h)."""
    _fields_ = [("A", c_void_p), ("ld", c_int64), ("rows", c_int64), ("cols", ctypes.c_int32), ("slot", c_void_p)]


class GemmSplit16(ctypes.Structure):
    """``gps_gemm_split16`` (include/gps_hip.h)."""
    _fields_ = [("W", c_void_p), ("ldw", c_int64), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("image_nt", c_void_p), ("image_tn", c_void_p), ("amax", c_void_p)]


class GpsHipError(RuntimeError):
    pass


def _hip_runtimes_mapped():
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


def load() -> ctypes.CDLL:
    """Load (once) and type the library.  Raises ``GpsHipError`` when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpsHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C graphgps_amd/csrc -j`. There is no CPU/PyTorch fallback "
            f"for the GPS hot path.")
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
    except OSError as e:
        raise GpsHipError(f"cannot dlopen {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GpsHipError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype, fn.argtypes = res, args
    if lib.gps_abi_version() != ABI_VERSION:
        raise GpsHipError(f"ABI mismatch: library {lib.gps_abi_version()} vs binding {ABI_VERSION}")
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise GpsHipError(
            "two HIP runtimes are mapped in this process (" + ", ".join(rts) + "): kernels would "
            "be launched on a runtime PyTorch does not own. Import torch before graphgps_amd.lib.")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gps_last_error()
        raise GpsHipError(f"{what or 'libgps_hip'} failed (code {rc}): "
                          f"{msg.decode() if msg else 'no message'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor as a ``c_void_p`` argument (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream(device: torch.device) -> Optional[int]:
    """Raw ``hipStream_t`` of torch's current stream on ``device`` (0 = the default stream).  Asked for ~230 times per
    training step: the raw query is ~0.3 us, ``torch.cuda.current_stream(device).cuda_stream`` builds a Stream object
    each time (~5 us, 1 ms of host time per step)."""
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    return _raw_stream(idx)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
    lambda idx: torch.cuda.current_stream(idx).cuda_stream)
