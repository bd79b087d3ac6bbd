"""Positional/structural encodings computed on the MI355X for a whole batch of graphs.

``rw_landing_probs`` is the batched GPU form of ``get_rw_landing_probs``
(``/root/reference/graphgps/transform/posenc_stats.py:184-230``), which the reference runs per graph on the
CPU at dataset-preprocessing time (``posenc_stats.py:93-99`` -> ``data.pestat_RWSE``, consumed by the
``RWSE`` node encoder, ``graphgps/encoder/kernel_pos_encoder.py:79-103``).  One HIP launch per batch
(csrc/rwse.hip); edge weights are not supported (the reference never passes any for RWSE).
"""
from typing import Optional, Sequence

import torch

from . import lib as _lib
from .lib import check, current_stream, ptr as _p
from .ops import GraphIndex, build_graph_index


def rw_landing_probs(ksteps: Sequence[int], edge_index: torch.Tensor, ptr: torch.Tensor,
                     num_nodes: Optional[int] = None, space_dim: float = 0,
                     gi: Optional[GraphIndex] = None) -> torch.Tensor:
    """``[num_nodes, len(ksteps)]`` random-walk landing probabilities ``diag(P^k)``, ``P = D^-1 A`` per
    graph (out-degree normalisation, ``1/0 -> 0``), times ``k ** (space_dim / 2)``.

    ``ptr`` are the graph offsets of the batch (``batch.ptr``, int64 ``[B+1]``) on the same device as
    ``edge_index`` (int64 ``[2, E]``)."""
    ksteps = [int(k) for k in ksteps]
    if not ksteps or min(ksteps) < 1:
        raise ValueError("ksteps must be positive integers")
    dev = edge_index.device
    if dev.type != "cuda":
        raise _lib.GpsHipError("rw_landing_probs is a HIP kernel: tensors must be on the MI355X")
    L = _lib.load()
    B = int(ptr.numel()) - 1
    N = int(num_nodes) if num_nodes is not None else int(ptr[-1])
    if gi is None:
        gi = build_graph_index(edge_index, N, B, ptr_vec=ptr)
    kmin, kmax = min(ksteps), max(ksteps)
    K = kmax - kmin + 1
    sizes = (ptr[1:] - ptr[:-1]).to(torch.int64)
    need = torch.where(sizes > L.gps_rwse_lds_nodes(), 3 * sizes * sizes, torch.zeros_like(sizes))
    off = torch.cumsum(need, 0) - need
    total = int(need.sum())                     # one sync: this is preprocessing, not the training step
    scratch = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    out = torch.zeros(N, K, dtype=torch.float32, device=dev)
    check(L.gps_rwse(_p(gi.rowptr_src), _p(gi.dst_by_src), _p(gi.ptr), B, N, kmin, kmax, float(space_dim),
                     _p(scratch), _p(off.contiguous()), _p(out), current_stream(dev)), "gps_rwse")
    if ksteps == list(range(kmin, kmax + 1)):
        return out
    return out[:, [k - kmin for k in ksteps]].contiguous()


def add_rwse(batch, ksteps: Sequence[int], space_dim: float = 0):
    """Set ``batch.pestat_RWSE`` for a batch that lives on the GPU (what ``compute_posenc_stats`` stores
    per graph at posenc_stats.py:93-99)."""
    batch.pestat_RWSE = rw_landing_probs(ksteps, batch.edge_index, batch.ptr, batch.x.shape[0], space_dim)
    return batch
