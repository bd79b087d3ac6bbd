"""Batched-graph container with the attribute surface the GPS path reads.

Stands in for ``torch_geometric.data.Batch`` (third-party, absent here).  The hot
path reads ``x``, ``edge_index``, ``edge_attr``, ``batch`` and assigns ``x`` /
``edge_attr`` (``/root/reference/graphgps/layer/gps_layer.py:156,167-174,199,231``);
loader batches additionally carry ``ptr`` (cumulative node counts), ``y`` and the
precomputed ``pestat_*`` tensors.  A real PyG ``Batch`` works wherever this class
does: nothing below relies on more than attribute access.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional

import torch


class Batch:
    def __init__(self, **kwargs: Any):
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- PyG-like conveniences -------------------------------------------
    def keys(self) -> List[str]:
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    @property
    def num_graphs(self) -> int:
        if "_num_graphs" in self.__dict__:
            return self.__dict__["_num_graphs"]
        if "ptr" in self.__dict__ and self.ptr is not None:
            return int(self.ptr.numel() - 1)
        if "y" in self.__dict__ and torch.is_tensor(self.y) and self.y.dim() >= 1:
            return int(self.y.shape[0])
        # last resort: one device->host sync, what PyG does as well
        return int(self.batch.max().item()) + 1 if self.batch.numel() else 0

    @num_graphs.setter
    def num_graphs(self, value: int) -> None:
        self.__dict__["_num_graphs"] = int(value)

    def to(self, device, non_blocking: bool = False) -> "Batch":
        # while ``ptr`` is still on the host: note the longest graph (a free kernel-selection hint for the device
        # side, ops._host_max_graph_nodes)
        p = self.__dict__.get("ptr")
        if torch.is_tensor(p) and not p.is_cuda and p.numel() > 1:
            self.__dict__.setdefault("_gps_meta", {})["nmax"] = int((p[1:] - p[:-1]).max())
        for k, v in list(self.__dict__.items()):
            if k == "_gps_index":  # device-side CSR cache is tied to the old device
                del self.__dict__[k]
            elif torch.is_tensor(v):
                self.__dict__[k] = v.to(device, non_blocking=non_blocking)
        return self

    def shallow_copy(self) -> "Batch":
        """A new batch object over the SAME tensors (no device copies) without the cached graph index:
        what a loader hands the step for the next batch of the same storage.  The GPS path never writes
        into the tensors it is given (it re-assigns ``batch.x`` / ``batch.edge_attr``)."""
        # ``_gps_meta``: a small host-side record shared BY REFERENCE by every shallow copy of this batch (what
        # the host knows about it without touching the device: e.g. the longest graph, ops._host_max_graph_nodes)
        self.__dict__.setdefault("_gps_meta", {})
        out = Batch()
        for k, v in self.__dict__.items():
            if k != "_gps_index":
                out.__dict__[k] = v
        return out

    def clone(self) -> "Batch":
        out = Batch()
        for k, v in self.__dict__.items():
            if k in ("_gps_index", "_gps_meta"):
                continue
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
        out.__dict__["_gps_meta"] = dict(self.__dict__.get("_gps_meta") or {})
        return out

    def __repr__(self) -> str:
        parts = []
        for k in self.keys():
            v = self.__dict__[k]
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return f"Batch({', '.join(parts)})"

    # -- construction ------------------------------------------------------
    @staticmethod
    def from_graph_list(graphs: Iterable[Dict[str, torch.Tensor]]) -> "Batch":
        """Block-diagonal batching (PyG ``Batch.from_data_list`` semantics): node
        tensors are concatenated, ``edge_index`` (any ``*index*`` key) is offset by the running
        node count, ``batch``/``ptr`` are emitted.  A key is a node tensor if its first
        dim equals the graph's ``x.shape[0]``, an edge tensor if it equals
        ``edge_index.shape[1]``, else per-graph."""
        graphs = list(graphs)
        cat: Dict[str, List[torch.Tensor]] = {}
        batch_vec, ptr, off = [], [0], 0
        for gi, g in enumerate(graphs):
            n = int(g["x"].shape[0])
            for k, v in g.items():
                if "index" in k:      # PyG: attributes named *index* hold node ids (edge_index, and
                    v = v + off       # Graphormer's all-pairs graph_index) -> shifted, joined on dim -1
                cat.setdefault(k, []).append(v)
            batch_vec.append(torch.full((n,), gi, dtype=torch.long))
            off += n
            ptr.append(off)
        out = Batch()
        for k, vs in cat.items():
            out.__dict__[k] = torch.cat(vs, dim=-1 if "index" in k else 0)
        out.batch = torch.cat(batch_vec) if batch_vec else torch.zeros(0, dtype=torch.long)
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.num_graphs = len(graphs)
        return out
