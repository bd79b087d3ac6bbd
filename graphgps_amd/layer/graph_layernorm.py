"""``gt.layer_norm=True``: PyG's graph-mode ``LayerNorm`` as the reference's ``GPSLayer`` uses it
(``/root/reference/graphgps/layer/gps_layer.py:129-134,148`` construct ``pygnn.norm.LayerNorm(dim_h)``; ``:191-192,209-210,
226-227`` call it as ``norm(h, batch.batch)``): every graph is normalised over ALL of its nodes and channels together
(mean and biased variance over ``n_g * dim_h`` elements, eps = 1e-5), then a per-channel affine (``weight`` = 1, ``bias`` = 0 at
construction: the reference's ``state_dict`` keys ``norm*.weight`` / ``norm*.bias``).

No ``configs/**/*.yaml`` of the reference sets ``gt.layer_norm``; the branch exists there, so it exists here -- on the
operator path (the fused blocks are the BatchNorm configurations): the per-graph sums are the deterministic ptr-segmented
HIP pooling (``csrc/segment_pool.hip``, which also carries the backward), the element-wise rest is device torch ops.
"""
import torch
import torch.nn as nn

from ..ops import GraphIndex, _node_graph, segment_pool


class GraphLayerNorm(nn.Module):
    def __init__(self, in_channels: int, eps: float = 1e-5, affine: bool = True, mode: str = 'graph'):
        super().__init__()
        if mode != 'graph':
            raise NotImplementedError("GraphLayerNorm: mode='graph' (PyG's default, what GPSLayer uses) only")
        self.in_channels, self.eps, self.mode = in_channels, eps, mode
        if affine:
            self.weight = nn.Parameter(torch.ones(in_channels))
            self.bias = nn.Parameter(torch.zeros(in_channels))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.ones_(self.weight)
            nn.init.zeros_(self.bias)

    def forward(self, x: torch.Tensor, gi: GraphIndex) -> torch.Tensor:
        """``x`` [N, C] fp32 on the device, ``gi`` the batch's graph index (``ptr`` segments = PyG's ``batch`` vector)."""
        C = x.shape[-1]
        node_graph = _node_graph(gi)[:gi.N].long()                       # graph id per node
        sizes = (gi.ptr[1:] - gi.ptr[:-1]).to(x.dtype).clamp(min=1) * C      # degree(batch).clamp_(min=1) * C
        mean = segment_pool(x, gi, "add").sum(dim=-1) / sizes            # [B]
        xc = x - mean.index_select(0, node_graph).unsqueeze(-1)
        var = segment_pool(xc * xc, gi, "add").sum(dim=-1) / sizes
        out = xc / (var + self.eps).sqrt().index_select(0, node_graph).unsqueeze(-1)
        if self.weight is not None:
            out = out * self.weight + self.bias
        return out

    def extra_repr(self):
        return f'{self.in_channels}, mode={self.mode}'
