"""GCNConv on the HIP sparse core: the ``GCN`` local model of ``GPSLayer``.

Stands in for ``torch_geometric.nn.GCNConv(dim_h, dim_h)`` (PyG 2.2, third-party), constructed at
``/root/reference/graphgps/layer/gps_layer.py:53-55`` and called as ``local_model(h, batch.edge_index)``
(:183).  Same parameters as the PyG module (``lin.weight`` [out, in] glorot-initialised, no bias inside
``lin``; ``bias`` [out] zeros), same arithmetic: ``D^-1/2 (A + I) D^-1/2 (x W^T) + b`` with ``gcn_norm``'s
degree (csrc/gcn.hip)."""
import math

import torch
import torch.nn as nn

from ..fused import linear
from ..ops import gcn_aggregate


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))      # PyG glorot
        nn.init.uniform_(self.lin.weight, -a, a)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward_tensors(self, x, gi):
        out = gcn_aggregate(linear(x, self.lin.weight, None), gi)
        return out if self.bias is None else out + self.bias

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}'
