"""Stub of torch_geometric.nn: ``Linear`` (== torch.nn.Linear for fixed in_channels: weight
[out,in], bias, kaiming-uniform(a=sqrt(5)) init) and the conv classes."""
import torch
from . import conv  # noqa: F401
from .conv import (MessagePassing, GINEConv, GCNConv, GINConv, GENConv, GATConv,  # noqa: F401
                   PNAConv)


class Linear(torch.nn.Linear):
    def __init__(self, in_channels, out_channels, bias=True, **kwargs):
        super().__init__(in_channels, out_channels, bias=bias)


class _Norm:
    class LayerNorm(torch.nn.Module):
        """torch_geometric.nn.norm.LayerNorm (PyG 2.x, ``mode='graph'`` -- the default, and what
        graphgps/layer/gps_layer.py:129-134,148 constructs with ``LayerNorm(dim_h)`` and calls as ``norm(h, batch.batch)``):
        every graph is normalised over ALL its nodes and channels together,
            x <- x - mean_g;  out = x / sqrt(var_g + eps)  with mean / var over n_g * C elements (biased),
        then the per-channel affine ``out * weight + bias`` (weight = 1, bias = 0 at reset).  Restated from the
        published implementation (the package is absent here): third-party semantics, see oracle/ref_stubs/README.md."""

        def __init__(self, in_channels, eps=1e-5, affine=True, mode='graph'):
            super().__init__()
            if mode != 'graph':
                raise NotImplementedError("stub: mode='graph' only (what the reference uses)")
            self.in_channels, self.eps, self.mode = in_channels, eps, mode
            if affine:
                self.weight = torch.nn.Parameter(torch.ones(in_channels))
                self.bias = torch.nn.Parameter(torch.zeros(in_channels))
            else:
                self.register_parameter('weight', None)
                self.register_parameter('bias', None)

        def reset_parameters(self):
            if self.weight is not None:
                torch.nn.init.ones_(self.weight)
                torch.nn.init.zeros_(self.bias)

        def forward(self, x, batch=None, batch_size=None):
            if batch is None:
                x = x - x.mean()
                out = x / (x.std(unbiased=False) + self.eps)
            else:
                if batch_size is None:
                    batch_size = int(batch.max()) + 1
                norm = torch.bincount(batch, minlength=batch_size).to(x.dtype).clamp_(min=1)
                norm = norm.mul_(x.size(-1)).view(-1, 1)
                mean = torch.zeros(batch_size, x.size(-1), dtype=x.dtype).index_add_(0, batch, x).sum(dim=-1, keepdim=True) / norm
                x = x - mean.index_select(0, batch)
                var = torch.zeros(batch_size, x.size(-1), dtype=x.dtype).index_add_(0, batch, x * x).sum(dim=-1, keepdim=True)
                var = var / norm
                out = x / (var + self.eps).sqrt().index_select(0, batch)
            if self.weight is not None:
                out = out * self.weight + self.bias
            return out


norm = _Norm


class inits:  # noqa: N801  (module-like namespace: torch_geometric.nn.inits)
    @staticmethod
    def reset(value):
        """PyG ``inits.reset``: reset_parameters() on the module, else recursively on its children."""
        if hasattr(value, 'reset_parameters'):
            value.reset_parameters()
        else:
            for child in value.children() if hasattr(value, 'children') else []:
                inits.reset(child)
