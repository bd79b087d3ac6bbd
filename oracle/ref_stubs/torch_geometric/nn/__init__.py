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
        def __init__(self, *a, **k):
            raise NotImplementedError


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
