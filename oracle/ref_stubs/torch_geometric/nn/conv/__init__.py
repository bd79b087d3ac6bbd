"""Stub of torch_geometric.nn.conv.MessagePassing / GINEConv (PyG 2.2 published behaviour).

``propagate`` restates PyG's signature-driven dispatch: arguments of ``message`` /
``aggregate`` / ``update`` are looked up by name in the ``propagate`` kwargs; a ``_j`` suffix
gathers rows ``edge_index[0]`` (source), ``_i`` gathers ``edge_index[1]`` (target) for the
default ``flow='source_to_target'``; ``index`` is ``edge_index[1]``, ``dim_size`` is the number
of target nodes; default aggregation is a scatter-sum (``aggr='add'``)."""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    _special = {"edge_index", "index", "dim_size", "ptr", "size", "size_i", "size_j",
                "edge_index_i", "edge_index_j", "adj_t"}

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert aggr in ("add", "sum") and flow == "source_to_target" and node_dim == -2
        self.aggr = aggr

    @staticmethod
    def _params(fn, skip_first=False):
        names = list(inspect.signature(fn).parameters)
        return names[1:] if skip_first else names

    def _collect(self, names, edge_index, n_nodes, kwargs):
        out = {}
        for name in names:
            if name == "index" or name == "edge_index_i":
                out[name] = edge_index[1]
            elif name == "edge_index_j":
                out[name] = edge_index[0]
            elif name == "edge_index":
                out[name] = edge_index
            elif name in ("dim_size", "size_i", "size_j"):
                out[name] = n_nodes
            elif name in ("ptr", "size"):
                out[name] = None
            elif name.endswith("_i") or name.endswith("_j"):
                data = kwargs.get(name[:-2])
                if isinstance(data, (tuple, list)):
                    data = data[1 if name.endswith("_i") else 0]
                if data is None:
                    out[name] = None
                else:
                    idx = edge_index[1] if name.endswith("_i") else edge_index[0]
                    out[name] = data.index_select(-2, idx)
            else:
                out[name] = kwargs.get(name)
        return out

    def propagate(self, edge_index, size=None, **kwargs):
        n_nodes = None
        for v in kwargs.values():
            if isinstance(v, (tuple, list)):
                v = v[1] if v[1] is not None else v[0]
            if torch.is_tensor(v) and v.dim() >= 2:
                n_nodes = v.size(-2)
                break
        # node count = rows of the first node-level tensor; the reference always passes one
        for key in ("x", "Bx", "Ax"):
            if key in kwargs and kwargs[key] is not None:
                t = kwargs[key]
                t = t[1] if isinstance(t, (tuple, list)) else t
                n_nodes = t.size(-2)
                break
        msg = self.message(**self._collect(self._params(self.message), edge_index, n_nodes, kwargs))
        agg = self.aggregate(msg, **self._collect(self._params(self.aggregate, True), edge_index,
                                                  n_nodes, kwargs))
        return self.update(agg, **self._collect(self._params(self.update, True), edge_index,
                                                n_nodes, kwargs))

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, dim_size=None):
        out = torch.zeros((dim_size,) + tuple(inputs.shape[1:]), dtype=inputs.dtype,
                          device=inputs.device)
        return out.index_add_(0, index, inputs)

    def update(self, inputs):
        return inputs


class GINEConv(MessagePassing):
    """out_i = nn((1 + eps) * x_i + sum_{j->i} relu(x_j + e_ji)); eps buffer, not trained."""

    def __init__(self, nn, eps=0.0, train_eps=False, edge_dim=None, **kwargs):
        super().__init__(aggr="add", **kwargs)
        assert not train_eps and edge_dim is None
        self.nn = nn
        self.register_buffer("eps", torch.Tensor([eps]))

    def forward(self, x, edge_index, edge_attr=None, size=None):
        if torch.is_tensor(x):
            x = (x, x)
        out = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size)
        if x[1] is not None:
            out = out + (1 + self.eps) * x[1]
        return self.nn(out)

    def message(self, x_j, edge_attr):
        return (x_j + edge_attr).relu()


def _unsupported(name):
    class _U(torch.nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is outside the stubbed surface")
    _U.__name__ = name
    return _U


class GINConv(torch.nn.Module):
    """PyG 2.2 GINConv, published behaviour: out_i = nn((1 + eps) * x_i + sum_{j->i} x_j) with the node axis at
    dim -2 (so [k, N, C] inputs aggregate over N); ``eps`` is a buffer unless train_eps."""

    def __init__(self, nn, eps=0.0, train_eps=False, **kwargs):
        super().__init__()
        assert not train_eps
        self.nn = nn
        self.register_buffer("eps", torch.Tensor([eps]))

    def forward(self, x, edge_index, size=None):
        agg = torch.zeros_like(x).index_add_(-2, edge_index[1], x.index_select(-2, edge_index[0]))
        return self.nn(agg + (1 + self.eps) * x)


class GCNConv(torch.nn.Module):
    """PyG 2.2 GCNConv with its defaults, published behaviour: ``lin`` (glorot weight, no bias), gcn_norm =
    input self loops dropped and ONE unit self loop per node added (add_remaining_self_loops), degree =
    scatter-add of ones over targets, weight(j->i) = deg_j^-1/2 deg_i^-1/2, sum aggregation, + ``bias``
    (zeros at init)."""

    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        assert not kwargs
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        a = (6.0 / (in_channels + out_channels)) ** 0.5
        torch.nn.init.uniform_(self.lin.weight, -a, a)
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_weight=None):
        assert edge_weight is None
        n = x.size(0)
        keep = edge_index[0] != edge_index[1]
        loops = torch.arange(n, dtype=edge_index.dtype, device=x.device)
        row = torch.cat([edge_index[0][keep], loops])
        col = torch.cat([edge_index[1][keep], loops])
        deg = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, col, torch.ones_like(col, dtype=x.dtype))
        dinv = deg.pow(-0.5)
        h = self.lin(x)
        msg = (dinv[row] * dinv[col]).unsqueeze(-1) * h.index_select(0, row)
        return torch.zeros_like(h).index_add_(0, col, msg) + self.bias


GENConv, GATConv, PNAConv = (_unsupported(n) for n in ("GENConv", "GATConv", "PNAConv"))
