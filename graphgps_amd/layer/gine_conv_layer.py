"""GINE on the HIP gather-relu-segment-sum kernel.

``GINEConv`` stands in for PyG's ``torch_geometric.nn.GINEConv`` as the reference constructs it
at ``/root/reference/graphgps/layer/gps_layer.py:62-69`` (``nn`` = Linear-act-Linear, eps = 0
buffer, no edge_dim): state_dict keys ``nn.0.*``, ``nn.2.*``, ``eps``.  ``GINEConvLayer`` and the
``register_layer('gineconv')`` wrapper mirror ``graphgps/layer/gine_conv_layer.py:90-132``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graphgym.register import register_layer
from ..ops import build_graph_index, gine_aggregate, graph_index_of


class GINEConv(nn.Module):
    def __init__(self, nn_module: nn.Module, eps: float = 0., train_eps: bool = False,
                 edge_dim=None, **kwargs):
        super().__init__()
        if train_eps or edge_dim is not None:
            raise NotImplementedError("GINEConv(train_eps / edge_dim) is not used by GraphGPS")
        self.nn = nn_module
        self.initial_eps = eps
        self.register_buffer('eps', torch.Tensor([eps]))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # ``eps`` is a (non-trained) buffer: a checkpoint may carry a value other than the constructor's.  The
        # kernels take eps as a host scalar (no device sync per call), so mirror a loaded buffer into the float.
        v = state_dict.get(prefix + 'eps')
        if v is not None:
            self.initial_eps = float(v.reshape(-1)[0])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward_tensors(self, x, edge_attr, gi):
        # eps is a constant buffer (never trained): read the Python float, no device sync
        return self.nn(gine_aggregate(x, edge_attr, gi, self.initial_eps))

    def forward(self, x, edge_index, edge_attr, gi=None):
        """PyG call convention ``conv(x, edge_index, edge_attr)``.  Without a shared per-batch
        index (``gi``) one is built for this call (CSR/CSC only; one pseudo-graph)."""
        if gi is None:
            n = x.shape[0]
            gi = build_graph_index(edge_index, n, 1,
                                   ptr_vec=torch.tensor([0, n], device=x.device))
        return self.forward_tensors(x, edge_attr, gi)

    def __repr__(self):
        return f'{self.__class__.__name__}(nn={self.nn})'


class GINEConvESLapPE(GINEConv):
    """``GINEConvESLapPE`` (graphgps/layer/gine_conv_layer.py:11-87): GINE whose messages are scaled by
    r_ij = MLP(||PE_i - PE_j||^2) (a per-edge scalar in (0,1)); parameter names ``nn.*``, ``eps``,
    ``mlp_r_ij.{0,2}.*``."""

    def __init__(self, nn_module: nn.Module, eps: float = 0., train_eps: bool = False, edge_dim=None,
                 **kwargs):
        super().__init__(nn_module, eps, train_eps, edge_dim, **kwargs)
        out_dim = nn_module[0].out_features
        self.mlp_r_ij = nn.Sequential(nn.Linear(1, out_dim), nn.ReLU(), nn.Linear(out_dim, 1),
                                      nn.Sigmoid())

    def forward_tensors(self, x, edge_attr, gi, pe_LapPE=None):
        if pe_LapPE is None:
            raise ValueError("GINEConvESLapPE needs pe_LapPE (batch.pe_EquivStableLapPE)")
        r = ((pe_LapPE.index_select(0, gi.edge_dst) - pe_LapPE.index_select(0, gi.edge_src)) ** 2) \
            .sum(dim=-1, keepdim=True)
        r = self.mlp_r_ij(r).view(-1)
        return self.nn(gine_aggregate(x, edge_attr, gi, self.initial_eps, r))

    def forward(self, x, edge_index, edge_attr, pe_LapPE=None, gi=None):
        if gi is None:
            n = x.shape[0]
            gi = build_graph_index(edge_index, n, 1, ptr_vec=torch.tensor([0, n], device=x.device))
        return self.forward_tensors(x, edge_attr, gi, pe_LapPE)


class GINEConvLayer(nn.Module):
    """graphgps/layer/gine_conv_layer.py:90-116."""

    def __init__(self, dim_in, dim_out, dropout, residual):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.dropout, self.residual = dropout, residual
        gin_nn = nn.Sequential(nn.Linear(dim_in, dim_out), nn.ReLU(), nn.Linear(dim_out, dim_out))
        self.model = GINEConv(gin_nn)

    def forward(self, batch):
        x_in = batch.x
        x = self.model.forward_tensors(batch.x, batch.edge_attr, graph_index_of(batch))
        x = F.relu(x)
        x = F.dropout(x, p=self.dropout, training=self.training)
        if self.residual:
            x = x_in + x
        batch.x = x
        return batch


@register_layer('gineconv', overwrite=True)
class GINEConvGraphGymLayer(nn.Module):
    """graphgps/layer/gine_conv_layer.py:119-132."""

    def __init__(self, layer_config, **kwargs):
        super().__init__()
        gin_nn = nn.Sequential(nn.Linear(layer_config.dim_in, layer_config.dim_out), nn.ReLU(),
                               nn.Linear(layer_config.dim_out, layer_config.dim_out))
        self.model = GINEConv(gin_nn)

    def forward(self, batch):
        batch.x = self.model.forward_tensors(batch.x, batch.edge_attr, graph_index_of(batch))
        return batch
