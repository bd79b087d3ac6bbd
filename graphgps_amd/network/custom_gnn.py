"""``custom_gnn`` behind ``register_network('custom_gnn')``: stacks of GatedGCN / GINE layers without
the global-attention half -- the ``configs/GatedGCN/*.yaml`` and ``configs/GINE/*.yaml`` baselines.

Drop-in for ``/root/reference/graphgps/network/custom_gnn.py:12-55``: same constructor
``(dim_in, dim_out)`` resolved from ``cfg``, same children names (``encoder``, ``pre_mp``,
``gnn_layers``, ``post_mp``), same ``build_conv_model`` strings and error text.  The layers are the
HIP-backed ``GatedGCNLayer`` / ``GINEConvLayer`` (other callers of the same sparse kernels,
SURVEY.md section 8f rank 3)."""
import torch

from ..graphgym import register
from ..graphgym.config import cfg
from ..graphgym.layers import GNNPreMP
from ..graphgym.register import register_network
from ..head import graph_head as _h0, ogb_code_graph as _h1, san_graph as _h2  # noqa: F401
from ..layer.gatedgcn_layer import GatedGCNLayer
from ..layer.gine_conv_layer import GINEConvLayer
from .gps_model import FeatureEncoder


@register_network('custom_gnn', overwrite=True)
class CustomGNN(torch.nn.Module):
    """GNN model that customizes GraphGym's GNN to support specific handling of new conv layers."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.encoder = FeatureEncoder(dim_in)
        dim_in = self.encoder.dim_in

        if cfg.gnn.layers_pre_mp > 0:
            self.pre_mp = GNNPreMP(dim_in, cfg.gnn.dim_inner, cfg.gnn.layers_pre_mp, cfg)
            dim_in = cfg.gnn.dim_inner

        assert cfg.gnn.dim_inner == dim_in, "The inner and hidden dims must match."

        conv_model = self.build_conv_model(cfg.gnn.layer_type)
        layers = []
        for _ in range(cfg.gnn.layers_mp):
            layers.append(conv_model(dim_in, dim_in, dropout=cfg.gnn.dropout,
                                     residual=cfg.gnn.residual))
        self.gnn_layers = torch.nn.Sequential(*layers)

        GNNHead = register.head_dict[cfg.gnn.head]
        self.post_mp = GNNHead(dim_in=cfg.gnn.dim_inner, dim_out=dim_out)

    def build_conv_model(self, model_type):
        if model_type == 'gatedgcnconv':
            return GatedGCNLayer
        elif model_type == 'gineconv':
            return GINEConvLayer
        else:
            raise ValueError("Model {} unavailable".format(model_type))

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch
