"""``custom_gnn`` behind ``register_network('custom_gnn')``: stacks of GatedGCN / GINE layers without the
global-attention half -- the ``configs/GatedGCN/*.yaml`` and ``configs/GINE/*.yaml`` baselines.

Drop-in for ``/root/reference/graphgps/network/custom_gnn.py:12-55``: ``(dim_in, dim_out)`` resolved from
``cfg.gnn.*``, children ``encoder`` / ``pre_mp`` / ``gnn_layers`` / ``post_mp`` (network/base.py), same
``build_conv_model`` strings and error text.  The layers are the HIP-backed ``GatedGCNLayer`` / ``GINEConvLayer``
(other callers of the same sparse kernels)."""
from ..graphgym.config import cfg
from ..graphgym.register import register_network
from ..head import heads as _heads  # noqa: F401
from ..layer.gatedgcn_layer import GatedGCNLayer
from ..layer.gine_conv_layer import GINEConvLayer
from .base import GraphGymNetwork

_CONV = {'gatedgcnconv': GatedGCNLayer, 'gineconv': GINEConvLayer}


@register_network('custom_gnn', overwrite=True)
class CustomGNN(GraphGymNetwork):
    """GraphGym's GNN with this package's conv layers in the message-passing stage."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        width = self._front(dim_in)
        assert cfg.gnn.dim_inner == width, "The inner and hidden dims must match."
        conv = self.build_conv_model(cfg.gnn.layer_type)
        self._stack('gnn_layers',
                    lambda: conv(width, width, dropout=cfg.gnn.dropout, residual=cfg.gnn.residual),
                    cfg.gnn.layers_mp)
        self._head(dim_out)

    def build_conv_model(self, model_type):
        if model_type not in _CONV:
            raise ValueError("Model {} unavailable".format(model_type))
        return _CONV[model_type]
