"""``SANTransformer`` behind ``register_network('SANTransformer')`` (https://arxiv.org/abs/2106.03893).

Drop-in for ``/root/reference/graphgps/network/san_transformer.py:11-56``: children ``encoder`` / ``pre_mp`` /
``trf_layers`` / ``post_mp`` (network/base.py); one ``Embedding(1, dim_hidden)`` for the fake edges shared by all
layers (registered through each layer's ``attention.fake_edge_emb``, as in the reference).  The layers are
torch-level (layer/san_layers.py), not HIP kernels."""
import torch

from ..graphgym.config import cfg
from ..graphgym.register import register_network
from ..head import heads as _heads  # noqa: F401
from ..layer.san_layers import SAN2Layer, SANLayer
from .base import GraphGymNetwork


@register_network('SANTransformer', overwrite=True)
class SANTransformer(GraphGymNetwork):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        width = self._front(dim_in)
        assert cfg.gt.dim_hidden == cfg.gnn.dim_inner == width, "The inner and hidden dims must match."
        fake_edge_emb = torch.nn.Embedding(1, cfg.gt.dim_hidden)
        layer_cls = {'SANLayer': SANLayer, 'SAN2Layer': SAN2Layer}.get(cfg.gt.layer_type)
        gt = cfg.gt
        self._stack('trf_layers',
                    lambda: layer_cls(gamma=gt.gamma, in_dim=gt.dim_hidden, out_dim=gt.dim_hidden,
                                      num_heads=gt.n_heads, full_graph=gt.full_graph,
                                      fake_edge_emb=fake_edge_emb, dropout=gt.dropout,
                                      layer_norm=gt.layer_norm, batch_norm=gt.batch_norm,
                                      residual=gt.residual),
                    gt.layers)
        self._head(dim_out)
