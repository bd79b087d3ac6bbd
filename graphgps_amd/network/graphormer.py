"""``GraphormerModel`` behind ``register_network('Graphormer')``.

Drop-in for ``/root/reference/graphgps/network/graphormer.py:10-52``: ``(dim_in, dim_out)`` resolved from the
global ``cfg`` (``cfg.graphormer.*``, :35-43), children ``encoder`` / ``pre_mp`` / ``layers`` / ``post_mp``
applied in order (:49-52)."""
import torch

from ..graphgym import register
from ..graphgym.config import cfg
from ..graphgym.layers import GNNPreMP
from ..graphgym.register import register_network
from ..head import graphormer_graph as _h  # noqa: F401
from ..layer.graphormer_layer import GraphormerLayer
from .gps_model import FeatureEncoder


@register_network('Graphormer', overwrite=True)
class GraphormerModel(torch.nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.encoder = FeatureEncoder(dim_in)
        dim_in = self.encoder.dim_in

        if cfg.gnn.layers_pre_mp > 0:
            self.pre_mp = GNNPreMP(dim_in, cfg.gnn.dim_inner, cfg.gnn.layers_pre_mp, cfg)
            dim_in = cfg.gnn.dim_inner

        if not cfg.graphormer.embed_dim == cfg.gnn.dim_inner == dim_in:
            raise ValueError(
                f"The inner and embed dims must match: "
                f"embed_dim={cfg.graphormer.embed_dim} "
                f"dim_inner={cfg.gnn.dim_inner} dim_in={dim_in}")

        layers = []
        for _ in range(cfg.graphormer.num_layers):
            layers.append(GraphormerLayer(
                embed_dim=cfg.graphormer.embed_dim,
                num_heads=cfg.graphormer.num_heads,
                dropout=cfg.graphormer.dropout,
                attention_dropout=cfg.graphormer.attention_dropout,
                mlp_dropout=cfg.graphormer.mlp_dropout))
        self.layers = torch.nn.Sequential(*layers)

        GNNHead = register.head_dict[cfg.gnn.head]
        self.post_mp = GNNHead(dim_in=cfg.gnn.dim_inner, dim_out=dim_out)

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch
