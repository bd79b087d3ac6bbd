"""``GraphormerModel`` behind ``register_network('Graphormer')``.

Drop-in for ``/root/reference/graphgps/network/graphormer.py:10-52``: ``(dim_in, dim_out)`` resolved from
``cfg.graphormer.*``; children ``encoder`` / ``pre_mp`` / ``layers`` / ``post_mp`` (network/base.py)."""
from ..graphgym.config import cfg
from ..graphgym.register import register_network
from ..head import heads as _heads  # noqa: F401
from ..layer.graphormer_layer import GraphormerLayer
from .base import GraphGymNetwork


@register_network('Graphormer', overwrite=True)
class GraphormerModel(GraphGymNetwork):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        width = self._front(dim_in)
        gcfg = cfg.graphormer
        if not gcfg.embed_dim == cfg.gnn.dim_inner == width:
            raise ValueError(
                f"The inner and embed dims must match: "
                f"embed_dim={gcfg.embed_dim} "
                f"dim_inner={cfg.gnn.dim_inner} dim_in={width}")
        self._stack('layers',
                    lambda: GraphormerLayer(embed_dim=gcfg.embed_dim, num_heads=gcfg.num_heads,
                                            dropout=gcfg.dropout, attention_dropout=gcfg.attention_dropout,
                                            mlp_dropout=gcfg.mlp_dropout),
                    gcfg.num_layers)
        self._head(dim_out)
