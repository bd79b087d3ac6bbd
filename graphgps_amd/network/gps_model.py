"""``GPSModel`` behind ``register_network('GPSModel')``.

Drop-in for ``/root/reference/graphgps/network/gps_model.py:54-108``: constructor ``(dim_in, dim_out)``
resolved from the global ``cfg`` (``gt.*``, ``gnn.act``, ``posenc_EquivStableLapPE.enable``, ``train.mode``),
``forward(batch) -> (pred, true)``, child names ``encoder`` / ``pre_mp`` / ``layers`` / ``post_mp``
(network/base.py)."""
from ..graphgym.config import cfg
from ..graphgym.register import register_network
from ..head import heads as _heads  # noqa: F401
from ..layer.gps_layer import GPSLayer
from .base import FeatureEncoder, GraphGymNetwork  # noqa: F401  (FeatureEncoder re-exported)


@register_network('GPSModel', overwrite=True)
class GPSModel(GraphGymNetwork):
    """General-Powerful-Scalable graph transformer, https://arxiv.org/abs/2205.12454"""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        width = self._front(dim_in)
        if not cfg.gt.dim_hidden == cfg.gnn.dim_inner == width:
            raise ValueError(
                f"The inner and hidden dims must match: "
                f"embed_dim={cfg.gt.dim_hidden} dim_inner={cfg.gnn.dim_inner} "
                f"dim_in={width}")
        halves = cfg.gt.layer_type.split('+') if isinstance(cfg.gt.layer_type, str) else []
        if len(halves) != 2:
            raise ValueError(f"Unexpected layer type: {cfg.gt.layer_type}")
        layer_kwargs = dict(
            dim_h=cfg.gt.dim_hidden, local_gnn_type=halves[0], global_model_type=halves[1],
            num_heads=cfg.gt.n_heads, act=cfg.gnn.act, pna_degrees=cfg.gt.pna_degrees,
            equivstable_pe=cfg.posenc_EquivStableLapPE.enable, dropout=cfg.gt.dropout,
            attn_dropout=cfg.gt.attn_dropout, layer_norm=cfg.gt.layer_norm, batch_norm=cfg.gt.batch_norm,
            bigbird_cfg=cfg.gt.bigbird, log_attn_weights=cfg.train.mode == 'log-attn-weights')
        self._stack('layers', lambda: GPSLayer(**layer_kwargs), cfg.gt.layers)
        self._head(dim_out)
