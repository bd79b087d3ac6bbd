"""``GPSModel`` behind ``register_network('GPSModel')``.

Drop-in for ``/root/reference/graphgps/network/gps_model.py:12-108``: constructor
``(dim_in, dim_out)`` resolved from the global ``cfg`` exactly as the reference does (:62-103),
``forward(batch) -> (pred, true)`` = sequential children (:105-108), child names ``encoder``,
``layers``, ``post_mp`` (the ``state_dict`` / fine-tuning contract, graphgps/finetuning.py:124-138).
"""
import torch

from ..encoder import encoders as _encoders  # noqa: F401  (fills the encoder registries)
from ..encoder.encoders import BatchNorm1dNode
from ..graphgym import register
from ..graphgym.config import cfg
from ..graphgym.layers import GNNPreMP
from ..graphgym.register import register_network
from ..head import inductive_node as _h0, ogb_code_graph as _h1, san_graph as _h2  # noqa: F401
from ..layer.gps_layer import GPSLayer


class FeatureEncoder(torch.nn.Module):
    """Encoding node and edge features (reference :12-51)."""

    def __init__(self, dim_in):
        super().__init__()
        self.dim_in = dim_in
        if cfg.dataset.node_encoder:
            NodeEncoder = register.node_encoder_dict[cfg.dataset.node_encoder_name]
            self.node_encoder = NodeEncoder(cfg.gnn.dim_inner)
            if cfg.dataset.node_encoder_bn:
                self.node_encoder_bn = BatchNorm1dNode(cfg.gnn.dim_inner, cfg.bn.eps, cfg.bn.mom)
            self.dim_in = cfg.gnn.dim_inner
        if cfg.dataset.edge_encoder:
            if 'PNA' in cfg.gt.layer_type:
                cfg.gnn.dim_edge = min(128, cfg.gnn.dim_inner)
            else:
                cfg.gnn.dim_edge = cfg.gnn.dim_inner
            EdgeEncoder = register.edge_encoder_dict[cfg.dataset.edge_encoder_name]
            self.edge_encoder = EdgeEncoder(cfg.gnn.dim_edge)
            if cfg.dataset.edge_encoder_bn:
                self.edge_encoder_bn = BatchNorm1dNode(cfg.gnn.dim_edge, cfg.bn.eps, cfg.bn.mom)

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch


@register_network('GPSModel', overwrite=True)
class GPSModel(torch.nn.Module):
    """General-Powerful-Scalable graph transformer, https://arxiv.org/abs/2205.12454"""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.encoder = FeatureEncoder(dim_in)
        dim_in = self.encoder.dim_in

        if cfg.gnn.layers_pre_mp > 0:                    # reference :67-70
            self.pre_mp = GNNPreMP(dim_in, cfg.gnn.dim_inner, cfg.gnn.layers_pre_mp, cfg)
            dim_in = cfg.gnn.dim_inner

        if not cfg.gt.dim_hidden == cfg.gnn.dim_inner == dim_in:
            raise ValueError(
                f"The inner and hidden dims must match: "
                f"embed_dim={cfg.gt.dim_hidden} dim_inner={cfg.gnn.dim_inner} "
                f"dim_in={dim_in}")

        try:
            local_gnn_type, global_model_type = cfg.gt.layer_type.split('+')
        except Exception:
            raise ValueError(f"Unexpected layer type: {cfg.gt.layer_type}")
        layers = []
        for _ in range(cfg.gt.layers):
            layers.append(GPSLayer(
                dim_h=cfg.gt.dim_hidden,
                local_gnn_type=local_gnn_type,
                global_model_type=global_model_type,
                num_heads=cfg.gt.n_heads,
                act=cfg.gnn.act,
                pna_degrees=cfg.gt.pna_degrees,
                equivstable_pe=cfg.posenc_EquivStableLapPE.enable,
                dropout=cfg.gt.dropout,
                attn_dropout=cfg.gt.attn_dropout,
                layer_norm=cfg.gt.layer_norm,
                batch_norm=cfg.gt.batch_norm,
                bigbird_cfg=cfg.gt.bigbird,
                log_attn_weights=cfg.train.mode == 'log-attn-weights',
            ))
        self.layers = torch.nn.Sequential(*layers)

        GNNHead = register.head_dict[cfg.gnn.head]
        self.post_mp = GNNHead(dim_in=cfg.gnn.dim_inner, dim_out=dim_out)

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch
