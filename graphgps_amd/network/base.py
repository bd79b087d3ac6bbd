"""Shared skeleton of the GraphGym networks this package registers.

Every network of the reference has the same shape -- ``FeatureEncoder`` -> optional ``GNNPreMP`` -> a stack of
layers -> a head from ``register.head_dict`` -- and the same ``forward``: the children applied in registration
order (graphgps/network/gps_model.py:62-108, custom_gnn.py:17-55, graphormer.py:18-52).  ``GraphGymNetwork``
holds that skeleton once; the subclasses only say which layers go in the middle and under which attribute name
(the child names ``encoder`` / ``pre_mp`` / ``layers`` | ``gnn_layers`` / ``post_mp`` are the checkpoint contract,
graphgps/finetuning.py:124-138)."""
import torch

from ..encoder import encoders as _encoders  # noqa: F401  (fills the encoder registries)
from ..encoder.encoders import BatchNorm1dNode
from ..graphgym import register
from ..graphgym.config import cfg
from ..graphgym.layers import GNNPreMP


class FeatureEncoder(torch.nn.Module):
    """Node / edge feature encoders picked by ``cfg.dataset.*`` (reference gps_model.py:12-51): children
    ``node_encoder`` [``node_encoder_bn``] ``edge_encoder`` [``edge_encoder_bn``], applied in that order."""

    def __init__(self, dim_in):
        super().__init__()
        self.dim_in = dim_in
        ds = cfg.dataset
        if ds.node_encoder:
            self.node_encoder = register.node_encoder_dict[ds.node_encoder_name](cfg.gnn.dim_inner)
            if ds.node_encoder_bn:
                self.node_encoder_bn = BatchNorm1dNode(cfg.gnn.dim_inner, cfg.bn.eps, cfg.bn.mom)
            self.dim_in = cfg.gnn.dim_inner
        if ds.edge_encoder:
            # PNA caps its edge width at 128 (reference :36-39); every other local model uses dim_inner
            cfg.gnn.dim_edge = min(128, cfg.gnn.dim_inner) if 'PNA' in cfg.gt.layer_type else cfg.gnn.dim_inner
            self.edge_encoder = register.edge_encoder_dict[ds.edge_encoder_name](cfg.gnn.dim_edge)
            if ds.edge_encoder_bn:
                # (the reference applies this BatchNorm to batch.x as well -- kept)
                self.edge_encoder_bn = BatchNorm1dNode(cfg.gnn.dim_edge, cfg.bn.eps, cfg.bn.mom)

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch


class GraphGymNetwork(torch.nn.Module):
    def _front(self, dim_in):
        """Registers ``encoder`` (+ ``pre_mp`` when ``cfg.gnn.layers_pre_mp > 0``); returns the width that
        reaches the layer stack."""
        self.encoder = FeatureEncoder(dim_in)
        width = self.encoder.dim_in
        if cfg.gnn.layers_pre_mp > 0:
            self.pre_mp = GNNPreMP(width, cfg.gnn.dim_inner, cfg.gnn.layers_pre_mp, cfg)
            width = cfg.gnn.dim_inner
        return width

    def _stack(self, name, make_layer, count):
        setattr(self, name, torch.nn.Sequential(*[make_layer() for _ in range(count)]))

    def _head(self, dim_out):
        self.post_mp = register.head_dict[cfg.gnn.head](dim_in=cfg.gnn.dim_inner, dim_out=dim_out)

    def forward(self, batch):
        for module in self.children():
            if isinstance(module, torch.nn.Sequential) and len(module) and hasattr(module[0], "local_gnn_type"):
                batch = self._run_stack(module, batch)
            else:
                batch = module(batch)
        return batch

    @staticmethod
    def _run_stack(layers, batch):
        """The GPS layer stack, bracketed so that the fused blocks share their per-step set-up (one weight-image
        launch, one BatchNorm counter launch for all layers: layer/gps_block.py stack_begin / stack_end)."""
        from ..layer import gps_block as _blk
        if not _blk.stack_begin(layers, batch):
            return layers(batch)
        try:
            return layers(batch)
        finally:
            _blk.stack_end()
