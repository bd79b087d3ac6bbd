"""Inductive node-prediction head: an MLP over ``batch.x``.
Mirror of ``/root/reference/graphgps/head/inductive_node.py:8-34`` (``layer_post_mp`` =
GraphGym ``MLP(new_layer_config(dim_in, dim_out, cfg.gnn.layers_post_mp, has_act=False, has_bias=True))``)."""
import torch.nn as nn

from ..graphgym.config import cfg
from ..graphgym.layers import MLP, new_layer_config
from ..graphgym.register import register_head


@register_head('inductive_node', overwrite=True)
class GNNInductiveNodeHead(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer_post_mp = MLP(new_layer_config(dim_in, dim_out, cfg.gnn.layers_post_mp,
                                                  has_act=False, has_bias=True, cfg=cfg))

    def _apply_index(self, batch):
        return batch.x, batch.y

    def forward(self, batch):
        batch = self.layer_post_mp(batch)
        pred, label = self._apply_index(batch)
        return pred, label
