"""GraphGym's default graph-level head (``head_dict['graph']``, ``GNNGraphHead``; PyG 2.2,
third-party): ``pooling_dict[cfg.model.graph_pooling]`` followed by
``MLP(new_layer_config(dim_in, dim_out, cfg.gnn.layers_post_mp, has_act=False, has_bias=True))``.
It is the head every ``configs/GatedGCN|GINE/*.yaml`` with ``dataset.task: graph`` resolves to
(``gnn.head`` 'default' -> ``dataset.task``).  Pooling uses the ptr-segmented HIP reduction when the
batch carries a graph index (same as head/san_graph.py)."""
import torch.nn as nn

from ..graphgym import register
from ..graphgym import pooling as _pooling  # noqa: F401
from ..graphgym.config import cfg
from ..graphgym.layers import MLP, new_layer_config
from ..graphgym.register import register_head


class GNNGraphHead(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer_post_mp = MLP(new_layer_config(dim_in, dim_out, cfg.gnn.layers_post_mp,
                                                  has_act=False, has_bias=True, cfg=cfg))
        self.pooling_fun = register.pooling_dict[cfg.model.graph_pooling]

    def _apply_index(self, batch):
        return batch.graph_feature, batch.y

    def forward(self, batch):
        gi = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
        try:
            graph_emb = self.pooling_fun(batch.x, batch.batch, batch.num_graphs, gi=gi)
        except TypeError:  # a user-registered pooling function with the plain GraphGym signature
            graph_emb = self.pooling_fun(batch.x, batch.batch)
        graph_emb = self.layer_post_mp(graph_emb)
        batch.graph_feature = graph_emb
        return self._apply_index(batch)


if 'graph' not in register.head_dict:        # real PyG registers its own
    register_head('graph', GNNGraphHead)
