"""SAN graph-prediction head: pool + (L+1)-layer halving MLP.
Mirror of ``/root/reference/graphgps/head/san_graph.py:8-42`` (parameter names
``FC_layers.<i>.{weight,bias}``).  Runs once per step after the layers; plain PyTorch except the
pooling, which uses the ptr-segmented HIP reduction when the batch carries a graph index."""
import torch.nn as nn

from ..graphgym import register
from ..graphgym import pooling as _pooling  # noqa: F401
from ..graphgym.config import cfg
from ..graphgym.register import register_head


@register_head('san_graph', overwrite=True)
class SANGraphHead(nn.Module):
    def __init__(self, dim_in, dim_out, L=2):
        super().__init__()
        self.pooling_fun = register.pooling_dict[cfg.model.graph_pooling]
        layers = [nn.Linear(dim_in // 2 ** l, dim_in // 2 ** (l + 1), bias=True) for l in range(L)]
        layers.append(nn.Linear(dim_in // 2 ** L, dim_out, bias=True))
        self.FC_layers = nn.ModuleList(layers)
        self.L = L
        self.activation = register.act_dict[cfg.gnn.act]()

    def _apply_index(self, batch):
        return batch.graph_feature, batch.y

    def forward(self, batch):
        gi = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
        try:
            graph_emb = self.pooling_fun(batch.x, batch.batch, batch.num_graphs, gi=gi)
        except TypeError:  # a user-registered pooling function with the plain GraphGym signature
            graph_emb = self.pooling_fun(batch.x, batch.batch)
        for l in range(self.L):
            graph_emb = self.activation(self.FC_layers[l](graph_emb))
        graph_emb = self.FC_layers[self.L](graph_emb)
        batch.graph_feature = graph_emb
        return self._apply_index(batch)
