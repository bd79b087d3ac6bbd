"""Graphormer graph-prediction head + the ``graph_token`` pooling it is used with.

Mirror of ``/root/reference/graphgps/head/graphormer_graph.py:8-39`` (``ln.*``, ``layers.0.*``) and
``graphgps/pooling/graph_token.py:5-12``.  The reference extracts the token with ``to_dense_batch(x)[:, 0]``;
the token is the FIRST row of every graph (``graphormer_encoder.py:add_graph_token`` sorts it there), so
here it is one gather at ``ptr[:-1]`` -- no padding."""
import torch

from ..graphgym import register
from ..graphgym import pooling as _pooling  # noqa: F401
from ..graphgym.config import cfg
from ..graphgym.register import register_head, register_pooling


@register_pooling('graph_token', overwrite=True)
def graph_token_pooling(x, batch, size=None, gi=None):
    if gi is not None:
        first = gi.ptr[:-1].long()
    else:   # batch is sorted: a graph's first row is where its id first appears
        size = int(batch.max().item()) + 1 if size is None else size
        counts = torch.bincount(batch, minlength=size)
        first = torch.cumsum(counts, 0) - counts
    return x[first]


@register_head('graphormer_graph', overwrite=True)
class GraphormerHead(torch.nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.pooling_fun = register.pooling_dict[cfg.model.graph_pooling]
        self.ln = torch.nn.LayerNorm(dim_in)
        self.layers = torch.nn.Sequential(torch.nn.Linear(dim_in, dim_out))

    def _apply_index(self, batch):
        return batch.graph_feature, batch.y

    def forward(self, batch):
        x = self.ln(batch.x)
        gi = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
        if gi is not None and gi.N != x.shape[0]:
            gi = None
        try:
            graph_emb = self.pooling_fun(x, batch.batch, batch.num_graphs, gi=gi)
        except TypeError:  # a user-registered pooling function with the plain GraphGym signature
            graph_emb = self.pooling_fun(x, batch.batch)
        batch.graph_feature = self.layers(graph_emb)
        return self._apply_index(batch)
