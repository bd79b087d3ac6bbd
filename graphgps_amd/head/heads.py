"""Prediction heads (plain torch, after the layers), all on two small bases.

Graph-level heads pool ``batch.x`` per graph -- through ``register.pooling_dict[cfg.model.graph_pooling]``, which
is the ptr-segmented HIP reduction when the batch carries a graph index (graphgym/pooling.py) -- and differ only
in what they put on top of the pooled embedding; node-level heads run GraphGym's post-MP ``MLP`` over ``batch.x``
and differ only in which rows they return.  Parameter names are the reference's (checkpoint contract):

  ====================  =======================================================  =============================
  ``head_dict`` key      reference                                                parameters
  ====================  =======================================================  =============================
  ``san_graph``          graphgps/head/san_graph.py:8-42                          ``FC_layers.<i>.*``
  ``ogb_code_graph``     graphgps/head/ogb_code_graph.py:8-45                     ``graph_pred_linear_list.<i>.*``
  ``graphormer_graph``   graphgps/head/graphormer_graph.py:9-37                   ``ln.*``, ``layers.0.*``
  ``graph``              GraphGym ``GNNGraphHead`` (PyG 2.2, third-party)         ``layer_post_mp.*``
  ``inductive_node``     graphgps/head/inductive_node.py:8-29                     ``layer_post_mp.*``
  ``node``               GraphGym ``GNNNodeHead`` (PyG 2.2, third-party)          ``layer_post_mp.*``
  ====================  =======================================================  =============================

``pooling_dict['graph_token']`` (graphgps/pooling/graph_token.py:5-12) lives here too: the reference extracts
the token with ``to_dense_batch(x)[:, 0]``; the token is the FIRST row of every graph, so it is one gather at
``ptr[:-1]``."""
import torch
import torch.nn as nn

from ..graphgym import pooling as _pooling  # noqa: F401  (registers add / mean / max)
from ..graphgym import register
from ..graphgym.config import cfg
from ..graphgym.layers import MLP, new_layer_config
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


def _post_mp(dim_in, dim_out):
    return MLP(new_layer_config(dim_in, dim_out, cfg.gnn.layers_post_mp, has_act=False, has_bias=True, cfg=cfg))


class _GraphLevelHead(nn.Module):
    """pool -> ``self.predict(graph_emb)`` -> ``batch.graph_feature``; returns (prediction, ``batch.y``)."""

    def __init__(self):
        super().__init__()
        self.pooling_fun = register.pooling_dict[cfg.model.graph_pooling]

    def pool(self, batch, x=None):
        x = batch.x if x is None else x
        gi = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
        if gi is not None and gi.N != x.shape[0]:
            gi = None
        try:
            return self.pooling_fun(x, batch.batch, batch.num_graphs, gi=gi)
        except TypeError:       # a user-registered pooling function with the plain GraphGym signature
            return self.pooling_fun(x, batch.batch)

    def _apply_index(self, batch):
        return batch.graph_feature, batch.y

    def forward(self, batch):
        batch.graph_feature = self.predict(self.pool(batch))
        return self._apply_index(batch)


@register_head('san_graph', overwrite=True)
class SANGraphHead(_GraphLevelHead):
    """L halving Linear+act stages and a final Linear."""

    def __init__(self, dim_in, dim_out, L=2):
        super().__init__()
        widths = [dim_in // 2 ** l for l in range(L + 1)]
        self.FC_layers = nn.ModuleList([nn.Linear(a, b, bias=True) for a, b in zip(widths[:-1], widths[1:])]
                                       + [nn.Linear(widths[-1], dim_out, bias=True)])
        self.L = L
        self.activation = register.act_dict[cfg.gnn.act]()

    def predict(self, emb):
        if emb.is_cuda and isinstance(self.activation, nn.ReLU):
            # one row per graph: the small-product kernel (csrc/small_gemm.hip), ReLU in its epilogue
            from ..fused import small_linear
            for fc in self.FC_layers[:-1]:
                emb = small_linear(emb, fc.weight, fc.bias, relu=True)
            fc = self.FC_layers[-1]
            return small_linear(emb, fc.weight, fc.bias)
        for fc in self.FC_layers[:-1]:
            emb = self.activation(fc(emb))
        return self.FC_layers[-1](emb)


@register_head('graphormer_graph', overwrite=True)
class GraphormerHead(_GraphLevelHead):
    """LayerNorm over the nodes, pool (the graph token), one Linear."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.ln = nn.LayerNorm(dim_in)
        self.layers = nn.Sequential(nn.Linear(dim_in, dim_out))

    def forward(self, batch):
        batch.graph_feature = self.layers(self.pool(batch, self.ln(batch.x)))
        return self._apply_index(batch)


class GNNGraphHead(_GraphLevelHead):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer_post_mp = _post_mp(dim_in, dim_out)

    def predict(self, emb):
        return self.layer_post_mp(emb)


@register_head('ogb_code_graph', overwrite=True)
class OGBCodeGraphHead(_GraphLevelHead):
    """ogbg-code2: 5 sub-token classifiers over the 5002-word vocabulary on the pooled embedding; returns
    (list of 5 logit tensors, {'y_arr', 'y'})."""

    def __init__(self, dim_in, dim_out, L=1):
        super().__init__()
        if L != 1:
            raise ValueError("Multilayer prediction heads are not supported.")
        self.L, self.max_seq_len = L, 5
        self.graph_pred_linear_list = nn.ModuleList(nn.Linear(dim_in, 5002) for _ in range(self.max_seq_len))

    def _apply_index(self, batch):
        return batch.pred_list, {'y_arr': batch.y_arr, 'y': batch.y}

    def forward(self, batch):
        emb = self.pool(batch)
        batch.pred_list = [lin(emb) for lin in self.graph_pred_linear_list]
        return self._apply_index(batch)


class _NodeLevelHead(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer_post_mp = _post_mp(dim_in, dim_out)

    def forward(self, batch):
        return self._apply_index(self.layer_post_mp(batch))


@register_head('inductive_node', overwrite=True)
class GNNInductiveNodeHead(_NodeLevelHead):
    """Every node of every graph is a labelled example."""

    def _apply_index(self, batch):
        return batch.x, batch.y


class GNNNodeHead(_NodeLevelHead):
    """Transductive: the rows selected by ``batch.<split>_mask``."""

    def _apply_index(self, batch):
        mask = getattr(batch, f'{batch.split}_mask')
        return batch.x[mask], batch.y[mask]


for _name, _cls in (('graph', GNNGraphHead), ('node', GNNNodeHead)):
    if _name not in register.head_dict:      # real PyG registers its own
        register_head(_name, _cls)
