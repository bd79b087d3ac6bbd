"""The slice of ``torch_geometric.graphgym.models.layer`` (PyG 2.2, third-party to the reference)
that the callers around the GPS hot path construct: ``LayerConfig`` / ``new_layer_config``,
``Linear``, ``GeneralLayer``, ``GeneralMultiLayer``, ``MLP`` and ``GNNPreMP``.

Used by ``custom_gnn`` (graphgps/network/custom_gnn.py:23-26 ``GNNPreMP``), by GraphGym's default
graph head (``GNNGraphHead``: pooling + ``MLP``) that every ``configs/GatedGCN|GINE/*.yaml`` with
``dataset.task: graph`` resolves to, and by ``GPSModel`` when ``gnn.layers_pre_mp > 0``
(graphgps/network/gps_model.py:67-70).  Module/parameter names follow GraphGym
(``Layer_<i>.layer.model.weight``, ``Layer_<i>.post_layer.<k>``) so checkpoints interchange.
Restated from the published PyG 2.2 source -- the package is absent here, so this is unpinned
third-party behaviour."""
import copy
from dataclasses import dataclass, replace
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import register
from . import act as _act  # noqa: F401


@dataclass
class LayerConfig:
    has_batchnorm: bool = False
    bn_eps: float = 1e-5
    bn_mom: float = 0.1
    mem_inplace: bool = False
    dim_in: int = -1
    dim_out: int = -1
    edge_dim: int = -1
    dim_inner: Optional[int] = None
    num_layers: int = 2
    has_bias: bool = True
    has_l2norm: bool = True
    dropout: float = 0.0
    has_act: bool = True
    final_act: bool = True
    act: str = 'relu'
    keep_edge: float = 0.5


def new_layer_config(dim_in, dim_out, num_layers, has_act, has_bias, cfg):
    return LayerConfig(
        has_batchnorm=cfg.gnn.batchnorm, bn_eps=cfg.bn.eps, bn_mom=cfg.bn.mom,
        mem_inplace=cfg.mem.inplace, dim_in=dim_in, dim_out=dim_out, edge_dim=cfg.dataset.edge_dim,
        has_l2norm=cfg.gnn.l2norm, dropout=cfg.gnn.dropout, has_act=has_act, final_act=True,
        act=cfg.gnn.act, has_bias=has_bias, keep_edge=cfg.gnn.keep_edge, dim_inner=cfg.gnn.dim_inner,
        num_layers=num_layers)


class Linear(nn.Module):
    """GraphGym ``Linear`` layer: ``self.model = Linear_pyg(dim_in, dim_out, bias=has_bias)``."""

    def __init__(self, layer_config: LayerConfig, **kwargs):
        super().__init__()
        self.model = nn.Linear(layer_config.dim_in, layer_config.dim_out, bias=layer_config.has_bias)

    def forward(self, batch):
        if isinstance(batch, torch.Tensor):
            return self.model(batch)
        batch.x = self.model(batch.x)
        return batch


if 'linear' not in register.layer_dict:      # real PyG registers its own
    register.register_layer('linear', Linear)


class GeneralLayer(nn.Module):
    """layer -> [BatchNorm1d] -> [Dropout] -> [act] (-> l2norm)."""

    def __init__(self, name, layer_config: LayerConfig, **kwargs):
        super().__init__()
        self.has_l2norm = layer_config.has_l2norm
        has_bn = layer_config.has_batchnorm
        layer_config.has_bias = not has_bn
        self.layer = register.layer_dict[name](layer_config, **kwargs)
        wrapper = []
        if has_bn:
            wrapper.append(nn.BatchNorm1d(layer_config.dim_out, eps=layer_config.bn_eps,
                                          momentum=layer_config.bn_mom))
        if layer_config.dropout > 0:
            wrapper.append(nn.Dropout(p=layer_config.dropout, inplace=layer_config.mem_inplace))
        if layer_config.has_act:
            wrapper.append(register.act_dict[layer_config.act]())
        self.post_layer = nn.Sequential(*wrapper)

    def forward(self, batch):
        batch = self.layer(batch)
        if isinstance(batch, torch.Tensor):
            batch = self.post_layer(batch)
            if self.has_l2norm:
                batch = F.normalize(batch, p=2, dim=1)
        else:
            batch.x = self.post_layer(batch.x)
            if self.has_l2norm:
                batch.x = F.normalize(batch.x, p=2, dim=1)
        return batch


class GeneralMultiLayer(nn.Module):
    def __init__(self, name, layer_config: LayerConfig, **kwargs):
        super().__init__()
        dim_inner = layer_config.dim_out if layer_config.dim_inner is None else layer_config.dim_inner
        for i in range(layer_config.num_layers):
            d_in = layer_config.dim_in if i == 0 else dim_inner
            d_out = layer_config.dim_out if i == layer_config.num_layers - 1 else dim_inner
            has_act = layer_config.final_act if i == layer_config.num_layers - 1 else True
            inter = copy.deepcopy(layer_config)
            inter.dim_in, inter.dim_out, inter.has_act = d_in, d_out, has_act
            self.add_module(f'Layer_{i}', GeneralLayer(name, inter, **kwargs))

    def forward(self, batch):
        for layer in self.children():
            batch = layer(batch)
        return batch


class MLP(nn.Module):
    """(num_layers - 1) hidden GeneralLayer('linear') + a final plain Linear with bias.

    As published in PyG 2.2 (``graphgym/models/layer.py``, ``MLP.__init__``) the hidden stack's sub-config is
    ``LayerConfig(num_layers=num_layers - 1, dim_in=dim_in, dim_out=dim_inner, dim_inner=dim_inner,
    final_act=True)`` and NOTHING else: the hidden layers therefore run on LayerConfig's own defaults -- no
    BatchNorm (so ``Layer_<i>.layer.model.bias`` exists), dropout 0, ReLU, ``has_l2norm=True`` -- whatever
    ``cfg.gnn.batchnorm / dropout / act / l2norm`` say.  (Not pinnable here: PyG is absent; restated from the
    published source)"""

    def __init__(self, layer_config: LayerConfig, **kwargs):
        super().__init__()
        dim_inner = layer_config.dim_in if layer_config.dim_inner is None else layer_config.dim_inner
        layer_config.has_bias = True
        layers = []
        if layer_config.num_layers > 1:
            sub = LayerConfig(num_layers=layer_config.num_layers - 1, dim_in=layer_config.dim_in,
                              dim_out=dim_inner, dim_inner=dim_inner, final_act=True)
            layers.append(GeneralMultiLayer('linear', sub))
            layer_config = replace(layer_config, dim_in=dim_inner)
        layers.append(Linear(layer_config))
        self.model = nn.Sequential(*layers)

    def forward(self, batch):
        if isinstance(batch, torch.Tensor):
            return self.model(batch)
        batch.x = self.model(batch.x)
        return batch


def GNNPreMP(dim_in, dim_out, num_layers, cfg):
    """``GeneralMultiLayer('linear', new_layer_config(dim_in, dim_out, num_layers, has_act=False,
    has_bias=False, cfg))`` (GraphGym ``GNNPreMP``)."""
    return GeneralMultiLayer('linear', new_layer_config(dim_in, dim_out, num_layers, has_act=False,
                                                        has_bias=False, cfg=cfg))
