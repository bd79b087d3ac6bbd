"""Activation registry defaults: what PyG 2.2's ``graphgym/models/act.py`` (third-party) and
``/root/reference/graphgps/act/example.py:22-26`` put into ``register.act_dict``.  Entries are
zero-argument factories, as the reference calls ``register.act_dict[act]()``
(graphgps/layer/gps_layer.py:33,63-65)."""
from functools import partial

import torch
import torch.nn as nn

from . import register
from .config import cfg


class SWISH(nn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return x.mul_(torch.sigmoid(x)) if self.inplace else x * torch.sigmoid(x)


_DEFAULTS = {
    "relu": lambda: nn.ReLU(inplace=cfg.mem.inplace),
    "selu": lambda: nn.SELU(inplace=cfg.mem.inplace),
    "prelu": nn.PReLU,
    "elu": lambda: nn.ELU(inplace=cfg.mem.inplace),
    "lrelu_01": lambda: nn.LeakyReLU(0.1, inplace=cfg.mem.inplace),
    "lrelu_025": lambda: nn.LeakyReLU(0.25, inplace=cfg.mem.inplace),
    "lrelu_05": lambda: nn.LeakyReLU(0.5, inplace=cfg.mem.inplace),
    "swish": partial(SWISH, inplace=False),
    "lrelu_03": partial(nn.LeakyReLU, 0.3),
    "gelu": nn.GELU,
}
for _k, _v in _DEFAULTS.items():
    register.act_dict.setdefault(_k, _v)
