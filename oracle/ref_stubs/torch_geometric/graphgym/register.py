"""Stub of torch_geometric.graphgym.register: re-exports the in-repo registry and fills
``act_dict`` the way PyG 2.2's graphgym/models/act.py does (zero-arg factories)."""
import torch.nn as nn
from graphgps_amd.graphgym.register import *  # noqa: F401,F403
from graphgps_amd.graphgym.register import act_dict

for _k, _v in {"relu": nn.ReLU, "selu": nn.SELU, "prelu": nn.PReLU, "elu": nn.ELU,
               "lrelu_01": (lambda: nn.LeakyReLU(0.1)), "lrelu_025": (lambda: nn.LeakyReLU(0.25)),
               "lrelu_05": (lambda: nn.LeakyReLU(0.5)), "gelu": nn.GELU}.items():
    act_dict.setdefault(_k, _v)
