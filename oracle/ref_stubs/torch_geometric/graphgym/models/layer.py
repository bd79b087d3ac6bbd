"""Stub: only the ``LayerConfig`` record the register_layer wrappers take
(graphgps/layer/gatedgcn_layer.py:145-152)."""
from dataclasses import dataclass


@dataclass
class LayerConfig:
    has_batchnorm: bool = False
    bn_eps: float = 1e-5
    bn_mom: float = 0.1
    mem_inplace: bool = False
    dim_in: int = -1
    dim_out: int = -1
    edge_dim: int = -1
    dim_inner: int = None
    num_layers: int = 2
    has_bias: bool = True
    has_l2norm: bool = True
    dropout: float = 0.0
    has_act: bool = True
    final_act: bool = True
    act: str = "relu"
    keep_edge: float = 0.5


def __getattr__(name):
    # GraphGym's own ``MLP`` / ``new_layer_config`` (third-party to the reference): the in-repo restatement,
    # resolved lazily so that importing this stub does not pull the package in before the registry is set up
    if name in ("MLP", "new_layer_config"):
        import graphgps_amd.graphgym.layers as _l
        return getattr(_l, name)
    raise AttributeError(name)
