"""GraphGym-compatible plugin registry.

Mirrors the public surface of ``torch_geometric.graphgym.register`` (PyG 2.2,
third-party to the reference) that GraphGPS plugs into: the ``*_dict``
registries and the ``register_*`` decorators used at e.g.
``/root/reference/graphgps/network/gps_model.py:54`` (``register_network``),
``graphgps/layer/gatedgcn_layer.py:139`` (``register_layer``),
``graphgps/config/gt_config.py:5`` (``register_config``).

If the real ``torch_geometric`` is importable the dictionaries below ARE its
dictionaries, so the reference's own ``main.py`` resolves
``network_dict['GPSModel']`` to the MI355X-native implementation without any
change to the reference tree.
"""
from typing import Any, Callable, Dict, Optional, Union

_NAMES = (
    "act", "node_encoder", "edge_encoder", "stage", "head", "layer", "pooling",
    "network", "config", "dataset", "loader", "optimizer", "scheduler", "loss",
    "train", "metric",
)

try:  # pragma: no cover - exercised only where PyG is installed
    import torch_geometric as _pyg  # type: ignore
    if str(getattr(_pyg, "__version__", "")).endswith("-stub"):
        raise ImportError("oracle/ref_stubs stand-in, not the real PyG")
    import torch_geometric.graphgym.register as _pyg_register  # type: ignore
except Exception:  # ModuleNotFoundError in this image
    _pyg_register = None

for _n in _NAMES:
    if _pyg_register is not None and hasattr(_pyg_register, f"{_n}_dict"):
        globals()[f"{_n}_dict"] = getattr(_pyg_register, f"{_n}_dict")
    else:
        globals()[f"{_n}_dict"] = {}


def register_base(mapping: Dict[str, Any], key: str, module: Any = None,
                  overwrite: bool = False) -> Union[None, Callable]:
    """Register ``module`` under ``key``; decorator form when ``module`` is None.

    Same error behaviour as GraphGym: a duplicate key raises ``KeyError``
    (``overwrite=True`` is what lets this package shadow a PyG-registered
    reference implementation with the HIP-backed one).
    """
    if module is not None:
        if key in mapping and not overwrite:
            raise KeyError(f"Module with '{key}' already defined")
        mapping[key] = module
        return None

    def wrapper(mod):
        register_base(mapping, key, mod, overwrite=overwrite)
        return mod

    return wrapper


def _make(name: str):
    def reg(key: str, module: Any = None, overwrite: bool = False):
        return register_base(globals()[f"{name}_dict"], key, module, overwrite)
    reg.__name__ = f"register_{name}"
    reg.__doc__ = f"Register into ``{name}_dict`` (GraphGym ``register_{name}``)."
    return reg


for _n in _NAMES:
    globals()[f"register_{_n}"] = _make(_n)

__all__ = [f"{n}_dict" for n in _NAMES] + [f"register_{n}" for n in _NAMES] + ["register_base"]
