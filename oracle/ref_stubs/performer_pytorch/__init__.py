"""Stub of the pip package ``performer_pytorch``: the reference vendors the same upstream file
at graphgps/layer/performer_layer.py, so ``SelfAttention`` is loaded from THAT file, by path."""
import importlib.util
import os

_ref = os.environ.get("GPS_REFERENCE_ROOT", "/root/reference")
_spec = importlib.util.spec_from_file_location(
    "_ref_performer_layer", os.path.join(_ref, "graphgps", "layer", "performer_layer.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
SelfAttention = _mod.SelfAttention
FastAttention = _mod.FastAttention
