"""Stub of torch_geometric (PyG 2.2) -- only what the reference's hot-path files import."""
__version__ = "2.2.0-stub"
