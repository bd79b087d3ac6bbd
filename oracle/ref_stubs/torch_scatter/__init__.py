"""Stub of torch_scatter.scatter (dim 0, reduce='sum'): published behaviour is
``zeros(dim_size, ...).scatter_add_(0, index, src)``."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None and reduce in ("sum", "add")
    if dim_size is None:
        dim_size = int(index.max()) + 1
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)
