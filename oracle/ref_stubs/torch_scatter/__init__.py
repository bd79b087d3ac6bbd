"""Stub of torch_scatter (dim 0 only), published behaviour:
``scatter(src, index, dim=0, out=None, dim_size=None, reduce)``: 'sum'/'add' accumulate into zeros (or into ``out``,
in place), 'mul' multiplies into ones; ``scatter_add`` = scatter(..., 'sum'); ``scatter_max`` returns
(max per group with 0 for empty groups, argmax) -- only the values are used by the reference."""
import torch


def _expand(index, src):
    return index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and reduce in ("sum", "add", "mul")
    if dim_size is None:
        dim_size = out.shape[0] if out is not None else int(index.max()) + 1
    if reduce == "mul":
        assert out is None
        res = torch.ones((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        return res.scatter_reduce(0, _expand(index, src), src, "prod", include_self=True)
    if out is not None:
        return out.index_add_(0, index, src)
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim=dim, out=out, dim_size=dim_size, reduce="sum")


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    if dim_size is None:
        dim_size = int(index.max()) + 1
    res = torch.full((dim_size,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype, device=src.device)
    res = res.scatter_reduce(0, _expand(index, src), src, "amax", include_self=True)
    res = torch.where(torch.isinf(res), torch.zeros_like(res), res)
    return res, None
