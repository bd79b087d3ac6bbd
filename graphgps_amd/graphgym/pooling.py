"""``register.pooling_dict`` defaults: GraphGym's 'add' / 'mean' / 'max' (PyG global_*_pool,
third-party), resolved by the heads via ``cfg.model.graph_pooling``
(``/root/reference/graphgps/head/san_graph.py:21``).  On the GPU 'add'/'mean' run the
deterministic ptr-segmented HIP reduction; signature stays ``fn(x, batch_vec, size=None)`` with
an optional ``gi=`` graph index (heads in this package pass it)."""
import torch

from . import register


def _pool(mode):
    def fn(x, batch, size=None, gi=None):
        if gi is not None and x.is_cuda and mode in ("add", "mean"):
            from ..ops import segment_pool
            return segment_pool(x, gi, mode)
        size = int(batch.max().item()) + 1 if size is None else size
        out = x.new_zeros(size, x.shape[1])
        if mode == "max":
            out = out.fill_(float("-inf")).scatter_reduce(0, batch[:, None].expand_as(x), x, "amax")
            return torch.where(torch.isinf(out), torch.zeros_like(out), out)
        out = out.index_add_(0, batch, x)
        if mode == "mean":
            cnt = torch.bincount(batch, minlength=size).clamp(min=1).to(x.dtype)
            out = out / cnt[:, None]
        return out
    fn.__name__ = f"global_{mode}_pool"
    return fn


for _m in ("add", "mean", "max"):
    register.pooling_dict.setdefault(_m, _pool(_m))
