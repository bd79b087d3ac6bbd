"""Stub of torch_geometric.utils.to_dense_batch (PyG 2.2 published behaviour: zero padding to
[B, Nmax, ...], Nmax = max graph size, bool mask of real rows)."""
import torch


def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    if batch is None:
        return x.unsqueeze(0), torch.ones(1, x.size(0), dtype=torch.bool, device=x.device)
    if batch_size is None:
        batch_size = int(batch.max()) + 1
    num_nodes = torch.zeros(batch_size, dtype=torch.long, device=x.device).index_add_(
        0, batch, torch.ones_like(batch))
    cum_nodes = torch.cat([batch.new_zeros(1), num_nodes.cumsum(dim=0)])
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    idx = torch.arange(batch.size(0), dtype=torch.long, device=x.device)
    idx = (idx - cum_nodes[batch]) + (batch * max_num_nodes)
    size = [batch_size * max_num_nodes] + list(x.size())[1:]
    out = x.new_full(size, fill_value)
    out[idx] = x
    out = out.view([batch_size, max_num_nodes] + list(x.size())[1:])
    mask = torch.zeros(batch_size * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[idx] = 1
    return out, mask.view(batch_size, max_num_nodes)
