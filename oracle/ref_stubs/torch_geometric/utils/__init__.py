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


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    """PyG 2.2 published behaviour: [B, Nmax, Nmax, *attr dims] with entry (b, i - first(b), j - first(b))
    = sum of edge_attr over the edges (i, j) of graph b (ones when edge_attr is None), zeros elsewhere."""
    if batch is None:
        batch = edge_index.new_zeros(int(edge_index.max()) + 1 if edge_index.numel() else 0)
    batch_size = int(batch.max()) + 1 if batch.numel() else 1
    num_nodes = torch.zeros(batch_size, dtype=torch.long, device=batch.device).index_add_(
        0, batch, torch.ones_like(batch))
    cum_nodes = torch.cat([batch.new_zeros(1), num_nodes.cumsum(dim=0)])
    idx0 = batch[edge_index[0]]
    idx1 = edge_index[0] - cum_nodes[batch][edge_index[0]]
    idx2 = edge_index[1] - cum_nodes[batch][edge_index[1]]
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    if edge_attr is None:
        edge_attr = torch.ones(idx0.numel(), device=edge_index.device)
    size = [batch_size, max_num_nodes, max_num_nodes] + list(edge_attr.size())[1:]
    flat = idx0 * max_num_nodes * max_num_nodes + idx1 * max_num_nodes + idx2
    adj = edge_attr.new_zeros([batch_size * max_num_nodes * max_num_nodes] + list(edge_attr.size())[1:])
    adj = adj.index_add(0, flat, edge_attr)
    return adj.view(size)


def to_networkx(data, *args, **kwargs):
    """PyG 2.2 published behaviour (defaults): a networkx.DiGraph with nodes 0..num_nodes-1 added in order
    and one edge per column of ``edge_index``, added in column order."""
    import networkx as nx
    G = nx.DiGraph()
    n = int(data.num_nodes) if getattr(data, "num_nodes", None) is not None else int(data.x.shape[0])
    G.add_nodes_from(range(n))
    for u, v in zip(data.edge_index[0].tolist(), data.edge_index[1].tolist()):
        G.add_edge(u, v)
    return G


def remove_self_loops(edge_index, edge_attr=None):
    """PyG published behaviour: drop the columns with source == target."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    return torch.zeros(n, dtype=dtype or torch.float).index_add_(0, index, torch.ones(index.numel(), dtype=dtype or torch.float))


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """PyG 2.3 ``utils.scatter`` (dim 0): sum / add / mul."""
    from torch_scatter import scatter as _scatter
    return _scatter(src, index, dim=dim, dim_size=dim_size, reduce=reduce)
