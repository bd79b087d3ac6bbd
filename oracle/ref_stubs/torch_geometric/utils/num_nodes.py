"""Stub of torch_geometric.utils.num_nodes.maybe_num_nodes (published behaviour)."""


def maybe_num_nodes(index, num_nodes=None):
    return int(index.max()) + 1 if num_nodes is None else num_nodes
