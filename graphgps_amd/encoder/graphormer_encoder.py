"""Graphormer structural encodings: the producer of ``batch.attn_bias`` for the biased attention kernel.

Mirrors ``/root/reference/graphgps/encoder/graphormer_encoder.py`` with identical parameter names
(``spatial_encoder``, ``edge_dis_encoder``, ``edge_encoder``, ``graph_token``; ``in_degree_encoder``,
``out_degree_encoder``, ``graph_token``), so checkpoints interchange:

  * ``graphormer_pre_processing`` (:15-99) -- one-off, per graph, on the host: in/out degrees, the
    all-pairs spatial types (clipped shortest-path lengths), the edge types along one shortest path per
    pair.  The reference walks networkx's ``shortest_path`` output; here a FIFO breadth-first search over
    adjacency lists kept in ``edge_index`` order picks the SAME path (first-discovered predecessor), checked
    against the reference on random graphs (tests/golden/graphormer_encoder_layer.pt).
  * ``BiasEncoder`` (:102-183) -- per batch, on the device.  The reference scatters the ragged per-pair
    lists into dense ``[B, n, n, ...]`` tensors three times (``to_dense_adj``) and contracts the padded
    ``[B, n, n, dist, H]`` edge tensor; here the embeddings and the ``dist x H x H`` contraction are
    evaluated on the ragged list (sum of n_g^2 rows, no padding) and scattered ONCE into the dense
    ``[B*H, nmax, nmax]`` layout the attention kernel reads.
  * ``NodeEncoder`` (:207-248) -- degree embeddings (+ the graph token rows).

The reference's ``add_graph_token`` (:186-204) relies on ``torch.sort`` keeping equal keys in order and
leaves ``edge_index`` pointing at the pre-token rows; here the token row is placed first in each graph by
construction and ``edge_index`` / ``ptr`` are shifted with the nodes.
"""
from collections import deque

import torch

from ..graphgym.config import cfg
from ..graphgym.register import register_node_encoder
from . import encoders as _enc


# ------------------------------------------------------------------------------------------------
# host pre-processing (per graph, once per dataset)
# ------------------------------------------------------------------------------------------------
def _unique_edges(edge_index):
    """Edges in first-occurrence order without duplicates (what a networkx ``DiGraph`` keeps)."""
    seen, out = set(), []
    for s, t in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        if (s, t) not in seen:
            seen.add((s, t))
            out.append((s, t))
    return out


def graphormer_pre_processing(data, distance):
    """Adds ``in_degrees``, ``out_degrees`` and, unless ``posenc_GraphormerBias.node_degrees_only``,
    ``spatial_types`` [n*n], ``graph_index`` [2, n*n] (every ordered pair, row-major) and -- when the graph
    has (1-D, integer) ``edge_attr`` -- ``shortest_path_types`` [n*n, distance] to one graph."""
    n = int(data.x.shape[0]) if getattr(data, "x", None) is not None else int(data.num_nodes)
    edges = _unique_edges(data.edge_index)
    adj = [[] for _ in range(n)]
    in_deg, out_deg = [0] * n, [0] * n
    for s, t in edges:
        adj[s].append(t)
        out_deg[s] += 1
        in_deg[t] += 1
    data.in_degrees = torch.tensor(in_deg, dtype=torch.long)
    data.out_degrees = torch.tensor(out_deg, dtype=torch.long)
    pecfg = cfg.posenc_GraphormerBias
    if n and max(in_deg) >= pecfg.num_in_degrees:
        raise ValueError(f"Encountered in_degree: {max(in_deg)}, set posenc_"
                         f"GraphormerBias.num_in_degrees to at least {max(in_deg) + 1}")
    if n and max(out_deg) >= pecfg.num_out_degrees:
        raise ValueError(f"Encountered out_degree: {max(out_deg)}, set posenc_"
                         f"GraphormerBias.num_out_degrees to at least {max(out_deg) + 1}")
    if pecfg.node_degrees_only:
        return data

    has_attr = getattr(data, "edge_attr", None) is not None
    spatial = torch.full((n * n,), int(distance), dtype=torch.long)     # unreachable pairs
    if has_attr:
        sp_types = torch.zeros(n * n, distance, dtype=torch.long)
        etype = {}
        for s, t, a in zip(data.edge_index[0].tolist(), data.edge_index[1].tolist(),
                           data.edge_attr.tolist()):
            etype[(s, t)] = a                                         # the last duplicate wins, as an
    rows = torch.arange(n).repeat_interleave(n)                       # indexed assignment does
    cols = torch.arange(n).repeat(n)
    for src in range(n):
        parent = {src: -1}
        depth = {src: 0}
        queue = deque([src])
        while queue:
            v = queue.popleft()
            for w in adj[v]:
                if w not in parent:
                    parent[w] = v
                    depth[w] = depth[v] + 1
                    queue.append(w)
        for dst, dep in depth.items():
            # a path of `dep` hops has dep + 1 nodes; the reference clips it to `distance` NODES
            hops = min(dep, distance - 1)
            spatial[src * n + dst] = hops
            if has_attr and dep > 0:
                path = [dst]
                while path[-1] != src:
                    path.append(parent[path[-1]])
                path.reverse()
                path = path[:distance]
                for k in range(len(path) - 1):
                    sp_types[src * n + dst, k] = etype[(path[k], path[k + 1])]
    data.spatial_types = spatial
    data.graph_index = torch.stack([rows, cols])
    if has_attr:
        data.shortest_path_types = sp_types
    return data


def add_graphormer_stats(batch, distance=None):
    """``graphormer_pre_processing`` for every graph of an already collated HOST batch (what the reference's
    loader does per graph before collation, master_loader.py -> posenc_stats.py): attaches the batched
    ``in_degrees``, ``out_degrees``, ``spatial_types``, ``graph_index``, ``shortest_path_types``."""
    from ..data import Batch
    distance = int(distance if distance is not None else cfg.posenc_GraphormerBias.num_spatial_types)
    ptr = batch.ptr.tolist()
    src_graph = batch.batch[batch.edge_index[0]]
    parts = []
    for g in range(len(ptr) - 1):
        sel = src_graph == g
        one = Batch(x=batch.x[ptr[g]:ptr[g + 1]], edge_index=batch.edge_index[:, sel] - ptr[g])
        if getattr(batch, "edge_attr", None) is not None and batch.edge_attr.dim() == 1:
            one.edge_attr = batch.edge_attr[sel]
        one = graphormer_pre_processing(one, distance)
        parts.append({k: v for k, v in one.__dict__.items()
                      if k in ("x", "in_degrees", "out_degrees", "spatial_types", "graph_index",
                               "shortest_path_types")})
    joined = Batch.from_graph_list(parts)
    for k in ("in_degrees", "out_degrees", "spatial_types", "graph_index", "shortest_path_types"):
        if k in joined.__dict__:
            setattr(batch, k, getattr(joined, k))
    return batch


# ------------------------------------------------------------------------------------------------
# device encoders
# ------------------------------------------------------------------------------------------------
def _graph_sizes(batch):
    """(ptr [B+1] on the batch's device, B, largest graph).  One host read of the maximum."""
    if getattr(batch, "ptr", None) is not None:
        ptr = batch.ptr.to(batch.batch.device)
        B = int(ptr.numel() - 1)
    else:
        B = int(batch.num_graphs)
        counts = torch.bincount(batch.batch, minlength=B)
        ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    nmax = int((ptr[1:] - ptr[:-1]).max().item()) if B else 0
    return ptr, B, nmax


class BiasEncoder(torch.nn.Module):
    def __init__(self, num_heads: int, num_spatial_types: int, num_edge_types: int,
                 use_graph_token: bool = True):
        super().__init__()
        self.num_heads = num_heads
        # + 1: the type of disconnected pairs
        self.spatial_encoder = torch.nn.Embedding(num_spatial_types + 1, num_heads)
        self.edge_dis_encoder = torch.nn.Embedding(num_spatial_types * num_heads * num_heads, 1)
        self.edge_encoder = torch.nn.Embedding(num_edge_types, num_heads)
        self.use_graph_token = use_graph_token
        if self.use_graph_token:
            self.graph_token = torch.nn.Parameter(torch.zeros(1, num_heads, 1))
        self.reset_parameters()

    def reset_parameters(self):
        for emb in (self.spatial_encoder, self.edge_encoder, self.edge_dis_encoder):
            emb.weight.data.normal_(std=0.02)
        if self.use_graph_token:
            self.graph_token.data.normal_(std=0.02)

    def forward(self, data):
        H = self.num_heads
        ptr, B, nmax = _graph_sizes(data)
        src, dst = data.graph_index[0], data.graph_index[1]
        g = data.batch[src]
        il, jl = src - ptr[g], dst - ptr[g]
        # per-pair bias values on the ragged list of all ordered pairs: [sum n_g^2, H]
        vals = self.spatial_encoder(data.spatial_types)                              # reference :150-154
        if hasattr(data, "shortest_path_types"):
            e = self.edge_encoder(data.shortest_path_types)                          # [M, dist, H]
            w = self.edge_dis_encoder.weight.reshape(-1, H, H)                       # [dist, H, H]
            hops = data.spatial_types.to(vals.dtype).clamp(min=1.0)                  # :165-168
            vals = vals + torch.einsum('mdh,dhk->mk', e, w) / hops[:, None]          # :170-176
        t = 1 if self.use_graph_token else 0
        n = nmax + t
        bias = vals.new_zeros(B, H, n, n)
        bias[g, :, il + t, jl + t] = vals                                            # one scatter
        if self.use_graph_token:                                                     # :178-181
            bias[:, :, 1:, 0] = self.graph_token
            bias[:, :, 0, :] = self.graph_token
        data.attn_bias = bias.reshape(B * H, n, n)
        return data


def add_graph_token(data, token):
    """One token row in front of every graph (reference :186-204); ``batch``, ``ptr`` and ``edge_index``
    move with the nodes."""
    ptr, B, _ = _graph_sizes(data)
    N, dev = data.x.shape[0], data.x.device
    new_pos = torch.arange(N, device=dev) + data.batch + 1          # node i of graph b -> i + b + 1
    tok_pos = ptr[:-1] + torch.arange(B, device=dev)
    x = data.x.new_empty(N + B, data.x.shape[1])
    x[new_pos] = data.x
    x[tok_pos] = token.expand(B, -1)
    bvec = data.batch.new_empty(N + B)
    bvec[new_pos] = data.batch
    bvec[tok_pos] = torch.arange(B, device=dev, dtype=data.batch.dtype)
    if getattr(data, "edge_index", None) is not None:
        data.edge_index = new_pos[data.edge_index]
    data.x, data.batch = x, bvec
    if getattr(data, "ptr", None) is not None:
        data.ptr = (ptr + torch.arange(B + 1, device=dev)).to(data.ptr.dtype)
    return data


class NodeEncoder(torch.nn.Module):
    def __init__(self, embed_dim, num_in_degree, num_out_degree, input_dropout=0.0,
                 use_graph_token: bool = True):
        super().__init__()
        self.in_degree_encoder = torch.nn.Embedding(num_in_degree, embed_dim)
        self.out_degree_encoder = torch.nn.Embedding(num_out_degree, embed_dim)
        self.use_graph_token = use_graph_token
        if self.use_graph_token:
            self.graph_token = torch.nn.Parameter(torch.zeros(1, embed_dim))
        self.input_dropout = torch.nn.Dropout(input_dropout)
        self.reset_parameters()

    def reset_parameters(self):
        self.in_degree_encoder.weight.data.normal_(std=0.02)
        self.out_degree_encoder.weight.data.normal_(std=0.02)
        if self.use_graph_token:
            self.graph_token.data.normal_(std=0.02)

    def forward(self, data):
        deg = self.in_degree_encoder(data.in_degrees) + self.out_degree_encoder(data.out_degrees)
        data.x = data.x + deg if data.x.size(1) > 0 else deg                          # :231-234
        if self.use_graph_token:
            data = add_graph_token(data, self.graph_token)
        data.x = self.input_dropout(data.x)
        return data


@register_node_encoder("GraphormerBias", overwrite=True)
class GraphormerEncoder(torch.nn.Sequential):
    def __init__(self, dim_emb, *args, **kwargs):
        encoders = [
            BiasEncoder(cfg.graphormer.num_heads, cfg.posenc_GraphormerBias.num_spatial_types,
                        cfg.dataset.edge_encoder_num_types, cfg.graphormer.use_graph_token),
            NodeEncoder(dim_emb, cfg.posenc_GraphormerBias.num_in_degrees,
                        cfg.posenc_GraphormerBias.num_out_degrees, cfg.graphormer.input_dropout,
                        cfg.graphormer.use_graph_token),
        ]
        if cfg.posenc_GraphormerBias.node_degrees_only:      # no attention-bias encoder
            encoders = encoders[1:]
        super().__init__(*encoders)


@register_node_encoder('LinearNode', overwrite=True)
class LinearNodeEncoder(torch.nn.Module):
    """graphgps/encoder/linear_node_encoder.py:6-15."""

    def __init__(self, emb_dim):
        super().__init__()
        self.encoder = torch.nn.Linear(cfg.share.dim_in, emb_dim)

    def forward(self, batch):
        batch.x = self.encoder(batch.x)
        return batch


def _compose(classes, pe_names):
    """``X+GraphormerBias`` and ``X+GraphormerBias+RWSE`` (composed_encoders.py:36-100,144-155): the dataset
    encoder gets what the PE encoders leave of ``dim_emb`` (GraphormerBias adds in place: dim_pe = 0)."""

    class Composed(torch.nn.Module):
        def __init__(self, dim_emb):
            super().__init__()
            pe_dims = [getattr(cfg, f"posenc_{n}").dim_pe for n in pe_names]
            left = dim_emb - sum(pe_dims)
            self.encoder1 = classes[0](left)
            for k, cls in enumerate(classes[1:]):
                left += pe_dims[k]
                setattr(self, f"encoder{k + 2}", cls(left, expand_x=False))

        def forward(self, batch):
            for k in range(len(classes)):
                batch = getattr(self, f"encoder{k + 1}")(batch)
            return batch

    Composed.__name__ = "+".join(c.__name__ for c in classes)
    return Composed


for _ds_name, _ds_cls in (('Atom', _enc.AtomEncoder), ('ASTNode', _enc.ASTNodeEncoder),
                          ('TypeDictNode', _enc.TypeDictNodeEncoder), ('LinearNode', LinearNodeEncoder)):
    register_node_encoder(f"{_ds_name}+GraphormerBias",
                          _compose([_ds_cls, GraphormerEncoder], ['GraphormerBias']), overwrite=True)
    register_node_encoder(f"{_ds_name}+GraphormerBias+RWSE",
                          _compose([_ds_cls, GraphormerEncoder, _enc.RWSENodeEncoder],
                                   ['GraphormerBias', 'RWSE']), overwrite=True)
