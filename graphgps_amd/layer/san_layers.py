"""SAN graph-transformer layers (``SANLayer`` / ``SAN2Layer``).

The attention over the REAL edges -- score, clamp-exp (SAN) or per-target softmax (SAN2), weighted sum of source
values -- is one HIP gather-gate-segment-reduce kernel (csrc/edge_attn.hip through ``ops.edge_attention``) whenever
the tensors are on the GPU and the head width is 4, 8, 16, 32 or 64; the complement-graph ("fake edge") half of the
full-graph variants, the gamma mixing and everything on the CPU use device torch ops at the reference's abstraction
level (gather, ``index_add_``, ``scatter_reduce``).  The modules carry the reference's parameter names (``attention.{Q,K,E,V,Q_2,K_2,E_2}``, SAN2: ``attention.gamma``;
``O_h``, ``FFN_h_layer{1,2}``, ``batch_norm{1,2}_h`` / ``layer_norm{1,2}_h``) and arithmetic
(``/root/reference/graphgps/layer/san_layer.py:10-216``, ``san2_layer.py:11-238``):

    real edges j->i:   s_ij = sum_c K_j Q_i E_ij / sqrt(d_h)          fake (complement) pairs: the *_2 projections
    SANLayer:          w = exp(clamp(s, -5, 5)) scaled 1/(gamma+1) | gamma/(gamma+1);  h_i = sum w V_j / (sum w + 1e-6)
    SAN2Layer:         w = softmax over each target's real (resp. fake) pairs, same gamma scaling;  h_i = sum w V_j

The complement of the adjacency (``graphgps/utils.py:12-66``: per graph, every ordered pair i != j that is not an
edge) is built vectorised from the batch's ``ptr`` instead of one dense n x n matrix per graph in a Python loop.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def complement_edge_index(edge_index, batch):
    """All ordered same-graph pairs (src, dst), src != dst, that are not columns of ``edge_index``."""
    n_graphs = int(batch.max().item()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=n_graphs)
    first = torch.cumsum(counts, 0) - counts
    n_of = counts[batch]                                             # size of each node's graph
    src = torch.repeat_interleave(torch.arange(batch.numel(), device=batch.device), n_of)
    # position of each pair within its source node's run -> the partner's local index
    run_start = torch.cumsum(n_of, 0) - n_of
    local = torch.arange(src.numel(), device=batch.device) - run_start[src]
    dst = first[batch[src]] + local
    N = batch.numel()
    taken = torch.isin(src * N + dst, edge_index[0] * N + edge_index[1])
    keep = (src != dst) & ~taken
    return torch.stack([src[keep], dst[keep]])


def _segment_softmax(score, index, num_nodes):
    """Softmax of ``score`` [M, H, 1] within the groups given by ``index`` (san2_layer.py:11-33)."""
    idx = index.view(-1, 1, 1).expand_as(score)
    top = torch.full((num_nodes,) + tuple(score.shape[1:]), float("-inf"), dtype=score.dtype,
                     device=score.device).scatter_reduce(0, idx, score, "amax", include_self=True)
    e = (score - top[index]).exp()
    tot = torch.zeros_like(top).index_add_(0, index, e)
    return e / (tot[index] + 1e-16)


class _SANAttention(nn.Module):
    def __init__(self, gamma, in_dim, out_dim, num_heads, full_graph, fake_edge_emb, use_bias, softmax):
        super().__init__()
        self.out_dim, self.num_heads, self.full_graph, self.softmax = out_dim, num_heads, full_graph, softmax
        if softmax:       # SAN2: a learnable gamma, clamped to [0, 1] at use
            self.gamma = nn.Parameter(torch.tensor(0.5, dtype=float), requires_grad=True)
        else:
            self.gamma = gamma
        width = out_dim * num_heads
        self.Q = nn.Linear(in_dim, width, bias=use_bias)
        self.K = nn.Linear(in_dim, width, bias=use_bias)
        self.E = nn.Linear(in_dim, width, bias=use_bias)
        if full_graph:
            self.Q_2 = nn.Linear(in_dim, width, bias=use_bias)
            self.K_2 = nn.Linear(in_dim, width, bias=use_bias)
            self.E_2 = nn.Linear(in_dim, width, bias=use_bias)
            self.fake_edge_emb = fake_edge_emb
        self.V = nn.Linear(in_dim, width, bias=use_bias)

    def _weights(self, q, k, e, pairs, n):
        """Unnormalised (SAN) or per-target softmax (SAN2) weight of every pair: [M, H, 1]."""
        s = (k[pairs[0]] * q[pairs[1]] * e / math.sqrt(self.out_dim)).sum(-1, keepdim=True)
        return _segment_softmax(s, pairs[1], n) if self.softmax else torch.exp(s.clamp(-5, 5))

    def forward(self, batch):
        H, D, n = self.num_heads, self.out_dim, batch.x.shape[0]
        x = batch.x
        if x.is_cuda and x.dtype == torch.float32:
            from ..ops import edge_attention_supported
            if edge_attention_supported(H, D):
                return self._forward_hip(batch)
        v = self.V(x).view(n, H, D)
        real = batch.edge_index
        w = self._weights(self.Q(x).view(n, H, D), self.K(x).view(n, H, D),
                          self.E(batch.edge_attr).view(-1, H, D), real, n)
        if self.full_graph:
            fake = complement_edge_index(real, batch.batch)
            e2 = self.E_2(self.fake_edge_emb(real.new_zeros(1))).view(1, H, D)    # one embedding for all
            w2 = self._weights(self.Q_2(x).view(n, H, D), self.K_2(x).view(n, H, D), e2, fake, n)
            g = torch.clamp(self.gamma, min=0.0, max=1.0) if self.softmax else self.gamma
            w, w2 = w / (g + 1), g * w2 / (g + 1)
        wv = torch.zeros_like(v).index_add_(0, real[1], v[real[0]] * w)
        z = w.new_zeros(n, H, 1).index_add_(0, real[1], w)
        if self.full_graph:
            wv = wv.index_add_(0, fake[1], v[fake[0]] * w2)
            z = z.index_add_(0, fake[1], w2)
        return wv if self.softmax else wv / (z + 1e-6)


    def _forward_hip(self, batch):
        """Real edges on the HIP kernel; fake pairs (full_graph) and the gamma mixing as above."""
        from ..ops import edge_attention, graph_index_of
        H, D, n = self.num_heads, self.out_dim, batch.x.shape[0]
        x = batch.x
        v2d = self.V(x)
        gi = graph_index_of(batch)
        wv, z = edge_attention(self.Q(x), self.K(x), v2d, self.E(batch.edge_attr), gi, H, self.softmax)
        wv, z = wv.view(n, H, D), z.view(n, H, 1)
        if self.full_graph:
            v = v2d.view(n, H, D)
            real = batch.edge_index
            fake = complement_edge_index(real, batch.batch)
            e2 = self.E_2(self.fake_edge_emb(real.new_zeros(1))).view(1, H, D)    # one embedding for all
            w2 = self._weights(self.Q_2(x).view(n, H, D), self.K_2(x).view(n, H, D), e2, fake, n)
            g = torch.clamp(self.gamma, min=0.0, max=1.0) if self.softmax else self.gamma
            w2 = g * w2 / (g + 1)
            wv = (wv / (g + 1)).index_add(0, fake[1], v[fake[0]] * w2)
            z = (z / (g + 1)).index_add(0, fake[1], w2)
        return wv if self.softmax else wv / (z + 1e-6)


class _SANBlock(nn.Module):
    """attention -> dropout -> O_h -> (+x) -> norm -> FFN (d -> 2d -> d, ReLU) -> (+) -> norm."""
    _softmax = False

    def __init__(self, gamma, in_dim, out_dim, num_heads, full_graph, fake_edge_emb, dropout=0.0,
                 layer_norm=False, batch_norm=True, residual=True, use_bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.num_heads = in_dim, out_dim, num_heads
        self.dropout, self.residual, self.layer_norm, self.batch_norm = dropout, residual, layer_norm, batch_norm
        self.attention = _SANAttention(gamma, in_dim, out_dim // num_heads, num_heads, full_graph,
                                       fake_edge_emb, use_bias, self._softmax)
        self.O_h = nn.Linear(out_dim, out_dim)
        if layer_norm:
            self.layer_norm1_h = nn.LayerNorm(out_dim)
        if batch_norm:
            self.batch_norm1_h = nn.BatchNorm1d(out_dim)
        self.FFN_h_layer1 = nn.Linear(out_dim, out_dim * 2)
        self.FFN_h_layer2 = nn.Linear(out_dim * 2, out_dim)
        if layer_norm:
            self.layer_norm2_h = nn.LayerNorm(out_dim)
        if batch_norm:
            self.batch_norm2_h = nn.BatchNorm1d(out_dim)

    def _norm(self, h, which):
        if self.layer_norm:
            h = getattr(self, f"layer_norm{which}_h")(h)
        if self.batch_norm:
            h = getattr(self, f"batch_norm{which}_h")(h)
        return h

    def forward(self, batch):
        x = batch.x
        h = self.attention(batch).reshape(-1, self.out_channels)
        h = self.O_h(F.dropout(h, self.dropout, training=self.training))
        h = self._norm(x + h if self.residual else h, 1)
        f = F.dropout(F.relu(self.FFN_h_layer1(h)), self.dropout, training=self.training)
        f = self.FFN_h_layer2(f)
        batch.x = self._norm(h + f if self.residual else f, 2)
        return batch

    def __repr__(self):
        return '{}(in_channels={}, out_channels={}, heads={}, residual={})'.format(
            self.__class__.__name__, self.in_channels, self.out_channels, self.num_heads, self.residual)


class SANLayer(_SANBlock):
    """exp-clamp weights normalised by their sum (san_layer.py)."""
    _softmax = False


class SAN2Layer(_SANBlock):
    """Per-target softmax over the real and over the fake pairs, learnable gamma (san2_layer.py)."""
    _softmax = True
