"""Performer ``SelfAttention`` on the HIP FAVOR+ kernels.

Stands in for ``performer_pytorch.SelfAttention(dim, heads, dropout, causal=False)`` as the
reference constructs it at ``/root/reference/graphgps/layer/gps_layer.py:111-114`` (arithmetic
spec: ``graphgps/layer/performer_layer.py:421-508``; the vendored file and the pip package are
the same upstream code).  Same parameter/buffer names, so checkpoints interchange:
``to_q/to_k/to_v.weight`` (no bias, ``qkv_bias=False`` :436), ``to_out.{weight,bias}``,
``fast_attention.projection_matrix`` (a fixed buffer: the redraw logic lives only in the
whole-model ``Performer`` class, which GPSLayer never instantiates -- SURVEY.md section 3.3).

The dense path (pad -> einsum feature maps -> un-pad) is replaced by ``ops.favor_attention``;
unsupported options of the upstream class (causal, local heads, generalized attention, rotary
embeddings, cross attention) raise.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..fused import LinearGroup, linear
from ..ops import favor_attention


def orthogonal_matrix_chunk(cols):
    """QR of a square Gaussian block, on the CPU as upstream does (performer_layer.py:163-170)."""
    q, _ = torch.linalg.qr(torch.randn((cols, cols)).cpu(), mode='reduced')
    return q.t()


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0):
    """performer_layer.py:172-195: stacked orthogonal blocks, rows rescaled by the norms of a
    fresh Gaussian matrix (scaling 0) or by sqrt(nb_columns) (scaling 1)."""
    blocks = [orthogonal_matrix_chunk(nb_columns) for _ in range(int(nb_rows / nb_columns))]
    remaining = nb_rows - int(nb_rows / nb_columns) * nb_columns
    if remaining > 0:
        blocks.append(orthogonal_matrix_chunk(nb_columns)[:remaining])
    final = torch.cat(blocks)
    if scaling == 0:
        multiplier = torch.randn((nb_rows, nb_columns)).norm(dim=1)
    elif scaling == 1:
        multiplier = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,))
    else:
        raise ValueError(f'Invalid scaling {scaling}')
    return torch.diag(multiplier) @ final


class FastAttention(nn.Module):
    """Holder of the random-feature projection (performer_layer.py:251-273)."""

    def __init__(self, dim_heads, nb_features=None, ortho_scaling=0):
        super().__init__()
        nb_features = nb_features if nb_features is not None else int(dim_heads * math.log(dim_heads))
        self.dim_heads = dim_heads
        self.nb_features = nb_features
        self.ortho_scaling = ortho_scaling
        self.register_buffer('projection_matrix',
                             gaussian_orthogonal_random_matrix(nb_features, dim_heads, ortho_scaling))

    @torch.no_grad()
    def redraw_projection_matrix(self, device=None):
        self.projection_matrix.copy_(
            gaussian_orthogonal_random_matrix(self.nb_features, self.dim_heads, self.ortho_scaling))


class SelfAttention(nn.Module):
    def __init__(self, dim, causal=False, heads=8, dim_head=64, local_heads=0,
                 local_window_size=256, nb_features=None, feature_redraw_interval=1000,
                 generalized_attention=False, kernel_fn=None, dropout=0., no_projection=False,
                 qkv_bias=False, attn_out_bias=True):
        super().__init__()
        assert dim % heads == 0, 'dimension must be divisible by number of heads'
        if causal or local_heads or generalized_attention or no_projection:
            raise NotImplementedError("only the non-causal softmax-kernel FAVOR+ path that "
                                      "GPSLayer uses is built (gps_layer.py:111-114)")
        if qkv_bias:
            raise NotImplementedError("qkv_bias=True makes padded key rows non-zero; GraphGPS never "
                                      "sets it (performer_layer.py:436 default False)")
        dim_head = dim_head if dim_head is not None else dim // heads
        inner_dim = dim_head * heads
        self.fast_attention = FastAttention(dim_head, nb_features)
        self.heads = heads
        self.global_heads = heads
        self.to_q = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_k = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_v = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_out = nn.Linear(inner_dim, dim, bias=attn_out_bias)
        self.dropout = nn.Dropout(dropout)
        self._qkv = None

    def forward_segments(self, x, gi):
        """[N, dim] node features + graph index -> [N, dim]; equals
        ``SelfAttention(to_dense_batch(x), mask=mask)[mask]`` of the reference (gps_layer.py:199,206)."""
        if self._qkv is None:
            self._qkv = LinearGroup([self.to_q, self.to_k, self.to_v])
        qkv = self._qkv(x)
        out = favor_attention(qkv, self.fast_attention.projection_matrix, gi, self.heads)
        return self.dropout(linear(out, self.to_out.weight, self.to_out.bias))

    def forward(self, x, mask=None, **kwargs):
        raise NotImplementedError(
            "dense [B, Nmax, d] call convention is the reference's padded path; the HIP path is "
            "varlen: call forward_segments(x, graph_index) (GPSLayer does)")
