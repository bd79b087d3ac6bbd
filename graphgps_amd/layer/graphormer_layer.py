"""GraphormerLayer on the varlen attention kernel with its additive-bias operand.

Drop-in for ``/root/reference/graphgps/layer/graphormer_layer.py:5-49`` (same constructor, same
parameter names: ``attention.*`` of a ``torch.nn.MultiheadAttention``, ``input_norm.*``, ``mlp.{0,1,4}.*``).
The reference pads with ``to_dense_batch``, runs the dense MHA with ``attn_mask=data.attn_bias`` and a key
padding mask and un-pads with a boolean index (:41-46); here the packed in-projection output goes straight
to ``csrc/seg_attention.hip`` with the dense ``[B*H, nmax, nmax]`` bias read in place.  LayerNorm, GELU and
the two d x d MLP GEMMs are library ops (this layer is a "next" row of the scope table).
"""
import torch

from ..fused import linear
from ..ops import graph_index_of, segment_attention


class GraphormerLayer(torch.nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout: float,
                 attention_dropout: float, mlp_dropout: float):
        super().__init__()
        self.ctor_kwargs = dict(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout,
                                attention_dropout=attention_dropout, mlp_dropout=mlp_dropout)
        # parameter container (names / init of the reference's module); its dense forward is never called
        self.attention = torch.nn.MultiheadAttention(embed_dim, num_heads, attention_dropout,
                                                     batch_first=True)
        self.num_heads = num_heads
        self.attention_dropout = attention_dropout
        self.input_norm = torch.nn.LayerNorm(embed_dim)
        self.dropout = torch.nn.Dropout(dropout)
        self.mlp = torch.nn.Sequential(
            torch.nn.LayerNorm(embed_dim),
            torch.nn.Linear(embed_dim, embed_dim),
            torch.nn.GELU(),
            torch.nn.Dropout(mlp_dropout),
            torch.nn.Linear(embed_dim, embed_dim),
            torch.nn.Dropout(dropout),
        )

    def forward(self, data):
        gi = graph_index_of(data)
        att = self.attention
        x = self.input_norm(data.x)                                              # reference :40
        qkv = linear(x, att.in_proj_weight, att.in_proj_bias)
        p = self.attention_dropout if self.training else 0.0
        o = segment_attention(qkv, gi, self.num_heads, p, bias=getattr(data, "attn_bias", None))  # :43-46
        x = linear(o, att.out_proj.weight, att.out_proj.bias)
        x = self.dropout(x) + data.x                                             # :47
        data.x = self.mlp(x) + x                                                 # :48
        return data
