"""GPSLayer: local MPNN || global attention -> sum -> FFN, on MI355X HIP kernels.

Drop-in for ``/root/reference/graphgps/layer/gps_layer.py:15-264``: identical constructor
signature (:20-24), ``forward(batch) -> batch`` contract (reads ``batch.x``, ``edge_index``,
``edge_attr``, ``batch``/``ptr``; assigns ``batch.x`` and, for GatedGCN, ``batch.edge_attr``),
identical parameter/buffer names so ``state_dict``s interchange with reference checkpoints
(SURVEY.md section 8b).

What runs where:
  * sparse local half  -> csrc/gatedgcn.hip / csrc/gine.hip (CSR segment reductions)
  * global half        -> csrc/seg_attention.hip (varlen MFMA attention off ``ptr``; the
                          reference's to_dense_batch padding, key-padding mask and [mask]
                          un-pad -- 3 host syncs per layer -- do not exist here)
                          or csrc/favor.hip (Performer FAVOR+)
  * dense projections / FFN / BatchNorm -> rocBLAS + MIOpen through torch
There is no CPU fallback: on a CPU tensor the ops raise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graphgym import register
from ..graphgym import act as _act  # noqa: F401
from ..fused import add_dropout, bn_act, linear, relu_dropout
from ..ops import graph_index_of, segment_attention
from .gatedgcn_layer import GatedGCNLayer
from .gcn_conv_layer import GCNConv
from .gine_conv_layer import GINEConv, GINEConvESLapPE
from .gps_block import (block_eval_supported, block_supported, gine_block_supported, gps_block, gps_block_eval,
                        gps_block_gine)

import os as _os
# single-node block path (layer/gps_block.py): merges the A|B|D|E and in-proj GEMMs; measured
# 16.9 -> 15.8 ms/step on MI355X.  GPS_FUSED_BLOCK=0 keeps the operator-by-operator path.
_BLOCK_ENABLED = _os.environ.get("GPS_FUSED_BLOCK", "1") != "0"
# inference form of the block (model.eval() under no_grad: eval_epoch, custom_train.py:50-77); GPS_EVAL_BLOCK=0 keeps
# the operator-by-operator path there
_EVAL_BLOCK = _os.environ.get("GPS_EVAL_BLOCK", "1") != "0"

_NEEDS_PYG = {"GIN", "GENConv", "GAT", "PNA"}


class GPSLayer(nn.Module):
    """Local MPNN + full graph attention x-former layer."""

    def __init__(self, dim_h,
                 local_gnn_type, global_model_type, num_heads, act='relu',
                 pna_degrees=None, equivstable_pe=False, dropout=0.0,
                 attn_dropout=0.0, layer_norm=False, batch_norm=True,
                 bigbird_cfg=None, log_attn_weights=False):
        super().__init__()
        self.ctor_kwargs = dict(dim_h=dim_h, local_gnn_type=local_gnn_type,
                                global_model_type=global_model_type, num_heads=num_heads,
                                act=act, equivstable_pe=equivstable_pe, dropout=dropout,
                                attn_dropout=attn_dropout, layer_norm=layer_norm,
                                batch_norm=batch_norm)
        self.dim_h = dim_h
        self.num_heads = num_heads
        self.attn_dropout = attn_dropout
        self.layer_norm = layer_norm
        self.batch_norm = batch_norm
        self.equivstable_pe = equivstable_pe
        self.activation = register.act_dict[act]

        self.log_attn_weights = log_attn_weights
        if log_attn_weights and global_model_type not in ['Transformer', 'BiasedTransformer']:
            raise NotImplementedError(
                f"Logging of attention weights is not supported "
                f"for '{global_model_type}' global attention model.")
        if log_attn_weights:
            raise NotImplementedError(
                "log_attn_weights materialises [B,H,n,n]; the varlen kernel never forms it")

        # Local message-passing model (reference :43-98).
        self.local_gnn_with_edge_attr = True
        if local_gnn_type == 'None':
            self.local_model = None
        elif local_gnn_type == 'GCN':            # MPNN without edge attributes (reference :50-52)
            self.local_gnn_with_edge_attr = False
            self.local_model = GCNConv(dim_h, dim_h)
        elif local_gnn_type == 'GINE':
            gin_nn = nn.Sequential(nn.Linear(dim_h, dim_h), self.activation(),
                                   nn.Linear(dim_h, dim_h))
            if self.equivstable_pe:          # specialised GINE layer for EquivStableLapPE (:66-67)
                self.local_model = GINEConvESLapPE(gin_nn)
            else:
                self.local_model = GINEConv(gin_nn)
        elif local_gnn_type == 'CustomGatedGCN':
            self.local_model = GatedGCNLayer(dim_h, dim_h, dropout=dropout, residual=True,
                                             act=act, equivstable_pe=equivstable_pe)
        elif local_gnn_type in _NEEDS_PYG:
            raise NotImplementedError(
                f"local_gnn_type={local_gnn_type!r} is a PyG-native conv outside the HIP hot path "
                f"(SURVEY.md section 8b item 4); supported: 'None', 'GCN', 'GINE', 'CustomGatedGCN'")
        else:
            raise ValueError(f"Unsupported local GNN model: {local_gnn_type}")
        self.local_gnn_type = local_gnn_type

        # Global attention transformer-style model (reference :100-123).
        if global_model_type == 'None':
            self.self_attn = None
        elif global_model_type in ('Transformer', 'BiasedTransformer'):
            # BiasedTransformer = the same module, called with attn_mask=batch.attn_bias (:201-203)
            if dim_h % num_heads != 0:
                raise AssertionError("embed_dim must be divisible by num_heads")
            # parameter container only (same names/init as the reference's module, :104-106);
            # its dense forward is never called
            self.self_attn = torch.nn.MultiheadAttention(
                dim_h, num_heads, dropout=self.attn_dropout, batch_first=True)
        elif global_model_type == 'Performer':
            try:
                from .performer_layer import SelfAttention
            except ImportError as e:
                raise NotImplementedError(
                    "global_model_type='Performer': the FAVOR+ HIP path (csrc/favor.hip) is not "
                    "built in this tree") from e
            self.self_attn = SelfAttention(dim=dim_h, heads=num_heads,
                                           dropout=self.attn_dropout, causal=False)
        elif global_model_type == 'BigBird':
            raise NotImplementedError(
                "global_model_type='BigBird' (block-sparse attention) is outside the HIP hot path; "
                "supported: 'None', 'Transformer', 'BiasedTransformer', 'Performer'")
        else:
            raise ValueError(f"Unsupported global x-former model: {global_model_type}")
        self.global_model_type = global_model_type

        if self.layer_norm and self.batch_norm:
            raise ValueError("Cannot apply two types of normalization together")
        if self.layer_norm:                  # reference :129-131: pygnn.norm.LayerNorm(dim_h), PyG's graph-mode LayerNorm
            from .graph_layernorm import GraphLayerNorm
            self.norm1_local = GraphLayerNorm(dim_h)
            self.norm1_attn = GraphLayerNorm(dim_h)
        if self.batch_norm:
            self.norm1_local = nn.BatchNorm1d(dim_h)
            self.norm1_attn = nn.BatchNorm1d(dim_h)
        self.dropout_local = nn.Dropout(dropout)
        self.dropout_attn = nn.Dropout(dropout)

        # Feed Forward block (reference :142-153).
        self.ff_linear1 = nn.Linear(dim_h, dim_h * 2)
        self.ff_linear2 = nn.Linear(dim_h * 2, dim_h)
        self.act_fn_ff = self.activation()
        if self.layer_norm:                  # reference :148
            from .graph_layernorm import GraphLayerNorm
            self.norm2 = GraphLayerNorm(dim_h)
        if self.batch_norm:
            self.norm2 = nn.BatchNorm1d(dim_h)
        self.ff_dropout1 = nn.Dropout(dropout)
        self.ff_dropout2 = nn.Dropout(dropout)

    def forward(self, batch):
        h = batch.x
        h_in1 = h  # for first residual connection
        gi = graph_index_of(batch)

        edge_attr = getattr(batch, 'edge_attr', None)      # absent for MPNNs without edge attributes
        if _BLOCK_ENABLED and block_supported(self, h, edge_attr):
            # measured configuration (CustomGatedGCN+Transformer, BN, ReLU, training): the whole
            # block as ONE autograd node -- same kernels, ~4x less host time (layer/gps_block.py)
            h, e_new = gps_block(self, h, batch.edge_attr, gi)
            batch.x = h
            batch.edge_attr = e_new
            return batch
        if _BLOCK_ENABLED and _EVAL_BLOCK and block_eval_supported(self, h, edge_attr):
            # model.eval() under no_grad (eval_epoch, inference): the same kernels on the running statistics
            h, e_new = gps_block_eval(self, h, batch.edge_attr, gi)
            batch.x = h
            batch.edge_attr = e_new
            return batch
        if _BLOCK_ENABLED and gine_block_supported(self, h, edge_attr):
            batch.x = gps_block_gine(self, h, batch.edge_attr, gi)      # GINE leaves edge_attr as is
            return batch
        if gi.n_real is not None and self.training:
            # a padded batch (loader.BucketPadding) that took none of the blocks above: every other path computes its
            # BatchNorm statistics over ALL rows it is given -- padding would silently change the batch
            from ..lib import GpsHipError
            raise GpsHipError("padded batches (batch.gps_counts) are served by the fused blocks only (CustomGatedGCN + "
                              "Transformer / Performer, GINE + Transformer: training mode, BatchNorm, ReLU); pad nothing "
                              "for this layer")

        h_local = h_attn = None
        if self.local_model is not None:
            if self.local_gnn_type == 'CustomGatedGCN':
                # GatedGCN does residual connection and dropout internally (reference :164-174)
                es = batch.pe_EquivStableLapPE if self.equivstable_pe else None      # :165-166
                h_local, e_new = self.local_model.forward_tensors(h, batch.edge_attr, gi, es)
                batch.edge_attr = e_new
            else:
                if not self.local_gnn_with_edge_attr:                                # :183
                    h_local = self.local_model.forward_tensors(h, gi)
                elif self.equivstable_pe:                                            # :177-181
                    h_local = self.local_model.forward_tensors(h, batch.edge_attr, gi,
                                                               batch.pe_EquivStableLapPE)
                else:
                    h_local = self.local_model.forward_tensors(h, batch.edge_attr, gi)
                # dropout_local + residual (reference :188-189)
                h_local = add_dropout(h_in1, h_local, self.dropout_local.p, self.training)
            if self.layer_norm:
                h_local = self.norm1_local(h_local, gi)                      # :191-192
            if self.batch_norm:
                h_local = bn_act(h_local, self.norm1_local)                  # :193-194

        if self.self_attn is not None:
            # the global branch attends over the PRE-layer h (reference :156,199)
            if self.global_model_type == 'Transformer':
                h_attn = self._sa_block(h, gi)
            elif self.global_model_type == 'BiasedTransformer':
                # Graphormer-like conditioning, requires `batch.attn_bias` (reference :201-203)
                h_attn = self._sa_block(h, gi, batch.attn_bias)
            elif self.global_model_type == 'Performer':
                h_attn = self.self_attn.forward_segments(h, gi)
            else:
                raise RuntimeError(f"Unexpected {self.global_model_type}")
            h_attn = add_dropout(h_in1, h_attn, self.dropout_attn.p, self.training)  # :212-213
            if self.layer_norm:
                h_attn = self.norm1_attn(h_attn, gi)                         # :209-210
            if self.batch_norm:
                # norm1_attn, with the branch sum h_local + h_attn (:222) folded into the
                # BN epilogue as its residual operand
                h_attn = bn_act(h_attn, self.norm1_attn, res=h_local)
                h_local = None

        # Combine local and global outputs (reference :222: sum of the branch outputs).
        parts = [t for t in (h_local, h_attn) if t is not None]
        h = parts[0] if len(parts) == 1 else sum(parts)

        # Feed Forward block + norm2 (reference :225-229).
        ff = self._ff_block(h)
        h = add_dropout(h, ff, self.ff_dropout2.p, self.training)
        if self.layer_norm:
            h = self.norm2(h, gi)                                            # :226-227
        if self.batch_norm:
            h = bn_act(h, self.norm2)

        batch.x = h
        return batch

    def _sa_block(self, x, gi, attn_bias=None):
        """Self-attention block: packed in-proj GEMM -> varlen MFMA attention -> out-proj GEMM.
        Same arithmetic as nn.MultiheadAttention(x, x, x, attn_mask=attn_bias,
        key_padding_mask=~mask)[mask] (reference :199-203,234-241) without the dense padding; the
        dense ``[B*H, nmax, nmax]`` bias is read in place, its n_g x n_g corners only."""
        sa = self.self_attn
        qkv = linear(x, sa.in_proj_weight, sa.in_proj_bias)
        p = self.attn_dropout if self.training else 0.0
        o = segment_attention(qkv, gi, self.num_heads, p, bias=attn_bias)
        return linear(o, sa.out_proj.weight, sa.out_proj.bias)

    def _ff_block(self, x):
        """ff_linear2(ff_dropout1(act(ff_linear1(x)))); ff_dropout2 is applied by the caller
        together with the residual add (reference :253-257)."""
        x = linear(x, self.ff_linear1.weight, self.ff_linear1.bias)
        if isinstance(self.act_fn_ff, nn.ReLU):
            x = relu_dropout(x, self.ff_dropout1.p, self.training)
        else:
            x = self.ff_dropout1(self.act_fn_ff(x))
        return linear(x, self.ff_linear2.weight, self.ff_linear2.bias)

    def extra_repr(self):
        return (f'summary: dim_h={self.dim_h}, local_gnn_type={self.local_gnn_type}, '
                f'global_model_type={self.global_model_type}, heads={self.num_heads}')
