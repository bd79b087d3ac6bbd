"""CPU oracle for the GPSLayer hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product path (``graphgps_amd``) never
does and fails loudly when its HIP extension is missing.

What it is: a pure-PyTorch (CPU, fp32 or fp64) restatement of the reference path,
at the reference's own op granularity (``index_select`` gathers, ``index_add_``
scatters, dense padding + ``torch.nn.MultiheadAttention``, einsum FAVOR+), each
piece citing the reference file:line it follows (paths relative to
``/root/reference``).

Pinning status (also in DESIGN.md):
* The reference ships no golden vectors / KATs for this path.
* In-tree arithmetic -- ``GatedGCNLayer`` (graphgps/layer/gatedgcn_layer.py),
  ``GPSLayer`` wiring incl. the BiasedTransformer branch (graphgps/layer/gps_layer.py), FAVOR+
  (graphgps/layer/performer_layer.py), ``GraphormerLayer`` (graphgps/layer/graphormer_layer.py) and the
  Graphormer / SignNet encoders (graphgps/encoder/{graphormer_encoder,signnet_pos_encoder}.py) -- IS pinned:
  ``oracle/gen_golden.py`` imports those reference files unmodified in this container (hosting them on the
  stub modules in ``oracle/ref_stubs``) and writes ``tests/golden/*.pt``;
  ``tests/test_oracle_golden.py`` checks this oracle against them.
* Third-party semantics the reference relies on but does not vendor
  (PyG 2.2 ``MessagePassing.propagate`` index convention, ``GINEConv`` / ``GCNConv`` / ``GINConv``,
  ``to_dense_batch`` / ``to_dense_adj``; ``torch_scatter.scatter``) are restated from their published
  behaviour both here and in the stubs -> that part is "parity unpinned".
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

ACT = {"relu": nn.ReLU, "gelu": nn.GELU, "elu": nn.ELU, "selu": nn.SELU,
       "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "prelu": nn.PReLU,
       "identity": nn.Identity}


# ---------------------------------------------------------------------------
# third-party primitives, restated
# ---------------------------------------------------------------------------
def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """``torch_scatter.scatter(src, index, 0, None, dim_size, reduce='sum')`` ==
    ``zeros.scatter_add_`` == ``index_add_`` (sequential edge order on CPU).
    Call sites: graphgps/layer/gatedgcn_layer.py:117-123."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def to_dense_batch(x: torch.Tensor, batch: torch.Tensor,
                   num_graphs: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """PyG ``to_dense_batch`` (call site graphgps/layer/gps_layer.py:199): pad ragged
    [N,d] to [B,Nmax,d] with zeros, return bool mask [B,Nmax]."""
    B = int(batch.max()) + 1 if num_graphs is None else num_graphs
    counts = torch.bincount(batch, minlength=B)
    nmax = int(counts.max())
    cum = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    idx = torch.arange(x.shape[0], device=x.device) - cum[batch] + batch * nmax
    dense = x.new_zeros((B * nmax,) + tuple(x.shape[1:]))
    dense[idx] = x
    mask = torch.zeros(B * nmax, dtype=torch.bool, device=x.device)
    mask[idx] = True
    return dense.view(B, nmax, *x.shape[1:]), mask.view(B, nmax)


# ---------------------------------------------------------------------------
# local MPNNs
# ---------------------------------------------------------------------------
class OracleGatedGCNLayer(nn.Module):
    """graphgps/layer/gatedgcn_layer.py:11-136 (equivstable_pe=False branch)."""

    def __init__(self, in_dim, out_dim, dropout, residual, act="relu", equivstable_pe=False):
        super().__init__()
        self.EquivStablePE = equivstable_pe                       # :28-35
        if equivstable_pe:
            self.mlp_r_ij = nn.Sequential(nn.Linear(1, out_dim), ACT[act](), nn.Linear(out_dim, 1),
                                          nn.Sigmoid())
        # :21-25  pyg_nn.Linear(bias=True) == nn.Linear for shapes/keys
        self.A = nn.Linear(in_dim, out_dim, bias=True)
        self.B = nn.Linear(in_dim, out_dim, bias=True)
        self.C = nn.Linear(in_dim, out_dim, bias=True)
        self.D = nn.Linear(in_dim, out_dim, bias=True)
        self.E = nn.Linear(in_dim, out_dim, bias=True)
        self.bn_node_x = nn.BatchNorm1d(out_dim)   # :37
        self.bn_edge_e = nn.BatchNorm1d(out_dim)   # :38
        self.act_fn_x = ACT[act]()
        self.act_fn_e = ACT[act]()
        self.dropout = dropout
        self.residual = residual

    def forward(self, x, e, edge_index, pe=None):
        x_in, e_in = x, e                                         # :54-55
        Ax, Bx, Ce, Dx, Ex = self.A(x), self.B(x), self.C(e), self.D(x), self.E(x)  # :57-61
        # propagate (:67-70): PyG flow source_to_target => _j = edge_index[0], _i = edge_index[1]
        j, i = edge_index[0], edge_index[1]
        Dx_i, Ex_j, Bx_j = Dx.index_select(0, i), Ex.index_select(0, j), Bx.index_select(0, j)
        e_ij = Dx_i + Ex_j + Ce                                   # :96
        sigma_ij = torch.sigmoid(e_ij)                            # :97
        if self.EquivStablePE:                                    # :101-104
            r_ij = ((pe.index_select(0, i) - pe.index_select(0, j)) ** 2).sum(dim=-1, keepdim=True)
            sigma_ij = sigma_ij * self.mlp_r_ij(r_ij)
        N = Bx.shape[0]                                           # :115
        num = scatter_sum(sigma_ij * Bx_j, i, N)                  # :117-119
        den = scatter_sum(sigma_ij, i, N)                         # :121-123
        x = Ax + num / (den + 1e-6)                               # :125,133
        e = e_ij                                                  # :134
        x = self.bn_node_x(x)                                     # :72
        e = self.bn_edge_e(e)                                     # :73
        x = self.act_fn_x(x)
        e = self.act_fn_e(e)
        x = F.dropout(x, self.dropout, training=self.training)    # :78
        e = F.dropout(e, self.dropout, training=self.training)    # :79
        if self.residual:
            x = x_in + x
            e = e_in + e
        return x, e


class OracleGINEConv(nn.Module):
    """PyG 2.2 ``GINEConv(nn, eps=0, train_eps=False, edge_dim=None)`` (third-party),
    constructed at graphgps/layer/gps_layer.py:62-69; message form mirrored in-tree
    at graphgps/layer/gine_conv_layer.py:70-84 (minus the r_ij factor)."""

    def __init__(self, nn_module: nn.Module, eps: float = 0.0, equivstable_pe: bool = False):
        super().__init__()
        self.nn = nn_module
        self.register_buffer("eps", torch.tensor([eps]))
        if equivstable_pe:   # GINEConvESLapPE, graphgps/layer/gine_conv_layer.py:43-48,80-84
            out_dim = nn_module[0].out_features
            self.mlp_r_ij = nn.Sequential(nn.Linear(1, out_dim), nn.ReLU(), nn.Linear(out_dim, 1),
                                          nn.Sigmoid())
        self.EquivStablePE = equivstable_pe

    def forward(self, x, edge_index, edge_attr, pe=None):
        j, i = edge_index[0], edge_index[1]
        msg = (x.index_select(0, j) + edge_attr).relu()
        if self.EquivStablePE:
            r_ij = ((pe.index_select(0, i) - pe.index_select(0, j)) ** 2).sum(dim=-1, keepdim=True)
            msg = msg * self.mlp_r_ij(r_ij)
        out = scatter_sum(msg, i, x.shape[0])
        out = out + (1 + self.eps) * x
        return self.nn(out)


# ---------------------------------------------------------------------------
# Performer (FAVOR+)
# ---------------------------------------------------------------------------
class OracleGCNConv(nn.Module):
    """PyG 2.2 ``GCNConv(in, out)`` with its defaults (third-party; published algorithm): ``lin`` without bias,
    ``gcn_norm`` = drop the input's self loops, add ONE unit self loop per node (add_remaining_self_loops),
    deg = scatter-add of ones over TARGETS, weight(j->i) = deg_j^-1/2 deg_i^-1/2; sum-aggregate; + ``bias``.
    Local model of ``gt.layer_type: GCN+...`` (graphgps/layer/gps_layer.py:53-55,183)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        n = x.shape[0]
        keep = edge_index[0] != edge_index[1]
        loops = torch.arange(n, dtype=edge_index.dtype)
        row = torch.cat([edge_index[0][keep], loops])          # sources
        col = torch.cat([edge_index[1][keep], loops])          # targets
        deg = torch.zeros(n, dtype=x.dtype).index_add_(0, col, torch.ones(col.numel(), dtype=x.dtype))
        dinv = deg.pow(-0.5)
        w = dinv[row] * dinv[col]
        h = self.lin(x)
        return scatter_sum(w[:, None] * h.index_select(0, row), col, n) + self.bias


def orthogonal_matrix_chunk(cols: int) -> torch.Tensor:
    # graphgps/layer/performer_layer.py:163-170
    q, _ = torch.linalg.qr(torch.randn(cols, cols), mode="reduced")
    return q.t()


def gaussian_orthogonal_random_matrix(nb_rows: int, nb_columns: int) -> torch.Tensor:
    # graphgps/layer/performer_layer.py:172-195 with scaling == 0
    blocks = [orthogonal_matrix_chunk(nb_columns) for _ in range(nb_rows // nb_columns)]
    rem = nb_rows - (nb_rows // nb_columns) * nb_columns
    if rem > 0:
        blocks.append(orthogonal_matrix_chunk(nb_columns)[:rem])
    final = torch.cat(blocks)
    multiplier = torch.randn(nb_rows, nb_columns).norm(dim=1)
    return torch.diag(multiplier) @ final


def softmax_kernel(data, projection_matrix, is_query: bool, eps: float = 1e-4):
    # graphgps/layer/performer_layer.py:119-144 (normalize_data=True)
    normalizer = data.shape[-1] ** -0.25
    ratio = projection_matrix.shape[0] ** -0.5
    proj = projection_matrix.to(data.dtype)
    data_dash = torch.einsum("...id,jd->...ij", normalizer * data, proj)
    diag = (data ** 2).sum(dim=-1)
    diag = (diag / 2.0) * (normalizer ** 2)
    diag = diag.unsqueeze(-1)
    if is_query:
        return ratio * (torch.exp(data_dash - diag - data_dash.amax(dim=-1, keepdim=True)) + eps)
    return ratio * (torch.exp(data_dash - diag - data_dash.amax(dim=(-1, -2), keepdim=True)) + eps)


def linear_attention(q, k, v):
    # graphgps/layer/performer_layer.py:200-205
    k_cumsum = k.sum(dim=-2)
    d_inv = 1.0 / torch.einsum("...nd,...d->...n", q, k_cumsum)
    context = torch.einsum("...nd,...ne->...de", k, v)
    return torch.einsum("...de,...nd,...n->...ne", context, q, d_inv)


class _FastAttention(nn.Module):
    def __init__(self, dim_heads: int, nb_features: Optional[int] = None):
        super().__init__()
        nb_features = nb_features or int(dim_heads * math.log(dim_heads))  # :261
        self.register_buffer("projection_matrix",
                             gaussian_orthogonal_random_matrix(nb_features, dim_heads))  # :272-273

    def forward(self, q, k, v):
        q = softmax_kernel(q, self.projection_matrix, is_query=True)    # :313-314
        k = softmax_kernel(k, self.projection_matrix, is_query=False)   # :315
        return linear_attention(q, k, v)


class OraclePerformerSelfAttention(nn.Module):
    """``performer_pytorch.SelfAttention(dim, heads, dropout, causal=False)`` as used at
    graphgps/layer/gps_layer.py:111-114,206; arithmetic per
    graphgps/layer/performer_layer.py:421-503 (dim_head defaults to 64, no qkv bias)."""

    def __init__(self, dim, heads, dropout=0.0, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.fast_attention = _FastAttention(dim_head)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, mask=None):
        b, n, _ = x.shape
        h = self.heads
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)                       # :476
        q, k, v = (t.view(b, n, h, -1).permute(0, 2, 1, 3) for t in (q, k, v))   # :478
        if mask is not None:
            v = v.masked_fill(~mask[:, None, :, None], 0.0)                      # :485-487
        out = self.fast_attention(q, k, v)                                       # :492
        out = out.permute(0, 2, 1, 3).reshape(b, n, -1)                          # :500
        return self.dropout(self.to_out(out))                                    # :501-503


# ---------------------------------------------------------------------------
# GPS block
# ---------------------------------------------------------------------------
class OracleGraphLayerNorm(nn.Module):
    """``pygnn.norm.LayerNorm(dim_h)`` as graphgps/layer/gps_layer.py:129-134,148 builds it and :191-192,209-210,226-227 call
    it (``norm(h, batch.batch)``): PyG's graph-mode LayerNorm -- mean and (biased) variance of every graph over all of its
    nodes AND channels, then the per-channel affine.  Third-party semantics (torch_geometric is absent): restated from the
    published implementation, eps = 1e-5, weight = 1 / bias = 0 at construction."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x, batch):
        B = int(batch.max()) + 1
        cnt = torch.bincount(batch, minlength=B).to(x.dtype).clamp(min=1) * x.shape[-1]
        mean = torch.zeros(B, dtype=x.dtype).index_add_(0, batch, x.sum(-1)) / cnt
        xc = x - mean[batch].unsqueeze(-1)
        var = torch.zeros(B, dtype=x.dtype).index_add_(0, batch, (xc * xc).sum(-1)) / cnt
        return xc / (var + self.eps).sqrt()[batch].unsqueeze(-1) * self.weight + self.bias


class OracleGPSLayer(nn.Module):
    """graphgps/layer/gps_layer.py:15-257 for local in {None, GCN, CustomGatedGCN, GINE} and
    global in {None, Transformer, BiasedTransformer, Performer}, batch_norm or no norm."""

    def __init__(self, dim_h, local_gnn_type, global_model_type, num_heads, act="relu",
                 pna_degrees=None, equivstable_pe=False, dropout=0.0, attn_dropout=0.0,
                 layer_norm=False, batch_norm=True, bigbird_cfg=None, log_attn_weights=False):
        super().__init__()
        if log_attn_weights:
            raise NotImplementedError("oracle covers the BASELINE.json configurations only")
        if layer_norm and batch_norm:
            raise ValueError("Cannot apply two types of normalization together")           # :126-127
        self.layer_norm = layer_norm
        self.equivstable_pe = equivstable_pe
        self.dim_h, self.num_heads = dim_h, num_heads
        self.batch_norm = batch_norm
        self.local_gnn_type, self.global_model_type = local_gnn_type, global_model_type
        if local_gnn_type == "None":
            self.local_model = None
        elif local_gnn_type == "GCN":
            self.local_model = OracleGCNConv(dim_h, dim_h)                              # :53-55
        elif local_gnn_type == "GINE":
            gin_nn = nn.Sequential(nn.Linear(dim_h, dim_h), ACT[act](), nn.Linear(dim_h, dim_h))
            self.local_model = OracleGINEConv(gin_nn, equivstable_pe=equivstable_pe)   # :62-69
        elif local_gnn_type == "CustomGatedGCN":
            self.local_model = OracleGatedGCNLayer(dim_h, dim_h, dropout=dropout, residual=True,
                                                   act=act, equivstable_pe=equivstable_pe)  # :91-96
        else:
            raise ValueError(f"Unsupported local GNN model: {local_gnn_type}")
        if global_model_type == "None":
            self.self_attn = None
        elif global_model_type in ("Transformer", "BiasedTransformer"):
            self.self_attn = nn.MultiheadAttention(dim_h, num_heads, dropout=attn_dropout,
                                                   batch_first=True)        # :104-106
        elif global_model_type == "Performer":
            self.self_attn = OraclePerformerSelfAttention(dim_h, num_heads,
                                                          dropout=attn_dropout)  # :111-114
        else:
            raise ValueError(f"Unsupported global x-former model: {global_model_type}")
        if layer_norm:                                                                     # :129-131,148
            self.norm1_local = OracleGraphLayerNorm(dim_h)
            self.norm1_attn = OracleGraphLayerNorm(dim_h)
        if batch_norm:
            self.norm1_local = nn.BatchNorm1d(dim_h)
            self.norm1_attn = nn.BatchNorm1d(dim_h)
        self.dropout_local = nn.Dropout(dropout)
        self.dropout_attn = nn.Dropout(dropout)
        self.ff_linear1 = nn.Linear(dim_h, dim_h * 2)
        self.ff_linear2 = nn.Linear(dim_h * 2, dim_h)
        self.act_fn_ff = ACT[act]()
        if layer_norm:
            self.norm2 = OracleGraphLayerNorm(dim_h)
        if batch_norm:
            self.norm2 = nn.BatchNorm1d(dim_h)
        self.ff_dropout1 = nn.Dropout(dropout)
        self.ff_dropout2 = nn.Dropout(dropout)

    def forward(self, batch):
        h = batch.x
        h_in1 = h
        outs = []
        if self.local_model is not None:
            if self.local_gnn_type == "CustomGatedGCN":
                pe = batch.pe_EquivStableLapPE if self.equivstable_pe else None
                h_local, e = self.local_model(h, batch.edge_attr, batch.edge_index, pe)  # :164-174
                batch.edge_attr = e
            else:
                pe = batch.pe_EquivStableLapPE if self.equivstable_pe else None
                if self.local_gnn_type == "GCN":
                    h_local = self.local_model(h, batch.edge_index)                  # :183
                else:
                    h_local = self.local_model(h, batch.edge_index, batch.edge_attr, pe)  # :177-185
                h_local = self.dropout_local(h_local)
                h_local = h_in1 + h_local                                            # :188-189
            if self.layer_norm:
                h_local = self.norm1_local(h_local, batch.batch)                     # :191-192
            if self.batch_norm:
                h_local = self.norm1_local(h_local)                                  # :193-194
            outs.append(h_local)
        if self.self_attn is not None:
            nb = getattr(batch, "num_graphs", None)
            h_dense, mask = to_dense_batch(h, batch.batch, nb)                       # :199
            if self.global_model_type == "Transformer":
                h_attn = self.self_attn(h_dense, h_dense, h_dense, attn_mask=None,
                                        key_padding_mask=~mask, need_weights=False)[0][mask]  # :201,238-241
            elif self.global_model_type == "BiasedTransformer":
                h_attn = self.self_attn(h_dense, h_dense, h_dense, attn_mask=batch.attn_bias,
                                        key_padding_mask=~mask, need_weights=False)[0][mask]  # :203
            else:
                h_attn = self.self_attn(h_dense, mask=mask)[mask]                    # :206
            h_attn = self.dropout_attn(h_attn)
            h_attn = h_in1 + h_attn
            if self.layer_norm:
                h_attn = self.norm1_attn(h_attn, batch.batch)                        # :209-210
            if self.batch_norm:
                h_attn = self.norm1_attn(h_attn)
            outs.append(h_attn)
        h = sum(outs)                                                                # :222
        h = h + self.ff_dropout2(self.ff_linear2(self.ff_dropout1(self.act_fn_ff(self.ff_linear1(h)))))
        if self.layer_norm:
            h = self.norm2(h, batch.batch)                                           # :226-227
        if self.batch_norm:
            h = self.norm2(h)
        batch.x = h
        return batch


class OracleGraphormerLayer(nn.Module):
    """graphgps/layer/graphormer_layer.py:5-49: pre-LN MHA with the additive ``attn_bias`` mask,
    dropout + residual, then the pre-LN GELU MLP (hidden = embed dim) + residual."""

    def __init__(self, embed_dim, num_heads, dropout, attention_dropout, mlp_dropout):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads, attention_dropout,
                                               batch_first=True)                     # :21-24
        self.input_norm = nn.LayerNorm(embed_dim)
        self.dropout = nn.Dropout(dropout)
        self.mlp = nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, embed_dim), nn.GELU(),
                                 nn.Dropout(mlp_dropout), nn.Linear(embed_dim, embed_dim),
                                 nn.Dropout(dropout))                                # :30-37

    def forward(self, data):
        x = self.input_norm(data.x)                                                  # :40
        x, real = to_dense_batch(x, data.batch, getattr(data, "num_graphs", None))   # :41
        bias = getattr(data, "attn_bias", None)
        x = self.attention(x, x, x, ~real, attn_mask=bias)[0][real]                  # :43-46
        x = self.dropout(x) + data.x                                                 # :47
        data.x = self.mlp(x) + x                                                     # :48
        return data


class _OracleGatedGCNBatchLayer(OracleGatedGCNLayer):
    """``GatedGCNLayer.forward(batch)`` (graphgps/layer/gatedgcn_layer.py:45-88) as custom_gnn calls it."""

    def forward(self, batch):
        x, e = super().forward(batch.x, batch.edge_attr, batch.edge_index)
        batch.x, batch.edge_attr = x, e
        return batch


class _OracleGINEConvLayer(nn.Module):
    """graphgps/layer/gine_conv_layer.py:90-116 (GINEConvLayer: GINE -> relu -> dropout -> residual)."""

    def __init__(self, dim_in, dim_out, dropout, residual):
        super().__init__()
        self.dropout, self.residual = dropout, residual
        gin_nn = nn.Sequential(nn.Linear(dim_in, dim_out), nn.ReLU(), nn.Linear(dim_out, dim_out))
        self.model = OracleGINEConv(gin_nn)

    def forward(self, batch):
        x_in = batch.x
        x = self.model(batch.x, batch.edge_index, batch.edge_attr)
        x = F.dropout(F.relu(x), p=self.dropout, training=self.training)
        batch.x = x_in + x if self.residual else x
        return batch


def to_oracle_model(model: nn.Module) -> nn.Module:
    """Swap every HIP-backed layer of a ``graphgps_amd`` ``GPSModel`` / ``GraphormerModel`` (``layers``:
    ``GPSLayer`` / ``GraphormerLayer``) or
    ``CustomGNN`` (``gnn_layers``: ``GatedGCNLayer`` / ``GINEConvLayer``) for its oracle twin with
    identical ``state_dict`` keys (encoders / heads are plain PyTorch in both).  Used by the parity
    tests and bench.py's cpu_baseline leg."""
    import copy
    model = copy.deepcopy(model).cpu()
    if hasattr(model, "gnn_layers"):
        new_layers = []
        for layer in model.gnn_layers:
            if hasattr(layer, "bn_node_x"):
                o = _OracleGatedGCNBatchLayer(layer.A.in_features, layer.A.out_features, layer.dropout,
                                              layer.residual)
            else:
                o = _OracleGINEConvLayer(layer.dim_in, layer.dim_out, layer.dropout, layer.residual)
            o.load_state_dict(layer.state_dict(), strict=True)
            o.train(layer.training)
            new_layers.append(o)
        model.gnn_layers = nn.Sequential(*new_layers)
        return model
    new_layers = []
    for layer in model.layers:
        if hasattr(layer, "input_norm"):          # graphgps_amd.layer.graphormer_layer.GraphormerLayer
            o = OracleGraphormerLayer(**layer.ctor_kwargs)
        else:
            o = OracleGPSLayer(**layer.ctor_kwargs)
        missing = o.load_state_dict(layer.state_dict(), strict=True)
        o.train(layer.training)
        new_layers.append(o)
    model.layers = nn.Sequential(*new_layers)
    return model
