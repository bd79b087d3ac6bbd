"""Input encoders used by the measured configs (run once per step, before the layers; plain
PyTorch embedding lookups -- SURVEY.md section 2a marks them out of the HIP scope).

Mirrors, with identical parameter names:
  * ``Atom`` / ``Bond``: GraphGym ``AtomEncoder`` / ``BondEncoder`` (PyG 2.2
    graphgym/models/encoder.py, third-party): one xavier-initialised ``nn.Embedding`` per OGB
    feature column (``atom_embedding_list`` / ``bond_embedding_list``), summed.
  * ``TypeDictNode`` / ``TypeDictEdge``: /root/reference/graphgps/encoder/type_dict_encoder.py:81-116
  * ``ASTNode`` / ``ASTEdge``: graphgps/encoder/ast_encoder.py:34-85
  * ``RWSE`` (+ ``HKdiagSE``, ``ElstaticSE``): graphgps/encoder/kernel_pos_encoder.py:8-110
  * ``X+RWSE`` compositions: graphgps/encoder/composed_encoders.py:19-58
  * ``BatchNorm1dNode``: GraphGym layer used for ``dataset.{node,edge}_encoder_bn``
    (graphgps/network/gps_model.py:27-46) -- note the reference applies it to ``batch.x`` for the
    edge case as well; kept.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graphgym.config import cfg
from ..graphgym.register import (node_encoder_dict, register_edge_encoder,
                                 register_node_encoder)
from ..synthetic import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS


class _MultihotMM(torch.autograd.Function):
    """``multihot @ table`` whose weight gradient ``multihot^T @ g`` -- a [V, R] x [R, emb] product with R = every node /
    edge of the batch and a handful of output rows -- runs on the split-K weight-gradient kernel (csrc/wgrad.hip:
    deterministic, fp32-exact products) instead of a library GEMM that leaves most of the chip idle on this shape."""

    @staticmethod
    def forward(ctx, multihot, table):
        ctx.save_for_backward(multihot)
        return multihot @ table

    @staticmethod
    def backward(ctx, g):
        from ..fused import _param_grads
        (multihot,) = ctx.saved_tensors
        g_w, _ = _param_grads(multihot, g.contiguous(), True, False)
        return None, g_w


# The gather-sum kernel CLAMPS out-of-range feature values (memory safety); ``nn.Embedding`` -- what the reference runs --
# raises on them.  So that a corrupt or unsupported feature value (a new atom type ...) cannot train silently against the
# wrong table row, the FIRST batch every table set sees is range-checked (one device read, never inside a capture);
# GPS_EMBED_CHECK=all checks every eager batch, =0 none.
_EMBED_CHECK = os.environ.get("GPS_EMBED_CHECK", "first")
_range_checked = set()


def _check_feature_range(feats, tables) -> None:
    if _EMBED_CHECK == "0" or torch.cuda.is_current_stream_capturing():
        return
    key = (tables[0].data_ptr(), len(tables))
    if _EMBED_CHECK != "all" and key in _range_checked:
        return
    vocab = torch.tensor([t.shape[0] for t in tables], device=feats.device)
    bad = ((feats.amin(0) < 0) | (feats.amax(0) >= vocab)).nonzero().flatten().tolist()
    if bad:
        raise IndexError(f"embedding index out of range in feature column(s) {bad}: values must lie in [0, vocabulary) = "
                         f"{[int(t.shape[0]) for t in tables]} (nn.Embedding raises here too)")
    _range_checked.add(key)


class _EmbedSum(torch.autograd.Function):
    """``sum_i tables[i][feats[:, i]]`` in ONE launch (csrc/embed.hip ``gps_embed_sum``: the tables read in place, columns
    added in index order -- the reference's own summation order, ogb's ``x_embedding += emb[i](x[:, i])``).  Backward: ONE
    launch writes the multi-hot matrix of the features (``gps_multihot_fill``) and the stacked table gradient is the GEMM
    ``multihot^T @ g`` -- deterministic, what the multi-hot form of rounds 1-4 was built for -- handed back as row slices.
    Replaces zeros + index add + ``scatter_`` + pad + ``cat`` + GEMM (6 launches, and ~0.35 ms of idle time per replayed
    pcqm4m step in front of the ATen index kernels of the edge encoder: DESIGN section 5)."""

    @staticmethod
    def forward(ctx, feats, *tables):
        import ctypes
        from .. import lib as _lib
        L = _lib.load()
        _check_feature_range(feats, tables)
        R, k = feats.shape
        emb = tables[0].shape[1]
        out = torch.empty(R, emb, dtype=torch.float32, device=feats.device)
        ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tables])
        vocab = (ctypes.c_int * k)(*[t.shape[0] for t in tables])
        _lib.check(L.gps_embed_sum(_lib.ptr(feats), feats.stride(0), R, k, ptrs, vocab, emb, _lib.ptr(out),
                                   _lib.current_stream(feats.device)), "gps_embed_sum")
        ctx.save_for_backward(feats)
        ctx.vocab = [t.shape[0] for t in tables]
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from .. import lib as _lib
        L = _lib.load()
        (feats,) = ctx.saved_tensors
        R, k = feats.shape
        vocab = (ctypes.c_int * k)(*ctx.vocab)
        vpad = L.gps_multihot_columns(k, vocab)
        multihot = torch.empty(R, vpad, dtype=torch.float32, device=g.device)
        _lib.check(L.gps_multihot_fill(_lib.ptr(feats), feats.stride(0), R, k, vocab, _lib.ptr(multihot),
                                       _lib.current_stream(g.device)), "gps_multihot_fill")
        g = g.contiguous()
        if _MULTIHOT_WGRAD and R >= 256:
            from ..fused import _param_grads
            g_w, _ = _param_grads(multihot, g, True, False)
        else:
            g_w = multihot.t() @ g
        outs, off = [], 0
        for v in ctx.vocab:
            outs.append(g_w[off:off + v])
            off += v
        return (None, *outs)


_EMBED_SUM = os.environ.get("GPS_EMBED_SUM", "1") != "0"      # 0: the multi-hot GEMM forward of rounds 1-4 (A/B)


def _embed_sum_ok(feats, embs) -> bool:
    w = embs[0].weight
    return (_EMBED_SUM and feats.is_cuda and feats.dtype == torch.int64 and feats.dim() == 2 and feats.stride(1) == 1
            and 1 <= len(embs) <= 16 and feats.shape[1] == len(embs) and feats.shape[0] > 0
            and all(e.weight.dtype == torch.float32 and e.weight.is_contiguous() and e.weight.data_ptr() % 16 == 0
                    and e.weight.shape[1] == w.shape[1] and e.padding_idx is None and e.max_norm is None for e in embs)
            and w.shape[1] % 4 == 0)


# The table gradients of the sum-of-embeddings encoders (multihot^T g) and the PE encoder's Linear: [V, emb] / [dim_pe, steps]
# results contracted over every node / edge of the batch.  On the split-K weight-gradient kernel (csrc/wgrad.hip) by default
# since round 6: the library picked 71 / 57 / 43 us kernels for them (profiles/r05_kernel_trace_stats_pcqm4m.txt:
# MT32x16x256, MT32x32x256, MT16x32x512).  GPS_MULTIHOT_WGRAD=0: the library GEMMs (A/B).
_MULTIHOT_WGRAD = os.environ.get("GPS_MULTIHOT_WGRAD", "1") != "0"


def _multihot_embedding(feats, embs, owner):
    """Sum over the columns of ``feats`` [R, k] of ``embs[i](feats[:, i])`` as ONE [R, sum(vocab)] x [sum(vocab), emb]
    GEMM over a multi-hot matrix (small vocabularies only).  The point is the backward: grad_W = multihot^T @ g is a
    deterministic GEMM instead of k sort-based ``embedding_dense_backward`` pipelines.  The vocabulary axis is padded to
    a multiple of 4 (zero columns / zero table rows) so that the weight-gradient kernel takes it."""
    if _embed_sum_ok(feats, embs):           # round 5: one gather-sum launch (csrc/embed.hip); the multi-hot matrix only
        return _EmbedSum.apply(feats, *[e.weight for e in embs])      # exists in the backward, for the table-gradient GEMM
    offs = getattr(owner, "_offsets", None)
    if offs is None or offs.device != feats.device:
        sizes = [e.num_embeddings for e in embs]
        offs = torch.tensor([sum(sizes[:i]) for i in range(len(sizes))], device=feats.device)
        owner._offsets, owner._vocab = offs, sum(sizes)
    vocab = owner._vocab
    vpad = -(-vocab // 4) * 4
    w = embs[0].weight
    multihot = torch.zeros(feats.shape[0], vpad, dtype=w.dtype, device=feats.device)
    multihot.scatter_(1, feats + offs, 1.0)
    tables = [e.weight for e in embs]
    if vpad != vocab:
        tables.append(w.new_zeros(vpad - vocab, w.shape[1]))
    table = torch.cat(tables, dim=0)
    if (_MULTIHOT_WGRAD and feats.is_cuda and torch.is_grad_enabled() and w.dtype == torch.float32
            and w.shape[1] % 4 == 0 and feats.shape[0] >= 256):
        return _MultihotMM.apply(multihot, table)
    return multihot @ table


def masked_batch_norm(bn: nn.BatchNorm1d, x: torch.Tensor, n_real: torch.Tensor) -> torch.Tensor:
    """``bn(x)`` of a PADDED batch in training mode: batch statistics over the first ``n_real[0]`` rows only (a device
    int32 word -- ``batch.gps_counts[0:1]`` of loader.BucketPadding -- so the arithmetic has static shapes and can sit in a
    captured step), every row normalised with them, running statistics updated as ``nn.BatchNorm1d`` does (unbiased
    variance, ``momentum`` or the cumulative average).  On an un-padded batch (n_real = rows) it equals ``bn(x)``."""
    R = x.shape[0]
    w = (torch.arange(R, device=x.device) < n_real).to(x.dtype).unsqueeze(1)         # [R, 1]: 1 on real rows
    n = n_real.to(x.dtype)
    mean = (x * w).sum(0) / n
    xc = (x - mean) * w
    var = (xc * xc).sum(0) / n
    y = (x - mean) * torch.rsqrt(var + bn.eps)
    if bn.affine:
        y = y * bn.weight + bn.bias
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked.add_(1)
            if bn.momentum is None:
                m = 1.0 / bn.num_batches_tracked.to(x.dtype)
            else:
                m = bn.momentum
            bn.running_mean.mul_(1.0 - m).add_(mean.detach() * m)
            bn.running_var.mul_(1.0 - m).add_(var.detach() * (n / (n - 1.0).clamp(min=1.0)) * m)
    return y


def _batch_norm(bn: nn.BatchNorm1d, x: torch.Tensor, batch) -> torch.Tensor:
    """``bn(x)`` over the node rows of ``batch``; on a padded batch in training mode the statistics skip the padding."""
    counts = getattr(batch, "gps_counts", None)
    if torch.is_tensor(counts) and bn.training:
        return masked_batch_norm(bn, x, counts[0:1])
    return bn(x)


class _OGBFeatureEncoder(nn.Module):
    _attr = None
    _dims = None

    def __init__(self, emb_dim, *args, **kwargs):
        super().__init__()
        embs = nn.ModuleList()
        for dim in self._dims:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            embs.append(emb)
        setattr(self, self._attr, embs)

    def _encode(self, feats):
        embs = getattr(self, self._attr)
        if feats.is_cuda and feats.shape[1] == len(embs):
            return self._encode_onehot(feats, embs)
        out = 0
        for i in range(feats.shape[1]):
            out = out + embs[i](feats[:, i])
        return out

    def _encode_onehot(self, feats, embs):
        """Same sum of per-column embedding rows, as ONE [R, sum(vocab)] x [sum(vocab), emb] GEMM
        over a multi-hot matrix (vocabularies are tiny: 173 atom / 13 bond entries).  The point
        is the backward: grad_W = multihot^T @ g is a deterministic GEMM instead of 9 (3)
        sort-based ``embedding_dense_backward`` pipelines (~1.3 ms per step on MI355X)."""
        return _multihot_embedding(feats, embs, self)


@register_node_encoder('Atom', overwrite=True)
class AtomEncoder(_OGBFeatureEncoder):
    _attr, _dims = 'atom_embedding_list', ATOM_FEATURE_DIMS

    def forward(self, batch):
        batch.x = self._encode(batch.x)
        return batch


@register_edge_encoder('Bond', overwrite=True)
class BondEncoder(_OGBFeatureEncoder):
    _attr, _dims = 'bond_embedding_list', BOND_FEATURE_DIMS

    def forward(self, batch):
        batch.edge_attr = self._encode(batch.edge_attr)
        return batch


@register_node_encoder('TypeDictNode', overwrite=True)
class TypeDictNodeEncoder(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        num_types = cfg.dataset.node_encoder_num_types
        if num_types < 1:
            raise ValueError(f"Invalid 'node_encoder_num_types': {num_types}")
        self.encoder = nn.Embedding(num_embeddings=num_types, embedding_dim=emb_dim)

    def forward(self, batch):
        # (nn.Embedding on purpose: the gather-sum / multi-hot path of the sum-of-embeddings encoders was tried here in round 6
        # -- zinc step 2.55 -> 2.52 ms -- and withdrawn: its table gradient is a library GEMM for batches under 256 rows, and
        # captured steps that held one died in hipGraph replay on about half of the boxes, profiles/r06_ab_step.txt)
        batch.x = self.encoder(batch.x[:, 0])  # only the first column
        return batch


@register_edge_encoder('TypeDictEdge', overwrite=True)
class TypeDictEdgeEncoder(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        num_types = cfg.dataset.edge_encoder_num_types
        if num_types < 1:
            raise ValueError(f"Invalid 'edge_encoder_num_types': {num_types}")
        self.encoder = nn.Embedding(num_embeddings=num_types, embedding_dim=emb_dim)

    def forward(self, batch):
        batch.edge_attr = self.encoder(batch.edge_attr)
        return batch


@register_node_encoder('ASTNode', overwrite=True)
class ASTNodeEncoder(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.max_depth = 20
        self.type_encoder = nn.Embedding(98, emb_dim)
        self.attribute_encoder = nn.Embedding(10030, emb_dim)
        self.depth_encoder = nn.Embedding(self.max_depth + 1, emb_dim)

    def forward(self, batch):
        x = batch.x
        depth = batch.node_depth.view(-1).clamp(max=self.max_depth)
        if x.is_cuda and torch.is_grad_enabled():
            # the two small tables (98 node types, 21 depths) as one multi-hot GEMM, the 10,030-entry attribute table
            # through the lookup whose weight gradient is the deterministic segmented sum (ops.embedding): no ATen
            # sort + sum_and_scatter (2 ms of the code2 step in round 2) on the path
            from ..ops import embedding
            small = _multihot_embedding(torch.stack([x[:, 0], depth], dim=1),
                                        [self.type_encoder, self.depth_encoder], self)
            batch.x = small + embedding(x[:, 1], self.attribute_encoder.weight)
            return batch
        batch.x = (self.type_encoder(x[:, 0]) + self.attribute_encoder(x[:, 1])
                   + self.depth_encoder(depth))
        return batch


@register_edge_encoder('ASTEdge', overwrite=True)
class ASTEdgeEncoder(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.embedding_type = nn.Embedding(2, emb_dim)
        self.embedding_direction = nn.Embedding(2, emb_dim)

    def forward(self, batch):
        ea = batch.edge_attr
        if ea.is_cuda and torch.is_grad_enabled():
            batch.edge_attr = _multihot_embedding(ea[:, :2], [self.embedding_type, self.embedding_direction], self)
            return batch
        batch.edge_attr = (self.embedding_type(ea[:, 0]) + self.embedding_direction(ea[:, 1]))
        return batch


class _LinearSplitKGrad(torch.autograd.Function):
    """``F.linear`` whose weight + bias gradient -- a [dim_pe, steps] result contracted over every node of the batch -- runs on
    the split-K kernel (csrc/wgrad.hip: exact fp32 products, bias gradient in the same pass) instead of a library GEMM
    (43 us: MT16x32x512 on a K = 7,569 contraction) + a column-sum reduction (28 us).  Forward and input gradient stay the
    library's plain fp32 products: at K = 16 .. 20 there is no accumulation to hide the fp16-form ring GEMM's 2^-22
    operand rounding behind (zinc model test: 2e-6 on the raw-norm gradient against 6e-9 for the CPU oracle)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        from ..fused import _param_grads
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        g_w, g_b = _param_grads(g, x.contiguous(), ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.params,
                                targets=ctx.params)
        return (g.mm(weight) if ctx.needs_input_grad[0] else None), g_w, g_b


class KernelPENodeEncoder(nn.Module):
    """Kernel-statistics PE encoder (RWSE etc.): raw-norm -> Linear/MLP -> concat to x."""
    kernel_type = None

    def __init__(self, dim_emb, expand_x=True):
        super().__init__()
        if self.kernel_type is None:
            raise ValueError(f"{self.__class__.__name__} has to be preconfigured by setting "
                             f"'kernel_type' class variable before calling the constructor.")
        dim_in = cfg.share.dim_in
        pecfg = getattr(cfg, f"posenc_{self.kernel_type}")
        dim_pe = pecfg.dim_pe
        num_rw_steps = len(pecfg.kernel.times)
        model_type = pecfg.model.lower()
        n_layers = pecfg.layers
        norm_type = pecfg.raw_norm_type.lower()
        self.pass_as_var = pecfg.pass_as_var
        if dim_emb - dim_pe < 0:
            raise ValueError(f"PE dim size {dim_pe} is too large for desired embedding size of "
                             f"{dim_emb}.")
        if expand_x and dim_emb - dim_pe > 0:
            self.linear_x = nn.Linear(dim_in, dim_emb - dim_pe)
        self.expand_x = expand_x and dim_emb - dim_pe > 0
        self.raw_norm = nn.BatchNorm1d(num_rw_steps) if norm_type == 'batchnorm' else None
        if model_type == 'mlp':
            layers = []
            if n_layers == 1:
                layers += [nn.Linear(num_rw_steps, dim_pe), nn.ReLU()]
            else:
                layers += [nn.Linear(num_rw_steps, 2 * dim_pe), nn.ReLU()]
                for _ in range(n_layers - 2):
                    layers += [nn.Linear(2 * dim_pe, 2 * dim_pe), nn.ReLU()]
                layers += [nn.Linear(2 * dim_pe, dim_pe), nn.ReLU()]
            self.pe_encoder = nn.Sequential(*layers)
        elif model_type == 'linear':
            self.pe_encoder = nn.Linear(num_rw_steps, dim_pe)
        else:
            raise ValueError(f"{self.__class__.__name__}: Does not support '{model_type}' "
                             f"encoder model.")

    def forward(self, batch):
        pestat_var = f"pestat_{self.kernel_type}"
        if not hasattr(batch, pestat_var):
            raise ValueError(f"Precomputed '{pestat_var}' variable is required for "
                             f"{self.__class__.__name__}; set config "
                             f"'posenc_{self.kernel_type}.enable' to True, and also set "
                             f"'posenc.kernel.times' values")
        pos_enc = getattr(batch, pestat_var)
        if self.raw_norm:
            pos_enc = _batch_norm(self.raw_norm, pos_enc, batch)
        if (_MULTIHOT_WGRAD and isinstance(self.pe_encoder, nn.Linear) and pos_enc.is_cuda and torch.is_grad_enabled()
                and pos_enc.shape[0] >= 256 and self.pe_encoder.bias is not None):
            pos_enc = _LinearSplitKGrad.apply(pos_enc, self.pe_encoder.weight, self.pe_encoder.bias)
        else:
            pos_enc = self.pe_encoder(pos_enc)
        h = self.linear_x(batch.x) if self.expand_x else batch.x
        batch.x = torch.cat((h, pos_enc), 1)
        if self.pass_as_var:
            setattr(batch, f'pe_{self.kernel_type}', pos_enc)
        return batch


def _kernel_encoder(name):
    cls = type(f"{name}NodeEncoder", (KernelPENodeEncoder,), {"kernel_type": name})
    register_node_encoder(name, cls, overwrite=True)
    return cls


RWSENodeEncoder = _kernel_encoder('RWSE')
HKdiagSENodeEncoder = _kernel_encoder('HKdiagSE')
ElstaticSENodeEncoder = _kernel_encoder('ElstaticSE')


@register_node_encoder('EquivStableLapPE', overwrite=True)
class EquivStableLapPENodeEncoder(nn.Module):
    """graphgps/encoder/equivstable_laplace_pos_encoder.py:8-50: k-dim node LapPE -> d-dim, kept apart from
    ``batch.x`` in ``batch.pe_EquivStableLapPE`` (used by the local GNN as an edge gate)."""

    def __init__(self, dim_emb):
        super().__init__()
        pecfg = cfg.posenc_EquivStableLapPE
        max_freqs = pecfg.eigen.max_freqs
        self.raw_norm = nn.BatchNorm1d(max_freqs) if pecfg.raw_norm_type.lower() == 'batchnorm' else None
        self.linear_encoder_eigenvec = nn.Linear(max_freqs, dim_emb)

    def forward(self, batch):
        if not (hasattr(batch, 'EigVals') and hasattr(batch, 'EigVecs')):
            raise ValueError("Precomputed eigen values and vectors are "
                             f"required for {self.__class__.__name__}; set "
                             f"config 'posenc_EquivStableLapPE.enable' to True")
        pos_enc = batch.EigVecs
        pos_enc = torch.where(torch.isnan(pos_enc), torch.zeros_like(pos_enc), pos_enc)
        if self.raw_norm:
            pos_enc = _batch_norm(self.raw_norm, pos_enc, batch)
        batch.pe_EquivStableLapPE = self.linear_encoder_eigenvec(pos_enc)
        return batch


def concat_node_encoders(enc1_cls, enc2_cls, enc2_name):
    """Dataset encoder to (dim_emb - dim_pe) then PE encoder appending dim_pe
    (composed_encoders.py:36-58, non-EquivStable branch)."""

    class Concat2NodeEncoder(nn.Module):
        def __init__(self, dim_emb):
            super().__init__()
            if cfg.posenc_EquivStableLapPE.enable:   # node feats and PE are not concatenated (:45-47)
                self.encoder1 = enc1_cls(dim_emb)
                self.encoder2 = enc2_cls(dim_emb)
            else:
                enc2_dim_pe = getattr(cfg, f"posenc_{enc2_name}").dim_pe
                self.encoder1 = enc1_cls(dim_emb - enc2_dim_pe)
                self.encoder2 = enc2_cls(dim_emb, expand_x=False)

        def forward(self, batch):
            return self.encoder2(self.encoder1(batch))

    Concat2NodeEncoder.__name__ = f"{enc1_cls.__name__}+{enc2_cls.__name__}"
    return Concat2NodeEncoder


for _ds_name, _ds_cls in (('Atom', AtomEncoder), ('ASTNode', ASTNodeEncoder),
                          ('TypeDictNode', TypeDictNodeEncoder)):
    for _pe_name, _pe_cls in (('RWSE', RWSENodeEncoder), ('HKdiagSE', HKdiagSENodeEncoder),
                              ('ElstaticSE', ElstaticSENodeEncoder),
                              ('EquivStableLapPE', EquivStableLapPENodeEncoder)):
        register_node_encoder(f"{_ds_name}+{_pe_name}",
                              concat_node_encoders(_ds_cls, _pe_cls, _pe_name), overwrite=True)


class BatchNorm1dNode(nn.Module):
    """GraphGym ``BatchNorm1dNode``: BN over ``batch.x`` (parameter prefix ``bn``)."""

    def __init__(self, dim_in, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(dim_in, eps=eps, momentum=momentum)

    def forward(self, batch):
        batch.x = _batch_norm(self.bn, batch.x, batch)
        return batch
