"""The remaining input encoders of ``configs/**``: plain torch modules run once per step before the layers
(outside the HIP scope), kept so that every GPS / Graphormer / GatedGCN / GINE config of the
reference constructs and feeds the HIP layers.  Parameter names follow the reference, so checkpoints interchange:

  * ``LapPE``: graphgps/encoder/laplace_pos_encoder.py:8-144 (``linear_x``, ``linear_A``, ``raw_norm``,
    ``pe_encoder``, ``post_mlp``)
  * ``VOCNode`` / ``VOCEdge``: encoder/voc_superpixels_encoder.py:18-43; ``PPANode`` / ``PPAEdge``:
    encoder/ppa_encoder.py:6-29; ``LinearEdge``: encoder/linear_edge_encoder.py:6-19; ``DummyEdge``:
    encoder/dummy_edge_encoder.py:5-17 (all ``encoder.*``)
  * the ``X+LapPE``, ``X+LapPE+RWSE``, ``X+GraphormerBias+LapPE`` and ``LinearNode|VOCNode|PPANode+PE``
    compositions of encoder/composed_encoders.py:104-155 (``encoder1`` / ``encoder2`` / ``encoder3``)
"""
import torch
import torch.nn as nn

from ..graphgym.config import cfg
from ..graphgym.register import register_edge_encoder, register_node_encoder
from . import encoders as _enc
from . import graphormer_encoder as _gph


def _relu_mlp(sizes, final_act=True):
    """Linear(sizes[0], sizes[1]) -> ReLU -> ... ; a ReLU after the last Linear iff ``final_act``."""
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2 or final_act:
            mods.append(nn.ReLU())
    return mods


@register_node_encoder('LapPE', overwrite=True)
class LapPENodeEncoder(nn.Module):
    """Per node: the k (eigenvector entry, eigenvalue) pairs -> DeepSet or Transformer over the k frequencies
    -> masked sum -> optional MLP -> appended to the node features."""

    def __init__(self, dim_emb, expand_x=True):
        super().__init__()
        pecfg = cfg.posenc_LapPE
        dim_pe, n_layers, post = pecfg.dim_pe, pecfg.layers, pecfg.post_layers
        if pecfg.model not in ('Transformer', 'DeepSet'):
            raise ValueError(f"Unexpected PE model {pecfg.model}")
        self.model_type = pecfg.model
        self.pass_as_var = pecfg.pass_as_var
        if dim_emb - dim_pe < 0:
            raise ValueError(f"LapPE size {dim_pe} is too large for "
                             f"desired embedding size of {dim_emb}.")
        self.expand_x = expand_x and dim_emb - dim_pe > 0
        if self.expand_x:
            self.linear_x = nn.Linear(cfg.share.dim_in, dim_emb - dim_pe)
        wide = self.model_type == 'DeepSet' and n_layers > 1
        self.linear_A = nn.Linear(2, 2 * dim_pe if wide else dim_pe)
        self.raw_norm = (nn.BatchNorm1d(pecfg.eigen.max_freqs)
                         if pecfg.raw_norm_type.lower() == 'batchnorm' else None)
        if self.model_type == 'Transformer':
            layer = nn.TransformerEncoderLayer(d_model=dim_pe, nhead=pecfg.n_heads, batch_first=True)
            self.pe_encoder = nn.TransformerEncoder(layer, num_layers=n_layers)
        elif n_layers == 1:
            self.pe_encoder = nn.Sequential(nn.ReLU())
        else:   # ReLU, (n_layers - 2) x [Linear(2p, 2p), ReLU], Linear(2p, p), ReLU
            self.pe_encoder = nn.Sequential(
                nn.ReLU(), *_relu_mlp([2 * dim_pe] * (n_layers - 1) + [dim_pe]))
        self.post_mlp = None
        if post == 1:
            self.post_mlp = nn.Sequential(*_relu_mlp([dim_pe, dim_pe]))
        elif post > 1:
            self.post_mlp = nn.Sequential(*_relu_mlp([dim_pe] + [2 * dim_pe] * (post - 1) + [dim_pe]))

    def forward(self, batch):
        if not (hasattr(batch, 'EigVals') and hasattr(batch, 'EigVecs')):
            raise ValueError("Precomputed eigen values and vectors are "
                             f"required for {self.__class__.__name__}; "
                             "set config 'posenc_LapPE.enable' to True")
        vecs = batch.EigVecs
        if self.training:       # random sign per frequency (the eigenvectors' sign is arbitrary)
            flip = torch.rand(vecs.size(1), device=vecs.device)
            vecs = vecs * torch.where(flip >= 0.5, 1.0, -1.0).to(vecs.dtype).unsqueeze(0)
        pe = torch.cat((vecs.unsqueeze(2), batch.EigVals), dim=2)            # [N, k, 2]
        empty = torch.isnan(pe)
        pe = torch.where(empty, torch.zeros_like(pe), pe)
        if self.raw_norm:
            pe = self.raw_norm(pe)
        pe = self.linear_A(pe)                                                # [N, k, dim_pe or 2 dim_pe]
        if self.model_type == 'Transformer':
            pe = self.pe_encoder(src=pe, src_key_padding_mask=empty[:, :, 0])
        else:
            pe = self.pe_encoder(pe)
        pe = pe.masked_fill(empty[:, :, 0].unsqueeze(2), 0.0).sum(1)          # [N, dim_pe]
        if self.post_mlp is not None:
            pe = self.post_mlp(pe)
        h = self.linear_x(batch.x) if self.expand_x else batch.x
        batch.x = torch.cat((h, pe), 1)
        if self.pass_as_var:
            batch.pe_LapPE = pe
        return batch


class _BatchLinear(nn.Module):
    """``encoder = Linear(dim_in, emb_dim)`` applied to one batch attribute."""
    attr, dim_in = 'x', None

    def __init__(self, emb_dim):
        super().__init__()
        self.encoder = nn.Linear(self._dim_in(), emb_dim)

    def _dim_in(self):
        return self.dim_in

    def forward(self, batch):
        setattr(batch, self.attr, self.encoder(self._input(batch)))
        return batch

    def _input(self, batch):
        return getattr(batch, self.attr)


@register_node_encoder('VOCNode', overwrite=True)
class VOCNodeEncoder(_BatchLinear):
    attr, dim_in = 'x', 14                   # superpixel node features


@register_edge_encoder('VOCEdge', overwrite=True)
class VOCEdgeEncoder(_BatchLinear):
    attr = 'edge_attr'

    def _dim_in(self):
        return 2 if cfg.dataset.name == 'edge_wt_region_boundary' else 1


@register_edge_encoder('PPAEdge', overwrite=True)
class PPAEdgeEncoder(_BatchLinear):
    attr, dim_in = 'edge_attr', 7


@register_edge_encoder('LinearEdge', overwrite=True)
class LinearEdgeEncoder(_BatchLinear):
    attr = 'edge_attr'

    def _dim_in(self):
        if cfg.dataset.name in ('MNIST', 'CIFAR10'):
            self.in_dim = 1
            return 1
        raise ValueError("Input edge feature dim is required to be hardset "
                         "or refactored to use a cfg option.")

    def _input(self, batch):
        return batch.edge_attr.view(-1, self.in_dim)


@register_node_encoder('PPANode', overwrite=True)
class PPANodeEncoder(nn.Module):
    """One shared embedding row: PPA graphs carry no node features."""

    def __init__(self, emb_dim):
        super().__init__()
        self.encoder = nn.Embedding(1, emb_dim)

    def forward(self, batch):
        batch.x = self.encoder(batch.x)
        return batch


@register_edge_encoder('DummyEdge', overwrite=True)
class DummyEdgeEncoder(nn.Module):
    """One shared embedding row for every edge of a dataset without edge features."""

    def __init__(self, emb_dim):
        super().__init__()
        self.encoder = nn.Embedding(num_embeddings=1, embedding_dim=emb_dim)

    def forward(self, batch):
        batch.edge_attr = self.encoder(batch.edge_index.new_zeros(batch.edge_index.shape[1]))
        return batch


# ---- compositions (composed_encoders.py:104-155) -------------------------------------------------
_DS = {'Atom': _enc.AtomEncoder, 'ASTNode': _enc.ASTNodeEncoder, 'PPANode': PPANodeEncoder,
       'TypeDictNode': _enc.TypeDictNodeEncoder, 'VOCNode': VOCNodeEncoder,
       'LinearNode': _gph.LinearNodeEncoder}
_PE = {'LapPE': LapPENodeEncoder, 'RWSE': _enc.RWSENodeEncoder, 'HKdiagSE': _enc.HKdiagSENodeEncoder,
       'ElstaticSE': _enc.ElstaticSENodeEncoder, 'EquivStableLapPE': _enc.EquivStableLapPENodeEncoder,
       'GraphormerBias': _gph.GraphormerEncoder}


def _compose2(ds_cls, pe_cls, pe_name):
    if pe_name == 'GraphormerBias':
        return _gph._compose([ds_cls, pe_cls], [pe_name])
    return _enc.concat_node_encoders(ds_cls, pe_cls, pe_name)


for _ds_name, _ds_cls in _DS.items():
    for _pe_name, _pe_cls in _PE.items():
        register_node_encoder(f"{_ds_name}+{_pe_name}", _compose2(_ds_cls, _pe_cls, _pe_name), overwrite=True)
    register_node_encoder(f"{_ds_name}+LapPE+RWSE",
                          _gph._compose([_ds_cls, LapPENodeEncoder, _enc.RWSENodeEncoder], ['LapPE', 'RWSE']),
                          overwrite=True)
    register_node_encoder(f"{_ds_name}+GraphormerBias+LapPE",
                          _gph._compose([_ds_cls, _gph.GraphormerEncoder, LapPENodeEncoder],
                                        ['GraphormerBias', 'LapPE']), overwrite=True)
    register_node_encoder(f"{_ds_name}+GraphormerBias+RWSE",
                          _gph._compose([_ds_cls, _gph.GraphormerEncoder, _enc.RWSENodeEncoder],
                                        ['GraphormerBias', 'RWSE']), overwrite=True)
