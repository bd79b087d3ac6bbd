"""SignNet positional encoder: ``rho([phi(v_k) + phi(-v_k)]_k)`` with ``phi`` a GIN over the graph's edges.

Mirror of ``/root/reference/graphgps/encoder/signnet_pos_encoder.py`` (parameter names ``linear_x``,
``sign_inv_net.enc.layers.<i>.nn.{lins,bns}.*`` + buffer ``eps``, ``sign_inv_net.enc.bns.<i>.*``,
``sign_inv_net.rho.{lins,bns}.*``), so checkpoints interchange.  Runs once per step before the layers.
The reference carries the eigenvector features as ``[k, N, C]`` and lets PyG's ``GINConv`` gather / scatter-add
over the node axis (:70-110,126-131); here they stay ``[N, k, C]`` and one GIN aggregation of all k*C columns is
a single CSR segment sum (``ops.gin_aggregate`` -> csrc/gcn.hip:k_adj_sum, deterministic) on GPU tensors; the
BatchNorms see the same ``k*N`` rows per channel in a different order, which does not change their statistics.
On CPU tensors (the oracle's side of the parity tests) the aggregation is an ``index_add_``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graphgym.config import cfg
from ..graphgym.register import register_node_encoder
from ..ops import gin_aggregate, graph_index_of
from . import encoders as _enc
from . import graphormer_encoder as _gph

_ACT = {'relu': nn.ReLU, 'elu': nn.ELU, 'tanh': nn.Tanh}


class MLP(nn.Module):
    """lins[0] -> act -> [bn] -> [ln] -> [+res] -> dropout -> ... -> lins[-1] [+res]  (reference :14-67)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, use_bn=False, use_ln=False,
                 dropout=0.5, activation='relu', residual=False):
        super().__init__()
        sizes = ([in_channels, out_channels] if num_layers == 1 else
                 [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels])
        self.lins = nn.ModuleList(nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:]))
        if use_bn:
            self.bns = nn.ModuleList(nn.BatchNorm1d(h) for h in sizes[1:-1])
        if use_ln:
            self.lns = nn.ModuleList(nn.LayerNorm(h) for h in sizes[1:-1])
        if activation not in _ACT:
            raise ValueError('Invalid activation')
        self.activation = _ACT[activation]()
        self.use_bn, self.use_ln, self.dropout, self.residual = use_bn, use_ln, dropout, residual

    def forward(self, x):
        prev = x
        for i, lin in enumerate(self.lins[:-1]):
            x = self.activation(lin(x))
            if self.use_bn:      # per channel over every leading position (2-D and 3-D inputs alike)
                x = self.bns[i](x.reshape(-1, x.shape[-1])).reshape(x.shape)
            if self.use_ln:
                x = self.lns[i](x)
            if self.residual and prev.shape == x.shape:
                x = x + prev
            x = F.dropout(x, p=self.dropout, training=self.training)
            prev = x
        x = self.lins[-1](x)
        return x + prev if self.residual and prev.shape == x.shape else x


class GINConv(nn.Module):
    """``nn((1 + eps) x_i + sum_{j->i} x_j)`` on node-major ``[N, ..., C]`` features (PyG GINConv, eps = 0)."""

    def __init__(self, net, eps: float = 0.0):
        super().__init__()
        self.nn = net
        self.register_buffer('eps', torch.tensor([float(eps)]))
        self._eps = float(eps)

    def forward(self, x, batch):
        flat = x.reshape(x.shape[0], -1)
        if flat.is_cuda:
            agg = gin_aggregate(flat, graph_index_of(batch), self._eps)
        else:
            src, dst = batch.edge_index[0], batch.edge_index[1]
            agg = ((1.0 + self._eps) * flat).index_add(0, dst, flat.index_select(0, src))
        return self.nn(agg.reshape(x.shape))


class GIN(nn.Module):
    """n_layers GINConv(MLP, 2 layers); dropout + BatchNorm between them (reference :70-110)."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_layers, use_bn=True, dropout=0.5,
                 activation='relu'):
        super().__init__()
        dims = [in_channels] + [hidden_channels] * (n_layers - 1)
        outs = [hidden_channels] * (n_layers - 1) + [out_channels]
        self.layers = nn.ModuleList(
            GINConv(MLP(a, hidden_channels, b, 2, use_bn=use_bn, dropout=dropout, activation=activation))
            for a, b in zip(dims, outs))
        if use_bn:
            self.bns = nn.ModuleList(nn.BatchNorm1d(hidden_channels) for _ in range(n_layers - 1))
        self.use_bn = use_bn
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, x, batch):
        for i, layer in enumerate(self.layers):
            if i != 0:
                x = self.dropout(x)
                if self.use_bn:
                    x = self.bns[i - 1](x.reshape(-1, x.shape[-1])).reshape(x.shape)
            x = layer(x, batch)
        return x


class GINDeepSigns(nn.Module):
    """Fixed k: rho = MLP over the concatenated k * out channels (reference :113-131)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, k, dim_pe, rho_num_layers,
                 use_bn=False, use_ln=False, dropout=0.5, activation='relu'):
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)
        self.rho = MLP(out_channels * k, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn,
                       dropout=dropout, activation=activation)

    def forward(self, x, batch):                       # x [N, k, in]
        z = self.enc(x, batch) + self.enc(-x, batch)
        return self.rho(z.reshape(z.shape[0], -1))


class MaskedGINDeepSigns(nn.Module):
    """All eigenvectors: frequencies k >= (size of the node's graph) are zeroed, sum over k, rho = MLP
    (reference :134-172)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dim_pe, rho_num_layers,
                 use_bn=False, use_ln=False, dropout=0.5, activation='relu'):
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)
        self.rho = MLP(out_channels, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)

    def forward(self, x, batch):
        z = self.enc(x, batch) + self.enc(-x, batch)                      # [N, k, out]
        ptr, _, _ = _gph._graph_sizes(batch)
        n_of_node = (ptr[1:] - ptr[:-1])[batch.batch]                     # size of each node's graph
        keep = torch.arange(z.shape[1], device=z.device)[None, :] < n_of_node[:, None]
        return self.rho((z * keep[:, :, None].to(z.dtype)).sum(dim=1))


@register_node_encoder('SignNet', overwrite=True)
class SignNetNodeEncoder(nn.Module):
    def __init__(self, dim_emb, expand_x=True):
        super().__init__()
        pecfg = cfg.posenc_SignNet
        dim_pe = pecfg.dim_pe
        if pecfg.model not in ('MLP', 'DeepSet'):
            raise ValueError(f"Unexpected SignNet model {pecfg.model}")
        self.model_type = pecfg.model
        if pecfg.post_layers < 1:
            raise ValueError("Num layers in rho model has to be positive.")
        self.pass_as_var = pecfg.pass_as_var
        if dim_emb - dim_pe < 1:
            raise ValueError(f"SignNet PE size {dim_pe} is too large for "
                             f"desired embedding size of {dim_emb}.")
        if expand_x:
            self.linear_x = nn.Linear(cfg.share.dim_in, dim_emb - dim_pe)
        self.expand_x = expand_x
        common = dict(in_channels=1, hidden_channels=pecfg.phi_hidden_dim, out_channels=pecfg.phi_out_dim,
                      num_layers=pecfg.layers, dim_pe=dim_pe, rho_num_layers=pecfg.post_layers, use_bn=True,
                      dropout=0.0, activation='relu')
        if self.model_type == 'MLP':
            self.sign_inv_net = GINDeepSigns(k=pecfg.eigen.max_freqs, **common)
        else:
            self.sign_inv_net = MaskedGINDeepSigns(**common)

    def forward(self, batch):
        if not (hasattr(batch, 'eigvals_sn') and hasattr(batch, 'eigvecs_sn')):
            raise ValueError("Precomputed eigen values and vectors are "
                             f"required for {self.__class__.__name__}; "
                             "set config 'posenc_SignNet.enable' to True")
        pe = batch.eigvecs_sn.unsqueeze(-1)                               # [N, k, 1]
        pe = torch.where(torch.isnan(pe), torch.zeros_like(pe), pe)
        pe = self.sign_inv_net(pe, batch)
        h = self.linear_x(batch.x) if self.expand_x else batch.x
        batch.x = torch.cat((h, pe), 1)
        if self.pass_as_var:
            batch.pe_SignNet = pe
        return batch


from . import extra_encoders as _extra  # noqa: E402  (dataset encoder table)

for _ds_name, _ds_cls in _extra._DS.items():
    register_node_encoder(f"{_ds_name}+SignNet",
                          _enc.concat_node_encoders(_ds_cls, SignNetNodeEncoder, 'SignNet'), overwrite=True)
    register_node_encoder(f"{_ds_name}+SignNet+RWSE",
                          _gph._compose([_ds_cls, SignNetNodeEncoder, _enc.RWSENodeEncoder], ['SignNet', 'RWSE']),
                          overwrite=True)
