"""Reference (CPU, torch) restatements used as checkers by the GPU parity tests."""
import torch

from oracle.gps_oracle import scatter_sum


def gatedgcn_core_ref(proj, ce, edge_index):
    """x_tilde, e_hat per graphgps/layer/gatedgcn_layer.py:90-136 from the fused projection."""
    d = proj.shape[1] // 4
    Ax, Bx, Dx, Ex = proj[:, :d], proj[:, d:2 * d], proj[:, 2 * d:3 * d], proj[:, 3 * d:]
    j, i = edge_index[0], edge_index[1]
    e_ij = Dx.index_select(0, i) + Ex.index_select(0, j) + ce
    s = torch.sigmoid(e_ij)
    num = scatter_sum(s * Bx.index_select(0, j), i, proj.shape[0])
    den = scatter_sum(s, i, proj.shape[0])
    return Ax + num / (den + 1e-6), e_ij


def gine_core_ref(x, e, edge_index, eps=0.0):
    j, i = edge_index[0], edge_index[1]
    return scatter_sum((x.index_select(0, j) + e).relu(), i, x.shape[0]) + (1 + eps) * x


def segment_attention_ref(qkv, ptr, H, keep=None, p_drop=0.0, bias=None):
    """Dense per-graph softmax attention; ``keep`` optional list (per graph) of bool [H,n,n]; ``bias`` the
    reference's dense additive mask [B*H, nmax, nmax] (plane g*H + h, its n x n corner is what counts)."""
    N, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    out = torch.zeros(N, d, dtype=qkv.dtype)
    outs = []
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        n = b - a
        if n == 0:
            continue
        q = qkv[a:b, :d].view(n, H, dh).transpose(0, 1)
        k = qkv[a:b, d:2 * d].view(n, H, dh).transpose(0, 1)
        v = qkv[a:b, 2 * d:].view(n, H, dh).transpose(0, 1)
        s = (q * dh ** -0.5) @ k.transpose(1, 2)
        if bias is not None:
            s = s + bias[g * H:(g + 1) * H, :n, :n]
        p = torch.softmax(s, dim=-1)
        if keep is not None:
            p = p * keep[g].to(p.dtype) / (1.0 - p_drop)
        outs.append(((p @ v).transpose(0, 1).reshape(n, d), a, b))
    if not outs:
        return qkv.new_zeros(N, d)
    return torch.cat([o for o, _, _ in outs], 0)
