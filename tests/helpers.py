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
    reference's dense additive mask [B*H, nmax, nmax] (plane g*H + h, its n x n corner is what counts).
    Graphs of the same size are evaluated together (one batched product per distinct size instead of one per graph:
    a 10-layer model over 256 graphs is 2,560 per-graph evaluations and ~10 autograd nodes each otherwise -- two minutes
    of the GPU suite were this loop's backward); every graph's rows are still a function of that graph alone."""
    N, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    sizes = (ptr[1:] - ptr[:-1]).tolist()
    starts = ptr[:-1].tolist()
    by_size = {}
    for g, n in enumerate(sizes):
        if n > 0:
            by_size.setdefault(int(n), []).append(g)
    if not by_size:
        return qkv.new_zeros(N, d)
    pieces, rows_all = [], []
    for n, gs in by_size.items():
        rows = (torch.tensor([int(starts[g]) for g in gs]).unsqueeze(1) + torch.arange(n).unsqueeze(0)).reshape(-1)   # [G*n]
        blk = qkv.index_select(0, rows).view(len(gs), n, 3, H, dh)
        q, k, v = (blk[:, :, i].transpose(1, 2) for i in range(3))          # [G, H, n, dh]
        s_ = (q * dh ** -0.5) @ k.transpose(2, 3)
        if bias is not None:
            s_ = s_ + torch.stack([bias[g * H:(g + 1) * H, :n, :n] for g in gs])
        p = torch.softmax(s_, dim=-1)
        if keep is not None:
            p = p * torch.stack([keep[g] for g in gs]).to(p.dtype) / (1.0 - p_drop)
        pieces.append((p @ v).transpose(1, 2).reshape(len(gs) * n, d))
        rows_all.append(rows)
    rows_all = torch.cat(rows_all)
    out = torch.cat(pieces, 0)
    inv = torch.empty_like(rows_all)
    inv[rows_all] = torch.arange(rows_all.numel())
    return out.index_select(0, inv)             # back to batch order (empty graphs own no rows: these are all N)


# ---------------------------------------------------------------------------------------------------------------
# Dropout-ON parity: the oracle with every dropout replaced by an INJECTED mask (the host model of the kernels'
# counter hash), so a training-mode layer with the measured configuration's dropout rates can be compared element by
# element.  graphgps/layer/gps_layer.py:139-140,152-153,253-257, gatedgcn_layer.py:78-79.
# ---------------------------------------------------------------------------------------------------------------
class MaskedSegmentMHA(torch.nn.Module):
    """Stands in for the oracle's ``nn.MultiheadAttention`` (same parameters) with the attention-probability dropout
    replaced by given keep masks: ``keep[g]`` bool [H, n_g, n_g], survivors scaled by 1 / (1 - p_eff)."""

    def __init__(self, mha, ptr, keep, p_eff):
        super().__init__()
        self.in_proj_weight, self.in_proj_bias, self.out_proj = mha.in_proj_weight, mha.in_proj_bias, mha.out_proj
        self.H, self.ptr, self.keep, self.p_eff = mha.num_heads, ptr, keep, p_eff

    def forward(self, q, k, v, attn_mask=None, key_padding_mask=None, need_weights=False):
        mask = ~key_padding_mask                                   # [B, Nmax]
        x = q[mask]                                                # ragged [N, d]
        qkv = torch.nn.functional.linear(x, self.in_proj_weight, self.in_proj_bias)
        o = segment_attention_ref(qkv, self.ptr, self.H, keep=self.keep, p_drop=self.p_eff)
        out = torch.zeros_like(q)
        out[mask] = self.out_proj(o)
        return out, None


class inject_dropout_masks:
    """``with inject_dropout_masks([m0, m1, ...]):`` every ``F.dropout`` call inside (``nn.Dropout`` modules included)
    multiplies its input by the next mask of the list (already scaled by 1 / (1 - p)) instead of drawing one."""

    def __init__(self, masks):
        self.masks = list(masks)

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.dropout
        queue = self.masks

        def fake(x, p=0.5, training=True, inplace=False):
            m = queue.pop(0)
            assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
            return x * m.to(x.dtype)
        F.dropout = fake
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.dropout = self._orig
        assert exc[0] is not None or not self.masks, f"{len(self.masks)} injected masks were not consumed"
        return False


def block_seeds(seed):
    """The seven per-stage seeds a fused block derives from its one drawn seed (layer/gps_block.py)."""
    return [(seed + 0x9E3779B97F4A7C15 * (i + 1)) & 0xFFFFFFFFFFFFFFFF for i in range(7)]


def row_mask(seed, R, d, p):
    """Keep mask of the row kernels / GEMM epilogues, keyed (row, column), as a multiplier (0 or 1 / (1 - p))."""
    from graphgps_amd.ops import attn_dropout_keep_mask
    if p == 0:
        return torch.ones(R, d, dtype=torch.float64)
    return attn_dropout_keep_mask(seed, torch.arange(R), 0, 1, torch.arange(d), p).double() / (1 - p)


def attention_keep(seed, ptr, H, p):
    from graphgps_amd.ops import attn_dropout_keep_mask
    keep = []
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        keep.append(torch.stack([attn_dropout_keep_mask(seed, torch.arange(a, b), h, H, torch.arange(b - a), p,
                                                        paired=True) for h in range(H)]))
    return keep
