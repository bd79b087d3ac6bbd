"""GPU parity tests of the individual HIP operators, through the C ABI (ctypes), against CPU
restatements.  Tolerances: bit-exact for indices; 1e-5 absolute on O(1) activations and 1e-5
relative-to-max on gradients (BASELINE.json north_star: 'within 1e-5 fp32')."""
import os

import numpy as np
import pytest
import torch

from conftest import Tol, assert_close
from helpers import gatedgcn_core_ref, gine_core_ref, segment_attention_ref

pytestmark = pytest.mark.gpu


def _structure(profile, nb, seed):
    from graphgps_amd.synthetic import make_structure
    sizes, ei, bvec, ptr, gen, _ = make_structure(profile, nb, seed)
    return sizes, ei, bvec, ptr, gen


def _index(ei, bvec, ptr, use_ptr=True, host_hint=True):
    """Graph index of a test batch.  ``host_hint``: record the longest graph as the host knows it (what a loader
    batch carries), which lets batches of <= 64-node graphs take the block-form attention kernels (csrc/sattn.hip);
    False = unknown, the per-tile kernels (csrc/seg_attention.hip) run whatever the sizes."""
    from graphgps_amd.ops import build_graph_index
    dev = torch.device("cuda:0")
    gi = build_graph_index(ei.to(dev), int(ptr[-1]), len(ptr) - 1,
                           batch_vec=bvec.to(dev), ptr_vec=ptr.to(dev) if use_ptr else None)
    if host_hint and len(ptr) > 1:
        gi.nmax_host = int((ptr[1:] - ptr[:-1]).max())
    return gi


@pytest.mark.parametrize("profile,nb,seed", [("P30", 64, 1), ("P14", 256, 2), ("CODE2_REAL", 8, 3)])
def test_graph_index_bit_exact(profile, nb, seed):
    sizes, ei, bvec, ptr, _ = _structure(profile, nb, seed)
    gi = _index(ei, bvec, ptr, use_ptr=False)
    N, E = int(ptr[-1]), ei.shape[1]
    src, dst = ei[0].numpy(), ei[1].numpy()
    for key, other, rowptr, oth_sorted, eid in (
            (dst, src, gi.rowptr_dst, gi.src_by_dst, gi.eid_by_dst),
            (src, dst, gi.rowptr_src, gi.dst_by_src, gi.eid_by_src)):
        perm = np.argsort(key, kind="stable")
        rp = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=N))])
        assert np.array_equal(rowptr.cpu().numpy(), rp.astype(np.int32))
        assert np.array_equal(eid.cpu().numpy(), perm.astype(np.int32))
        assert np.array_equal(oth_sorted.cpu().numpy(), other[perm].astype(np.int32))
    assert torch.equal(gi.ptr.cpu(), ptr.to(torch.int32))  # derived from batch.batch on device
    # tile map: every 16-row tile of every graph exactly once
    tg, tr = gi.tile_graph.cpu().numpy()[:gi.max_tiles], gi.tile_row0.cpu().numpy()[:gi.max_tiles]
    got = sorted((int(g), int(r)) for g, r in zip(tg, tr) if g >= 0)
    want = sorted((g, r) for g in range(nb) for r in range(int(ptr[g]), int(ptr[g + 1]), 16))
    assert got == want


def test_graph_index_hub_and_isolated_nodes():
    # star graph (hub degree 300 -> heapsort path) + isolated nodes + empty graph in the middle
    n = 301
    hub_edges = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)])
    ei = torch.cat([hub_edges, hub_edges.flip(0)], 1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))]
    N = n + 5
    bvec = torch.cat([torch.zeros(n, dtype=torch.long), torch.full((5,), 2, dtype=torch.long)])
    ptr = torch.tensor([0, n, n, N])
    gi = _index(ei, bvec, ptr, use_ptr=False)
    perm = np.argsort(ei[1].numpy(), kind="stable")
    assert np.array_equal(gi.eid_by_dst.cpu().numpy(), perm.astype(np.int32))
    assert torch.equal(gi.ptr.cpu(), ptr.to(torch.int32))


def _hub_batch(deg, extra_graphs=3, seed=0):
    """A star of ``deg`` leaves around node 0 (both directions, shuffled) followed by a few P30-like graphs: the degree
    MalNet's function-call graphs reach (configs/GPS/malnettiny-GPS.yaml; hubs of thousands of callers)."""
    gen = torch.Generator().manual_seed(seed)
    n = deg + 1
    hub = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)])
    parts, sizes, off = [hub, hub.flip(0)], [n], n
    for g in range(extra_graphs):
        k = 20 + 7 * g
        a = torch.arange(k - 1)
        chain = torch.stack([a, a + 1]) + off
        parts += [chain, chain.flip(0)]
        sizes.append(k)
        off += k
    ei = torch.cat(parts, 1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=gen)]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return ei, bvec, ptr, gen


@pytest.mark.parametrize("case,d,p", [("p30", 384, 0.1), ("p30", 64, 0.0), ("hub", 128, 0.25), ("p30_padded", 256, 0.2),
                                      ("ast", 256, 0.2)])
def test_gatedgcn_bwd_bn_through_the_c_abi(case, d, p):
    """gps_gatedgcn_bwd_bn (include/gps_hip.h, ABI v9) against the three launches it stands for -- gps_norm_bwd_apply of
    bn_node_x and bn_edge_e, then gps_gatedgcn_bwd -- on the same operands and seeds: molecule-sized graphs (in-degrees 1 .. 6:
    the one-pass chunks and, at degree >= 4 with both folds, the two-pass form), a hub of degree 300 (the long-segment path,
    index slices past the LDS staging limit, out-of-block targets), and a padded batch whose last rows are junk beyond the
    device-side real-row words (their folded gradients must come out exactly as the apply kernel's zeros make them), and an
    AST batch (24 graphs of 600 - 1000 nodes at d = 256: a node block's CSR slice outgrows the per-edge LDS stash, the
    per-node a_i stash form of the backward runs -- csrc/gatedgcn.hip ASTASH).
    Reference: gatedgcn_layer.py:72-83 and its autograd backward."""
    import ctypes
    from graphgps_amd import lib as _lib, norm as _norm
    from graphgps_amd.lib import check, current_stream, ptr as P_
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5 + d)
    if case == "hub":
        ei, bvec, ptr, _ = _hub_batch(300, extra_graphs=6, seed=3)
    elif case == "ast":
        _, ei, bvec, ptr, _ = _structure("CODE2_LONG", 24, 4)
    else:
        b = layer_batch("P30", 48, d, seed=21)
        ei, bvec, ptr = b.edge_index, b.batch, b.ptr
    N, E = int(ptr[-1]), ei.shape[1]
    gi = _index(ei, bvec, ptr, use_ptr=False)
    n_real, e_real = (N - 37, E - 90) if case == "p30_padded" else (N, E)
    rn = torch.tensor([n_real], dtype=torch.int32, device=dev) if case == "p30_padded" else None
    re_ = torch.tensor([e_real], dtype=torch.int32, device=dev) if case == "p30_padded" else None
    f = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
    proj, xt, eh = f(N, 4 * d), f(N, d), f(E, d)
    g_x1, g_e1 = f(N, d), f(E, d)
    L = _lib.load()
    st = current_stream(dev)
    bns = []
    for _ in range(2):
        m = torch.nn.BatchNorm1d(d).to(dev)
        with torch.no_grad():
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.5, 0.5)
        bns.append(m)
    stats = torch.empty(4, d, device=dev)
    stats[0], stats[2] = xt[:n_real].mean(0), eh[:e_real].mean(0)
    stats[1] = (xt[:n_real].var(0, unbiased=False) + 1e-5).rsqrt()
    stats[3] = (eh[:e_real].var(0, unbiased=False) + 1e-5).rsqrt()
    bnx, bne = _norm.bn_desc(bns[0], stats[0], stats[1]), _norm.bn_desc(bns[1], stats[2], stats[3])
    sums = torch.zeros(4, d, device=dev)            # (g_gamma_x, g_beta_x, g_gamma_e, g_beta_e)
    sx, se = 0x1111222233334444, 0x5555666677778888

    class Owner:
        pass
    sync = _norm.sync_arena(Owner(), dev)
    g_xt, g_eh = torch.empty(N, d, device=dev), torch.empty(E, d, device=dev)
    tasks = [_norm.bwd_task(xt, g_x1, bnx, N, sums[0], sums[1], relu=True, p=p, seed=sx, g_z=g_xt, rdev=rn),
             _norm.bwd_task(eh, g_e1, bne, E, sums[2], sums[3], relu=True, p=p, seed=se, g_z=g_eh, rdev=re_)]
    _norm.bwd_partial(tasks, d, dev, sync.site(0))
    _norm.bwd_apply(tasks, d, dev, None)
    fs = 4 * d
    outs = []
    for folded in (False, True):
        g_proj, g_ce = torch.zeros(N, 4 * d, device=dev), torch.zeros(E, d, device=dev)
        Pp, G = proj.data_ptr(), g_proj.data_ptr()
        common = (Pp, Pp + fs, 4 * d, P_(xt), P_(gi.rowptr_dst), P_(gi.src_by_dst), P_(gi.eid_by_dst), P_(gi.rowptr_src),
                  P_(gi.dst_by_src), P_(gi.eid_by_src), N, E, d, P_(g_ce), G, G + fs, G + 2 * fs, G + 3 * fs, 4 * d, None, None, None)
        if folded:
            fx = _lib.BnBwdFold(ctypes.addressof(bnx), sums[1].data_ptr(), sums[0].data_ptr(), p, sx, 1, P_(rn))
            fe = _lib.BnBwdFold(ctypes.addressof(bne), sums[3].data_ptr(), sums[2].data_ptr(), p, se, 1, P_(re_))
            check(L.gps_gatedgcn_bwd_bn(P_(g_x1), d, P_(g_e1), P_(eh), *common, ctypes.byref(fx), ctypes.byref(fe), st),
                  "gps_gatedgcn_bwd_bn")
        else:
            check(L.gps_gatedgcn_bwd(P_(g_xt), d, P_(g_eh), P_(eh), *common, st), "gps_gatedgcn_bwd")
        outs.append((g_proj, g_ce))
    torch.cuda.synchronize()
    for name, a, c in (("g_Ax|g_Bx|g_Dx|g_Ex", outs[1][0], outs[0][0]), ("g_Ce", outs[1][1], outs[0][1])):
        scale = float(c.abs().max())
        err = float((a - c).abs().max()) / scale
        assert err <= 2e-6, f"{case}: {name} folded vs launched differ by {err:.2e} of max|.|"
    if case == "p30_padded":        # the padding rows' folded g_x~ (stored as g_Ax) are exactly zero, as the apply kernel makes them
        assert float(outs[1][0][n_real:, :d].abs().max()) == 0.0


def test_graph_index_and_gatedgcn_with_a_degree_5000_hub():
    """Graph index bit-exact against numpy's stable argsort and the GatedGCN core against the CPU restatement when one node
    has 5,000 in- and out-edges (the index sorts each CSR segment; the kernels walk a segment per node)."""
    from graphgps_amd.ops import gatedgcn_aggregate
    ei, bvec, ptr, gen = _hub_batch(5000)
    N, E = int(ptr[-1]), ei.shape[1]
    gi = _index(ei, bvec, ptr, use_ptr=False)
    src, dst = ei[0].numpy(), ei[1].numpy()
    for key, other, rowptr, oth_sorted, eid in ((dst, src, gi.rowptr_dst, gi.src_by_dst, gi.eid_by_dst),
                                                (src, dst, gi.rowptr_src, gi.dst_by_src, gi.eid_by_src)):
        perm = np.argsort(key, kind="stable")
        rp = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=N))])
        assert np.array_equal(rowptr.cpu().numpy(), rp.astype(np.int32))
        assert np.array_equal(eid.cpu().numpy(), perm.astype(np.int32))
        assert np.array_equal(oth_sorted.cpu().numpy(), other[perm].astype(np.int32))
    d = 128
    proj = torch.randn(N, 4 * d, generator=gen)
    ce = torch.randn(E, d, generator=gen)
    wx, we = torch.randn(N, d, generator=gen), torch.randn(E, d, generator=gen)
    pr, cr = proj.clone().double().requires_grad_(True), ce.clone().double().requires_grad_(True)
    xr, er = gatedgcn_core_ref(pr, cr, ei)
    ((xr * wx.double()).sum() + (er * we.double()).sum()).backward()
    dev = torch.device("cuda:0")
    pg, cg = proj.to(dev).requires_grad_(True), ce.to(dev).requires_grad_(True)
    xg, eg = gatedgcn_aggregate(pg, cg, gi)
    ((xg * wx.to(dev)).sum() + (eg * we.to(dev)).sum()).backward()
    assert_close(xg, xr, Tol.ACT, "x_tilde (hub)")
    assert_close(eg, er, Tol.ACT, "e_hat (hub)")
    assert_close(pg.grad, pr.grad, Tol.GRAD_REL, "g_proj (hub)", rel_to_max=True)
    assert_close(cg.grad, cr.grad, Tol.GRAD_REL, "g_Ce (hub)", rel_to_max=True)


@pytest.mark.parametrize("d,profile,nb", [(16, "P30", 48), (52, "P30", 48), (384, "P30", 48), (256, "CODE2_LONG", 24),
                                          (384, "CODE2_LONG", 16), (384, "P30", 1500)],
                         ids=["16", "52", "384", "ast-256", "ast-384", "1500-molecules"])
def test_gatedgcn_core(d, profile, nb):
    """(the two AST cases: node blocks whose CSR slices outgrow the per-edge LDS stash of the backward -- the per-node a_i
    stash form, csrc/gatedgcn.hip ASTASH; 1,500 molecules: ~45k nodes, where one dispatch round's 176-node blocks outgrow
    the a_i stash too and the backward launches smaller blocks over several rounds)"""
    from graphgps_amd.ops import gatedgcn_aggregate
    sizes, ei, bvec, ptr, gen = _structure(profile, nb, 5)
    N, E = int(ptr[-1]), ei.shape[1]
    proj = torch.randn(N, 4 * d, generator=gen)
    ce = torch.randn(E, d, generator=gen)
    wx, we = torch.randn(N, d, generator=gen), torch.randn(E, d, generator=gen)
    pr, cr = proj.clone().requires_grad_(True), ce.clone().requires_grad_(True)
    xr, er = gatedgcn_core_ref(pr, cr, ei)
    ((xr * wx).sum() + (er * we).sum()).backward()
    dev = torch.device("cuda:0")
    gi = _index(ei, bvec, ptr)
    pg, cg = proj.to(dev).requires_grad_(True), ce.to(dev).requires_grad_(True)
    xg, eg = gatedgcn_aggregate(pg, cg, gi)
    ((xg * wx.to(dev)).sum() + (eg * we.to(dev)).sum()).backward()
    assert_close(xg, xr, Tol.ACT, "x_tilde")
    assert_close(eg, er, Tol.ACT, "e_hat")
    assert_close(pg.grad, pr.grad, Tol.GRAD_REL, "g_proj", rel_to_max=True)
    assert_close(cg.grad, cr.grad, Tol.GRAD_REL, "g_Ce", rel_to_max=True)
    # determinism: bitwise identical on a second run
    xg2, eg2 = gatedgcn_aggregate(pg.detach(), cg.detach(), gi)
    assert torch.equal(xg2, xg.detach()) and torch.equal(eg2, eg.detach())


def test_gatedgcn_core_with_edge_gate_at_the_benchmark_batch():
    """The EquivStableLapPE form (sigma_ij * r_ij gates and normalises, gatedgcn_layer.py:101-104) at P30 x 256, d = 384:
    32-node blocks of ~65 CSR entries, i.e. the backward's per-node a_i stash with the gate (k_gatedgcn_bwd<4, true, 4>),
    against the fp64 restatement; r_ij receives no gradient here (the layer differentiates it through PyTorch)."""
    from graphgps_amd.ops import gatedgcn_aggregate
    d = 384
    sizes, ei, bvec, ptr, gen = _structure("P30", 256, 9)
    N, E = int(ptr[-1]), ei.shape[1]
    proj = torch.randn(N, 4 * d, generator=gen)
    ce = torch.randn(E, d, generator=gen)
    r = torch.rand(E, generator=gen) * 0.9 + 0.05
    wx, we = torch.randn(N, d, generator=gen), torch.randn(E, d, generator=gen)
    pr, cr = proj.double().requires_grad_(True), ce.double().requires_grad_(True)
    Ax, Bx, Dx, Ex = pr[:, :d], pr[:, d:2 * d], pr[:, 2 * d:3 * d], pr[:, 3 * d:]
    j, i = ei[0], ei[1]
    e_ij = Dx.index_select(0, i) + Ex.index_select(0, j) + cr
    s = torch.sigmoid(e_ij) * r.double()[:, None]
    num = torch.zeros(N, d, dtype=torch.float64).index_add_(0, i, s * Bx.index_select(0, j))
    den = torch.zeros(N, d, dtype=torch.float64).index_add_(0, i, s)
    xr = Ax + num / (den + 1e-6)
    ((xr * wx.double()).sum() + (e_ij * we.double()).sum()).backward()
    dev = torch.device("cuda:0")
    gi = _index(ei, bvec, ptr)
    pg, cg = proj.to(dev).requires_grad_(True), ce.to(dev).requires_grad_(True)
    xg, eg = gatedgcn_aggregate(pg, cg, gi, r.to(dev))
    ((xg * wx.to(dev)).sum() + (eg * we.to(dev)).sum()).backward()
    assert_close(xg, xr, Tol.ACT, "x_tilde (gated)")
    assert_close(eg, e_ij, Tol.ACT, "e_hat (gated)")
    assert_close(pg.grad, pr.grad, Tol.GRAD_REL, "g_proj (gated)", rel_to_max=True)
    assert_close(cg.grad, cr.grad, Tol.GRAD_REL, "g_Ce (gated)", rel_to_max=True)


def test_gatedgcn_no_edges_and_odd_dim():
    from graphgps_amd.ops import build_graph_index, gatedgcn_aggregate
    dev = torch.device("cuda:0")
    N, d = 7, 6  # d % 4 != 0 -> float2 path; no edges -> x_tilde = Ax
    ei = torch.zeros(2, 0, dtype=torch.long, device=dev)
    gi = build_graph_index(ei, N, 1, ptr_vec=torch.tensor([0, N], device=dev))
    proj = torch.randn(N, 4 * d, device=dev)
    x, e = gatedgcn_aggregate(proj, torch.zeros(0, d, device=dev), gi)
    assert torch.equal(x, proj[:, :d]) and e.shape == (0, d)


@pytest.mark.parametrize("d", [64, 50])
def test_gine_core(d):
    from graphgps_amd.ops import gine_aggregate
    sizes, ei, bvec, ptr, gen = _structure("ZINC", 32, 6)
    N, E = int(ptr[-1]), ei.shape[1]
    x, e = torch.randn(N, d, generator=gen), torch.randn(E, d, generator=gen)
    w = torch.randn(N, d, generator=gen)
    xr, er = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    (gine_core_ref(xr, er, ei) * w).sum().backward()
    dev = torch.device("cuda:0")
    gi = _index(ei, bvec, ptr)
    xg, eg = x.to(dev).requires_grad_(True), e.to(dev).requires_grad_(True)
    out = gine_aggregate(xg, eg, gi, 0.0)
    (out * w.to(dev)).sum().backward()
    assert_close(out, gine_core_ref(x, e, ei), Tol.ACT, "gine out")
    assert_close(xg.grad, xr.grad, Tol.GRAD_REL, "g_x", rel_to_max=True)
    assert_close(eg.grad, er.grad, Tol.GRAD_REL, "g_e", rel_to_max=True)


ATTN_CASES = [  # (H, dh, graph sizes)
    (4, 16, [23, 9, 37, 16, 1, 17]),
    (16, 24, [30, 45, 4, 64, 29]),
    (4, 13, [12, 33, 20]),
    (8, 6, [14, 14, 7]),
    (4, 76, [40, 18]),
    (2, 32, [150, 70, 129]),      # > 64 keys: several online-softmax blocks
    (4, 96, [31, 66]),
    (4, 64, [1000, 3, 601]),      # code2-size graphs (master_loader.py:366-368 clips at 1000 nodes)
    (2, 64, [4500, 3]),           # MalNet-sized (configs/GPS/malnettiny-GPS.yaml: up to ~5,000 nodes): 282 row tiles per head
    (1, 128, [48, 2]),            # widest compiled head
    (2, 4, [5, 16, 32, 64]),      # narrowest compiled head, sizes exactly on the tile boundaries
    (8, 10, [12, 33, 7]),         # zinc-Graphormer: embed 80 / 8 heads
    (4, 20, [31, 5, 48]),
    (8, 8, [64, 1, 63, 17, 16]),  # block form (graphs <= 64, dh in {8,16,24,32}): every tile count, single rows
    (4, 32, [33, 48, 2, 64]),
    (16, 24, [64, 64, 5, 49, 32, 1, 15]),
]


def _attn_inputs(H, dh, sizes, seed=0):
    gen = torch.Generator().manual_seed(seed)
    N, d = sum(sizes), H * dh
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    qkv = torch.randn(N, 3 * d, generator=gen)
    w = torch.randn(N, d, generator=gen)
    ei = torch.zeros(2, 0, dtype=torch.long)
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return qkv, w, ptr, ei, bvec


@pytest.mark.parametrize("hint", [True, False])
@pytest.mark.parametrize("H,dh,sizes", ATTN_CASES)
def test_segment_attention_fwd_bwd(H, dh, sizes, hint):
    """Both kernel families against the dense fp64 softmax attention: with the host-side size hint the batches of
    <= 64-node graphs with dh in {8, 16, 24, 32} take the block form (one launch forward, one backward), without
    it everything takes the per-tile kernels."""
    from graphgps_amd.ops import segment_attention
    if not hint and (max(sizes) > 64 or dh not in (8, 16, 24, 32)):
        pytest.skip("same kernels as with the hint")
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes)
    qr = qkv.clone().double().requires_grad_(True)
    ref = segment_attention_ref(qr, ptr, H)
    (ref * w.double()).sum().backward()
    dev = torch.device("cuda:0")
    gi = _index(ei, bvec, ptr, host_hint=hint)
    qg = qkv.to(dev).requires_grad_(True)
    out = segment_attention(qg, gi, H, 0.0)
    (out * w.to(dev)).sum().backward()
    assert_close(out, ref, Tol.ACT, "attn out")
    assert_close(qg.grad, qr.grad, Tol.GRAD_REL, "d_qkv", rel_to_max=True)


@pytest.mark.parametrize("H,dh,nb", [(16, 24, 256), (4, 16, 37), (8, 8, 1000), (16, 24, 2)])
def test_attention_graph_order_is_a_balanced_permutation_and_scheduling_only(H, dh, nb):
    """gps_attn_graph_order (ABI v11): the dispatch order of the block-form attention kernels is a permutation of the graphs
    that deals long and short graphs to the same CUs (slots t, t + cols, ... share a CU: their summed tile-pair work stays
    near the mean), and it changes scheduling only: forward and backward are bit-identical with and without it."""
    import graphgps_amd.ops as ops
    from graphgps_amd.ops import segment_attention
    gen = torch.Generator().manual_seed(nb + H)
    sizes = (torch.randn(nb, generator=gen) * 7.5 + 30).round().clamp(4, 64).long().tolist()
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes)
    dev = torch.device("cuda:0")
    gi = _index(ei, bvec, ptr, host_hint=True)
    order = gi.attn_order(H)
    assert order is not None and sorted(order.cpu().tolist()) == list(range(nb))
    nt2 = ((torch.tensor(sizes) + 15) // 16) ** 2
    cols = max(1, 256 * 4 // H)
    if nb >= 4 * cols:          # whole rows: column sums of the snake against those of the identity order
        def worst(perm):
            v = nt2[perm][:nb // cols * cols].view(-1, cols).sum(0)
            return float(v.max()) / float(v.float().mean())
        assert worst(order.cpu().long()) <= min(worst(torch.arange(nb)), 1.25)      # (measured 1.10 / 1.16; identity 1.52 / 1.59)
    outs = []
    for use in (True, False):
        gi.orders = None if use else {H: None}
        if not use:
            assert gi.attn_order(H) is None
        qg = qkv.to(dev).requires_grad_(True)
        out = segment_attention(qg, gi, H, 0.0)
        (out * w.to(dev)).sum().backward()
        outs.append((out.detach().clone(), qg.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_segment_attention_spiked_scores():
    """Online-softmax rescale path: a key block whose max jumps far above the previous ones."""
    from graphgps_amd.ops import segment_attention
    H, dh, sizes = 2, 16, [200]
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes, seed=3)
    d = H * dh
    qkv[5, :d] *= 6.0
    qkv[170, d:2 * d] = qkv[5, :d] * 1.5      # huge q.k in the last key block
    qkv[70, d:2 * d] = -qkv[5, :d]
    ref = segment_attention_ref(qkv.double(), ptr, H)
    gi = _index(ei, bvec, ptr)
    out = segment_attention(qkv.cuda(), gi, H, 0.0)
    assert torch.isfinite(out).all()
    assert_close(out, ref, Tol.ACT, "attn out (spiked)")


@pytest.mark.parametrize("H,dh,sizes", [(4, 16, [23, 9, 37]), (16, 24, [30, 45, 70]), (16, 24, [30, 64, 17, 5])])
def test_segment_attention_dropout_shared_mask(H, dh, sizes):
    """Dropout parity by injecting the kernel's own counter-based mask into the reference."""
    from graphgps_amd.ops import attn_dropout_effective_p, attn_dropout_keep_mask, segment_attention
    p, seed = 0.3, 0x1234_5678_9ABC_DEF1
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes, seed=1)
    keep = []
    for g in range(len(sizes)):
        a, b = int(ptr[g]), int(ptr[g + 1])
        keep.append(torch.stack([attn_dropout_keep_mask(seed, torch.arange(a, b), h, H,
                                                        torch.arange(b - a), p, paired=True) for h in range(H)]))
    rate = torch.cat([k.flatten() for k in keep]).float().mean().item()
    assert abs(rate - (1 - p)) < 0.02, rate
    # the two decisions taken from one hash are independent: P(both kept) = (1-p)^2 over adjacent key pairs
    allk = torch.cat([k[:, :, :(k.shape[2] // 2) * 2].reshape(-1, 2) for k in keep]).float()
    assert abs((allk[:, 0] * allk[:, 1]).mean().item() - (1 - p) ** 2) < 0.02
    qr = qkv.clone().double().requires_grad_(True)
    ref = segment_attention_ref(qr, ptr, H, keep=keep, p_drop=attn_dropout_effective_p(p))
    (ref * w.double()).sum().backward()
    gi = _index(ei, bvec, ptr)
    qg = qkv.cuda().requires_grad_(True)
    out = segment_attention(qg, gi, H, p, seed=seed)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "attn out (dropout)")
    assert_close(qg.grad, qr.grad, Tol.GRAD_REL, "d_qkv (dropout)", rel_to_max=True)


@pytest.mark.parametrize("H,dh,sizes,pad,p", [
    (4, 16, [23, 9, 37, 16, 1, 17], 0, 0.0),     # zinc-GPSwGraphormer head shape; sizes on/around tile edges
    (8, 10, [12, 33, 20, 5], 0, 0.0),            # zinc-Graphormer: embed 80 / 8 heads
    (4, 20, [31, 64, 2], 3, 0.0),                # bias padded wider than the largest graph
    (16, 24, [30, 45, 70], 0, 0.25),             # two key blocks + attention dropout (shared mask)
    (2, 64, [150, 129], 1, 0.0),                 # several online-softmax blocks
])
def test_segment_attention_additive_bias(H, dh, sizes, pad, p):
    """softmax(q k^T / sqrt(dh) + bias) with the reference's dense [B*H, nmax, nmax] bias operand
    (gps_layer.py:201-203, graphormer_layer.py:43-44): output, d_qkv and d_bias against the dense per-graph
    fp64 reference; the gradient of the padded region is exactly zero."""
    from graphgps_amd.ops import attn_dropout_effective_p, attn_dropout_keep_mask, segment_attention
    seed = 0x0BADC0DE12345678
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes, seed=4)
    nmax = max(sizes) + pad
    gen = torch.Generator().manual_seed(9)
    bias = torch.randn(len(sizes) * H, nmax, nmax, generator=gen) * 2.0
    keep = None
    if p > 0:
        keep = []
        for g in range(len(sizes)):
            a, b = int(ptr[g]), int(ptr[g + 1])
            keep.append(torch.stack([attn_dropout_keep_mask(seed, torch.arange(a, b), h, H,
                                                            torch.arange(b - a), p, paired=True) for h in range(H)]))
    qr = qkv.clone().double().requires_grad_(True)
    br = bias.clone().double().requires_grad_(True)
    ref = segment_attention_ref(qr, ptr, H, keep=keep, p_drop=attn_dropout_effective_p(p), bias=br)
    (ref * w.double()).sum().backward()
    gi = _index(ei, bvec, ptr)
    qg = qkv.cuda().requires_grad_(True)
    bg = bias.cuda().requires_grad_(True)
    out = segment_attention(qg, gi, H, p, seed=seed, bias=bg)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "attn out (bias)")
    assert_close(qg.grad, qr.grad, Tol.GRAD_REL, "d_qkv (bias)", rel_to_max=True)
    assert_close(bg.grad, br.grad, Tol.GRAD_REL, "d_bias", rel_to_max=True)
    outside = torch.ones_like(bias, dtype=torch.bool)
    for g, n in enumerate(sizes):
        outside[g * H:(g + 1) * H, :n, :n] = False
    assert (bg.grad.cpu()[outside] == 0).all()
    # same result as without a bias when the bias is zero
    plain = segment_attention(qkv.cuda(), gi, H, 0.0)
    zero = segment_attention(qkv.cuda(), gi, H, 0.0, bias=torch.zeros_like(bg))
    assert_close(zero, plain, 1e-6, "zero bias == no bias")


def test_segment_attention_bias_shape_errors():
    from graphgps_amd.lib import GpsHipError
    from graphgps_amd.ops import segment_attention
    H, dh, sizes = 2, 16, [9, 20]
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, sizes)
    gi = _index(ei, bvec, ptr)
    with pytest.raises(GpsHipError):      # padded narrower than the largest graph
        segment_attention(qkv.cuda(), gi, H, 0.0, bias=torch.zeros(len(sizes) * H, 19, 19).cuda())
    with pytest.raises(GpsHipError):      # wrong number of planes
        segment_attention(qkv.cuda(), gi, H, 0.0, bias=torch.zeros(H, 20, 20).cuda())
    with pytest.raises(GpsHipError):      # CPU operand
        segment_attention(qkv.cuda(), gi, H, 0.0, bias=torch.zeros(len(sizes) * H, 20, 20))


def test_segment_attention_padding_invariance():
    """A graph's output must not depend on which other graphs share the batch."""
    from graphgps_amd.ops import segment_attention
    H, dh = 16, 24
    qkv, w, ptr, ei, bvec = _attn_inputs(H, dh, [30, 61, 12])
    gi = _index(ei, bvec, ptr)
    full = segment_attention(qkv.cuda(), gi, H, 0.0)
    a, b = int(ptr[1]), int(ptr[2])
    solo_ptr = torch.tensor([0, b - a])
    gi1 = _index(ei, torch.zeros(b - a, dtype=torch.long), solo_ptr)
    solo = segment_attention(qkv[a:b].cuda(), gi1, H, 0.0)
    assert torch.equal(full[a:b], solo)


def test_unsupported_head_dim_fails_loudly():
    from graphgps_amd.lib import GpsHipError
    from graphgps_amd.ops import segment_attention
    qkv, w, ptr, ei, bvec = _attn_inputs(2, 7, [5])
    gi = _index(ei, bvec, ptr)
    with pytest.raises(GpsHipError):
        segment_attention(qkv.cuda(), gi, 2, 0.0)


@pytest.mark.parametrize("mode", ["mean", "add"])
def test_segment_pool(mode):
    from graphgps_amd.ops import segment_pool
    sizes, ei, bvec, ptr, gen = _structure("P30", 40, 9)
    N, d = int(ptr[-1]), 384
    x, w = torch.randn(N, d, generator=gen), torch.randn(40, d, generator=gen)
    xr = x.clone().requires_grad_(True)
    ref = torch.zeros(40, d).index_add_(0, bvec, xr)
    if mode == "mean":
        ref = ref / torch.bincount(bvec, minlength=40).clamp(min=1)[:, None]
    (ref * w).sum().backward()
    gi = _index(ei, bvec, ptr)
    xg = x.cuda().requires_grad_(True)
    out = segment_pool(xg, gi, mode)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "pool")
    assert_close(xg.grad, xr.grad, Tol.GRAD_REL, "pool grad", rel_to_max=True)


@pytest.mark.parametrize("d", [256, 384, 52, 6, 1028])
@pytest.mark.parametrize("mode", ["mean", "add"])
def test_segment_pool_over_row_slices(mode, d):
    """Round 5, csrc/segment_pool.hip k_pool_slices / k_pool_merge (one workgroup per 32-row slice of a graph, partial rows
    merged in slice order) against fp64 index_add: graphs of 1, 31, 32, 33, 64, 65 and 1,000 rows, EMPTY graphs (first,
    middle, last: pooled to 0, mean count clamped to 1 as PyG does), widths with 4- / 2-float lanes, more lanes per row than
    a workgroup has threads; bitwise reproducible; same results as the one-lane-group-per-graph kernel to rounding."""
    from graphgps_amd import ops as _ops
    from graphgps_amd.ops import segment_pool
    sizes = torch.tensor([0, 1, 31, 32, 0, 33, 64, 65, 1000, 7, 0])
    B = len(sizes)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(sizes, 0)])
    N = int(ptr[-1])
    bvec = torch.repeat_interleave(torch.arange(B), sizes)
    gen = torch.Generator().manual_seed(d)
    ei = torch.stack([torch.arange(N), torch.arange(N)])                   # self loops: the index only needs ptr here
    x, w = torch.randn(N, d, generator=gen), torch.randn(B, d, generator=gen)
    xr = x.double().requires_grad_(True)
    ref = torch.zeros(B, d, dtype=torch.float64).index_add_(0, bvec, xr)
    if mode == "mean":
        ref = ref / sizes.clamp(min=1)[:, None]
    (ref * w.double()).sum().backward()
    gi = _index(ei, bvec, ptr)
    xg = x.cuda().requires_grad_(True)
    out = segment_pool(xg, gi, mode)
    (out * w.cuda()).sum().backward()
    scale = 1.0 if mode == "mean" else 30.0                                # |sum of 1,000 N(0,1)| ~ 30
    assert_close(out, ref, Tol.ACT * scale, "pool over slices")
    assert_close(xg.grad, xr.grad, Tol.GRAD_REL, "pool grad", rel_to_max=True)
    assert float(out[0].abs().max()) == 0.0 and float(out[4].abs().max()) == 0.0 and float(out[-1].abs().max()) == 0.0
    again = segment_pool(x.cuda(), gi, mode)
    assert torch.equal(again, out.detach()), "the sliced pooling is not bitwise reproducible"
    old = _ops._POOL_SLICED
    try:
        _ops._POOL_SLICED = False
        one = segment_pool(x.cuda(), gi, mode)
    finally:
        _ops._POOL_SLICED = old
    assert_close(one, ref, Tol.ACT * scale, "pool, one lane group per graph")


@pytest.mark.parametrize("kind,emb,R", [("atom", 364, 7569), ("bond", 384, 15348), ("ast", 256, 3001), ("narrow", 52, 130)])
def test_embed_sum_and_multihot_gradient(kind, emb, R):
    """Round 5, csrc/embed.hip: the sum-of-embeddings encoders in one launch (gps_embed_sum) and their table gradients through
    the one-launch multi-hot matrix (gps_multihot_fill) + a GEMM, against ``sum_i F.embedding(feats[:, i], table_i)``: the
    forward is the SAME fp32 additions in the same order (bit-exact against the torch loop on the device), the gradients
    match an fp64 evaluation to 1e-6 of their maximum; features taken as a strided column slice of a wider matrix."""
    import torch.nn.functional as F
    from graphgps_amd.encoder.encoders import _EmbedSum, _embed_sum_ok
    from graphgps_amd.synthetic import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS
    dims = {"atom": ATOM_FEATURE_DIMS, "bond": BOND_FEATURE_DIMS, "ast": [98, 21], "narrow": [5, 1, 7]}[kind]
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(emb + R)
    wide = torch.stack([torch.randint(0, v, (R,), generator=gen) for v in dims] + [torch.zeros(R, dtype=torch.long)], 1)
    feats = wide.to(dev)[:, :len(dims)]                       # row stride len(dims) + 1
    embs = torch.nn.ModuleList([torch.nn.Embedding(v, emb) for v in dims]).to(dev)
    assert _embed_sum_ok(feats, embs)
    out = _EmbedSum.apply(feats, *[e.weight for e in embs])
    ref = 0
    for i, e in enumerate(embs):
        ref = ref + F.embedding(feats[:, i], e.weight)
    assert torch.equal(out, ref), float((out - ref).abs().max())
    w = torch.randn(R, emb, generator=gen).to(dev)
    (out * w).sum().backward()
    for i, e in enumerate(embs):
        g64 = torch.zeros(dims[i], emb, dtype=torch.float64).index_add_(0, wide[:, i], w.double().cpu())
        assert_close(e.weight.grad, g64, Tol.GRAD_REL, f"table {i} gradient", rel_to_max=True)
    # determinism of the gradient (a GEMM over a fixed multi-hot matrix)
    g1 = [e.weight.grad.clone() for e in embs]
    for e in embs:
        e.weight.grad = None
    (_EmbedSum.apply(feats, *[e.weight for e in embs]) * w).sum().backward()
    assert all(torch.equal(a, e.weight.grad) for a, e in zip(g1, embs))


def test_embed_sum_rejects_out_of_range_features_like_nn_embedding():
    """The gather-sum kernel clamps out-of-range indices (memory safety); the first batch a table set sees is range-checked
    on the host so that corrupt features fail as ``nn.Embedding`` does instead of training against the wrong row."""
    from graphgps_amd.encoder.encoders import _EmbedSum
    dev = torch.device("cuda:0")
    embs = torch.nn.ModuleList([torch.nn.Embedding(v, 16) for v in (5, 3)]).to(dev)
    bad = torch.tensor([[0, 1], [4, 3]], device=dev)            # column 1 holds a 3: vocabulary is 3
    with pytest.raises(IndexError, match="column"):
        _EmbedSum.apply(bad, *[e.weight for e in embs])
    neg = torch.tensor([[0, 1], [-1, 2]], device=dev)
    with pytest.raises(IndexError):
        _EmbedSum.apply(neg, *[e.weight for e in embs])
    good = torch.tensor([[0, 1], [4, 2]], device=dev)
    out = _EmbedSum.apply(good, *[e.weight for e in embs])
    assert torch.equal(out, embs[0].weight[good[:, 0]] + embs[1].weight[good[:, 1]])


@pytest.mark.parametrize("M,K,N,relu", [(256, 384, 192, True), (256, 192, 96, True), (256, 96, 1, False), (37, 52, 5, True),
                                        (1, 7, 3, False), (2048, 64, 16, True), (33, 130, 10, False)])
def test_small_linear_fwd_bwd(M, K, N, relu):
    """csrc/small_gemm.hip (round 6): the Linear (+ ReLU) stages of the graph-level heads (graphgps/head/san_graph.py:36-41,
    one row per graph) and their three gradients against an fp64 evaluation; exact fp32 MFMA products, so the bar is
    fp32 accumulation rounding; x taken as a strided slice; deterministic."""
    from graphgps_amd.fused import _SmallLinear, small_linear
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M * 7 + K + N)
    wide = torch.randn(M, K + 3, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    gy = torch.randn(M, N, generator=gen)
    xr, wr, br = wide[:, :K].double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    yr = yr.relu() if relu else yr
    (yr * gy.double()).sum().backward()
    xg = wide.to(dev)[:, :K].requires_grad_(True)          # row stride K + 3
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    yg = small_linear(xg, wg, bg, relu=relu)
    assert yg.grad_fn is not None and type(yg.grad_fn).__name__.startswith("_SmallLinear")
    (yg * gy.to(dev)).sum().backward()
    assert_close(yg, yr, 2e-6 * max(1.0, float(yr.abs().max())), "small linear out")
    assert_close(xg.grad, xr.grad, 2e-6, "small linear g_x", rel_to_max=True)
    assert_close(wg.grad, wr.grad, 2e-6, "small linear g_w", rel_to_max=True)
    assert_close(bg.grad, br.grad, 2e-6, "small linear g_b", rel_to_max=True)
    y2 = small_linear(xg.detach(), wg.detach(), bg.detach(), relu=relu)          # the no-grad form, same launch
    assert torch.equal(y2, yg.detach())
    y3 = small_linear(xg.detach(), wg.detach(), None, relu=relu)
    ref3 = torch.nn.functional.linear(xr.detach(), wr.detach())
    assert_close(y3, ref3.relu() if relu else ref3, 2e-6 * max(1.0, float(ref3.abs().max())), "small linear, no bias")


def test_cpu_tensor_is_rejected():
    from graphgps_amd.lib import GpsHipError
    from graphgps_amd.ops import build_graph_index
    with pytest.raises(GpsHipError):
        build_graph_index(torch.zeros(2, 3, dtype=torch.long), 4, 1, ptr_vec=torch.tensor([0, 4]))


def _favor_ref(qkv, proj, ptr, H):
    """The reference's padded path: to_dense_batch -> mask V -> softmax_kernel -> linear_attention
    -> [mask] (oracle restatement of performer_layer.py, itself pinned by the golden fixture)."""
    from oracle.gps_oracle import linear_attention, softmax_kernel, to_dense_batch
    N, inner3 = qkv.shape
    inner = inner3 // 3
    B = len(ptr) - 1
    bvec = torch.repeat_interleave(torch.arange(B), ptr[1:] - ptr[:-1])
    dense, mask = to_dense_batch(qkv, bvec, B)
    b, n = mask.shape
    q, k, v = (dense[..., j * inner:(j + 1) * inner].view(b, n, H, -1).permute(0, 2, 1, 3)
               for j in range(3))
    v = v.masked_fill(~mask[:, None, :, None], 0.0)
    out = linear_attention(softmax_kernel(q, proj, True), softmax_kernel(k, proj, False), v)
    return out.permute(0, 2, 1, 3).reshape(b, n, inner)[mask]


@pytest.mark.parametrize("H,sizes", [(2, [40, 7, 64, 33]), (4, [25, 25, 25]), (1, [130]),
                                     (2, [3, 300, 17, 1]),
                                     # BASELINE configs[4] sizes (ogbg-code2-GPS.yaml: 4 heads x dim_head 64, graphs
                                     # clipped at 1000 nodes, master_loader.py:366-368): a 3-node graph next to a
                                     # 1000-node one is 997 padded key rows short of Nmax, so the closed-form
                                     # padded-key term (performer_layer.py:485-487: v masked, k not) dominates its D
                                     (4, [1000, 3, 601]),
                                     (4, [60, 999, 1, 16]),
                                     # MalNet-sized graphs (configs/GPS/malnettiny-GPS.yaml:40 runs CustomGatedGCN+Performer
                                     # on graphs of up to ~5,000 nodes, mean ~1,400): 313 row tiles per (graph, head)
                                     (4, [5000, 1400, 37])])
def test_favor_attention_fwd_bwd(H, sizes):
    from graphgps_amd.ops import favor_attention
    from oracle.gps_oracle import gaussian_orthogonal_random_matrix
    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    dh, m = 64, 266
    proj = gaussian_orthogonal_random_matrix(m, dh)
    N, inner = sum(sizes), H * dh
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    qkv = torch.randn(N, 3 * inner, generator=gen) * 0.7
    if len(sizes) > 1:
        qkv[int(ptr[1]):int(ptr[2]), inner:2 * inner] *= 0.02   # a graph whose real key max < 0 is possible
    w = torch.randn(N, inner, generator=gen)
    qr = qkv.clone().double().requires_grad_(True)
    ref = _favor_ref(qr, proj.double(), ptr, H)
    (ref * w.double()).sum().backward()
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    gi = _index(torch.zeros(2, 0, dtype=torch.long), bvec, ptr)
    qg = qkv.cuda().requires_grad_(True)
    out = favor_attention(qg, proj.cuda(), gi, H)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "favor out")
    assert_close(qg.grad, qr.grad, Tol.GRAD_REL, "favor d_qkv", rel_to_max=True)
    out2 = favor_attention(qg.detach(), proj.cuda(), gi, H)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize("H,sizes,m", [(4, [1000, 3, 601, 17, 333], 266), (2, [60, 999, 1, 16], 100),
                                       (2, [1 + (37 * i * i + 11 * i) % 90 for i in range(300)], 100),
                                       (4, [5000, 1400, 37], 266)],
                         ids=["long", "ragged", "300-graphs", "malnet"])
def test_favor_projection_in_lds_is_bit_identical(monkeypatch, H, sizes, m):
    """csrc/favor.hip round 5: the per-tile kernels of FAVOR+ with the projection staged in LDS and one workgroup per CU
    walking the work items (LP: GPS_FAVOR_LDS=1, the default from 2,048 work items on), and with the context record of a
    (graph, head) staged next to it, a workgroup per chunk of 4 / 8 tiles of one graph (LC: GPS_FAVOR_LC=1, the default for
    long graphs), against one wavefront per (16-row tile, head) reading everything through L2: the same arithmetic in the
    same order -- outputs and every gradient element bit-identical; graphs shorter than a tile, shorter than a chunk,
    feature counts that leave the last tiles empty."""
    from graphgps_amd.ops import favor_attention
    from oracle.gps_oracle import gaussian_orthogonal_random_matrix
    gen = torch.Generator().manual_seed(23)
    torch.manual_seed(23)
    dh = 64
    proj = gaussian_orthogonal_random_matrix(m, dh).cuda()
    N, inner = sum(sizes), H * dh
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    qkv = (torch.randn(N, 3 * inner, generator=gen) * 0.7).cuda()
    w = torch.randn(N, inner, generator=gen).cuda()
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    gi = _index(torch.zeros(2, 0, dtype=torch.long), bvec, ptr)
    res = {}
    # plain / projection staged (LP) / projection + the (graph, head) record staged, chunks of 4 or 8 tiles (LC): default
    # wavefront counts, all kernels at 4 (the register-prefetch form), all at 8, 4 without the prefetch
    # (GPS_FAVOR_CTX_LDS: the two context kernels one wavefront per (graph, head, feature tile, slice) / a workgroup per
    # (graph, head, slice) with the rows staged in LDS)
    for mode, env in {"plain": dict(GPS_FAVOR_LDS="0", GPS_FAVOR_LC="0", GPS_FAVOR_CTX_LDS="0"),
                      "ctx_staged": dict(GPS_FAVOR_LDS="0", GPS_FAVOR_LC="0", GPS_FAVOR_CTX_LDS="1"),
                      "lp": dict(GPS_FAVOR_LDS="1", GPS_FAVOR_LC="0", GPS_FAVOR_CTX_LDS="0"),
                      "lc": dict(GPS_FAVOR_LDS="1", GPS_FAVOR_LC="1", GPS_FAVOR_CTX_LDS="1"),
                      "lc_plain_ctx": dict(GPS_FAVOR_LDS="1", GPS_FAVOR_LC="1", GPS_FAVOR_CTX_LDS="0")}.items():
        for k in ("GPS_FAVOR_LDS", "GPS_FAVOR_LC", "GPS_FAVOR_CTX_LDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        q = qkv.clone().requires_grad_(True)
        out = favor_attention(q, proj, gi, H)
        (out * w).sum().backward()
        res[mode] = (out.detach(), q.grad)
    for mode in res:
        assert torch.equal(res["plain"][0], res[mode][0]), mode
        assert torch.equal(res["plain"][1], res[mode][1]), mode
    assert bool(torch.isfinite(res["lc"][1]).all()) and float(res["lc"][1].abs().max()) > 0


def _drop_mask(seed, R, d, p):
    from graphgps_amd.ops import attn_dropout_keep_mask
    # the BN/elementwise kernels key the same hash by (row, column)
    return attn_dropout_keep_mask(seed, torch.arange(R), 0, 1, torch.arange(d), p)


@pytest.mark.parametrize("R,d,relu,p,with_res", [
    (7569, 384, True, 0.1, True),      # GatedGCN node stream at P30
    (15348, 384, True, 0.0, True),     # edge stream, dropout off
    (3001, 64, False, 0.0, False),     # plain norm (ZINC width)
    (257, 52, True, 0.25, False),      # d % 4 == 0 but odd sizes, one partial row block
    (130, 6, False, 0.5, True),        # float2 path
])
def test_bn_act_fused(R, d, relu, p, with_res):
    """res + dropout(relu(BatchNorm1d(z))) fwd/bwd, batch statistics and running stats, against
    torch.nn.BatchNorm1d in fp64 with the kernel's own dropout mask injected."""
    from graphgps_amd.fused import bn_act
    gen = torch.Generator().manual_seed(R + d)
    z = torch.randn(R, d, generator=gen) * 1.7 + 0.6
    res = torch.randn(R, d, generator=gen) if with_res else None
    w = torch.randn(R, d, generator=gen)
    seed = 0xABCDEF0123456789
    bn_ref = torch.nn.BatchNorm1d(d).double().train()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5, generator=gen)
        bn_ref.bias.uniform_(-0.5, 0.5, generator=gen)
    bn_gpu = torch.nn.BatchNorm1d(d)
    bn_gpu.load_state_dict({k: v.float() if v.is_floating_point() else v
                            for k, v in bn_ref.state_dict().items()})
    bn_gpu.cuda().train()
    import copy
    bn_probe = copy.deepcopy(bn_gpu)
    zg = z.cuda().requires_grad_(True)
    rg = res.cuda().requires_grad_(True) if with_res else None
    yg = bn_act(zg, bn_gpu, relu=relu, p_drop=p, res=rg, seed=seed)
    (yg * w.cuda()).sum().backward()
    with torch.no_grad():   # same op without the residual: its zeros ARE the ReLU/dropout decisions
        branch = bn_act(z.cuda(), bn_probe, relu=relu, p_drop=p, res=None, seed=seed).cpu().double()
    # fp64 reference.  The forward is checked against the true ReLU; for the BACKWARD the ReLU
    # decision the GPU actually took is injected (read off its output), because of ~3e6
    # pre-activations one or two sit within fp32 rounding of the kink, and a single flipped
    # element shifts a whole column of g_z through BatchNorm's mean terms by ~1/R.
    zr = z.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    pre = bn_ref(zr)
    keep = _drop_mask(seed, R, d, p).double() / (1 - p) if p > 0 else 1.0
    y_true = (pre.relu() if relu else pre) * keep
    if with_res:
        y_true = rr + y_true
    assert_close(yg, y_true, Tol.ACT, "bn_act out")
    if relu:
        gate = (branch != 0).double() if p == 0 else ((branch != 0) | (keep == 0)).double()
        y = pre * gate * keep
        if with_res:
            y = rr + y
    else:
        y = y_true
    (y * w.double()).sum().backward()
    assert_close(zg.grad, zr.grad, Tol.GRAD_REL, "bn_act g_z", rel_to_max=True)
    assert_close(bn_gpu.weight.grad, bn_ref.weight.grad, 1e-4, "g_gamma", rel_to_max=True)
    assert_close(bn_gpu.bias.grad, bn_ref.bias.grad, 1e-4, "g_beta", rel_to_max=True)
    if with_res:
        assert_close(rg.grad, rr.grad, Tol.GRAD_REL, "g_res", rel_to_max=True)
    assert_close(bn_gpu.running_mean, bn_ref.running_mean, 1e-6, "running_mean")
    assert_close(bn_gpu.running_var, bn_ref.running_var, 1e-5, "running_var")
    assert int(bn_gpu.num_batches_tracked) == 1


def test_bn_stats_large_mean_is_accurate():
    """Shifted/Chan-merged statistics: a column with |mean| >> std must not lose the variance."""
    from graphgps_amd.fused import bn_act
    gen = torch.Generator().manual_seed(0)
    z = torch.randn(20000, 64, generator=gen) * 0.01 + 100.0
    bn = torch.nn.BatchNorm1d(64).cuda().train()
    y = bn_act(z.cuda(), bn)
    ref = torch.nn.functional.batch_norm(z.double(), None, None, training=True)
    # the fp32 input quantum at 100 is 7.6e-6 = 7.6e-4 std, so a few e-3 is the floor for ANY fp32 BN;
    # a naive E[x^2]-E[x]^2 reduction is off by O(1) here
    assert_close(y, ref, 6e-3, "normalised output at mean/std = 1e4")


@pytest.mark.parametrize("relu,with_a", [(False, True), (True, False)])
def test_act_drop_add(relu, with_a):
    from graphgps_amd.fused import add_dropout, relu_dropout
    gen = torch.Generator().manual_seed(4)
    R, d, p, seed = 7569, 768 if relu else 384, 0.1, 77
    a, b, w = (torch.randn(R, d, generator=gen) for _ in range(3))
    keep = _drop_mask(seed, R, d, p).double()
    br = b.double().requires_grad_(True)
    ref = (br.relu() if relu else br) * keep / (1 - p)
    if with_a:
        ref = a.double() + ref
    (ref * w.double()).sum().backward()
    bg = b.cuda().requires_grad_(True)
    out = relu_dropout(bg, p, True, seed=seed) if relu else add_dropout(a.cuda(), bg, p, True, seed=seed)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "act_drop_add out")
    assert_close(bg.grad, br.grad, Tol.GRAD_REL, "act_drop_add g_b", rel_to_max=True)


@pytest.mark.parametrize("R,M,Nn", [(7569, 384, 384), (15348, 384, 384), (7569, 1536, 384),
                                    (7569, 384, 768), (738, 64, 64), (1000, 100, 52), (3, 8, 4)])
def test_wgrad_split_k(R, M, Nn):
    """gW = g^T x and gb = colsum(g) from the split-K MFMA kernel vs fp64 (1e-5 of max|gW|)."""
    from graphgps_amd import lib as L_
    from graphgps_amd.lib import check, current_stream, ptr
    L = L_.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(R + M)
    g, x = torch.randn(R, M, generator=gen), torch.randn(R, Nn, generator=gen)
    gd, xd = g.to(dev), x.to(dev)
    gw, gb = torch.empty(M, Nn, device=dev), torch.empty(M, device=dev)
    ws = torch.empty(max(L.gps_wgrad_workspace_floats(R, M, Nn), 4), device=dev)
    check(L.gps_wgrad(ptr(gd), M, ptr(xd), Nn, R, M, Nn, ptr(gw), ptr(gb), ptr(ws), current_stream(dev)))
    assert_close(gw, g.double().t() @ x.double(), Tol.GRAD_REL, "gW", rel_to_max=True)
    assert_close(gb, g.double().sum(0), Tol.GRAD_REL, "gb", rel_to_max=True)
    gw2 = torch.empty_like(gw)
    check(L.gps_wgrad(ptr(gd), M, ptr(xd), Nn, R, M, Nn, ptr(gw2), None, ptr(ws), current_stream(dev)))
    assert torch.equal(gw, gw2)     # deterministic, and the bias output is optional
    # fp16 form (round 4): the same bound, with the operands' max|.| words; gradient-sized g (1e-7) as well
    from graphgps_amd.gemm import absmax
    for gs in (1.0, 1e-7):
        g3 = gd * gs
        words = absmax([g3, xd])
        gw3, gb3 = torch.empty_like(gw), torch.empty_like(gb)
        check(L.gps_wgrad16(ptr(g3), M, ptr(xd), Nn, R, M, Nn, ptr(words[0]), ptr(words[1]), ptr(gw3), ptr(gb3), ptr(ws),
                            current_stream(dev)))
        assert_close(gw3 / gs, g.double().t() @ x.double(), Tol.GRAD_REL, "gW (fp16 form)", rel_to_max=True)
        assert_close(gb3 / gs, g.double().sum(0), Tol.GRAD_REL, "gb (fp16 form)", rel_to_max=True)


@pytest.mark.parametrize("f16", [True, False], ids=["f16x3", "bf16x6"])
def test_wgrad_grouped_and_bf16_split_exactness(f16):
    """(1) The grouped launch returns, per problem, what the single-problem launch returns to fp32
    rounding.  (2) The contraction runs on the bf16 pipe through an exact 3-way split (or, round 4, on the fp16 pipe through
    two fp16 pieces under a power-of-two scale): its error against
    fp64 must not exceed that of an fp32 GEMM (torch.mm through rocBLAS) on the same data -- including
    data with a large common offset, where a lossy split would show."""
    from graphgps_amd import lib as L_
    from graphgps_amd.gemm import absmax
    from graphgps_amd.lib import check, current_stream, ptr
    L = L_.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    shapes = [(3001, 384, 2688), (4002, 384, 384), (3001, 768, 384)]          # (rows, in, out)
    pairs = [((torch.randn(R, n, generator=gen) * 3 + 100.0).to(dev), (torch.randn(R, k, generator=gen) + 7.0).to(dev))
             for R, k, n in shapes]
    probs = (L_.WgradProblem * len(pairs))()
    outs, keep = [], []
    for q, (g, x) in zip(probs, pairs):
        gw, gb = torch.empty(g.shape[1], x.shape[1], device=dev), torch.empty(g.shape[1], device=dev)
        q.g, q.x, q.gw, q.gb = g.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr()
        q.ldg, q.ldx, q.R, q.M, q.Nn = g.stride(0), x.stride(0), g.shape[0], g.shape[1], x.shape[1]
        if f16:
            w = absmax([g, x])
            keep.append(w)
            q.g_amax, q.x_amax = w[0].data_ptr(), w[1].data_ptr()
        outs.append((gw, gb))
    ws = torch.empty(max(L.gps_wgrad_grouped_workspace_floats(len(pairs), probs), 4), device=dev)
    check(L.gps_wgrad_grouped(len(pairs), probs, ptr(ws), current_stream(dev)), "gps_wgrad_grouped")
    for (g, x), (gw, gb) in zip(pairs, outs):
        ref = g.double().t() @ x.double()
        scale = ref.abs().max()
        err = ((gw.double() - ref).abs().max() / scale).item()
        err_lib = ((g.t().mm(x).double() - ref).abs().max() / scale).item()
        assert err <= max(1.5 * err_lib, 2e-6), (err, err_lib)
        assert_close(gb, g.double().sum(0), Tol.GRAD_REL, "gb", rel_to_max=True)


@pytest.mark.parametrize("R,K,M", [(7569, 384, 2688), (7569, 2688, 384), (15348, 384, 384), (1000, 52, 100),
                                   (37, 64, 20), (129, 4, 130)])
def test_gemm_nt_bf16_split(R, K, M):
    """C = A B^T + bias + Cin on the bf16 pipe through the exact 3-way split (csrc/gemm_split.hip): error
    against fp64 no larger than the fp32 library GEMM's; partial tiles, K tails, in-place residual."""
    from graphgps_amd import lib as L_
    from graphgps_amd.lib import check, current_stream, ptr
    L = L_.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(R * 7 + K)
    a = (torch.randn(R, K, generator=gen) + 2.0).to(dev)
    b = (torch.randn(M, K, generator=gen) / K ** 0.5).to(dev)
    bias, cin = torch.randn(M, generator=gen).to(dev), torch.randn(R, M, generator=gen).to(dev)
    ref = a.double() @ b.double().t() + bias.double() + cin.double()
    out = cin.clone()                                      # C aliases Cin (residual accumulate in place)
    check(L.gps_gemm_nt(ptr(a), K, ptr(b), K, R, M, K, ptr(bias), ptr(out), M, ptr(out), M,
                        current_stream(dev)), "gps_gemm_nt")
    scale = ref.abs().max()
    err = ((out.double() - ref).abs().max() / scale).item()
    err_lib = (((torch.addmm(bias, a, b.t()) + cin).double() - ref).abs().max() / scale).item()
    assert err <= max(1.5 * err_lib, 2e-6), (err, err_lib)
    plain = torch.empty(R, M, device=dev)
    check(L.gps_gemm_nt(ptr(a), K, ptr(b), K, R, M, K, None, None, 0, ptr(plain), M, current_stream(dev)))
    assert_close(plain, a.double() @ b.double().t(), Tol.GRAD_REL, "A B^T", rel_to_max=True)


def _rw_landing_probs_ref(ksteps, edge_index, num_nodes, space_dim=0):
    """graphgps/transform/posenc_stats.py:184-230 restated with plain dense torch ops in fp64 (the
    reference uses torch_scatter.scatter + PyG to_dense_adj, absent here: multi-edges ADD in the dense
    adjacency, out-degree normalisation, 1/0 -> 0)."""
    src, dst = edge_index[0], edge_index[1]
    A = torch.zeros(num_nodes, num_nodes, dtype=torch.float64)
    A.index_put_((src, dst), torch.ones(src.numel(), dtype=torch.float64), accumulate=True)
    deg = torch.zeros(num_nodes, dtype=torch.float64).index_add_(0, src, torch.ones(src.numel(), dtype=torch.float64))
    deg_inv = deg.pow(-1.0)
    deg_inv[deg_inv == float("inf")] = 0
    P = torch.diag(deg_inv) @ A
    rws = []
    Pk = torch.linalg.matrix_power(P, min(ksteps))
    for k in range(min(ksteps), max(ksteps) + 1):
        rws.append(torch.diagonal(Pk) * (k ** (space_dim / 2)))
        Pk = Pk @ P
    full = torch.stack(rws, 1)
    return full[:, [k - min(ksteps) for k in ksteps]]


@pytest.mark.parametrize("ksteps,space_dim", [(list(range(1, 17)), 0), (list(range(1, 21)), 0),
                                              ([2, 3, 4], 2.0), ([1, 4, 9], 0)])
def test_rwse_landing_probabilities(ksteps, space_dim):
    """Batched RWSE kernel vs the per-graph restatement of get_rw_landing_probs: molecule-like graphs,
    a single-node graph, isolated nodes, a multi-edge, a directed (asymmetric) edge, and one graph larger
    than the LDS limit (global-scratch path)."""
    from graphgps_amd.transform import rw_landing_probs
    dev = torch.device("cuda:0")
    b = _ragged_batch_ops([1, 17, 2, 33, 5, 64, 140, 9], seed=4)
    ei, ptr = b["edge_index"], b["ptr"]
    got = rw_landing_probs(ksteps, ei.to(dev), ptr.to(dev), space_dim=space_dim).cpu()
    assert got.shape == (int(ptr[-1]), len(ksteps))
    for g in range(len(ptr) - 1):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        m = (ei[0] >= n0) & (ei[0] < n1)
        ref = _rw_landing_probs_ref(ksteps, ei[:, m] - n0, n1 - n0, space_dim)
        assert_close(got[n0:n1], ref, 1e-6, f"graph {g} (n={n1 - n0})")


def _ragged_batch_ops(sizes, seed):
    gen = torch.Generator().manual_seed(seed)
    src, dst, ptr = [], [], [0]
    for gi, n in enumerate(sizes):
        base = ptr[-1]
        for v in range(1, n):
            if gi == 4 and v == n - 1:
                continue                                   # leaves an isolated node in graph 4
            u = int(torch.randint(max(0, v - 4), v, (1,), generator=gen))
            src += [base + u, base + v]
            dst += [base + v, base + u]
        if n > 5:
            src += [base, base + n - 1, base, base + 1]    # chord both ways + a multi-edge + one directed edge
            dst += [base + n - 1, base, base + 1, base + 3]
        ptr.append(base + n)
    return dict(edge_index=torch.tensor([src, dst], dtype=torch.int64), ptr=torch.tensor(ptr, dtype=torch.int64))


def _gcn_dense_ref(x, edge_index):
    """D^-1/2 (A + I) D^-1/2 x with gcn_norm's conventions, as a dense fp64 matrix product."""
    n = x.shape[0]
    A = torch.zeros(n, n, dtype=torch.float64)
    keep = edge_index[0] != edge_index[1]
    A.index_put_((edge_index[1][keep], edge_index[0][keep]), torch.ones(int(keep.sum()), dtype=torch.float64),
                 accumulate=True)                      # A[target, source], duplicates add up
    A += torch.eye(n, dtype=torch.float64)             # exactly one unit loop per node
    dinv = A.sum(1).pow(-0.5)
    return (dinv[:, None] * A * dinv[None, :]) @ x.double()


@pytest.mark.parametrize("d", [64, 50, 7])
def test_gcn_aggregate(d):
    """csrc/gcn.hip vs the dense normalised-adjacency product (PyG GCNConv.propagate after gcn_norm): directed
    edges, duplicate edges, input self loops (replaced by one unit loop), isolated nodes; forward and the
    transposed product the backward runs."""
    from graphgps_amd.ops import gcn_aggregate
    gen = torch.Generator().manual_seed(3)
    sizes = [40, 1, 17, 90]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    N = int(ptr[-1])
    parts = []
    for g, n in enumerate(sizes):
        if n == 1:
            continue
        m = 3 * n
        ei = torch.randint(0, n - 1 if g == 3 else n, (2, m), generator=gen)   # last node of graph 3 isolated
        ei = torch.cat([ei, ei[:, :5], torch.arange(4).repeat(2, 1)], dim=1)   # 5 duplicates + 4 self loops
        parts.append(ei + int(ptr[g]))
    ei = torch.cat(parts, dim=1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=gen)]
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(N, d, generator=gen)
    w = torch.randn(N, d, generator=gen)
    xr = x.clone().double().requires_grad_(True)
    ref = _gcn_dense_ref(xr, ei)
    (ref * w.double()).sum().backward()
    gi = _index(ei, bvec, ptr)
    xg = x.cuda().requires_grad_(True)
    out = gcn_aggregate(xg, gi)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, Tol.ACT, "gcn out")
    assert_close(xg.grad, xr.grad, Tol.GRAD_REL, "gcn d_x", rel_to_max=True)
    out2 = gcn_aggregate(x.cuda(), gi)
    assert torch.equal(out2, out.detach())            # fixed reduction order: bitwise reproducible


def test_gin_aggregate_matches_index_add():
    """csrc/gcn.hip:k_adj_sum = PyG GINConv's (1 + eps) x_i + sum_{j->i} x_j over EVERY stored edge (self loops
    and duplicates count), forward and transpose, on the wide [N, k * C] feature layout SignNet uses."""
    from graphgps_amd.ops import gin_aggregate
    gen = torch.Generator().manual_seed(5)
    sizes = [30, 2, 51]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    N = int(ptr[-1])
    parts = []
    for g, n in enumerate(sizes):
        ei = torch.randint(0, n, (2, 3 * n), generator=gen)
        parts.append(torch.cat([ei, ei[:, :4]], dim=1) + int(ptr[g]))        # duplicates; random self loops
    ei = torch.cat(parts, dim=1)
    bvec = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    for d, eps in ((36, 0.0), (5, 0.25)):
        x = torch.randn(N, d, generator=gen)
        w = torch.randn(N, d, generator=gen)
        xr = x.clone().double().requires_grad_(True)
        ref = ((1 + eps) * xr).index_add(0, ei[1], xr.index_select(0, ei[0]))
        (ref * w.double()).sum().backward()
        gi = _index(ei, bvec, ptr)
        xg = x.cuda().requires_grad_(True)
        out = gin_aggregate(xg, gi, eps)
        (out * w.cuda()).sum().backward()
        assert_close(out, ref, Tol.ACT, "gin out")
        assert_close(xg.grad, xr.grad, Tol.GRAD_REL, "gin d_x", rel_to_max=True)


_WGRAD_ID_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from graphgps_amd import lib as L_, gemm as G
from graphgps_amd.lib import check, current_stream
dev = torch.device('cuda:0'); L = L_.load()
gen = torch.Generator().manual_seed(1)
out = []
for (R, M, Nn) in ((7569, 2688, 384), (15348, 384, 384), (7569, 384, 768), (1000, 128, 128), (333, 256, 128), (64, 128, 128), (70, 128, 256)):
    g = torch.randn(R, M, generator=gen).to(dev); x = torch.randn(R, Nn, generator=gen).to(dev)
    w = G.absmax([g, x])
    gw = torch.empty(M, Nn, device=dev); gb = torch.empty(M, device=dev)
    ws = torch.empty(max(L.gps_wgrad_workspace_floats(R, M, Nn), 4), device=dev)
    check(L.gps_wgrad16(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), R, M, Nn, w[0].data_ptr(), w[1].data_ptr(),
                        gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), current_stream(dev)), 'gps_wgrad16')
    torch.cuda.synchronize()
    ref = g.double().t() @ x.double()
    assert float((gw.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((gb.double() - g.double().sum(0)).abs().max() / g.double().sum(0).abs().max()) < 2e-6
    out.append((gw.cpu(), gb.cpu()))
torch.save(out, sys.argv[1])
"""


def test_wgrad_register_path_is_bit_identical_to_the_lds_path(tmp_path):
    """csrc/wgrad.hip round 6: k_wgrad_direct (operands through registers, fragments = the four values of a 16-byte load,
    outputs stored as the permutation that implies) against k_wgrad_stream (LDS-DMA ring + ds_read fragments,
    GPS_WGRAD_DIRECT=0): every output element is the same sum of the same products in the same order -- weight and bias
    gradients bit-identical at the block's shapes, short slices and ragged row counts (the tail stage); both within 2e-6 of
    fp64.  The switch is read once per process, so each form runs in a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "wg.py"
    script.write_text(_WGRAD_ID_SCRIPT)
    outs = []
    for v in ("1", "0"):
        f = tmp_path / f"wg{v}.pt"
        env = dict(os.environ, GPS_WGRAD_DIRECT=v)
        r = subprocess.run([sys.executable, str(script), str(f), root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for (gw1, gb1), (gw0, gb0) in zip(*outs):
        assert torch.equal(gw1, gw0) and torch.equal(gb1, gb0)


@pytest.mark.parametrize("p0,p1,cin", [
    ((7569, 384, 2688), (15348, 384, 384), False),      # the block's forward pair: merged node projection | C(e); 1,080 tiles
                                                        # on 256 CUs: the last 56 re-cut as 64-row tiles (tail balancing)
    ((7569, 384, 2688), (15301, 384, 384), True),       # the same with addends and a ragged last 64-row tile
    ((5000, 384, 768), (9000, 384, 384), True),         # 160 + 142 tiles: a tail of 46 behind one full round
    ((7569, 2688, 384), (15348, 384, 384), True),       # its backward pair: g_pq Wcat (64-row tiles) | g_ce W_C (128-row tiles)
    ((300, 384, 384), (5000, 384, 768), True), ((25013, 256, 256), (7000, 256, 1792), False),     # 128-column panels
    ((1000, 384, 384), (1000, 256, 256), False),        # different panel widths: two launches behind the same call
    ((743, 304, 304), (900, 384, 384), False),          # an edge shape: two launches
    ((64, 384, 384), (0, 384, 384), False)])            # an empty problem
def test_gemm_panel_pair_is_bit_identical(p0, p1, cin):
    """gps_gemm16_panel_pair (two independent products in ONE dispatch, csrc/gemm_panel.hip k_gemm_ring16_pair) against the
    two gps_gemm16_panel launches it replaces: the same tiles run the same code, so the results are bit-identical --
    including the max|C| records and in-place accumulation into the addend."""
    from graphgps_amd.gemm import absmax, amax_records, gemm_panel, gemm_panel_pair, split_weights
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(sum(p0) + sum(p1))
    probs = []
    for (M, K, N) in (p0, p1):
        a = torch.randn(M, K, generator=gen).to(dev)
        w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=gen).to(dev)
        add = torch.randn(M, N, generator=gen).to(dev) if cin else None
        (img, _), = split_weights([w], tn=False, f16=True)
        rec = amax_records(1, dev)
        if M:
            absmax([a], out=rec)
        probs.append((a, img, N, b, add, rec[0]))
    singles, recs1 = [], amax_records(2, dev)
    for i, (a, img, N, b, add, rec) in enumerate(probs):
        singles.append(gemm_panel(a, img, N, bias=b, addend=add.clone() if cin else None, a_amax=rec, c_amax=recs1[i]))
    recs2 = amax_records(2, dev)
    outs = [q[4].clone() if cin else None for q in probs]
    pair = gemm_panel_pair(*[dict(a=a, image=img, N=N, bias=b, addend=outs[i], out=outs[i], a_amax=rec, c_amax=recs2[i])
                             for i, (a, img, N, b, add, rec) in enumerate(probs)])
    torch.cuda.synchronize()
    for i in range(2):
        assert pair[i].shape == singles[i].shape
        assert torch.equal(pair[i], singles[i]), f"problem {i}: {float((pair[i] - singles[i]).abs().max()):.3e}"
        if cin:
            assert pair[i].data_ptr() == outs[i].data_ptr()
    assert torch.equal(recs1.view(2, -1).max(dim=1).values, recs2.view(2, -1).max(dim=1).values)


_PERSIST_ID_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from graphgps_amd.gemm import absmax, amax_records, gemm_panel, gemm_panel_pair, split_weights
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(7)
out = []
# (M, K, N, epilogue, addend): FF1 and its input gradient at the AST block's size, the molecule block's merged projection,
# a 16-stage contraction with an addend -- every one two dispatch rounds or more of 128-row tiles
for (M, K, N, epi, cin) in ((25365, 256, 512, 1, False), (25365, 512, 256, 0, True), (25365, 256, 512, 2, False),
                            (7569, 384, 2688, 0, False), (40000, 512, 768, 2, True), (33000, 384, 768, 1, False)):
    a = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    add = torch.randn(M, N, generator=gen).to(dev) if cin else None
    mask = torch.randn(M, N, generator=gen).to(dev) if epi == 2 else None
    (img, _), = split_weights([w], tn=False, f16=True)
    rec, crec = amax_records(1, dev), amax_records(1, dev)
    absmax([a], out=rec)
    c = gemm_panel(a, img, N, bias=b, addend=add, epilogue=epi, mask_src=mask, p_drop=0.1 if epi else 0.0, seed=99,
                   a_amax=rec[0], c_amax=crec[0])
    torch.cuda.synchronize()
    if epi == 0:
        ref = a.double() @ w.double().t() + b.double() + (add.double() if cin else 0)
        assert float((c.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    out.append((c.cpu(), crec.cpu()))
# the pair dispatch WITH max|C| records (the first problem persistent)
probs = []
for (M, K, N) in ((25365, 256, 1792), (75856, 256, 256)):
    a = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    (img, _), = split_weights([w], tn=False, f16=True)
    rec = amax_records(1, dev)
    absmax([a], out=rec)
    probs.append(dict(a=a, image=img, N=N, a_amax=rec[0], c_amax=amax_records(1, dev)[0]))
pair = gemm_panel_pair(*probs)
torch.cuda.synchronize()
out += [(pair[i].cpu(), probs[i]['c_amax'].cpu()) for i in range(2)]
torch.save(out, sys.argv[1])
"""


def test_gemm_persistent_tiles_are_bit_identical_to_one_workgroup_per_tile(tmp_path):
    """csrc/gemm_panel.hip ring16_body PERSIST -- one workgroup per CU walking the tiles of a launch of two dispatch rounds or
    more, the k-loop's past-the-end DMA fetching the next tile's first stages -- against the one-workgroup-per-tile launches
    (GPS_GEMM_SCHED=1): epilogues 0 / 1 (ReLU + dropout) / 2 (mask of a saved activation), addends in place, max|C| records
    (one atomic per workgroup from the running maximum over its tiles), and the pair dispatch with records.  Same tiles, same
    arithmetic: outputs AND records bit-identical.  The switch is read once per process: each form runs in a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "pt.py"
    script.write_text(_PERSIST_ID_SCRIPT)
    outs = []
    for v in ("3", "1"):
        f = tmp_path / f"pt{v}.pt"
        r = subprocess.run([sys.executable, str(script), str(f), root], env=dict(os.environ, GPS_GEMM_SCHED=v),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for i, ((c1, r1), (c0, r0)) in enumerate(zip(*outs)):
        assert torch.equal(c1, c0), f"case {i}: outputs differ by {float((c1 - c0).abs().max()):.3e}"
        assert torch.equal(r1.max(), r0.max()), f"case {i}: max|C| records differ"


@pytest.mark.parametrize("p0,p1,cin", [
    ((7569, 384, 2688), (15348, 384, 384), False),      # the molecule block's forward pair: 840 tiles of 12 k-stages on a 4-slot ring
    ((25365, 256, 1792), (75856, 256, 256), False),     # the AST block's: 2,786 tiles of 8 k-stages, 128-column panels (4 of 5 slots)
    ((25365, 256, 1792), (9000, 256, 256), True),       # with addends; a short second problem behind the persistent one
    ((40000, 512, 1024), (3000, 512, 256), False)])     # 16 k-stages
def test_gemm_panel_pair_persistent_tiles_bit_identical(p0, p1, cin):
    """The pair dispatch WITHOUT max|C| records -- the form the blocks' forward pair runs: the first problem's tiles are walked by
    one persistent workgroup per CU when they span two dispatch rounds or more (csrc/gemm_panel.hip ring16_body PERSIST: the
    k-loop's past-the-end DMA fetches the next tile's first stages) -- against the two single launches: same tiles, same
    arithmetic, bit-identical."""
    from graphgps_amd.gemm import absmax, amax_records, gemm_panel, gemm_panel_pair, split_weights
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(sum(p0) + sum(p1) + 1)
    probs = []
    for (M, K, N) in (p0, p1):
        a = torch.randn(M, K, generator=gen).to(dev)
        w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=gen).to(dev)
        add = torch.randn(M, N, generator=gen).to(dev) if cin else None
        (img, _), = split_weights([w], tn=False, f16=True)
        rec = amax_records(1, dev)
        absmax([a], out=rec)
        probs.append((a, img, N, b, add, rec[0]))
    singles = [gemm_panel(a, img, N, bias=b, addend=add.clone() if cin else None, a_amax=rec) for a, img, N, b, add, rec in probs]
    outs = [q[4].clone() if cin else None for q in probs]
    pair = gemm_panel_pair(*[dict(a=a, image=img, N=N, bias=b, addend=outs[i], out=outs[i], a_amax=rec)
                             for i, (a, img, N, b, add, rec) in enumerate(probs)])
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(pair[i], singles[i]), f"problem {i}: {float((pair[i] - singles[i]).abs().max()):.3e}"


@pytest.mark.parametrize("M,K,N", [(7569, 384, 2688), (1000, 384, 384), (15348, 384, 384), (7569, 768, 384),
                                   (333, 2688, 384), (64, 128, 192), (65, 256, 768),
                                   # round 3: 128- and 64-column panels, any number of k-stages (d = 256: code2 / GPS-deep;
                                   # d = 64: ZINC)
                                   (7569, 256, 1792), (7569, 1792, 256), (25013, 256, 256), (1000, 256, 512),
                                   (1000, 512, 256), (743, 64, 448), (743, 448, 64), (743, 64, 64), (130, 32, 64),
                                   (130, 64, 128), (300, 160, 192), (2000, 608, 1216),
                                   # EDGE variants: a partial last column panel and / or a half-empty last k-stage
                                   # (d = 304: GPS-small PCQM4Mv2 / peptides; 96, 48, 16: narrow configs)
                                   (7569, 304, 2128), (7569, 2128, 304), (2000, 304, 304), (2000, 304, 608),
                                   (2000, 608, 304), (1000, 304, 1216), (743, 96, 672), (743, 96, 96), (743, 48, 48),
                                   (300, 48, 336), (130, 16, 16), (500, 384, 80), (500, 80, 384),
                                   # N, K multiples of 4 only (d = 52: ZINC GPS-small variants, 72: peptides SAN/GPS)
                                   (500, 52, 364), (500, 364, 52), (743, 72, 72), (300, 52, 52), (500, 20, 36)])
@pytest.mark.parametrize("f16", [True, False], ids=["f16x3", "bf16x6"])
def test_gemm_panel_split_products(M, K, N, f16):
    """csrc/gemm_panel.hip at the block's projection shapes (N, K in {384, 768, 2688}; M = nodes / edges, ragged last
    row tile): C = A W^T + bias against fp64, through the weight image (forward) and through the transposed image
    (input gradient), with the addend and both epilogues; the error is that of an fp32 GEMM (a few 1e-7 of the
    result scale), also with a +100 offset on the operands where a lossy split would show at 1e-3.  Both arithmetic
    forms (round 4: two fp16 pieces / 3 products, the default; round 2: three bf16 pieces / 6 products) meet the SAME
    bounds."""
    import functools
    from graphgps_amd import gemm as _g
    _sw = functools.partial(_g.split_weights, f16=f16)
    from graphgps_amd.gemm import gemm_panel, split_weights
    from graphgps_amd.ops import attn_dropout_keep_mask
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    add = torch.randn(M, N, generator=gen)
    ag, wg, bg = a.to(dev), w.to(dev), b.to(dev)
    from graphgps_amd.gemm import supported as _sup
    (img_nt, img_tn), = _sw([wg], tn=_sup(K, N))
    ref = a.double() @ w.double().t()
    scale = float(ref.abs().max())
    # yardstick: the library's own fp32 GEMM on the same operands (an fp32 dot product of K terms carries
    # ~sqrt(K) * 6e-8 * sum|a||b|, more than 2e-6 of max|result| once K reaches the hundreds)
    lib_err = float((torch.mm(ag, wg.t()).double().cpu() - ref).abs().max())
    tol = max(4e-6 * scale, 1.5 * lib_err)     # K = 2688: one fp32 accumulation chain of 1008 MFMA steps
    out = gemm_panel(ag, img_nt, N)
    err = float((out.double().cpu() - ref).abs().max())
    print(f"gemm_panel {M}x{K}x{N}: max err {err:.2e} (library fp32 GEMM {lib_err:.2e}, scale {scale:.2f})")
    assert err <= tol, (err, lib_err)
    out = gemm_panel(ag, img_nt, N, bias=bg, addend=add.to(dev))
    assert_close(out, ref + b.double() + add.double(), tol, "A W^T + b + C")
    # the transposed image: G [M, N] @ W [N, K]
    g = torch.randn(M, N, generator=gen)
    gref = g.double() @ w.double()
    from graphgps_amd.gemm import supported
    tn_ok = supported(K, N)                      # the transposed image serves G [M, N] x W [N, K]: "N" = K, "K" = N
    gout = gemm_panel(g.to(dev), img_tn, K) if tn_ok else None
    if gout is not None:
        gtol = max(2e-6 * float(gref.abs().max()),
                   1.5 * float((torch.mm(g.to(dev), wg).double().cpu() - gref).abs().max()))
        assert_close(gout, gref, gtol, "G W")
    # in-place accumulation into a column slice of a wider buffer (the block's g_x += g_pq W pattern)
    if tn_ok:
        wide = torch.zeros(M, K + 64, device=dev)
        wide[:, :K] = 1.0
        gemm_panel(g.to(dev), img_tn, K, addend=wide[:, :K], out=wide[:, :K])
        assert_close(wide[:, :K], gref + 1.0, gtol, "accumulate in place")
        assert float(wide[:, K:].abs().max()) == 0.0
    # epilogue 1: relu + dropout keyed (row, col) like gps_act_drop_add; epilogue 2: the mask of a saved activation
    p, seed = 0.25, 0x1234ABCD5678
    keep = attn_dropout_keep_mask(seed, torch.arange(M), 0, 1, torch.arange(N), p).double()
    t_ref = (ref + b.double()).clamp(min=0) * keep / (1 - p)
    t = gemm_panel(ag, img_nt, N, bias=bg, epilogue=1, p_drop=p, seed=seed)
    assert_close(t, t_ref, 2 * tol, "relu + dropout epilogue")
    m_ref = ref * (t_ref > 0) * keep / (1 - p)
    mo = gemm_panel(ag, img_nt, N, epilogue=2, mask_src=t, p_drop=p, seed=seed)
    settled = ((ref + b.double()).abs() > 1e-4).double()      # the ReLU side of a pre-activation within rounding of 0
    assert_close(mo.double().cpu() * settled, m_ref * settled, 2 * tol, "mask epilogue")   # is anybody's guess
    # operands with a common offset (mean / std = 3).  The split itself is exact; what grows with the offset is the
    # MFMA's own accumulation error: every product is TRUNCATED against the running accumulator (measured, for the
    # fp32-input MFMA of the library GEMM as well: error ~ 0.28 ulp(acc) per product), and the 6-term form adds 6x
    # as many products per k-step.  For the block's operands (BatchNorm / ReLU outputs, weights: |mean| <~ std) this
    # sits at the library's own ~1e-6; at mean/std = 100 it reaches 2e-4 of the result (library: 3e-5).
    a2, w2 = a + 3.0, w + 3.0 / K ** 0.5
    (i2, _), = _sw([w2.to(dev)], tn=False)
    ref2 = a2.double() @ w2.double().t()
    out2 = gemm_panel(a2.to(dev), i2, N)
    lib2 = float((torch.mm(a2.to(dev), w2.to(dev).t()).double().cpu() - ref2).abs().max())
    err2 = float((out2.double().cpu() - ref2).abs().max())
    print(f"  offset operands: max err {err2:.2e} (library {lib2:.2e}, scale {float(ref2.abs().max()):.1f})")
    assert err2 <= max(4e-6 * float(ref2.abs().max()), 16.0 * lib2), (err2, lib2)


@pytest.mark.parametrize("softmax", [False, True])
@pytest.mark.parametrize("H,D,profile,nb", [(4, 8, "P14", 12), (8, 16, "CODE2_REAL", 3), (2, 64, "ZINC", 5), (3, 4, "P30", 4)])
def test_edge_attention_real_edges(H, D, profile, nb, softmax):
    """ops.edge_attention (csrc/edge_attn.hip) vs the reference formulas in fp64 (san_layer.py:44-92,
    san2_layer.py:11-33,65-105): per-head K.Q.E score, clamp-exp or per-target softmax (eps 1e-16), weighted sum of
    V[src] and of the weights; isolated targets get zeros; all four input gradients."""
    from graphgps_amd.ops import edge_attention
    sizes, ei, bvec, ptr, gen = _structure(profile, nb, 5)
    N, E, HD = int(ptr[-1]), ei.shape[1], H * D
    q, k, v = (torch.randn(N, HD, generator=gen) for _ in range(3))
    e = torch.randn(E, HD, generator=gen)
    wgt, wz = torch.randn(N, HD, generator=gen), torch.randn(N, H, generator=gen)
    ref_in = [t.clone().double().requires_grad_(True) for t in (q, k, v, e)]
    qr, kr, vr, er = (t.view(-1, H, D) for t in ref_in)
    s = (kr[ei[0]] * qr[ei[1]] * er / D ** 0.5).sum(-1, keepdim=True)
    if softmax:
        idx = ei[1].view(-1, 1, 1).expand_as(s)
        top = torch.full((N, H, 1), float("-inf"), dtype=torch.float64).scatter_reduce(0, idx, s, "amax")
        ex = (s - top[ei[1]]).exp()
        w = ex / (torch.zeros(N, H, 1, dtype=torch.float64).index_add_(0, ei[1], ex)[ei[1]] + 1e-16)
    else:
        w = torch.exp(s.clamp(-5, 5))
    wv_ref = torch.zeros(N, H, D, dtype=torch.float64).index_add_(0, ei[1], vr[ei[0]] * w).view(N, HD)
    z_ref = torch.zeros(N, H, 1, dtype=torch.float64).index_add_(0, ei[1], w).view(N, H)
    loss = (wv_ref * wgt.double()).sum() + (0.0 if softmax else (z_ref * wz.double()).sum())
    loss.backward()
    gi = _index(ei, bvec, ptr)
    dev_in = [t.cuda().requires_grad_(True) for t in (q, k, v, e)]
    wv, z = edge_attention(*dev_in, gi, H, softmax)
    lg = (wv * wgt.cuda()).sum() + (0.0 if softmax else (z * wz.cuda()).sum())
    lg.backward()
    assert_close(wv, wv_ref, Tol.ACT * max(1.0, float(wv_ref.abs().max())), "wv")
    if not softmax:
        assert_close(z, z_ref, Tol.ACT * max(1.0, float(z_ref.abs().max())), "z")
    for name, a_, b_ in zip("QKVE", dev_in, ref_in):
        assert_close(a_.grad, b_.grad, Tol.GRAD_REL, f"grad {name}", rel_to_max=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mag", [1e-30, 3e-9, 1.0, 7e5, 2e30])
def test_gemm16_operand_scales(mag):
    """fp16 form of the ring GEMM: the per-tensor power-of-two scale keeps the result at fp32 grade whatever the
    magnitude of the operands (gradient-sized 3e-9, near-denormal 1e-30, huge 2e30), for rows far below the tensor's
    maximum (absolute error stays relative to the maximum), all-zero operands, and a word LARGER than the true maximum;
    ``absmax`` itself is bit-exact against torch."""
    from graphgps_amd import gemm as _g
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(31)
    M, K, N = 1000, 384, 384
    a = torch.randn(M, K, generator=gen) * mag
    a[::7] *= 1e-4                                  # rows far below the maximum
    a[5] = 0.0
    w = torch.randn(N, K, generator=gen) / K ** 0.5 * (1.0 / mag if 1e-20 < mag < 1e20 else 1.0)
    ag, wg = a.to(dev), w.to(dev)
    words = _g.absmax([ag, wg, ag[:, :128]])            # [3, 512] records: the maximum is the max over a row
    want = torch.stack([ag.abs().max(), wg.abs().max(), ag[:, :128].abs().max()]).view(torch.int32)
    assert torch.equal(words.max(dim=1).values.cpu(), want.cpu())
    assert _g.amax_value(words[1]) == float(wg.abs().max())
    (img, _), = _g.split_weights([wg], tn=False, f16=True)
    ref = a.double() @ w.double().t()
    scale = float(ref.abs().max())
    out = _g.gemm_panel(ag, img, N)
    err = float((out.double().cpu() - ref).abs().max())
    lib_err = float((torch.mm(ag, wg.t()).double().cpu() - ref).abs().max())
    print(f"gemm16 |a| ~ {mag:g}: max err {err:.2e} (library {lib_err:.2e}, scale {scale:.3g})")
    assert err <= max(4e-6 * scale, 1.5 * lib_err)
    assert float(out[5].abs().max()) == 0.0
    # a word twice / 64 times the true maximum: one / six bits of precision less, never a wrong result
    for k, bound in ((1, 8e-6), (6, 2e-4)):
        big = _g.record_of(float(ag.abs().max()) * 2.0 ** k, dev)
        out2 = _g.gemm_panel(ag, img, N, a_amax=big)
        assert float((out2.double().cpu() - ref).abs().max()) <= bound * scale
    z = torch.zeros(M, K, device=dev)
    assert float(_g.gemm_panel(z, img, N).abs().max()) == 0.0


@pytest.mark.gpu
def test_dma_kernels_race_screen():
    """The ring GEMM and the streaming weight-gradient kernel order their LDS-DMA against their LDS reads with counted
    vmcnt waits and (GEMM) one barrier per stage; a read placed one wait too early returns stale LDS only when the DMA
    happens to land late -- rare wrong tiles that come and go with timing.  Screen: the same launch 30 times, with
    unrelated memory traffic in between to move the timing, must reproduce the first result bit for bit (both kernels
    are deterministic by construction), and that result must be the right one."""
    from graphgps_amd import lib as L_
    from graphgps_amd.gemm import gemm_panel, split_weights
    from graphgps_amd.lib import check, current_stream, ptr
    L = L_.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(99)
    noise = torch.empty(64 << 20, device=dev)
    for M, K, N, f16 in ((7569, 384, 2688, True), (7569, 2688, 384, True), (15348, 384, 384, True), (7569, 768, 384, True),
                         (7569, 256, 1792, True), (7569, 384, 2688, False), (7569, 2688, 384, False), (15348, 384, 384, False)):
        a = torch.randn(M, K, generator=gen).to(dev)
        w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
        (img, _), = split_weights([w], tn=False, f16=f16)
        first = gemm_panel(a, img, N).clone()
        ref = a.double() @ w.double().t()
        assert float((first.double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-5
        for it in range(30):
            if it % 3 == 0:
                noise.normal_()                     # evict caches, shift arrival times
            out = gemm_panel(a, img, N)
            assert torch.equal(out, first), f"ring GEMM {M}x{K}x{N}: run {it} differs from run 0"
    R = 7569
    shapes = [(R, 384, 2688), (15348, 384, 384), (R, 384, 384), (R, 384, 768), (R, 768, 384)]   # (rows, in, out)
    pairs = [(torch.randn(r, n, generator=gen).to(dev), torch.randn(r, k, generator=gen).to(dev)) for r, k, n in shapes]
    probs = (L_.WgradProblem * len(pairs))()
    outs = []
    for q, (g, x) in zip(probs, pairs):
        gw, gb = torch.empty(g.shape[1], x.shape[1], device=dev), torch.empty(g.shape[1], device=dev)
        q.g, q.x, q.gw, q.gb = g.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr()
        q.ldg, q.ldx, q.R, q.M, q.Nn = g.stride(0), x.stride(0), g.shape[0], g.shape[1], x.shape[1]
        outs.append((gw, gb))
    ws = torch.empty(max(L.gps_wgrad_grouped_workspace_floats(len(pairs), probs), 4), device=dev)
    from graphgps_amd.gemm import absmax
    words = absmax([t for pr in pairs for t in pr])
    for f16 in (False, True):
        for i, q in enumerate(probs):
            q.g_amax = words[2 * i].data_ptr() if f16 else None
            q.x_amax = words[2 * i + 1].data_ptr() if f16 else None
        firsts = None
        for it in range(30):
            if it % 3 == 0:
                noise.normal_()
            check(L.gps_wgrad_grouped(len(pairs), probs, ptr(ws), current_stream(dev)), "gps_wgrad_grouped")
            cur = [(gw.clone(), gb.clone()) for gw, gb in outs]
            if firsts is None:
                firsts = cur
                for (g, x), (gw, gb) in zip(pairs, cur):
                    assert_close(gw, g.double().t() @ x.double(), Tol.GRAD_REL, "gW", rel_to_max=True)
                    assert_close(gb, g.double().sum(0), Tol.GRAD_REL, "gb", rel_to_max=True)
            else:
                for i, ((gw, gb), (fw, fb)) in enumerate(zip(cur, firsts)):
                    assert torch.equal(gw, fw) and torch.equal(gb, fb), f"streaming wgrad (f16={f16}) problem {i}: run {it} differs"


@pytest.mark.parametrize("n,V,d,hub", [(25000, 10030, 256, 0.5), (25000, 10030, 256, 0.0), (77000, 2, 256, 0.0),
                                       (743, 28, 64, 0.0), (130, 98, 128, 0.9), (1, 7, 64, 0.0), (64, 5, 64, 0.0),
                                       (65, 3, 256, 0.0)])
def test_embedding_weight_gradient(n, V, d, hub):
    """ops.embedding: forward = nn.Embedding lookup (bitwise); weight gradient by the stable-sort + fixed-unit segmented
    sum (csrc/segment_pool.hip) vs F.embedding's in fp64 -- runs inside one 64-entry unit, runs spanning many units (a
    token that takes half of the lookups: the ASTNode 'no attribute' id), unit-aligned boundaries; bitwise run to run."""
    from graphgps_amd.ops import embedding
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n + V)
    w = torch.randn(V, d, generator=gen)
    idx = torch.randint(0, V, (n,), generator=gen)
    idx[0] = V - 1                                          # the last token is hit, token 0 maybe not
    if hub > 0:
        idx[torch.rand(n, generator=gen) < hub] = V // 2
    g = torch.randn(n, d, generator=gen)
    wr = w.double().requires_grad_(True)
    torch.nn.functional.embedding(idx, wr).backward(g.double())
    grads = []
    for _ in range(2):
        wd = w.to(dev).requires_grad_(True)
        out = embedding(idx.to(dev), wd)
        assert torch.equal(out.detach().cpu(), w[idx])
        out.backward(g.to(dev))
        grads.append(wd.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert_close(grads[0], wr.grad, Tol.GRAD_REL, "g_weight", rel_to_max=True)
