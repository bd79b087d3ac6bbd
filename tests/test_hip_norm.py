"""GPU parity tests of the task-list norm kernels (csrc/block_norm.hip) and of the in-launch column reductions
(csrc/col_tree.hpp) behind them, the GatedGCN forward's own batch statistics and the ring GEMM's residual + dropout +
statistics epilogue -- all through the C ABI, against fp64 restatements of the reference stages
(graphgps/layer/gatedgcn_layer.py:72-83, graphgps/layer/gps_layer.py:191-194,212-229).  Dropout masks are injected from
the host model of the kernels' counter hash (ops.attn_dropout_keep_mask), so every comparison is element by element."""
import ctypes

import pytest
import torch

from conftest import Tol, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mask(seed, R, d, p):
    from graphgps_amd.ops import attn_dropout_keep_mask
    if p == 0:
        return torch.ones(R, d, dtype=torch.float64)
    return attn_dropout_keep_mask(seed, torch.arange(R), 0, 1, torch.arange(d), p).double() / (1 - p)


def _bn(d, gen, dev=DEV):
    bn = torch.nn.BatchNorm1d(d)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=gen)
        bn.bias.uniform_(-0.5, 0.5, generator=gen)
        bn.running_mean.uniform_(-1, 1, generator=gen)
        bn.running_var.uniform_(0.5, 2, generator=gen)
    return bn.to(dev).train()


class _Owner:
    pass


def _desc(bn, d):
    from graphgps_amd import norm
    st = torch.empty(2, d, device=DEV)
    return norm.bn_desc(bn, st[0], st[1]), st


def _ref_stats(v, bn_before, eps=1e-5, mom=0.1):
    v = v.double()
    mean, var = v.mean(0), v.var(0, unbiased=False)
    n = v.shape[0]
    rm = (1 - mom) * bn_before[0].double() + mom * mean
    rv = (1 - mom) * bn_before[1].double() + mom * var * n / max(n - 1, 1)
    return mean, 1.0 / torch.sqrt(var + eps), rm, rv


@pytest.mark.parametrize("R,d", [(2, 64), (9, 52), (1000, 256), (7569, 384), (15348, 384), (200000, 64), (5000, 1024),
                                 (4097, 8)])
def test_norm_statistics_tree(R, d):
    """Batch statistics of a [R, d] tensor completed inside ONE launch (level-0 records -> groups -> root): mean, rstd
    and both running statistics against fp64; large-mean columns keep their variance; bitwise reproducible; the
    arrival counters are zero again afterwards."""
    from graphgps_amd import norm
    gen = torch.Generator().manual_seed(R * 7 + d)
    z = torch.randn(R, d, generator=gen) * 1.3 + 0.4
    z[:, 0] = z[:, 0] * 0.01 + 100.0                      # |mean| >> std
    bn = _bn(d, gen)
    before = (bn.running_mean.clone().cpu(), bn.running_var.clone().cpu())
    desc, st = _desc(bn, d)
    own = _Owner()
    sync = norm.sync_arena(own, torch.device(DEV))
    zg = z.to(DEV)
    norm.fwd([norm.fwd_task(norm.LOAD, zg, R, stats=desc)], d, zg.device, sync.site(0))
    mean, rstd, rm, rv = _ref_stats(z, before)
    assert_close(st[0], mean, 2e-6 * max(1.0, float(mean.abs().max())), "mean")
    rel = ((st[1].double().cpu() - rstd) / rstd).abs()
    assert float(rel[1:].max() if d > 1 else 0) < 3e-6, float(rel[1:].max())
    if R >= 1000:
        assert float(rel[0]) < 2e-3, float(rel[0])       # the quantum of fp32 inputs at 100 against a std of 0.013
    assert_close(bn.running_mean, rm, 2e-6 * max(1.0, float(rm.abs().max())), "running_mean")
    assert float(((bn.running_var.double().cpu() - rv) / rv).abs()[1:].max()) < 5e-6
    assert int(sync.buf.abs().sum()) == 0, "arrival counters must be left at zero"
    first = st.clone()
    for _ in range(3):
        norm.fwd([norm.fwd_task(norm.LOAD, zg, R, stats=desc)], d, zg.device, sync.site(0))
        assert torch.equal(st, first)


def test_norm_fwd_kinds_with_shared_masks():
    """The four row-stream kinds of one list launch (a residual + dropout(relu(BN)) stream with statistics, the same on
    an edge-sized stream, an add + dropout stream with statistics) and the dual apply that consumes them."""
    from graphgps_amd import norm
    gen = torch.Generator().manual_seed(11)
    N, E, d, p, pl = 1237, 2511, 384, 0.1, 0.25
    xt, x, ao = (torch.randn(N, d, generator=gen) for _ in range(3))
    eh, e = (torch.randn(E, d, generator=gen) * 1.5 + 0.3 for _ in range(2))
    bns = [_bn(d, gen) for _ in range(4)]                 # bn_x, bn_e, bn_l, bn_a
    own = _Owner()
    sync = norm.sync_arena(own, torch.device(DEV))
    (dx, sx), (de, se), (dl, sl), (da, sa_) = (_desc(b, d) for b in bns)
    g = lambda t: t.to(DEV)
    xtg, xg, aog, ehg, eg = g(xt), g(x), g(ao), g(eh), g(e)
    norm.fwd([norm.fwd_task(norm.LOAD, xtg, N, stats=dx), norm.fwd_task(norm.LOAD, ehg, E, stats=de)], d, xg.device,
             sync.site(0))
    x1, e1, za, h = (torch.empty_like(t) for t in (xg, eg, xg, xg))
    s0, s1, s3 = 0x1111222233334444, 0x5555666677778888, 0x9999AAAABBBBCCCC
    norm.fwd([norm.fwd_task(norm.BN_ACT, xtg, N, res=xg, bn1=dx, relu=True, p=p, seed=s0, out=x1, stats=dl),
              norm.fwd_task(norm.BN_ACT, ehg, E, res=eg, bn1=de, relu=True, p=p, seed=s1, out=e1),
              norm.fwd_task(norm.ADD_DROP, xg, N, b=aog, p=pl, seed=s3, out=za, stats=da)], d, xg.device, sync.site(1))
    norm.fwd([norm.fwd_task(norm.BN_DUAL, x1, N, b=za, bn1=dl, bn2=da, out=h)], d, xg.device, None)

    def bn64(v, bn):
        v = v.double()
        return ((v - v.mean(0)) / torch.sqrt(v.var(0, unbiased=False) + 1e-5) * bn.weight.double().cpu()
                + bn.bias.double().cpu())
    x1r = x.double() + bn64(xt, bns[0]).relu() * _mask(s0, N, d, p)
    e1r = e.double() + bn64(eh, bns[1]).relu() * _mask(s1, E, d, p)
    zar = x.double() + ao.double() * _mask(s3, N, d, pl)
    hr = bn64(x1r, bns[2]) + bn64(zar, bns[3])
    assert_close(x1, x1r, Tol.ACT, "x1")
    assert_close(e1, e1r, Tol.ACT, "e1")
    assert_close(za, zar, Tol.ACT, "za")
    assert_close(h, hr, 2 * Tol.ACT, "h = BN_l(x1) + BN_a(za)")
    assert int(sync.buf.abs().sum()) == 0


@pytest.mark.parametrize("N,E,d", [(1237, 2511, 384), (743, 1590, 64), (300, 20000, 256)])
def test_norm_backward_lists_and_chain(N, E, d):
    _norm_backward_case(N, E, d)


@pytest.mark.parametrize("N,E,d,pn,pe", [(1237, 2511, 384, 43, 177), (743, 1590, 64, 25, 10), (300, 20000, 256, 84, 480),
                                         (1000, 2000, 384, 1, 0)])
def test_norm_lists_on_padded_batches_see_the_real_rows_only(N, E, d, pn, pe):
    """The same stages on PADDED streams (loader.BucketPadding: ``pn`` / ``pe`` rows of junk appended, the real row
    counts in device words -- gps_norm_fwd_task.rdev / gps_norm_bwd_task.rdev): statistics, running statistics and every
    gradient equal the fp64 reference over the REAL rows alone (divisors 1 / R_real), and the gradients the apply kernels
    write on padding rows are exactly zero whatever the BatchNorm backward's mean terms are -- what keeps padding out of
    the weight-gradient contractions downstream.  Sizes with whole row blocks of padding (300 + 84) and with one row."""
    _norm_backward_case(N, E, d, pn, pe)


def _norm_backward_case(N, E, d, pn=0, pe=0):
    """The block's backward norm stages exactly as gps_block.py issues them -- {norm2, bn_edge_e} partial + apply,
    the dual {norm1_local, norm1_attn} partial + apply CHAINED into bn_node_x's column sums, bn_node_x's apply --
    against autograd in fp64 on the same function with the same dropout masks and the ReLU decisions the GPU took
    (read off its forward: of ~1e6 pre-activations one sits within fp32 rounding of the kink)."""
    from graphgps_amd import norm
    gen = torch.Generator().manual_seed(N + d)
    p, pl, pf2 = 0.1, 0.15, 0.2
    s0, s1, s3, s5 = 101, 0xABCDEF0123, 303, 0x77777777FFFF
    xt, x, za, z2, g_out, w_h = (torch.randn(N, d, generator=gen) for _ in range(6))   # w_h: the network above h
    eh, e, g_e1 = (torch.randn(E, d, generator=gen) for _ in range(3))
    padded = pn > 0 or pe > 0
    NR, ER = N, E                                  # real rows; from here on N / E are the row counts the kernels are given
    if padded:
        junk = lambda t, k: torch.cat([t, torch.randn(k, d, generator=gen) * 3.0 + 2.0])     # far from the data
        zero = lambda t, k: torch.cat([t, torch.zeros(k, d)])                                  # no gradient arrives there
        xt, x, za, z2 = (junk(t, pn) for t in (xt, x, za, z2))
        eh, e = junk(eh, pe), junk(e, pe)
        g_out, w_h, g_e1 = zero(g_out, pn), zero(w_h, pn), zero(g_e1, pe)
        N, E = N + pn, E + pe
    rn = torch.tensor([NR], dtype=torch.int32, device=DEV) if padded else None
    re_ = torch.tensor([ER], dtype=torch.int32, device=DEV) if padded else None
    bnx, bne, bnl, bna, bn2 = (_bn(d, gen) for _ in range(5))
    own = _Owner()
    dev = torch.device(DEV)
    sync = norm.sync_arena(own, dev)
    g = lambda t: t.to(DEV).contiguous()

    # ---- HIP: the forward pieces the backward needs, then the backward lists -------------------------------------
    xtg, xg, zag, z2g, ehg, eg = g(xt), g(x), g(za), g(z2), g(eh), g(e)
    descs = [_desc(b, d) for b in (bnx, bne, bnl, bna, bn2)]       # (descriptor, its [2, d] statistics buffer): keep both alive
    dx, de, dl, da, d2 = (q[0] for q in descs)
    x1g, brx, bre = torch.empty_like(xg), torch.empty_like(xg), torch.empty_like(eg)
    before = {k: (b.running_mean.clone().cpu(), b.running_var.clone().cpu()) for k, b in (("x", bnx), ("e", bne), ("l", bnl))}
    norm.fwd([norm.fwd_task(norm.LOAD, xtg, N, stats=dx, rdev=rn), norm.fwd_task(norm.LOAD, ehg, E, stats=de, rdev=re_),
              norm.fwd_task(norm.LOAD, zag, N, stats=da, rdev=rn), norm.fwd_task(norm.LOAD, z2g, N, stats=d2, rdev=rn)],
             d, dev, sync.site(0))
    norm.fwd([norm.fwd_task(norm.BN_ACT, xtg, N, res=xg, bn1=dx, relu=True, p=p, seed=s0, out=x1g, stats=dl, rdev=rn),
              norm.fwd_task(norm.BN_ACT, xtg, N, bn1=dx, relu=True, p=p, seed=s0, out=brx),      # the branches alone:
              norm.fwd_task(norm.BN_ACT, ehg, E, bn1=de, relu=True, p=p, seed=s1, out=bre)],     # their zeros = decisions
             d, dev, sync.site(1))
    gp = torch.empty(10, d, device=DEV)
    g_bxw, g_bxb, g_bew, g_beb, g_nlw, g_nlb, g_naw, g_nab, g_n2w, g_n2b = gp.unbind(0)
    g_z2, g_f2, g_eh = torch.empty_like(xg), torch.empty_like(xg), torch.empty_like(eg)
    g_outg, g_e1g, w_hg = g(g_out), g(g_e1), g(w_h)       # tasks hold raw pointers: the tensors must outlive the launches
    b1 = [norm.bwd_task(z2g, g_outg, d2, N, g_n2w, g_n2b, g_z=g_z2, g_drop=g_f2, p2=pf2, seed2=s5, rdev=rn),
          norm.bwd_task(ehg, g_e1g, de, E, g_bew, g_beb, relu=True, p=p, seed=s1, g_z=g_eh, rdev=re_)]
    norm.bwd_partial(b1, d, dev, sync.site(2))
    norm.bwd_apply(b1, d, dev, None)
    g_x1, g_xres, g_ao, g_xt = (torch.empty_like(xg) for _ in range(4))
    b3 = [norm.bwd_task(x1g, w_hg, dl, N, g_nlw, g_nlb, z2=zag, bn2=da, g_gamma2=g_naw, g_beta2=g_nab, g_z=g_x1,
                        g_sum=g_xres, g_drop=g_ao, p2=pl, seed2=s3, cz=xtg, cbn=dx, crelu=True, cp=p, cseed=s0,
                        cg_gamma=g_bxw, cg_beta=g_bxb, rdev=rn)]
    norm.bwd_partial(b3, d, dev, sync.site(3))
    norm.bwd_apply(b3, d, dev, sync.site(4))
    norm.bwd_apply([norm.bwd_task(xtg, g_x1, dx, N, g_bxw, g_bxb, relu=True, p=p, seed=s0, g_z=g_xt, rdev=rn)], d, dev, None)
    torch.cuda.synchronize()
    assert int(sync.buf.abs().sum()) == 0
    if padded:
        # padding rows: exactly zero in everything the apply kernels wrote; then drop them -- the reference below is
        # the un-padded computation
        for name, t, k in (("g_z2", g_z2, NR), ("g_f2", g_f2, NR), ("g_eh", g_eh, ER), ("g_x1", g_x1, NR),
                           ("g_xres", g_xres, NR), ("g_ao", g_ao, NR), ("g_xt", g_xt, NR)):
            assert not t[k:].any(), f"{name}: non-zero gradient on a padding row"
        assert torch.isfinite(x1g).all()
        g_z2, g_f2, g_x1, g_xres, g_ao, g_xt, x1g, brx = (t[:NR] for t in (g_z2, g_f2, g_x1, g_xres, g_ao, g_xt, x1g, brx))
        g_eh, bre = g_eh[:ER], bre[:ER]
        xt, x, za, z2, g_out, w_h = (t[:NR] for t in (xt, x, za, z2, g_out, w_h))
        eh, e, g_e1 = (t[:ER] for t in (eh, e, g_e1))
        N, E = NR, ER
        for k, b, v in (("x", bnx, xt), ("e", bne, eh)):          # running statistics: over the real rows
            _, _, rm, rv = _ref_stats(v, before[k])
            assert_close(b.running_mean, rm, 2e-6 * max(1.0, float(rm.abs().max())), f"running_mean {k}")
            assert float(((b.running_var.double().cpu() - rv) / rv).abs().max()) < 5e-6, k

    # ---- fp64 reference (autograd), BatchNorm parameters as leaves too ----------------------------------------------
    leaf = lambda t: t.detach().double().cpu().clone().requires_grad_(True)
    par = {name: (leaf(b.weight), leaf(b.bias)) for name, b in (("x", bnx), ("e", bne), ("l", bnl), ("a", bna), ("2", bn2))}

    def BN(v, name):
        wgt, bias = par[name]
        return (v - v.mean(0)) / torch.sqrt(v.var(0, unbiased=False) + 1e-5) * wgt + bias
    xt_r, za_r, z2_r, eh_r = leaf(xt), leaf(za), leaf(z2), leaf(eh)
    m0, m1, m3, m5 = _mask(s0, N, d, p), _mask(s1, E, d, p), _mask(s3, N, d, pl), _mask(s5, N, d, pf2)
    gx = ((brx.cpu() != 0) | (m0 == 0)).double()
    ge = ((bre.cpu() != 0) | (m1 == 0)).double()
    x1_r = x.double() + BN(xt_r, "x") * gx * m0
    x1_r.retain_grad()
    e1_r = e.double() + BN(eh_r, "e") * ge * m1
    h_r = BN(x1_r, "l") + BN(za_r, "a")
    loss = (h_r * w_h.double()).sum() + (BN(z2_r, "2") * g_out.double()).sum() + (e1_r * g_e1.double()).sum()
    loss.backward()
    assert_close(x1g, x1_r, Tol.ACT, "x1 (forward)")

    tol = 2e-5          # gradients through up to two BatchNorm backwards; relative to the largest element
    assert_close(g_z2, z2_r.grad, tol, "g_z2", rel_to_max=True)
    assert_close(g_f2, z2_r.grad * m5, tol, "g_f2 = dropmask(g_z2)", rel_to_max=True)
    assert_close(g_eh, eh_r.grad, tol, "g_eh", rel_to_max=True)
    assert_close(g_x1, x1_r.grad, tol, "g_x1", rel_to_max=True)
    assert_close(g_xt, xt_r.grad, tol, "g_xt (sums from the chain)", rel_to_max=True)
    assert_close(g_ao, za_r.grad * m3, tol, "g_ao = dropmask(g_za)", rel_to_max=True)
    assert_close(g_xres, x1_r.grad + za_r.grad, tol, "g_xres = g_x1 + g_za", rel_to_max=True)
    for name, gw, gb in (("x", g_bxw, g_bxb), ("e", g_bew, g_beb), ("l", g_nlw, g_nlb), ("a", g_naw, g_nab),
                         ("2", g_n2w, g_n2b)):
        assert_close(gw, par[name][0].grad, 1e-4, f"g_gamma {name}", rel_to_max=True)
        assert_close(gb, par[name][1].grad, 1e-4, f"g_beta {name}", rel_to_max=True)


def test_tree_race_screen():
    """The in-launch reductions hand data between workgroups (write-through records, arrival tickets, sc1 loads): a
    protocol error shows as a STALE record -- rarely, and only under the right timing.  Screen: alternate two different
    inputs through the same workspace sites 40 times each with unrelated memory traffic and a concurrent stream of
    launches in between; every result must equal, bit for bit, the first result of its input (the trees are deterministic
    by construction), the first results must be right, and the counters must end at zero."""
    from graphgps_amd import norm
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(5)
    N, E, d = 7569, 15348, 384
    own = _Owner()
    sync = norm.sync_arena(own, dev)
    bn_a, bn_b = _bn(d, gen), _bn(d, gen)
    (da, sa_), (db, sb) = _desc(bn_a, d), _desc(bn_b, d)
    inputs = []
    for k in range(2):
        zn = (torch.randn(N, d, generator=gen) * (1 + k) + k).to(dev)
        ze = (torch.randn(E, d, generator=gen) * (2 - k) - k).to(dev)
        gy = torch.randn(N, d, generator=gen).to(dev)
        inputs.append((zn, ze, gy))
    noise = torch.empty(48 << 20, device=dev)
    side = torch.cuda.Stream(device=dev)
    gpar = torch.empty(4, d, device=dev)
    gz = torch.empty(N, d, device=dev)
    firsts = {}
    for it in range(80):
        k = it & 1
        zn, ze, gy = inputs[k]
        if it % 3 == 0:
            noise.normal_()
        with torch.cuda.stream(side):            # unrelated concurrent load: uneven arrival order
            noise[: 8 << 20].mul_(1.0001)
        norm.fwd([norm.fwd_task(norm.LOAD, zn, N, stats=da), norm.fwd_task(norm.LOAD, ze, E, stats=db)], d, dev, sync.site(0))
        tasks = [norm.bwd_task(zn, gy, da, N, gpar[0], gpar[1], relu=True, p=0.1, seed=7, g_z=gz)]
        norm.bwd_partial(tasks, d, dev, sync.site(1))
        norm.bwd_apply(tasks, d, dev, None)
        cur = (sa_.clone(), sb.clone(), gpar[:2].clone(), gz.clone())
        if k not in firsts:
            firsts[k] = cur
            mean, rstd, _, _ = _ref_stats(zn.cpu(), (torch.zeros(d), torch.ones(d)))
            assert_close(cur[0][0], mean, 3e-6 * max(1.0, float(mean.abs().max())), "mean")
            assert float(((cur[0][1].double().cpu() - rstd) / rstd).abs().max()) < 3e-6
        else:
            for a, b, what in zip(cur, firsts[k], ("node stats", "edge stats", "column sums", "g_z")):
                assert torch.equal(a, b), f"iteration {it}: {what} differ from the first run of input {k}"
    torch.cuda.synchronize()
    assert int(sync.buf.abs().sum()) == 0


@pytest.mark.parametrize("profile,nb,d", [("P30", 256, 384), ("P14", 64, 64), ("CODE2_REAL", 6, 256)])
def test_gatedgcn_forward_emits_batch_statistics(profile, nb, d):
    """gps_gatedgcn_fwd_stats = gps_gatedgcn_fwd (bitwise the same x~ / e^) + the batch statistics of both outputs
    (bn_node_x, bn_edge_e: gatedgcn_layer.py:72-73): per-node-block records out of the forward, combined by a second
    launch of the same call (round 6)."""
    from graphgps_amd import lib as L_, norm
    from graphgps_amd.lib import check, current_stream, ptr
    from test_hip_ops import _index, _structure
    L = L_.load()
    sizes, ei, bvec, pt, gen = _structure(profile, nb, 3)
    N, E = int(pt[-1]), ei.shape[1]
    gi = _index(ei, bvec, pt)
    proj = torch.randn(N, 4 * d, generator=gen).to(DEV)
    ce = (torch.randn(E, d, generator=gen) * 1.2 + 0.2).to(DEV)
    st = current_stream(proj.device)
    P, fs = proj.data_ptr(), 4 * d
    xt0, eh0 = torch.empty(N, d, device=DEV), torch.empty(E, d, device=DEV)
    check(L.gps_gatedgcn_fwd(P, P + fs, P + 2 * fs, P + 3 * fs, 4 * d, ptr(ce), ptr(gi.rowptr_dst), ptr(gi.src_by_dst),
                             ptr(gi.eid_by_dst), N, E, d, ptr(xt0), ptr(eh0), None, st), "fwd")
    bnx, bne = _bn(d, gen), _bn(d, gen)
    before = [(b.running_mean.clone().cpu(), b.running_var.clone().cpu()) for b in (bnx, bne)]
    (dx, sx), (de, se) = _desc(bnx, d), _desc(bne, d)
    wsf = L.gps_gatedgcn_stats_floats(N, d)
    ws = torch.empty(wsf, device=DEV)
    xt, eh = torch.empty(N, d, device=DEV), torch.empty(E, d, device=DEV)
    first = None
    for it in range(4):
        check(L.gps_gatedgcn_fwd_stats(P, P + fs, P + 2 * fs, P + 3 * fs, 4 * d, ptr(ce), ptr(gi.rowptr_dst),
                                       ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh), None,
                                       ctypes.byref(dx), ctypes.byref(de), ptr(ws), wsf, None, st), "fwd_stats")
        if first is None:
            first = (sx.clone(), se.clone())
            # same arithmetic, but a different instantiation: the compiler contracts / orders a few operations differently
            assert float((xt - xt0).abs().max()) < 2e-6 and float((eh - eh0).abs().max()) < 2e-6
            for got, v, bn, bef in ((sx, xt0, bnx, before[0]), (se, eh0, bne, before[1])):
                mean, rstd, rm, rv = _ref_stats(v.cpu(), bef)
                assert_close(got[0], mean, 3e-6 * max(1.0, float(mean.abs().max())), "mean")
                assert float(((got[1].double().cpu() - rstd) / rstd).abs().max()) < 3e-6
                assert_close(bn.running_mean, rm, 3e-6 * max(1.0, float(rm.abs().max())), "running_mean")
                assert float(((bn.running_var.double().cpu() - rv) / rv).abs().max()) < 5e-6
        else:
            assert torch.equal(sx, first[0]) and torch.equal(se, first[1])
    # padded batches: with the real-node count on the device, the trailing rows and their incoming edges are not counted
    n_real = int(pt[-2])                                         # the last graph plays the padding
    e_real = int((ei[1] < n_real).sum())
    assert bool((ei[0][ei[1] >= n_real] >= n_real).all())        # its edges stay inside it (block-diagonal batch)
    for b in (bnx, bne):
        b.running_mean.zero_(); b.running_var.fill_(1.0)
    cnt = torch.tensor([n_real], dtype=torch.int32, device=DEV)
    check(L.gps_gatedgcn_fwd_stats(P, P + fs, P + 2 * fs, P + 3 * fs, 4 * d, ptr(ce), ptr(gi.rowptr_dst),
                                   ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh), None,
                                   ctypes.byref(dx), ctypes.byref(de), ptr(ws), wsf, ptr(cnt), st), "fwd_stats, padded")
    real_edges = (ei[1] < n_real)
    zero = (torch.zeros(d), torch.ones(d))
    for got, v in ((sx, xt0.cpu()[:n_real]), (se, eh0.cpu()[real_edges])):
        mean, rstd, _, _ = _ref_stats(v, zero)
        assert_close(got[0], mean, 3e-6 * max(1.0, float(mean.abs().max())), "mean over the real rows")
        assert float(((got[1].double().cpu() - rstd) / rstd).abs().max()) < 3e-6
    assert e_real == int(real_edges.sum())
    # (the statistics above were checked against the PLAIN kernel's outputs; the stats kernel's own agree to 2e-6)


@pytest.mark.parametrize("M,K,N", [(7569, 768, 384), (7569, 384, 384), (15348, 384, 384), (130, 384, 192), (64, 384, 384),
                                   (5000, 512, 256), (25013, 256, 256), (743, 64, 64), (743, 128, 64)])
@pytest.mark.parametrize("f16", [True, False], ids=["f16x3", "bf16x6"])
def test_gemm_epilogue_residual_dropout_statistics(M, K, N, f16):
    """gps_gemm_panel_stats: C = Cin + dropout(A W^T + b) with the host model of the mask, and the batch statistics of C
    (norm2 / norm1_attn: gps_layer.py:212-217,225-229) against fp64; repeated launches reproduce bit for bit."""
    from graphgps_amd import gemm, norm
    if not gemm.stats_supported(M, N, K):
        pytest.skip("shape not served by the ring kernel")
    gen = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen) * 1.5 + 0.5
    p, seed = 0.1, 0xFEEDFACE12345
    (img, _), = gemm.split_weights([w.to(DEV)], tn=False, f16=f16)
    bn = _bn(N, gen)
    before = (bn.running_mean.clone().cpu(), bn.running_var.clone().cpu())
    desc, st = _desc(bn, N)
    own = _Owner()
    sync = norm.sync_arena(own, torch.device(DEV))
    ag, bg, rg = a.to(DEV), b.to(DEV), res.to(DEV)
    out = gemm.gemm_panel_stats(ag, img, N, bg, rg, p, seed, desc, sync.site(0))
    ref = res.double() + (a.double() @ w.double().t() + b.double()) * _mask(seed, M, N, p)
    scale = float(ref.abs().max())
    assert_close(out, ref, 4e-6 * scale, "Cin + dropout(A W^T + b)")
    mean, rstd, rm, rv = _ref_stats(out.cpu(), before)
    assert_close(st[0], mean, 3e-6 * max(1.0, float(mean.abs().max())), "mean")
    assert float(((st[1].double().cpu() - rstd) / rstd).abs().max()) < 3e-6
    assert_close(bn.running_mean, rm, 3e-6 * max(1.0, float(rm.abs().max())), "running_mean")
    first = (out.clone(), st.clone())
    for _ in range(5):
        out2 = gemm.gemm_panel_stats(ag, img, N, bg, rg, p, seed, desc, sync.site(0))
        assert torch.equal(out2, first[0]) and torch.equal(st, first[1])
    assert int(sync.buf.abs().sum()) == 0


@pytest.mark.parametrize("M,K,N,real", [(7680, 768, 384, 7569), (7680, 384, 384, 7569), (2048, 768, 384, 1900), (192, 384, 384, 130),
                                        (7680, 384, 384, 7680), (1024, 256, 256, 3)])
def test_gemm_statistics_epilogue_skips_padding_rows(M, K, N, real):
    """gps_gemm16_panel_stats with ``m_dev`` (padded batches): every row of C is produced as before, the batch
    statistics cover rows [0, m_dev[0]) only -- tiles that straddle the boundary, tiles of padding only (count-0
    records in the tree), no padding at all (m_dev = M), and a near-empty batch."""
    from graphgps_amd import gemm, norm
    if not gemm.stats_supported(M, N, K):
        pytest.skip("shape not served by the ring kernel")
    gen = torch.Generator().manual_seed(M + K + real)
    a = torch.randn(M, K, generator=gen)
    a[real:] = a[real:] * 4.0 + 3.0                       # padding far from the data
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen) * 1.5 + 0.5
    p, seed = 0.1, 0xFEEDFACE12345
    (img, _), = gemm.split_weights([w.to(DEV)], tn=False, f16=True)
    bn = _bn(N, gen)
    before = (bn.running_mean.clone().cpu(), bn.running_var.clone().cpu())
    desc, st = _desc(bn, N)
    own = _Owner()
    sync = norm.sync_arena(own, torch.device(DEV))
    ag, bg, rg = a.to(DEV), b.to(DEV), res.to(DEV)
    m_dev = torch.tensor([real], dtype=torch.int32, device=DEV)
    out = gemm.gemm_panel_stats(ag, img, N, bg, rg, p, seed, desc, sync.site(0), m_dev=m_dev)
    ref = res.double() + (a.double() @ w.double().t() + b.double()) * _mask(seed, M, N, p)
    assert_close(out, ref, 4e-6 * float(ref.abs().max()), "Cin + dropout(A W^T + b), padding rows included")
    mean, rstd, rm, rv = _ref_stats(out[:real].cpu(), before)
    assert_close(st[0], mean, 3e-6 * max(1.0, float(mean.abs().max())), "mean over the real rows")
    assert float(((st[1].double().cpu() - rstd) / rstd).abs().max()) < 3e-6
    assert_close(bn.running_mean, rm, 3e-6 * max(1.0, float(rm.abs().max())), "running_mean")
    if real > 1:
        assert float(((bn.running_var.double().cpu() - rv) / rv).abs().max()) < 1e-5
    first = (out.clone(), st.clone())
    out2 = gemm.gemm_panel_stats(ag, img, N, bg, rg, p, seed, desc, sync.site(0), m_dev=m_dev)
    assert torch.equal(out2, first[0]) and torch.equal(st, first[1])
    assert int(sync.buf.abs().sum()) == 0


_POISON_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from graphgps_amd import lib as L_, norm
from graphgps_amd.lib import check, current_stream
import test_hip_norm as T
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
R, d = 5000, 64
z = torch.randn(R, d, generator=gen).to(dev)
bn = T._bn(d, gen)
desc, st = T._desc(bn, d)
own = T._Owner()
sync = norm.sync_arena(own, dev)
norm.fwd([norm.fwd_task(norm.LOAD, z, R, stats=desc)], d, dev, sync.site(0))
torch.cuda.synchronize()
ref = z.double().mean(0)
assert float((st[0].double() - ref).abs().max()) < 1e-5
print("CLEAN-LAUNCH-OK", flush=True)
L = L_.load()
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
check(L.gps_sync_nonzero(sync.buf.data_ptr(), sync.buf.numel(), cnt.data_ptr(), current_stream(dev)))
assert int(cnt) == 0
mode = sys.argv[1]
sync.buf[1] = 1                          # a stale group counter: what a launch that died mid-tree leaves behind
check(L.gps_sync_nonzero(sync.buf.data_ptr(), sync.buf.numel(), cnt.data_ptr(), current_stream(dev)))
assert int(cnt) == 1
print("SELF-CHECK-SEES-IT", flush=True)
if mode == "reset":
    check(L.gps_sync_reset(sync.buf.data_ptr(), sync.buf.numel(), current_stream(dev)))
norm.fwd([norm.fwd_task(norm.LOAD, z, R, stats=desc)], d, dev, sync.site(0))
torch.cuda.synchronize()
assert float((st[0].double() - ref).abs().max()) < 1e-5
print("SECOND-LAUNCH-RETURNED", flush=True)
"""


@pytest.mark.parametrize("mode", ["poisoned", "reset"])
def test_stale_arrival_counter_fails_loudly(mode):
    """the in-launch reductions need their arrival counters at zero; a launch that died mid-tree leaves
    one non-zero.  The NEXT launch must not publish statistics of incomplete records silently: a ticket beyond the group
    size traps (csrc/col_tree.hpp), which kills the launch and the process loudly.  In a child process: a clean launch,
    ``gps_sync_nonzero`` sees the poisoned word, then either the launch on the poisoned counters dies (no wrong result is
    ever returned) or, after ``gps_sync_reset``, it runs and is right."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = _POISON_CHILD.format(root=os.path.dirname(here), tests=here)
    r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "CLEAN-LAUNCH-OK" in r.stdout and "SELF-CHECK-SEES-IT" in r.stdout, out[-2000:]
    if mode == "reset":
        assert r.returncode == 0 and "SECOND-LAUNCH-RETURNED" in r.stdout, out[-2000:]
    else:
        assert r.returncode != 0 and "SECOND-LAUNCH-RETURNED" not in r.stdout, (
            "a launch on a stale arrival counter returned instead of failing loudly:\n" + out[-2000:])


def _colsum_refs(g, z, st):
    """fp64 S1 = sum g, S2 = sum g zhat (block_norm.hip run_bwd_partial)."""
    zh = (z.double() - st[0].double().cpu()) * st[1].double().cpu()
    return g.double().sum(0), (g.double() * zh).sum(0)


@pytest.mark.parametrize("M,K,N", [(7569, 768, 384), (7569, 384, 384), (130, 384, 192), (64, 384, 384), (5000, 512, 256),
                                   (25013, 256, 256), (2, 384, 384)])
def test_gemm_epilogue_dual_batchnorm_backward_sums(M, K, N):
    """gps_gemm16_panel_sums, two-BatchNorm form: g_h = g_z2 + g_f1 W1 (in place) leaves with sum g, sum g zhat(x1),
    sum g zhat(za) -- the column sums of norm1_local / norm1_attn's backward (gps_layer.py:219-222) -- against fp64 and
    against gps_norm_bwd_partial over the stored output; bit-reproducible; counters zero afterwards."""
    from graphgps_amd import gemm, norm
    if not gemm.colsums_supported(M, N, K):
        pytest.skip("shape not served")
    gen = torch.Generator().manual_seed(M * 3 + K)
    a = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    res = torch.randn(M, N, generator=gen)
    z1 = torch.randn(M, N, generator=gen) * 1.3 + 0.7
    z2 = torch.randn(M, N, generator=gen) * 0.6 - 2.0
    (img, _), = gemm.split_weights([w.to(DEV)], tn=False, f16=True)
    bn1, bn2 = _bn(N, gen), _bn(N, gen)
    d1, st1 = _desc(bn1, N)
    d2, st2 = _desc(bn2, N)
    for st, z in ((st1, z1), (st2, z2)):
        st[0].copy_(z.mean(0)); st[1].copy_(1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-5))
    own = _Owner()
    sync = norm.sync_arena(own, torch.device(DEV))
    ag, z1g, z2g = a.to(DEV), z1.to(DEV), z2.to(DEV)
    sums = torch.full((4, N), float("nan"), device=DEV)

    def run():
        c = res.to(DEV)
        out = gemm.gemm_panel_sums(dict(a=ag, image=img, N=N, addend=c, out=c),
                                   dict(z=z1g, bn=d1, sum_g=sums[0], sum_gz=sums[1], z2=z2g, bn2=d2, sum_g2=sums[2],
                                        sum_gz2=sums[3]), sync.site(0))
        assert out.data_ptr() == c.data_ptr()
        return out
    out = run()
    ref = res.double() + a.double() @ w.double().t()
    assert_close(out, ref, 4e-6 * float(ref.abs().max()), "Cin + A W^T")
    g = out.cpu()
    s1, s2a = _colsum_refs(g, z1, st1)
    _, s2b = _colsum_refs(g, z2, st2)
    scale = float(max(s1.abs().max(), s2a.abs().max(), s2b.abs().max(), 1.0))
    tol = 2e-6 * scale * max(1.0, (M / 1000.0) ** 0.5)
    assert_close(sums[0], s1, tol, "sum g")
    assert_close(sums[1], s2a, tol, "sum g zhat(z1)")
    assert torch.equal(sums[2], sums[0])
    assert_close(sums[3], s2b, tol, "sum g zhat(z2)")
    # the row pass it replaces, over the same stored gradient
    if M >= 2:
        ps = torch.empty(4, N, device=DEV)
        t = norm.bwd_task(z1g, out, d1, M, ps[1], ps[0], z2=z2g, bn2=d2, g_gamma2=ps[3], g_beta2=ps[2])
        norm.bwd_partial([t], N, torch.device(DEV), sync.site(1))
        assert_close(sums, ps.double().cpu(), tol, "vs gps_norm_bwd_partial")
    first = (out.clone(), sums.clone())
    for _ in range(3):
        out2 = run()
        assert torch.equal(out2, first[0]) and torch.equal(sums, first[1])
    assert int(sync.buf.abs().sum()) == 0
