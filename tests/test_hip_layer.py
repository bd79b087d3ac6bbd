"""GPU parity of the HIP-backed GPSLayer / GPSModel: against the reference-generated golden
fixtures, against the CPU oracle at BASELINE.json sizes, and through size-independent
properties (edge-permutation invariance, graph-order invariance, run-to-run determinism)."""
import os

import pytest
import torch

from conftest import (Tol, assert_close, assert_close_kink_tolerant, assert_fp32_grade, fp32_grade, golden_names,
                      load_golden)

pytestmark = pytest.mark.gpu

TRANSFORMER_GOLDEN = [n for n in golden_names() if "performer" not in n]
PERFORMER_GOLDEN = [n for n in golden_names() if "performer" in n]


def _run_layer(layer, fix, dev):
    from graphgps_amd.data import Batch
    x = fix["x"].to(dev).requires_grad_(True)
    e = fix["edge_attr"].to(dev).requires_grad_(True)
    b = Batch(x=x, edge_index=fix["edge_index"].to(dev), edge_attr=e, batch=fix["batch"].to(dev),
              ptr=fix["ptr"].to(dev))
    pe = None
    if "pe" in fix:                       # EquivStableLapPE fixtures
        pe = fix["pe"].to(dev).requires_grad_(True)
        b.pe_EquivStableLapPE = pe
    bias = None
    if "attn_bias" in fix:                # BiasedTransformer fixtures: dense [B*H, nmax, nmax] operand
        bias = fix["attn_bias"].to(dev).requires_grad_(True)
        b.attn_bias = bias
    out = layer(b)
    ((out.x * fix["wx"].to(dev)).sum() + (out.edge_attr * fix["we"].to(dev)).sum()).backward()
    if pe is not None:
        assert_close(pe.grad, fix["grad_pe"], Tol.GRAD_REL, "grad pe", rel_to_max=True)
    if bias is not None:
        assert_close(bias.grad, fix["grad_attn_bias"], Tol.GRAD_REL, "grad attn_bias", rel_to_max=True)
    return out, x, e


def _check_against_fixture(name):
    from graphgps_amd.data import Batch
    from graphgps_amd.layer.gps_layer import GPSLayer
    dev = torch.device("cuda:0")
    fix = load_golden(name)
    layer = GPSLayer(**fix["ctor"])
    layer.load_state_dict(fix["state_dict"], strict=True)   # checkpoint interchange contract
    layer.to(dev).train()
    out, x, e = _run_layer(layer, fix, dev)
    assert_close(out.x, fix["out_x"], Tol.ACT, "out.x")
    assert_close(out.edge_attr, fix["out_edge_attr"], Tol.ACT, "out.edge_attr")
    assert_close(x.grad, fix["grad_x"], Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(e.grad, fix["grad_edge_attr"], Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    got = dict(layer.named_parameters())
    for k, g in fix["param_grads"].items():
        # scale floor: 1 % of the layer's largest parameter gradient (biases that feed a
        # BatchNorm have a mathematically zero gradient = rounding residue on both sides)
        gs = max(float(v.abs().max()) for v in fix["param_grads"].values())
        a_, b_ = got[k].grad.detach().double().cpu(), g.double()
        assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"
    after = layer.state_dict()
    for k, v in fix["state_dict_after"].items():
        if v.dtype.is_floating_point:
            assert_close(after[k], v, Tol.ACT, f"state {k}")
    layer.eval()
    with torch.no_grad():
        eb = Batch(x=fix["x"].to(dev), edge_index=fix["edge_index"].to(dev),
                   edge_attr=fix["edge_attr"].to(dev), batch=fix["batch"].to(dev))
        if "pe" in fix:
            eb.pe_EquivStableLapPE = fix["pe"].to(dev)
        if "attn_bias" in fix:
            eb.attn_bias = fix["attn_bias"].to(dev)
        ob = layer(eb)
    assert_close(ob.x, fix["eval_out_x"], Tol.ACT, "eval out.x")
    assert_close(ob.edge_attr, fix["eval_out_edge_attr"], Tol.ACT, "eval out.edge_attr")


@pytest.mark.parametrize("name", TRANSFORMER_GOLDEN)
def test_gpslayer_matches_reference_fixture(name):
    _check_against_fixture(name)


@pytest.mark.parametrize("name", PERFORMER_GOLDEN)
def test_gpslayer_performer_matches_reference_fixture(name):
    pytest.importorskip("graphgps_amd.layer.performer_layer")
    _check_against_fixture(name)


def _double_batch(b):
    """The same batch with every floating tensor in fp64 (integer features / indices untouched)."""
    out = b.clone()
    for k, v in list(out.__dict__.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            out.__dict__[k] = v.double()
    return out


def _oracle_layer_like(layer):
    from oracle.gps_oracle import OracleGPSLayer
    o = OracleGPSLayer(**layer.ctor_kwargs)
    o.load_state_dict({k: v.cpu() for k, v in layer.state_dict().items()}, strict=True)
    return o


@pytest.mark.parametrize("local,glob,d,H,profile,nb", [
    ("CustomGatedGCN", "Transformer", 384, 16, "P30", 256),   # BASELINE configs[2] layer shape
    ("CustomGatedGCN", "Transformer", 384, 16, "P14", 256),
    ("GINE", "Transformer", 64, 4, "ZINC", 32),               # BASELINE configs[1] layer shape
    ("CustomGatedGCN", "Transformer", 304, 4, "P30", 256),    # pcqm4m-GPS.yaml (GPS-small): 4.75 column panels, 9.5 k-stages
    ("CustomGatedGCN", "Transformer", 304, 4, "P14", 128),
    ("CustomGatedGCN", "Transformer", 96, 4, "P30", 64),      # peptides-*-GPS.yaml width: 1.5 panels
    ("CustomGatedGCN", "Transformer", 52, 4, "P30", 64),      # a width that is a multiple of 4 only (dh = 13)
])
def test_gpslayer_vs_oracle_baseline_sizes(local, glob, d, H, profile, nb):
    from graphgps_amd.layer.gps_layer import GPSLayer
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layer = GPSLayer(d, local, glob, H, dropout=0.0, attn_dropout=0.0)
    import copy
    oracle = _oracle_layer_like(layer).train()
    o64 = copy.deepcopy(oracle).double().train()          # the same oracle in fp64: the yardstick of the gradient bars
    layer.to(dev).train()
    state0 = {k: v.clone() for k, v in layer.state_dict().items()}          # (BN running statistics move per forward)
    b = layer_batch(profile, nb, d, seed=77)
    gen = torch.Generator().manual_seed(5)
    wx = torch.randn(b.x.shape, generator=gen)
    we = torch.randn(b.edge_attr.shape, generator=gen)

    def run(wx, we):
        """One forward + backward of the loss sum(wx * x') + sum(we * e') on all three sides."""
        for m in (oracle, o64, layer):
            m.zero_grad(set_to_none=True)
        layer.load_state_dict(state0)
        oracle.load_state_dict({k: v.cpu() for k, v in state0.items()})
        o64.load_state_dict({k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in state0.items()})
        b6 = _double_batch(b)
        b6.x.requires_grad_(True); b6.edge_attr.requires_grad_(True)
        o6 = o64(b6)
        ((o6.x * wx.double()).sum() + (o6.edge_attr * we.double()).sum()).backward()
        bc = b.clone()
        bc.x.requires_grad_(True); bc.edge_attr.requires_grad_(True)
        xo, eo = bc.x, bc.edge_attr             # (the layer re-binds batch.x / batch.edge_attr to its outputs)
        oo = oracle(bc)
        ((oo.x * wx).sum() + (oo.edge_attr * we).sum()).backward()
        bg = b.clone().to(dev)
        bg.x.requires_grad_(True); bg.edge_attr.requires_grad_(True)
        xg, eg = bg.x, bg.edge_attr
        og = layer(bg)
        ((og.x * wx.to(dev)).sum() + (og.edge_attr * we.to(dev)).sum()).backward()
        return oo, og, xo.grad, eo.grad, xg.grad, eg.grad

    def check_params(strict, floor=1e-5):
        """Parameter gradients: sums over ~8k rows / ~16k edges of fp32 products, the reduction order differs between
        the HIP kernels and the CPU BLAS (bar 1e-4 of max|g|), and a ReLU-kink flip at (row r, channel c) lands
        undamped in row c of a weight gradient, so a few outlier rows per parameter are allowed
        (assert_close_kink_tolerant).  Biases that feed a BatchNorm have a mathematically zero gradient: what both
        sides hold is the rounding residue of a 7.5k-term cancelling sum, so the scale floor is 1 % of the layer's
        largest parameter gradient rather than the parameter's own (noise) magnitude.  Returns the failures."""
        op, o6p = dict(oracle.named_parameters()), dict(o64.named_parameters())
        gscale = max(float(q.grad.abs().max()) for q in o6p.values() if q.grad is not None)
        bad = []
        for k, p in layer.named_parameters():
            if op[k].grad is None:
                continue
            try:
                # the bar is what the reference's own fp32 arithmetic achieves against fp64 on this parameter (x 5: a
                # max over a few hundred rows is a noisier statistic than the rms the model tests grade with x 3; floor
                # 1e-5 = north_star), not a constant: VERDICT r3 -- a fixed 1e-4 would pass a 3x regression
                ms = max(1.0, 0.01 * gscale)
                c32 = _kink_free_err(op[k].grad, o6p[k].grad, ms)
                r = assert_close_kink_tolerant(p.grad, o6p[k].grad, max(floor, 5.0 * c32), f"grad {k} (cpu fp32 {c32:.1e})",
                                               min_scale=ms)
                if r[2]:
                    print(f"grad {k}: {r[2]} kink rows, max rel {r[0]:.2e} outside them")
            except AssertionError as exc:
                if strict:
                    raise
                bad.append(str(exc))
        return bad

    oo, og, gxo, geo, gxg, geg = run(wx, we)
    assert_close(og.x, oo.x, Tol.ACT, "out.x")
    assert_close(og.edge_attr, oo.edge_attr, Tol.ACT, "out.edge_attr")
    # a kink flip inside the attention / FFN path of one graph perturbs every node row of that
    # graph (attention mixes them): allow a few graphs' worth of rows
    gmax = int((b.ptr[1:] - b.ptr[:-1]).max())
    rx = assert_close_kink_tolerant(gxg, gxo, Tol.GRAD_REL, "grad x", min_allowed_rows=4 * gmax)
    re_ = assert_close_kink_tolerant(geg, geo, Tol.GRAD_REL, "grad e")
    print(f"grad x: max rel {rx[0]:.2e} outside {rx[2]} kink rows; "
          f"grad e: max rel {re_[0]:.2e} outside {re_[2]} kink rows")
    bad = check_params(strict=False)
    if not bad:
        return
    # Some parameter gradient is off by more than a few rows' worth.  The one legitimate cause is a kink flip UPSTREAM of
    # the attention: every node of that graph then carries a kink-sized gradient error (the rows counted above), and in
    # a small batch (P14 x 128: 1.8k rows) one graph's rows weigh enough to move every row of a weight gradient past
    # 1e-4.  Attribution instead of a wider tolerance: give the graphs that own the kink rows zero weight in the loss
    # (on both sides; the flips themselves stay where they are) -- every parameter gradient must then meet the bar.
    scale_x, scale_e = max(float(gxo.abs().max()), 1.0), max(float(geo.abs().max()), 1.0)
    kx = ((gxg.cpu() - gxo).abs().amax(dim=1) / scale_x > Tol.GRAD_REL).nonzero().flatten()
    ke = ((geg.cpu() - geo).abs().amax(dim=1) / scale_e > Tol.GRAD_REL).nonzero().flatten()
    graphs = set(b.batch[kx].tolist()) | set(b.batch[b.edge_index[1][ke]].tolist())
    assert graphs, "parameter gradients differ although no input-gradient row does:\n" + "\n".join(bad)
    assert len(graphs) <= 4, f"{len(graphs)} graphs with kink-sized gradient rows"
    print(f"{len(bad)} parameter gradients past 1e-4 with kink rows in graphs {sorted(graphs)}; second pass without them")
    keep_n = torch.ones(b.x.shape[0], 1)
    for g_ in graphs:
        keep_n[b.batch == g_] = 0.0
    keep_e = keep_n[b.edge_index[1]]
    run(wx * keep_n, we * keep_e)
    # (the flips themselves are still in the HIP forward: through BatchNorm's batch-wide column sums a flipped gate of a
    # zero-weight graph still reaches every row at O(R^-1.5) of the gradient scale -- 1e-5 at R = 1.8k rows -- so this pass
    # keeps round 3's 1e-4 floor; the first pass, which every flip-free configuration ends in, holds the relative bar)
    check_params(strict=True, floor=1e-4)


def _kink_free_err(a32, ref64, min_scale, frac=1e-3):
    """max|a32 - ref64| / max(max|ref64|, min_scale) over all rows but the worst ``frac`` of them (the rows a ReLU-kink flip
    of the CPU fp32 evaluation lands in: the same allowance ``assert_close_kink_tolerant`` gives the HIP side)."""
    e = (a32.detach().double().cpu() - ref64.detach().double().cpu()).abs()
    scale = max(float(ref64.abs().max()), min_scale)
    rows = e.reshape(e.shape[0], -1).amax(dim=1) if e.dim() > 1 else e
    k = max(int(rows.numel() * frac), 2 if rows.numel() > 8 else 0)
    if k and rows.numel() > k:
        rows = rows.sort().values[:rows.numel() - k]
    return float(rows.max()) / scale if rows.numel() else 0.0


def _masked_oracle_run(oracle, b, seeds_per_layer, H, p, p_attn, wx, we, dtype, local, glob="Transformer"):
    """Forward + backward of a stack of oracle layers with every dropout replaced by the mask the fused block draws
    from its seed (tests/helpers.py): returns (x_out, e_out, grad x, grad e, {param: grad} per layer)."""
    import copy
    from graphgps_amd.ops import attn_dropout_effective_p
    from helpers import MaskedSegmentMHA, attention_keep, block_seeds, inject_dropout_masks, row_mask
    layers = [copy.deepcopy(o).to(dtype).train() for o in oracle]
    bc = b.clone()
    bc.x = bc.x.to(dtype).requires_grad_(True)
    bc.edge_attr = bc.edge_attr.to(dtype).requires_grad_(True)
    x0, e0 = bc.x, bc.edge_attr
    N, d = bc.x.shape
    E = bc.edge_attr.shape[0]
    masks = _block_masks(layers, b, seeds_per_layer, N, E, d, H, p, p_attn, local, glob)
    with inject_dropout_masks(masks):
        for lay in layers:
            bc = lay(bc)
    loss = (bc.x * wx.to(dtype)).sum()
    if local == "CustomGatedGCN":
        loss = loss + (bc.edge_attr * we.to(dtype)).sum()
    loss.backward()
    grads = [{k: q.grad for k, q in lay.named_parameters() if q.grad is not None} for lay in layers]
    return bc.x, bc.edge_attr, x0.grad, e0.grad, grads


def _block_masks(layers, b, seeds_per_layer, N, E, d, H, p, p_attn, local, glob="Transformer", cache=None):
    """The dropout masks a stack of fused blocks draws from its per-layer seeds, in the order the oracle layers call
    ``F.dropout``; swaps each oracle layer's attention module for the masked one (Transformer).  ``cache`` (a dict): the
    masks are a function of the seeds alone, so a second oracle (the fp64 evaluation of the same step) reuses the first
    one's instead of re-running the host model of the hash (~2 s per layer at 256 graphs)."""
    from graphgps_amd.ops import attn_dropout_effective_p
    from helpers import MaskedSegmentMHA, attention_keep, block_seeds, row_mask
    if cache is not None and "masks" in cache:
        if glob == "Transformer":
            for lay, keep in zip(layers, cache["keeps"]):
                lay.self_attn = MaskedSegmentMHA(lay.self_attn, b.ptr, keep, attn_dropout_effective_p(p_attn))
        return list(cache["masks"])
    masks, keeps = [], []
    for lay, seed in zip(layers, seeds_per_layer):
        s = block_seeds(seed)
        if glob == "Transformer":
            keep = attention_keep(s[2], b.ptr, H, p_attn) if p_attn > 0 else None
            keeps.append(keep)
            lay.self_attn = MaskedSegmentMHA(lay.self_attn, b.ptr, keep, attn_dropout_effective_p(p_attn))
        if local == "CustomGatedGCN":       # call order: gatedgcn x, e | dropout_attn | ff_dropout1 | ff_dropout2
            masks += [row_mask(s[0], N, d, p), row_mask(s[1], E, d, p)]
        else:                               # GINE: dropout_local | dropout_attn | ff_dropout1 | ff_dropout2
            masks += [row_mask(s[0], N, d, p)]
        if glob == "Performer":
            # The Performer drops its OUTPUT (performer_layer.py:500-503), GPSLayer's dropout_attn drops it again
            # (gps_layer.py:212): two independent keep decisions with their scalings are ONE Bernoulli mask with keep
            # probability (1 - p_attn)(1 - p) -- the form the fused block draws (one hash per element).  Injected into
            # the reference's two calls as (that mask, all ones).
            p_eff = 1.0 - (1.0 - p_attn) * (1.0 - p)
            sizes = (b.ptr[1:] - b.ptr[:-1]).tolist()
            dense = torch.ones(len(sizes), max(sizes), d, dtype=torch.float64)     # the reference drops the PADDED batch
            rm = row_mask(s[3], N, d, p_eff)
            for g_, n_ in enumerate(sizes):
                dense[g_, :n_] = rm[int(b.ptr[g_]):int(b.ptr[g_]) + n_]
            masks += [dense, torch.ones(N, d, dtype=torch.float64)]
        else:
            masks += [row_mask(s[3], N, d, p)]
        masks += [row_mask(s[4], N, 2 * d, p), row_mask(s[5], N, d, p)]
    if cache is not None:
        cache["masks"], cache["keeps"] = list(masks), keeps
    return masks


@pytest.mark.parametrize("local,d,H,profile,nb,p,p_attn,n_layers", [
    ("CustomGatedGCN", 384, 16, "P30", 256, 0.1, 0.1, 1),     # the MEASURED configuration (pcqm4m-GPSmedium: dropout 0.1 / 0.1)
    ("CustomGatedGCN", 384, 16, "P14", 64, 0.1, 0.1, 2),      # two stacked layers, each with its own seed
    ("GINE", 64, 4, "ZINC", 32, 0.0, 0.5, 1),                 # zinc-GPS+RWSE.yaml: dropout 0.0, attn_dropout 0.5
    ("GINE", 64, 4, "ZINC", 32, 0.2, 0.5, 2),
])
def test_fused_block_with_dropout_on_vs_masked_oracle(local, d, H, profile, nb, p, p_attn, n_layers, monkeypatch):
    """Parity of the configuration bench.py measures, dropout ON: the fused block's seven seeds give seven masks (row /
    (row, column) hash of the norm kernels and GEMM epilogues, the paired-key hash of the attention kernels); the same
    masks, from the host model of the hash, are injected into the oracle (graphgps/layer/gps_layer.py:139-140,152-153,
    253-257, gatedgcn_layer.py:78-79).  Outputs to 1e-5; gradients to 1e-5 of their largest element outside ReLU-kink
    rows; three-way against the fp64 evaluation for the parameter gradients."""
    from graphgps_amd.layer import gps_block
    from graphgps_amd.layer.gps_layer import GPSLayer
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = [GPSLayer(d, local, "Transformer", H, dropout=p, attn_dropout=p_attn) for _ in range(n_layers)]
    oracle = [_oracle_layer_like(l) for l in layers]
    for l in layers:
        l.to(dev).train()
    b = layer_batch(profile, nb, d, seed=91)
    gen = torch.Generator().manual_seed(6)
    wx = torch.randn(b.x.shape, generator=gen)
    we = torch.randn(b.edge_attr.shape, generator=gen)
    seeds = [0x0123456789ABCDEF + 977 * i for i in range(n_layers)]
    it = iter(seeds)
    monkeypatch.setattr(gps_block, "draw_dropout_seed", lambda: next(it))
    bg = b.clone().to(dev)
    bg.x.requires_grad_(True); bg.edge_attr.requires_grad_(True)
    xg, eg = bg.x, bg.edge_attr
    og = bg
    for l in layers:
        assert (gps_block.block_supported(l, og.x, og.edge_attr) if local == "CustomGatedGCN"
                else gps_block.gine_block_supported(l, og.x, og.edge_attr)), "the fused block path must be the one under test"
        og = l(og)
    loss = (og.x * wx.to(dev)).sum()
    if local == "CustomGatedGCN":
        loss = loss + (og.edge_attr * we.to(dev)).sum()
    loss.backward()
    r32 = _masked_oracle_run(oracle, b, seeds, H, p, p_attn, wx, we, torch.float32, local)
    r64 = _masked_oracle_run(oracle, b, seeds, H, p, p_attn, wx, we, torch.float64, local)
    assert_close(og.x, r64[0], Tol.ACT * n_layers, "out.x (dropout on)")
    if local == "CustomGatedGCN":
        assert_close(og.edge_attr, r64[1], Tol.ACT * n_layers, "out.edge_attr (dropout on)")
    gmax = int((b.ptr[1:] - b.ptr[:-1]).max())
    rx = assert_close_kink_tolerant(xg.grad, r64[2], Tol.GRAD_REL * n_layers, "grad x (dropout on)",
                                    min_allowed_rows=4 * gmax * n_layers)
    print(f"dropout on, {local} x{n_layers}: grad x max rel {rx[0]:.2e} outside {rx[2]} kink rows")
    if local == "CustomGatedGCN":
        re_ = assert_close_kink_tolerant(eg.grad, r64[3], Tol.GRAD_REL * n_layers, "grad e (dropout on)")
        print(f"   grad e max rel {re_[0]:.2e} outside {re_[2]} kink rows")
    # parameter gradients: as in the dropout-off test above, a kink flip at (row r, channel c) lands undamped in row c of a
    # weight gradient (and in element c of a bias / BatchNorm gradient), so a few outlier rows per parameter are allowed and
    # the bar holds outside them; the CPU fp32 evaluation of the SAME masked function is graded next to it
    gscale = max(float(g.abs().max()) for gl in r64[4] for g in gl.values())
    worst, flips = 0.0, 0
    for li, l in enumerate(layers):
        for k, q in l.named_parameters():
            if k not in r64[4][li]:
                continue
            # the bar is the reference's own fp32 arithmetic, not a constant: outside the kink rows the HIP
            # gradient may be no further from fp64 than 5 x what the CPU fp32 evaluation of the same masked function is
            # (max-norm over <= 768 rows: noisier than the rms the model tests grade with x 3; floor: north_star's 1e-5).
            # A fixed 1e-4 would also have passed a 3x regression.
            ms = max(1.0, 0.01 * gscale)
            c32 = _kink_free_err(r32[4][li][k], r64[4][li][k], ms)
            rr = assert_close_kink_tolerant(q.grad, r64[4][li][k], max(1e-5, 5.0 * c32),
                                            f"layer {li} grad {k} (dropout on; cpu fp32 {c32:.1e})", min_scale=ms)
            worst, flips = max(worst, rr[0]), flips + rr[2]
    c_err = max(_kink_free_err(r32[4][li][k], r64[4][li][k], max(1.0, 0.01 * gscale))
                for li in range(n_layers) for k in r64[4][li])
    print(f"   parameter gradients: max rel error vs fp64 {worst:.2e} outside {flips} kink rows "
          f"(cpu-fp32 masked oracle: max {c_err:.2e})")


@pytest.mark.parametrize("d,H,profile,nb,p", [(384, 16, "P30", 256, 0.1), (384, 16, "P14", 64, 0.0), (64, 4, "ZINC", 32, 0.25),
                                              (256, 4, "CODE2_REAL", 8, 0.2)])
def test_gatedgcn_backward_bn_fold_matches_the_apply_launches(d, H, profile, nb, p, monkeypatch):
    """gps_gatedgcn_bwd_bn (round 6: the bn_node_x / bn_edge_e backward applies evaluated inside the GatedGCN backward's
    loads, csrc/gatedgcn.hip FOLD) against the launches it replaces (gps_norm_bwd_apply tasks + gps_gatedgcn_bwd): the same
    layer, the same dropout seeds, every gradient to 2e-6 of its largest element (the arithmetic per element is the same
    expression; only fp32 contraction may differ between the two kernels).  Reference: gatedgcn_layer.py:72-83."""
    from graphgps_amd.layer import gps_block
    from graphgps_amd.layer.gps_layer import GPSLayer
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    layer = GPSLayer(d, "CustomGatedGCN", "Transformer", H, dropout=p, attn_dropout=0.0).to(dev).train()
    b = layer_batch(profile, nb, d, seed=17)
    gen = torch.Generator().manual_seed(8)
    wx = torch.randn(b.x.shape, generator=gen).to(dev)
    we = torch.randn(b.edge_attr.shape, generator=gen).to(dev)
    res = {}
    for fold in (3, 2, 1, 0):
        monkeypatch.setattr(gps_block, "_GG_BN_FOLD", fold)
        monkeypatch.setattr(gps_block, "draw_dropout_seed", lambda: 0x0F1E2D3C4B5A6978)
        layer.zero_grad(set_to_none=True)
        bg = b.clone().to(dev)
        bg.x.requires_grad_(True); bg.edge_attr.requires_grad_(True)
        xg, eg = bg.x, bg.edge_attr
        assert gps_block.block_supported(layer, bg.x, bg.edge_attr)
        og = layer(bg)
        ((og.x * wx).sum() + (og.edge_attr * we).sum()).backward()
        res[fold] = dict(x=xg.grad.clone(), e=eg.grad.clone(), **{k: q.grad.clone() for k, q in layer.named_parameters()
                                                                  if q.grad is not None})
    worst = 0.0
    # (the bias gradients of A / B / D / E are column sums of a BatchNorm backward output -- zero in exact arithmetic, pure
    # rounding in fp32: every tensor is graded on a scale no smaller than 1e-3 of the largest gradient of the layer)
    floor = 1e-3 * max(float(g.abs().max()) for g in res[0].values())
    for k, ref in res[0].items():
        scale = max(float(ref.abs().max()), floor)
        if k.endswith(".bias") and k[:-4] + "weight" in res[0]:     # ... and a bias on the scale of its weight's gradient
            scale = max(scale, float(res[0][k[:-4] + "weight"].abs().max()))
        for fold in (3, 2, 1):
            err = float((res[fold][k] - ref).abs().max()) / scale
            worst = max(worst, err)
            assert err <= 2e-6, f"{k}: folded ({fold}) vs launched BatchNorm backward differ by {err:.2e} of max|grad|"
    print(f"BN-backward fold in gps_gatedgcn_bwd_bn, {profile} d={d} p={p}: worst relative difference {worst:.2e}")


@pytest.mark.parametrize("d,H,profile,nb,p,p_attn,n_layers", [
    (256, 4, "CODE2_REAL", 32, 0.2, 0.5, 1),      # ogbg-code2-GPS.yaml: d = 256, 4 x 64-wide heads, dropout 0.2 / 0.5
    (256, 4, "CODE2_REAL", 16, 0.2, 0.5, 2),
    (64, 2, "P30", 64, 0.1, 0.0, 1),              # a narrow layer: inner = 128 != d
])
def test_performer_block_with_dropout_on_vs_masked_oracle(d, H, profile, nb, p, p_attn, n_layers, monkeypatch):
    """The CustomGatedGCN+Performer layer as ONE autograd node (round 4: layer/gps_block.py; BASELINE configs[4]) with the
    config's dropout rates ON, against the oracle with the block's masks injected -- graphgps/layer/gps_layer.py:111-114,
    206,212-229, performer_layer.py:119-144,200-205,500-503.  Same bars as the Transformer block's test; FAVOR+ couples
    all rows of a graph, so a ReLU-kink flip moves a whole graph's gradient rows: those rows are counted, not widened."""
    from graphgps_amd.layer import gps_block
    from graphgps_amd.layer.gps_layer import GPSLayer
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = [GPSLayer(d, "CustomGatedGCN", "Performer", H, dropout=p, attn_dropout=p_attn) for _ in range(n_layers)]
    oracle = [_oracle_layer_like(l) for l in layers]
    for l in layers:
        l.to(dev).train()
    b = layer_batch(profile, nb, d, seed=93)
    gen = torch.Generator().manual_seed(8)
    wx = torch.randn(b.x.shape, generator=gen)
    we = torch.randn(b.edge_attr.shape, generator=gen)
    seeds = [0x0FEDCBA987654321 + 131 * i for i in range(n_layers)]
    it = iter(seeds)
    monkeypatch.setattr(gps_block, "draw_dropout_seed", lambda: next(it))
    bg = b.clone().to(dev)
    bg.x.requires_grad_(True); bg.edge_attr.requires_grad_(True)
    xg, eg = bg.x, bg.edge_attr
    og = bg
    for l in layers:
        assert gps_block.block_supported(l, og.x, og.edge_attr), "the fused block path must be the one under test"
        og = l(og)
    ((og.x * wx.to(dev)).sum() + (og.edge_attr * we.to(dev)).sum()).backward()
    r32 = _masked_oracle_run(oracle, b, seeds, H, p, p_attn, wx, we, torch.float32, "CustomGatedGCN", "Performer")
    r64 = _masked_oracle_run(oracle, b, seeds, H, p, p_attn, wx, we, torch.float64, "CustomGatedGCN", "Performer")
    # the CPU fp32 evaluation of the same masked function is the yardstick: FAVOR+'s exp / normaliser chain at 100-1000
    # keys is worth a few 1e-6 in fp32 on either side
    o32 = float((r32[0].double() - r64[0]).abs().max())
    tol = max(Tol.ACT * n_layers, 3.0 * o32)
    assert_close(og.x, r64[0], tol, "out.x (Performer block, dropout on)")
    assert_close(og.edge_attr, r64[1], Tol.ACT * n_layers, "out.edge_attr")
    gmax = int((b.ptr[1:] - b.ptr[:-1]).max())
    g32 = float((r32[2].double() - r64[2]).abs().max() / max(1.0, float(r64[2].abs().max())))
    rx = assert_close_kink_tolerant(xg.grad, r64[2], max(Tol.GRAD_REL * n_layers, 3.0 * g32), "grad x",
                                    min_allowed_rows=4 * gmax * n_layers)
    re_ = assert_close_kink_tolerant(eg.grad, r64[3], Tol.GRAD_REL * n_layers, "grad e")
    print(f"Performer block x{n_layers}, dropout on: out.x err {float((og.x.double().cpu() - r64[0]).abs().max()):.2e} "
          f"(cpu fp32 {o32:.2e}); grad x {rx[0]:.2e} outside {rx[2]} rows (cpu fp32 {g32:.2e}); grad e {re_[0]:.2e}")
    gscale = max(float(g.abs().max()) for gl in r64[4] for g in gl.values())
    worst = 0.0
    for li, l in enumerate(layers):
        for k, q in l.named_parameters():
            if k not in r64[4][li]:
                continue
            c32 = float((r32[4][li][k].double() - r64[4][li][k]).abs().max()) / max(1.0, 0.01 * gscale)
            rr = assert_close_kink_tolerant(q.grad, r64[4][li][k], max(2e-5, 3.0 * c32), f"layer {li} grad {k}",
                                            min_scale=max(1.0, 0.01 * gscale))
            worst = max(worst, rr[0])
    print(f"   parameter gradients: max rel error vs fp64 {worst:.2e}")


def test_code2_model_vs_oracle():
    """The 4-layer ``ogbg-code2-GPS.yaml`` model (ASTNode/ASTEdge encoders, 4 x CustomGatedGCN+Performer at
    d=256, ogb_code_graph head, sub-token cross entropy) on 32 code2-long graphs, dropout off: every one of the
    5 prediction heads, the loss and all parameter gradients vs the oracle model -- three-way (HIP, the CPU fp32
    oracle, its fp64 evaluation).

    Pass 1 compares the whole batch.  Its parameter gradients carry a floor of 1e-4 rms, and round 2 only ASSERTED
    why: a ReLU pre-activation within fp32 rounding of 0 lands on different sides in two fp32 evaluation orders, the
    forward moves by 1e-7, but the gate of that element flips, and FAVOR+ couples all ~800 rows of its graph, so that
    graph's whole gradient contribution moves.  Pass 2 DEMONSTRATES it: the error of the gradient w.r.t. the first
    layer's input is attributed graph by graph (a flip is confined to its own graph: the only cross-graph coupling is
    BatchNorm's batch statistics, O(1/rows)); graphs whose error stands out are the flipped ones; with exactly those
    graphs' loss terms weighted 0 -- same inputs, same forward, same flips -- every parameter gradient must agree with
    the fp64 evaluation to 5x what the reference's own fp32 arithmetic achieves, floor 2e-6."""
    import torch.nn.functional as F
    from graphgps_amd.synthetic import model_batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("code2_gps.yaml", 2, 5002, ["gt.dropout", 0.0, "gt.attn_dropout", 0.0])
    assert type(model.layers[0].self_attn).__name__ == "SelfAttention" and len(model.layers) == 4
    model.train()
    oracle = to_oracle_model(model)
    model.to(dev)
    b = model_batch("code2", 32, seed=4321)
    B = len(b.ptr) - 1
    assert int((b.ptr[1:] - b.ptr[:-1]).max()) > 900
    import copy
    o64 = copy.deepcopy(oracle).double()

    def run(m, batch, w):
        """(preds, loss, d loss / d (input of layer 0) [N, d], {parameter: grad}); the loss is the reference's
        graphgps/loss/subtoken_prediction_loss.py:6-20 (mean over graphs of the mean over the 5 heads) without its
        .to(float32) cast, with per-graph weights ``w`` (all ones = the reference's loss)."""
        m.zero_grad(set_to_none=True)     # (training mode: the running statistics the passes move play no role)
        hold = {}

        def grab(mod, args):
            args[0].x.retain_grad()
            hold["x"] = args[0].x
        h = m.layers[0].register_forward_pre_hook(grab)
        pred, true = m(batch)
        h.remove()
        w = w.to(pred[0].device, pred[0].dtype)
        loss = sum((F.cross_entropy(p_, true['y_arr'][:, i], reduction='none') * w).sum() / w.sum()
                   for i, p_ in enumerate(pred)) / len(pred)
        loss.backward()
        return pred, loss, hold["x"].grad, {k: q.grad for k, q in m.named_parameters() if q.grad is not None}

    ones = torch.ones(B)
    legs = {"hip": (model, b.clone().to(dev)), "cpu": (oracle, b.clone()), "f64": (o64, _double_batch(b))}
    r = {k: run(m, bb, ones) for k, (m, bb) in legs.items()}
    for i, (a_, b_, c_) in enumerate(zip(r["hip"][0], r["cpu"][0], r["f64"][0])):
        rg, rc, mg, mc = assert_fp32_grade(a_, b_, c_, f"pred[{i}]", floor=1e-5)
        assert_close(a_, b_, max(1e-5, 4 * (mg + mc)), f"pred[{i}] (hip vs cpu oracle)")
    assert_close(r["hip"][1], r["cpu"][1], 1e-5, "loss")

    def grade(res, floor, factor, tag):
        # every parameter's error relative to ITS OWN largest gradient element, but never below 1 % of the largest
        # gradient element of the whole model (round 2 divided by max(1, ...): with gradients of O(1e-2) that hid two
        # orders of magnitude)
        gscale = max(float(q.abs().max()) for q in res["f64"][3].values())
        worst = (0.0, 0.0, "")
        for k, g64 in res["f64"][3].items():
            if k not in res["hip"][3] or k not in res["cpu"][3]:
                continue
            rg, rc, _, _ = assert_fp32_grade(res["hip"][3][k], res["cpu"][3][k], g64, f"grad {k} ({tag})", floor=floor,
                                             factor=factor, min_scale=0.01 * gscale)
            worst = max(worst, (rg, rc, k))
        print(f"code2 model, {tag}: largest rms error vs fp64 relative to the parameter's own scale (model max {gscale:.2e}): "
              f"hip {worst[0]:.2e} (cpu-fp32 oracle {worst[1]:.2e}) on {worst[2]}")
        return worst
    # (floor: the flipped graphs' contribution, a chaotic quantity -- WHICH pre-activations sit within rounding of 0 changes with
    # any reordering of fp32 arithmetic upstream.  Rounds 2 - 5 measured 0.6 - 0.9e-4; round 6, with the statistics of x~ / e^
    # coming out of the GatedGCN forward's own accumulation order, 1.09e-4 on one parameter.  Pass 2 is the assertion.)
    grade(r, 2e-4, 5.0, "all 32 graphs")

    # ---- attribution: whose gradient moved?  error of d loss / d x0 per graph, relative to the largest fp64 element ----
    g64 = r["f64"][2]
    scale = float(g64.abs().max())
    ptr = b.ptr.tolist()

    def per_graph(err):
        e = err.detach().double().cpu().abs().max(dim=1).values / scale
        return torch.stack([e[ptr[g]:ptr[g + 1]].max() for g in range(B)])
    eh, ec = per_graph(r["hip"][2].cpu() - g64), per_graph(r["cpu"][2] - g64)
    med = float(torch.cat([eh, ec]).median())
    thr = max(20 * med, 2e-6)
    out_h = [g for g in range(B) if float(eh[g]) > thr]
    out_c = [g for g in range(B) if float(ec[g]) > thr]
    print(f"   per-graph error of d loss / d x0 vs fp64 (rel. to max {scale:.2e}): median {med:.1e}; above {thr:.1e}: "
          f"hip {len(out_h)} graphs {out_h} (max {float(eh.max()):.1e}), cpu-fp32 {len(out_c)} graphs {out_c} "
          f"(max {float(ec.max()):.1e})")
    # mechanism, checked where every intermediate is visible (the reference arithmetic itself): ReLU pre-activations whose
    # SIGN differs between the fp32 and the fp64 evaluation of the oracle, per graph
    def relu_signs(m, batch):
        rec, hooks = [], []
        for lay in m.layers:
            for mod in (lay.local_model.act_fn_x, lay.local_model.act_fn_e, lay.act_fn_ff):
                hooks.append(mod.register_forward_hook(lambda _m, inp, _o: rec.append(inp[0].detach() > 0)))
        with torch.no_grad():
            m(batch)
        for h in hooks:
            h.remove()
        return rec
    s32, s64 = relu_signs(oracle, b.clone()), relu_signs(o64, _double_batch(b))
    node_graph = b.batch
    edge_graph = b.batch[b.edge_index[1]]
    flips = torch.zeros(B, dtype=torch.long)
    for a_, c_ in zip(s32, s64):
        rows = (a_ != c_).any(dim=1).nonzero().flatten()
        gmap = node_graph if a_.shape[0] == node_graph.shape[0] else edge_graph
        flips += torch.bincount(gmap[rows], minlength=B)
    flipped = [g for g in range(B) if int(flips[g]) > 0]
    print(f"   ReLU sign flips, CPU fp32 oracle vs its fp64 evaluation: {int(flips.sum())} rows in graphs {flipped}")
    missing = [g for g in out_c if g not in flipped]
    assert not missing, f"cpu-fp32 outlier graphs {missing} contain no flipped pre-activation: another mechanism is at work"
    # graphs WITHOUT a flip agree to the noise level on the CPU leg: the outliers ARE the flips
    quiet = [g for g in range(B) if g not in flipped]
    if quiet:
        assert float(ec[quiet].max()) <= thr, "a graph without any flipped pre-activation is an outlier on the CPU leg"
    # and the HIP evaluation is hit no more often / no harder than the reference's own fp32 arithmetic
    assert len(out_h) <= 2 * len(out_c) + 4, (len(out_h), len(out_c))
    assert float(eh.max()) <= 10 * max(float(ec.max()), thr), (float(eh.max()), float(ec.max()))


def test_gpslayer_edge_permutation_and_determinism():
    """Shuffling edge order must not change node outputs (and permutes edge outputs); two
    identical runs are bitwise identical (no atomics anywhere on the path)."""
    from graphgps_amd.layer.gps_layer import GPSLayer
    from graphgps_amd.synthetic import layer_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    layer = GPSLayer(384, "CustomGatedGCN", "Transformer", 16).to(dev).train()
    b = layer_batch("P30", 256, 384, seed=3).to(dev)
    perm = torch.randperm(b.edge_index.shape[1], device=dev)
    b1, b2, b3 = b.clone(), b.clone(), b.clone()
    b3.edge_index = b.edge_index[:, perm].contiguous()
    b3.edge_attr = b.edge_attr[perm].contiguous()
    with torch.no_grad():
        o1, o2, o3 = layer(b1), layer(b2), layer(b3)
    assert torch.equal(o1.x, o2.x) and torch.equal(o1.edge_attr, o2.edge_attr)
    # BN batch statistics are sums over all rows in a different order -> tiny fp32 differences
    assert_close(o3.x, o1.x, Tol.ACT, "x under edge permutation")
    assert_close(o3.edge_attr, o1.edge_attr[perm], Tol.ACT, "edge_attr under edge permutation")


def _build_model(cfg_name, dim_in, dim_out, overrides=()):
    import graphgps_amd as g
    return g.create_model(os.path.join(g.CONFIG_DIR, cfg_name), list(overrides), dim_in, dim_out)


@pytest.mark.parametrize("cfg_name,kind,dim_in,nb", [
    ("pcqm4m_gpsmedium_rwse.yaml", "pcqm4m", 9, 256),
    ("zinc_gps_rwse.yaml", "zinc", 1, 32),
])
def test_full_model_vs_oracle(cfg_name, kind, dim_in, nb):
    """10-layer stack, dropout off (train mode, BN batch stats): prediction, loss and every parameter gradient
    vs the CPU oracle model -- and vs the SAME oracle evaluated in fp64, which is what makes the tolerance a
    statement: the fp32 CPU oracle is itself only an approximation of the function (per-layer rounding ~1e-6
    compounds through 10 BN-normalised layers; a ReLU pre-activation within rounding of 0 flips sides), so
    (i) the HIP prediction / gradients must be as close to fp64 as the CPU fp32 oracle is (rms, factor 3), and
    (ii) the direct HIP-vs-CPU-oracle difference is bounded by the figures that (i) supports."""
    import copy
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.synthetic import model_batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model(cfg_name, dim_in, 1, ["gt.dropout", 0.0, "gt.attn_dropout", 0.0])
    model.train()
    oracle = to_oracle_model(model)
    o64 = copy.deepcopy(oracle).double()
    model.to(dev)
    b = model_batch(kind, nb, seed=1234)
    po, to_ = oracle(b.clone())
    lo, _ = compute_loss(po, to_)
    lo.backward()
    p64, t64 = o64(_double_batch(b))
    l64, _ = compute_loss(p64, t64)
    l64.backward()
    pg, tg = model(b.clone().to(dev))
    lg, _ = compute_loss(pg, tg)
    lg.backward()
    rg, rc, mg, mc = assert_fp32_grade(pg, po, p64, "pred")
    print(f"pred  vs fp64: hip rms {rg:.2e} max {mg:.2e} | cpu-fp32 rms {rc:.2e} max {mc:.2e}")
    assert_close(pg, po, max(1e-5, 4 * (mg + mc)), "pred (hip vs cpu oracle)")
    assert_close(lg, lo, 1e-5, "loss")
    op, o6 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    gscale = max(float(q.grad.abs().max()) for q in o6.values() if q.grad is not None)
    worst_ratio, worst_hip, worst_cpu, worst_direct = (0.0, ""), 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        if op[k].grad is None or p.grad is None:
            continue
        rg, rc, mg, mc = assert_fp32_grade(p.grad, op[k].grad, o6[k].grad, f"grad {k}",
                                           min_scale=max(1.0, 0.01 * gscale))
        worst_ratio = max(worst_ratio, (rg / max(rc, 1e-9), k))
        worst_hip, worst_cpu = max(worst_hip, mg), max(worst_cpu, mc)
        # the direct comparison, bounded by what the two fp64 distances allow (triangle inequality)
        worst_direct = max(worst_direct, assert_close(p.grad, op[k].grad, max(1e-5, 1.5 * (mg + mc)), f"grad {k}",
                                                      rel_to_max=True))
    print(f"parameter gradients vs fp64 (max over parameters of max|err|/max|g|): hip {worst_hip:.2e}, "
          f"cpu-fp32 oracle {worst_cpu:.2e}; worst hip/cpu rms ratio {worst_ratio[0]:.2f} ({worst_ratio[1]}); "
          f"hip vs cpu oracle directly {worst_direct:.2e}")


def test_full_model_with_dropout_on_vs_masked_oracle(monkeypatch):
    """The MEASURED configuration at model depth: the 10-layer pcqm4m-GPSmedium+RWSE ``GPSModel``
    -- encoders, 10 fused blocks with dropout 0.1 / attention dropout 0.1, ``san_graph`` head, L1 -- on 256 P30 graphs
    against the oracle model with every block's seven masks injected (per-layer seeds; graphgps/layer/gps_layer.py:139-140,
    152-153, network/gps_model.py:12-51), evaluated in fp32 and in fp64.  Prediction, loss and every parameter gradient
    three-way: the HIP error against fp64 may not exceed 3 x the CPU fp32 oracle's own (rms; floor 1e-6)."""
    import copy
    from graphgps_amd.layer import gps_block
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.synthetic import model_batch
    from helpers import inject_dropout_masks
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("pcqm4m_gpsmedium_rwse.yaml", 9, 1)          # the config's own dropout: 0.1 / 0.1
    model.train()
    n_layers = len(model.layers)
    H, d = model.layers[0].num_heads, model.layers[0].dim_h
    p, p_attn = float(model.layers[0].dropout_attn.p), float(model.layers[0].attn_dropout)
    assert n_layers == 10 and p == 0.1 and p_attn == 0.1
    b = model_batch("pcqm4m", 256, seed=1234)
    N, E = b.x.shape[0], b.edge_index.shape[1]
    seeds = [0x1BADB002 + 7919 * i for i in range(n_layers)]
    it = iter(seeds)
    monkeypatch.setattr(gps_block, "draw_dropout_seed", lambda: next(it))

    mask_cache = {}

    def oracle_run(dtype):
        o = to_oracle_model(model)
        if dtype == torch.float64:
            o = o.double()
        o.train()
        masks = _block_masks(list(o.layers), b, seeds, N, E, d, H, p, p_attn, "CustomGatedGCN", cache=mask_cache)
        with inject_dropout_masks(masks):
            pred, true = o(_double_batch(b) if dtype == torch.float64 else b.clone())
        loss, _ = compute_loss(pred, true)
        loss.backward()
        return pred, loss, {k: q.grad for k, q in o.named_parameters() if q.grad is not None}
    p32, l32, g32 = oracle_run(torch.float32)
    p64, l64, g64 = oracle_run(torch.float64)
    model.to(dev)
    pg, tg = model(b.clone().to(dev))
    lg, _ = compute_loss(pg, tg)
    lg.backward()
    rg, rc, mg, mc = assert_fp32_grade(pg, p32, p64, "pred (10 layers, dropout on)")
    print(f"pred vs fp64: hip rms {rg:.2e} max {mg:.2e} | cpu-fp32 masked oracle rms {rc:.2e} max {mc:.2e}")
    assert abs(float(lg) - float(l64)) <= max(1e-5, 3 * abs(float(l32) - float(l64))) * max(1.0, abs(float(l64)))
    gscale = max(float(g.abs().max()) for g in g64.values())
    worst = (0.0, "")
    n = 0
    for k, q in model.named_parameters():
        if k not in g64 or q.grad is None:
            continue
        rg, rc, mg, mc = assert_fp32_grade(q.grad, g32[k], g64[k], f"grad {k}", min_scale=max(1.0, 0.01 * gscale))
        worst = max(worst, (rg / max(rc, 1e-9), k))
        n += 1
    assert n > 250
    print(f"{n} parameter gradients three-way; worst hip/cpu rms ratio {worst[0]:.2f} ({worst[1]})")


def test_full_model_train_step_with_dropout_runs():
    """The measured configuration (dropout 0.1 / attn_dropout 0.1): finite loss and gradients,
    dropout actually active (two steps differ), eval mode deterministic."""
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.synthetic import model_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("pcqm4m_gpsmedium_rwse.yaml", 9, 1).to(dev).train()
    b = model_batch("pcqm4m", 64, seed=5).to(dev)
    losses = []
    for _ in range(2):
        pred, true = model(b.clone())
        loss, _ = compute_loss(pred, true)
        loss.backward()
        losses.append(loss.item())
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert losses[0] != losses[1]
    # the blocks' BatchNorm counters are bumped once per forward by ONE launch for the whole stack (gps_block.stack_end),
    # and no layer keeps a stale set of pre-split weight images
    counters = {k: int(v) for k, v in model.state_dict().items() if k.endswith("num_batches_tracked")}
    assert counters and set(counters.values()) == {2}, counters
    assert not any("_presplit" in vars(m) for m in model.modules())
    model.eval()
    with torch.no_grad():
        p1, _ = model(b.clone())
        p2, _ = model(b.clone())
    assert torch.equal(p1, p2)
    assert {int(v) for k, v in model.state_dict().items() if k.endswith("num_batches_tracked")} == {2}


@pytest.mark.parametrize("block", [True, False])
def test_block_and_operator_paths_match_fixture(monkeypatch, block):
    """Two host-side implementations of the same block: the one-autograd-node path
    (layer/gps_block.py: merged [N,d]x[d,7d] projection GEMM, hand-written backward; default) and
    the operator-by-operator path (GPS_FUSED_BLOCK=0).  Both must meet the reference fixtures."""
    import graphgps_amd.layer.gps_layer as gl
    monkeypatch.setattr(gl, "_BLOCK_ENABLED", block)
    calls, calls_gine = [], []
    orig, orig_gine = gl.gps_block, gl.gps_block_gine
    monkeypatch.setattr(gl, "gps_block", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(gl, "gps_block_gine", lambda *a, **k: (calls_gine.append(1), orig_gine(*a, **k))[1])
    for name in ("gatedgcn_transformer_d32h4", "gatedgcn_transformer_d48h2", "gine_transformer_d32h2"):
        _check_against_fixture(name)
    assert bool(calls) == block and bool(calls_gine) == block


@pytest.mark.parametrize("nb,profile", [(64, "P30"), (48, "P14")])
def test_eval_mode_block_matches_operator_path_and_oracle(monkeypatch, nb, profile):
    """model.eval() under no_grad -- eval_epoch / inference (graphgps/train/custom_train.py:50-77): the inference form of
    the block (layer/gps_block.py gps_block_eval: BatchNorms on their running statistics, no dropout, no statistics
    tasks) against the operator-by-operator path and against the CPU oracle in eval mode, on a 3-layer GPS-medium model
    whose running statistics are NOT the initial (0, 1): layer outputs (node and edge streams) and predictions."""
    import graphgps_amd.layer.gps_layer as gl
    from graphgps_amd.synthetic import model_batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("pcqm4m_gpsmedium_rwse.yaml", 9, 1, ["gt.layers", 3])
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) * 1.5 + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    model.eval()
    oracle = to_oracle_model(model).eval()
    model.to(dev)
    b = model_batch("pcqm4m", nb, seed=77, profile=profile)
    calls = []
    orig = gl.gps_block_eval
    monkeypatch.setattr(gl, "gps_block_eval", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])

    def run(m, batch):
        with torch.no_grad():
            batch = m.encoder(batch)
            batch = m._run_stack(m.layers, batch) if hasattr(m, "_run_stack") and batch.x.is_cuda else m.layers(batch)
            x, e = batch.x.clone(), batch.edge_attr.clone()
            pred, _ = m.post_mp(batch)
        return x, e, pred
    xo, eo, po = run(oracle, b.clone())
    monkeypatch.setattr(gl, "_EVAL_BLOCK", False)
    x0, e0, p0 = run(model, b.clone().to(dev))
    assert not calls
    monkeypatch.setattr(gl, "_EVAL_BLOCK", True)
    x1, e1, p1 = run(model, b.clone().to(dev))
    assert len(calls) == 3, calls
    torch.cuda.synchronize()
    tol = 2e-5          # three BatchNorm-normalised layers of unit-scale activations (Tol.ACT per layer and stream)
    for what, got, ref in (("x (block vs oracle)", x1, xo), ("e (block vs oracle)", e1, eo), ("pred (block vs oracle)", p1, po),
                           ("x (operator path vs oracle)", x0, xo), ("pred (operator path vs oracle)", p0, po),
                           ("x (block vs operator path)", x1, x0), ("e (block vs operator path)", e1, e0),
                           ("pred (block vs operator path)", p1, p0)):
        assert torch.isfinite(got).all(), what
        assert_close(got, ref, tol * max(1.0, float(ref.abs().max())), what)
    # the training-mode path is untouched by an eval pass in between: one training step still runs
    model.train()
    pt, _ = model(b.clone().to(dev))
    assert torch.isfinite(pt).all()


@pytest.mark.parametrize("layer_type,residual", [("gatedgcnconv", True), ("gineconv", True),
                                                 ("gatedgcnconv", False)])
def test_custom_gnn_vs_oracle(layer_type, residual):
    """custom_gnn (graphgps/network/custom_gnn.py): a 3-layer GatedGCN / GINE stack + GraphGym's default
    graph head on the HIP kernels vs the oracle: prediction, loss and every parameter gradient."""
    from graphgps_amd.synthetic import model_batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("pcqm4m_gpsmedium_rwse.yaml", 9, 3,
                         ["model.type", "custom_gnn", "gnn.layer_type", layer_type, "gnn.layers_mp", 3,
                          "gnn.dim_inner", 64, "gt.dim_hidden", 64, "gnn.head", "graph",
                          "gnn.layers_post_mp", 2, "gnn.residual", residual, "gnn.dropout", 0.0])
    assert type(model).__name__ == "CustomGNN"
    model.train()
    oracle = to_oracle_model(model)
    model.to(dev)
    b = model_batch("pcqm4m", 48, seed=11)
    po, _ = oracle(b.clone())
    lo = (po ** 2).mean() + po.sum() * 0.01
    lo.backward()
    pg, _ = model(b.clone().to(dev))
    lg = (pg ** 2).mean() + pg.sum() * 0.01
    lg.backward()
    assert_close(pg, po, 1e-4, "pred")
    assert_close(lg, lo, 1e-5, "loss")
    op = dict(oracle.named_parameters())
    checked = 0
    for k, p in model.named_parameters():
        if op[k].grad is None or p.grad is None:
            continue
        assert_close(p.grad, op[k].grad, 1e-3, f"grad {k}", rel_to_max=True)
        checked += 1
    assert checked > 20


def _ragged_batch(sizes, d, seed, isolated=()):
    """Graphs of the given sizes: random tree + a chord each way, both directions adjacent (OGB order);
    graphs listed in ``isolated`` get no edges at all (isolated nodes: empty CSR segments)."""
    from graphgps_amd.data import Batch
    gen = torch.Generator().manual_seed(seed)
    src, dst, ptr = [], [], [0]
    for gi, n in enumerate(sizes):
        base = ptr[-1]
        if gi not in isolated:
            for v in range(1, n):
                u = int(torch.randint(max(0, v - 4), v, (1,), generator=gen))
                src += [base + u, base + v]
                dst += [base + v, base + u]
            if n > 5:
                a, b = 0, n - 1
                src += [base + a, base + b]
                dst += [base + b, base + a]
        ptr.append(base + n)
    N = ptr[-1]
    ei = torch.tensor([src, dst], dtype=torch.int64)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    b = Batch(x=torch.randn(N, d, generator=gen), edge_index=ei,
              edge_attr=torch.randn(ei.shape[1], d, generator=gen), batch=batch,
              ptr=torch.tensor(ptr, dtype=torch.int64))
    b.num_graphs = len(sizes)
    return b


@pytest.mark.parametrize("local,glob,d,H,es", [("CustomGatedGCN", "Transformer", 32, 4, False),   # block path
                                               ("GINE", "Transformer", 48, 2, False),            # operator path
                                               ("CustomGatedGCN", "Performer", 64, 1, False),
                                               ("CustomGatedGCN", "Transformer", 32, 4, True),   # EquivStableLapPE gate
                                               ("GINE", "Transformer", 48, 2, True)])
def test_gpslayer_ragged_and_boundary_graph_sizes(local, glob, d, H, es):
    """Graph sizes straddling every tile boundary of the kernels (1, 2, 15..17, 31..33, 63..65, 129),
    single-node graphs, a graph made of isolated nodes, the last graph ending exactly at N: outputs and
    gradients of one training-mode layer vs the oracle."""
    from graphgps_amd.layer.gps_layer import GPSLayer
    sizes = [1, 17, 2, 16, 33, 1, 15, 64, 31, 65, 32, 63, 129, 5, 1]
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    layer = GPSLayer(d, local, glob, H, dropout=0.0, attn_dropout=0.0, equivstable_pe=es)
    oracle = _oracle_layer_like(layer).train()
    layer.to(dev).train()
    b = _ragged_batch(sizes, d, seed=9, isolated=(13,))
    gen = torch.Generator().manual_seed(6)
    if es:
        b.pe_EquivStableLapPE = torch.randn(b.x.shape[0], d, generator=gen) * 0.5
    wx, we = torch.randn(b.x.shape, generator=gen), torch.randn(b.edge_attr.shape, generator=gen)
    bc = b.clone()
    bc.x.requires_grad_(True); bc.edge_attr.requires_grad_(True)
    xo, eo = bc.x, bc.edge_attr
    if es:
        bc.pe_EquivStableLapPE = bc.pe_EquivStableLapPE.clone().requires_grad_(True)
    oo = oracle(bc)
    loss_o = (oo.x * wx).sum() + ((oo.edge_attr * we).sum() if local == "CustomGatedGCN" else 0.0)
    loss_o.backward()
    bg = b.clone().to(dev)
    bg.x.requires_grad_(True); bg.edge_attr.requires_grad_(True)
    xg, eg = bg.x, bg.edge_attr
    if es:
        bg.pe_EquivStableLapPE = bg.pe_EquivStableLapPE.clone().requires_grad_(True)
    og = layer(bg)
    loss_g = (og.x * wx.to(dev)).sum() + ((og.edge_attr * we.to(dev)).sum() if local == "CustomGatedGCN" else 0.0)
    loss_g.backward()
    assert_close(og.x, oo.x, Tol.ACT, "out.x")
    assert_close(og.edge_attr, oo.edge_attr, Tol.ACT, "out.edge_attr")
    assert_close(xg.grad, xo.grad, Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(eg.grad, eo.grad, Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    if es:
        assert_close(bg.pe_EquivStableLapPE.grad, bc.pe_EquivStableLapPE.grad, Tol.GRAD_REL, "grad pe",
                     rel_to_max=True)
    op = dict(oracle.named_parameters())
    gscale = max(float(q.grad.abs().max()) for q in op.values() if q.grad is not None)
    for k, p in layer.named_parameters():
        if op[k].grad is None:
            continue
        diff = (p.grad.detach().cpu().double() - op[k].grad.double()).abs().max().item()
        assert diff <= 1e-4 * max(float(op[k].grad.abs().max()), 0.01 * gscale, 1.0), f"grad {k}: {diff:.3e}"


@pytest.mark.parametrize("variant", ["plain", "token"])
def test_graphormer_layer_and_encoders_match_reference_fixture(variant):
    """Reference graphormer_encoder.py + graphormer_layer.py fixture (oracle/gen_golden.py:run_graphormer) on
    the GPU: BiasEncoder -> NodeEncoder (device torch ops) -> GraphormerLayer (varlen MFMA attention with the
    dense bias operand, dh = 10): attn_bias, encoded x, layer output, d(attn_bias) from the kernel, and every
    parameter gradient of the three modules (the embedding tables get theirs THROUGH the kernel's d_bias)."""
    from conftest import GRAPHORMER_GOLDEN
    from graphgps_amd.data import Batch
    from graphgps_amd.encoder.graphormer_encoder import BiasEncoder, NodeEncoder
    from graphgps_amd.layer.graphormer_layer import GraphormerLayer
    dev = torch.device("cuda:0")
    fix = load_golden(GRAPHORMER_GOLDEN)[variant]
    token = variant == "token"
    H, D = fix["H"], fix["D"]
    data = Batch.from_graph_list(fix["pre"])
    data.x = fix["x0"].clone()
    data = data.to(dev)
    bias_enc = BiasEncoder(H, fix["dist"], 4, use_graph_token=token)
    node_enc = NodeEncoder(D, 16, 16, input_dropout=0.0, use_graph_token=token)
    layer = GraphormerLayer(D, H, dropout=0.0, attention_dropout=0.0, mlp_dropout=0.0)
    bias_enc.load_state_dict(fix["state"]["bias"], strict=True)
    node_enc.load_state_dict(fix["state"]["node"], strict=True)
    layer.load_state_dict(fix["state"]["layer"], strict=True)      # checkpoint interchange contract
    for m in (bias_enc, node_enc, layer):
        m.to(dev).train()
    data = node_enc(bias_enc(data))
    assert_close(data.attn_bias, fix["attn_bias"], Tol.ACT, "attn_bias")
    assert_close(data.x, fix["x_enc"], Tol.ACT, "encoded x")
    assert torch.equal(data.batch.cpu(), fix["batch_after"])
    bias = data.attn_bias
    bias.retain_grad()
    data = layer(data)
    (data.x * fix["w"].to(dev)).sum().backward()
    assert_close(data.x, fix["out_x"], Tol.ACT, "layer out")
    assert_close(bias.grad, fix["grad_attn_bias"], Tol.GRAD_REL, "grad attn_bias", rel_to_max=True)
    for part, mod in (("bias", bias_enc), ("node", node_enc), ("layer", layer)):
        got = dict(mod.named_parameters())
        gs = max(float(v.abs().max()) for v in fix["grads"][part].values())
        for k, g in fix["grads"][part].items():
            a_, b_ = got[k].grad.detach().double().cpu(), g.double()
            assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
                f"grad {part}.{k}: {(a_ - b_).abs().max().item():.3e}"


@pytest.mark.parametrize("cfg_name,overrides", [
    # configs/Graphormer/zinc-Graphormer.yaml: graph token, graph_token pooling, 8 heads x dh 10
    ("zinc_graphormer.yaml", ["graphormer.num_layers", 3, "graphormer.attention_dropout", 0.0,
                              "graphormer.mlp_dropout", 0.0, "graphormer.input_dropout", 0.0]),
    # configs/GPS/zinc-GPSwGraphormer.yaml: GINE + BiasedTransformer GPS layers, bias from the same encoder
    ("zinc_gps_graphormer_rwse.yaml", ["gt.layers", 3, "gt.attn_dropout", 0.0]),
])
def test_graphormer_models_vs_oracle(cfg_name, overrides):
    """Whole models on the Graphormer attention bias: encoder (host shortest-path statistics -> device
    BiasEncoder) -> layers on the HIP kernels -> head, against the oracle model: prediction, loss and every
    parameter gradient (the bias embeddings receive theirs through the kernel's d_bias)."""
    from graphgps_amd.encoder.graphormer_encoder import add_graphormer_stats
    from graphgps_amd.synthetic import model_batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model(cfg_name, 1, 1, overrides)
    model.train()
    oracle = to_oracle_model(model)
    model.to(dev)
    b = add_graphormer_stats(model_batch("zinc", 24, seed=5))
    po, _ = oracle(b.clone())
    lo = (po ** 2).mean() + po.sum() * 0.01
    lo.backward()
    pg, _ = model(b.clone().to(dev))
    lg = (pg ** 2).mean() + pg.sum() * 0.01
    lg.backward()
    assert_close(pg, po, 1e-4, "pred")
    assert_close(lg, lo, 1e-5, "loss")
    op = dict(oracle.named_parameters())
    checked = 0
    for k, p in model.named_parameters():
        if op[k].grad is None or p.grad is None:
            continue
        assert_close(p.grad, op[k].grad, 1e-3, f"grad {k}", rel_to_max=True)
        checked += 1
    assert checked > 20
    assert any("spatial_encoder" in k and p.grad is not None and float(p.grad.abs().max()) > 0
               for k, p in model.named_parameters())


def _random_digraph_batch(sizes, d, seed):
    """Directed random graphs with a few duplicate edges and self loops (GCN's normalisation cases)."""
    from graphgps_amd.data import Batch
    gen = torch.Generator().manual_seed(seed)
    parts, off = [], 0
    for n in sizes:
        ei = torch.randint(0, n, (2, 4 * n), generator=gen)
        parts.append(torch.cat([ei, ei[:, :3]], dim=1) + off)
        off += n
    ei = torch.cat(parts, dim=1)
    ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)))
    b = Batch(x=torch.randn(off, d, generator=gen), edge_index=ei,
              batch=torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)), ptr=ptr)
    b.num_graphs = len(sizes)
    return b, gen


@pytest.mark.parametrize("glob,bn,act", [("Transformer", False, "gelu"), ("None", True, "relu")])
def test_gpslayer_gcn_local_model_vs_oracle(glob, bn, act):
    """GPSLayer with the GCN local model (gps_layer.py:53-55,183; configs/GPS/actor-GPS.yaml: GCN+Transformer,
    GELU, no normalisation, no edge attributes) on the HIP sparse core vs the oracle's restatement of PyG
    GCNConv: output, input gradient and every parameter gradient."""
    from graphgps_amd.layer.gps_layer import GPSLayer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    d, H = 64, 4
    layer = GPSLayer(d, "GCN", glob, H, act=act, dropout=0.0, attn_dropout=0.0, batch_norm=bn)
    with torch.no_grad():
        layer.local_model.bias.uniform_(-0.2, 0.2)
    assert {"local_model.lin.weight", "local_model.bias"} <= set(layer.state_dict())
    oracle = _oracle_layer_like(layer).train()
    layer.to(dev).train()
    b, gen = _random_digraph_batch([150, 33, 1, 64], d, seed=9)
    wx = torch.randn(b.x.shape, generator=gen)
    bc = b.clone()
    bc.x.requires_grad_(True)
    xo = bc.x
    oo = oracle(bc)
    (oo.x * wx).sum().backward()
    bg = b.clone().to(dev)
    bg.x.requires_grad_(True)
    xg = bg.x
    og = layer(bg)
    (og.x * wx.to(dev)).sum().backward()
    assert_close(og.x, oo.x, Tol.ACT, "out.x")
    assert_close(xg.grad, xo.grad, Tol.GRAD_REL, "grad x", rel_to_max=True)
    op = dict(oracle.named_parameters())
    gs = max(float(p.grad.abs().max()) for p in op.values() if p.grad is not None)
    for k, p in layer.named_parameters():
        if op[k].grad is None:
            continue
        a_, b_ = p.grad.detach().double().cpu(), op[k].grad.double()
        if float(b_.abs().max()) < 1e-5 * gs:
            # a bias that feeds a BatchNorm: mathematically zero gradient, rounding residue on both sides
            assert float(a_.abs().max()) < 1e-5 * gs, k
            continue
        assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"


def test_actor_gps_model_vs_oracle():
    """configs/GPS/actor-GPS.yaml shape: LapPE DeepSet encoder -> 2 x GPSLayer(GCN+Transformer, GELU, no norm)
    -> GraphGym node head with the split mask, one transductive graph (attention over all 1,200 nodes of it):
    prediction and parameter gradients vs the oracle model.  Eval mode: the LapPE encoder's training-mode sign
    flips draw from different generators on the two devices."""
    from graphgps_amd.data import Batch
    from oracle.gps_oracle import to_oracle_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _build_model("actor_gps.yaml", 32, 5)
    assert type(model.post_mp).__name__ == "GNNNodeHead"
    model.eval()
    oracle = to_oracle_model(model).eval()
    model.to(dev)
    gen = torch.Generator().manual_seed(1)
    N = 1200
    vecs = torch.randn(N, 4, generator=gen)
    vecs[:7, 3] = float("nan")                       # graphs with fewer eigenvectors pad with NaN
    vals = torch.randn(N, 4, 1, generator=gen)
    vals[:7, 3] = float("nan")
    b = Batch(x=torch.randn(N, 32, generator=gen), edge_index=torch.randint(0, N, (2, 6 * N), generator=gen),
              batch=torch.zeros(N, dtype=torch.long), ptr=torch.tensor([0, N]), EigVals=vals, EigVecs=vecs,
              y=torch.randint(0, 5, (N,), generator=gen), train_mask=torch.rand(N, generator=gen) < 0.6)
    b.split, b.num_graphs = "train", 1
    po, yo = oracle(b.clone())
    lo = torch.nn.functional.cross_entropy(po, yo)
    lo.backward()
    pg, yg = model(b.clone().to(dev))
    lg = torch.nn.functional.cross_entropy(pg, yg)
    lg.backward()
    assert torch.equal(yg.cpu(), yo)
    assert_close(pg, po, 1e-4, "pred")
    assert_close(lg, lo, 1e-5, "loss")
    op = dict(oracle.named_parameters())
    checked = 0
    for k, p in model.named_parameters():
        if op[k].grad is None or p.grad is None:
            continue
        assert_close(p.grad, op[k].grad, 1e-3, f"grad {k}", rel_to_max=True)
        checked += 1
    assert checked > 15


@pytest.mark.parametrize("model", ["MLP", "DeepSet"])
def test_signnet_encoder_matches_reference_fixture(model):
    """SignNet encoder with its GIN aggregations on the CSR segment-sum kernel (csrc/gcn.hip:k_adj_sum)."""
    from test_oracle_golden import _signnet_case
    # gradient tolerance 1e-4: the phi network's first-layer gradients are heavily cancelling sums through
    # BatchNorm over the +v / -v branches (the reference's own float32 value is 2e-3 off its float64 one, see
    # oracle/gen_golden.py:run_signnet); the library BN / GEMM kernels of the device round them differently
    # from the CPU's (1.2e-5 measured); the HIP aggregation itself is an exact-order segment sum
    _signnet_case(model, torch.device("cuda:0"), grad_tol=1e-4)


@pytest.mark.parametrize("name", ["SANLayer", "SAN2Layer"])
@pytest.mark.parametrize("full_graph", [True, False])
def test_san_layers_on_the_edge_attention_kernel(name, full_graph):
    """SANLayer / SAN2Layer with the real-edge attention on csrc/edge_attn.hip (score -> clamp-exp | per-target
    softmax -> weighted sum, one gather-gate-segment-reduce launch; two launches backward).  full_graph=True: the
    reference-generated fixture (san_layer.py / san2_layer.py run by oracle/gen_golden.py: output, input and
    edge-feature gradients, every parameter gradient); full_graph=False (sparse attention only, no fixture): the same
    module on the CPU, i.e. the torch-op path that the CPU fixture test pins."""
    import copy
    from conftest import SAN_GOLDEN
    from graphgps_amd.data import Batch
    from graphgps_amd.layer import san_layers
    dev = torch.device("cuda:0")
    fix = load_golden(SAN_GOLDEN)[name]
    d, H = fix["d"], fix["H"]
    layer = getattr(san_layers, name)(gamma=fix["gamma"], in_dim=d, out_dim=d, num_heads=H, full_graph=full_graph,
                                      fake_edge_emb=torch.nn.Embedding(1, d), dropout=0.0, layer_norm=False,
                                      batch_norm=True, residual=True)
    if full_graph:
        layer.load_state_dict(fix["state_dict"], strict=True)
    layer.train()
    cpu_layer = copy.deepcopy(layer)
    layer.to(dev)

    def run(mod, device):
        x = fix["x"].clone().to(device).requires_grad_(True)
        e = fix["edge_attr"].clone().to(device).requires_grad_(True)
        b = Batch(x=x, edge_index=fix["edge_index"].to(device), edge_attr=e, batch=fix["batch"].to(device),
                  ptr=fix["ptr"].to(device))
        out = mod(b)
        (out.x * fix["w"].to(device)).sum().backward()
        return out.x.detach().cpu(), x.grad.cpu(), e.grad.cpu(), {k: p.grad.detach().cpu() for k, p in
                                                                  mod.named_parameters() if p.grad is not None}

    calls = []
    import graphgps_amd.ops as ops
    orig = ops.edge_attention
    ops.edge_attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        got = run(layer, dev)
    finally:
        ops.edge_attention = orig
    assert calls, "the HIP edge-attention kernel was not used"
    if full_graph:
        want = (fix["out_x"], fix["grad_x"], fix["grad_edge_attr"], fix["grads"])
    else:
        want = run(cpu_layer, torch.device("cpu"))
    assert_close(got[0], want[0], Tol.ACT, "out.x")
    assert_close(got[1], want[1], Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(got[2], want[2], Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    gs = max(float(v.abs().max()) for v in want[3].values())
    for k, g in want[3].items():
        a_, b_ = got[3][k].double(), g.double()
        assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"
