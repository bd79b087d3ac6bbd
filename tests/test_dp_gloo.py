"""World-size-2 data-parallel step on CPU (gloo): the bucketed, backward-overlapped gradient
all-reduce of graphgps_amd/dp.py must produce, on every rank, the mean over ranks of the
gradients each rank's sub-batch yields on its own (SURVEY.md section 8e parity statement)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_oracle_model():
    import graphgps_amd as g
    from oracle.gps_oracle import to_oracle_model
    torch.manual_seed(0)
    m = g.create_model(os.path.join(g.CONFIG_DIR, "zinc_gps_rwse.yaml"),
                       ["gt.layers", 2, "gt.attn_dropout", 0.0], 1, 1)
    return to_oracle_model(m).train()


def _local_grads(model, rank):
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.synthetic import model_batch
    model.zero_grad(set_to_none=True)
    pred, true = model(model_batch("zinc", 8, seed=1234 + rank))
    loss, _ = compute_loss(pred, true)
    loss.backward()
    return {k: p.grad.clone() for k, p in model.named_parameters()}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphgps_amd.dp import GradBucketReducer
        from graphgps_amd.loss.losses import compute_loss
        from graphgps_amd.synthetic import model_batch
        model = _make_oracle_model()
        reducer = GradBucketReducer(model)
        names = sorted(b.name for b in reducer.buckets)
        assert names == ["_rest", "layers.0", "layers.1"], names
        for _ in range(2):                      # two steps: buffers are re-armed correctly
            reducer.zero_grad()
            pred, true = model(model_batch("zinc", 8, seed=1234 + rank))
            loss, _ = compute_loss(pred, true)
            loss.backward()
            reducer.finish()
        got = {k: p.grad.clone() for k, p in model.named_parameters()}
        # expected: mean over ranks of single-process gradients
        ref_model = _make_oracle_model()
        per_rank = [_local_grads(ref_model, r) for r in range(world)]
        worst = 0.0
        for k in got:
            want = sum(pr[k] for pr in per_rank) / world
            worst = max(worst, (got[k] - want).abs().max().item() / max(want.abs().max().item(), 1e-30))
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        q.put((rank, worst, reducer.num_bytes, 4 * n_params))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, nbytes, want_bytes in results:
        assert worst < 1e-6, (rank, worst)
        # every trainable parameter is exchanged exactly once, as fp32: the per-step all-reduce volume
        assert nbytes == want_bytes and nbytes > 4 * 80_000, (nbytes, want_bytes)


def test_single_process_reducer_is_a_noop_wrapper():
    from graphgps_amd.dp import GradBucketReducer
    model = _make_oracle_model()
    want = _local_grads(model, 0)
    reducer = GradBucketReducer(model)
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.synthetic import model_batch
    reducer.zero_grad()
    pred, true = model(model_batch("zinc", 8, seed=1234))
    compute_loss(pred, true)[0].backward()
    reducer.finish()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, want[k]), k


# ---------------------------------------------------------------------------------------------
# flat-arena exchange (optim.ParamArena + dp.FlatGradExchange): the N > 1 product path
# ---------------------------------------------------------------------------------------------
def _flat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphgps_amd.dp import FlatGradExchange
        from graphgps_amd.loss.losses import compute_loss
        from graphgps_amd.optim import ParamArena
        from graphgps_amd.synthetic import model_batch
        model = _make_oracle_model()
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        arena = ParamArena(model.parameters())
        for k, p in model.named_parameters():   # adoption moves storage, never values
            assert torch.equal(p, before[k]), k
        ex = FlatGradExchange(arena)
        for _ in range(2):
            for p in arena.params:
                p.grad = None
            pred, true = model(model_batch("zinc", 8, seed=1234 + rank))
            compute_loss(pred, true)[0].backward()
            arena.pack_grads()
            ex.all_reduce()
        got = {k: p.grad.clone() for k, p in model.named_parameters()}
        for p, v in zip(arena.params, arena.grad_views):
            assert p.grad.data_ptr() == v.data_ptr()
        ref_model = _make_oracle_model()
        per_rank = [_local_grads(ref_model, r) for r in range(world)]
        worst = 0.0
        for k in got:
            want = sum(pr[k] for pr in per_rank) / world
            worst = max(worst, (got[k] - want).abs().max().item() / max(want.abs().max().item(), 1e-30))
        q.put((rank, worst, ex.num_bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_arena_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, nbytes in results:
        assert worst < 1e-6, (rank, worst)
        assert nbytes >= 4 * 80_000


# ---------------------------------------------------------------------------------------------
# the whole N > 1 step API: train.TrainStep + optim.FlatAdamW arena + dp.FlatGradExchange, world 2
# ---------------------------------------------------------------------------------------------
def _cpu_flat_adamw():
    """FlatAdamW whose ``step()`` is a torch restatement of csrc/optim.hip over the SAME flat arena (test stand-in:
    the product's step is a HIP kernel and refuses CPU parameters).  Everything else -- arena adoption, gradient
    packing, ``zero_grad``, the buffer the exchange all-reduces -- is the product code."""
    from graphgps_amd.optim import FlatAdamW

    class CpuFlatAdamW(FlatAdamW):
        @torch.no_grad()
        def step(self, closure=None):
            a, g = self.arena, self.param_groups[0]
            if not self._packed:
                self.pack_grads()
            grad = a.flat_g
            if g.get("max_grad_norm"):
                norm = grad.double().norm().float()
                grad = grad * torch.clamp(g["max_grad_norm"] / (norm + 1e-6), max=1.0)
            self.hyper[6] += 1
            t = float(self.hyper[6])
            b1, b2 = g["betas"]
            a.flat_p.mul_(1 - g["lr"] * g["weight_decay"])
            self.exp_avg.mul_(b1).add_(grad, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
            denom = (self.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(g["eps"])
            a.flat_p.addcdiv_(self.exp_avg, denom, value=-g["lr"] / (1 - b1 ** t))
            self._packed = False

    return CpuFlatAdamW


def _trainstep_worker(rank, world, port, q, split=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphgps_amd.dp import FlatGradExchange
        from graphgps_amd.loss.losses import compute_loss
        from graphgps_amd.synthetic import model_batch
        from graphgps_amd.train import TrainStep
        model = _make_oracle_model()                       # the oracle model stands in for the HIP model
        opt = _cpu_flat_adamw()(model.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        ex = FlatGradExchange(opt.arena)
        assert ex.active and ex.num_bytes == opt.arena.num_bytes
        ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=ex, backward_split=model.layers[0] if split else None)
        if split:
            # the cut sits behind layers.0: its range is a proper suffix of the arena, and the two collectives really are
            # started at different points of the backward (the upper one while layers.0 / the encoders have no gradient yet)
            started = []
            real_start = ex.start

            def spy(lo=0, hi=None):
                lower_done = all(p.grad is not None for p in model.layers[0].parameters())
                started.append((lo, hi, lower_done))
                return real_start(lo, hi)
            ex.start = spy
            assert ts._cut is not None and 0 < ts._cut_index < len(opt.arena.params)
        losses = [float(ts(model_batch("zinc", 8, seed=1234 + 10 * step + rank))) for step in range(3)]
        # single-process reference: torch.optim.AdamW + clip_grad_norm_ fed the rank-mean gradient
        ref = _make_oracle_model()
        o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-5)
        for step in range(3):
            grads = []
            for r in range(world):
                ref.zero_grad(set_to_none=True)
                pred, true = ref(model_batch("zinc", 8, seed=1234 + 10 * step + r))
                compute_loss(pred, true)[0].backward()
                grads.append([p.grad.clone() for p in ref.parameters()])
            for i, p in enumerate(ref.parameters()):
                p.grad = sum(g[i] for g in grads) / world
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
            o_ref.step()
        # Adam turns ANY non-zero gradient into a +-lr step, so parameters whose gradient is mathematically zero
        # (biases feeding a BatchNorm: rounding residue on both sides) random-walk apart; the first moment is
        # linear in the gradients and is compared for every parameter, the weights for those with a real signal
        gmax = max(float(o_ref.state[p]["exp_avg"].abs().max()) for p in ref.parameters())
        worst_m = worst = 0.0
        n_cmp = 0
        for off, p_got, p_ref in zip(opt.arena.offsets, opt.arena.params, ref.parameters()):
            m_ref = o_ref.state[p_ref]["exp_avg"]
            m_got = opt.exp_avg[off:off + p_got.numel()].view_as(p_got)
            worst_m = max(worst_m, (m_got - m_ref).abs().max().item() / gmax)
            if float(m_ref.abs().max()) > 1e-3 * gmax and float(m_ref.abs().min()) > 1e-6 * gmax:
                worst = max(worst, (p_got - p_ref).abs().max().item())
                n_cmp += 1
        assert worst_m < 1e-6, worst_m
        assert n_cmp >= 5, n_cmp
        flat = opt.arena.flat_p.detach().clone()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        if split:
            off = int(opt.arena.offsets[ts._cut_index])
            assert ts._cut is not None, "the cut must survive: no parameter of this model is used on both sides of it"
            assert len(started) == 6 and started[0::2] == [(off, None, False)] * 3 and started[1::2] == [(0, off, True)] * 3, started
        q.put((rank, worst, bool(torch.equal(both[0], both[1])), losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trainstep_split_backward_overlapped_exchange_world2_gloo():
    """The exchange SURVEY.md section 8e specifies, in the form the replayed step uses it: the backward cut behind
    ``layers.0`` (TrainStep ``backward_split``), the gradient arena all-reduced as two ranges -- the upper layers' range
    started between the two halves of the backward, before ``layers.0`` and the encoders have any gradient -- and the
    same end state as the single-collective step: ranks bit-identical, weights equal to a single-process AdamW run on the
    rank-mean gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainstep_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same, losses in results:
        assert same, "ranks diverged"
        assert worst < 2e-6, (rank, worst)
        assert all(l == l for l in losses)


@pytest.mark.timeout(300)
def test_trainstep_flat_exchange_world2_gloo():
    """TrainStep([fwd + bwd + pack] | FlatGradExchange.all_reduce | [clip + AdamW]) on two gloo ranks, three steps
    with different per-rank batches: both ranks end with bit-identical weights, equal (to fp32 rounding) to a
    single-process AdamW run fed the mean of the per-rank gradients (SURVEY.md section 8e parity statement)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainstep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same, losses in results:
        assert same, "ranks diverged"
        assert worst < 2e-6, (rank, worst)
        assert all(l == l for l in losses)


def test_param_arena_keeps_linear_group_stacks_and_follows_moves():
    """A LinearGroup's members share one storage; the arena must move them as a block (their
    stacked view stays a view), notice when parameters leave it, and re-adopt."""
    from graphgps_amd.fused import LinearGroup
    from graphgps_amd.optim import ParamArena
    torch.manual_seed(0)
    lins = [torch.nn.Linear(8, 8) for _ in range(3)]
    extra = torch.nn.Linear(8, 3)
    grp = LinearGroup(lins)
    w0, b0 = grp._stacked()
    want_w, want_b = w0.clone(), b0.clone()
    params = [p for l in lins for p in l.parameters()] + list(extra.parameters())
    arena = ParamArena(params)
    assert arena.intact()
    w1, b1 = grp._stacked()                       # re-derived as views of the arena
    assert torch.equal(w1, want_w) and torch.equal(b1, want_b)
    lo, hi = arena.flat_p.data_ptr(), arena.flat_p.data_ptr() + arena.flat_p.numel() * 4
    assert lo <= w1.data_ptr() < hi and lo <= b1.data_ptr() < hi
    for l, rows in zip(lins, range(0, 24, 8)):
        assert l.weight.data_ptr() == w1.data_ptr() + rows * 8 * 4
    # writing through the arena is writing the parameters
    arena.flat_p.mul_(2.0)
    assert torch.equal(lins[1].weight, want_w[8:16] * 2)
    # chunk table covers every parameter element exactly once
    covered = torch.zeros_like(arena.flat_p, dtype=torch.int32)
    for o, n, i in zip(arena.chunk_off.tolist(), arena.chunk_len.tolist(), arena.chunk_param.tolist()):
        covered[o:o + n] += 1
        p = arena.params[i]
        assert arena.offsets[i] <= o and o + n <= arena.offsets[i] + p.numel()
    assert int(covered.sum()) == sum(p.numel() for p in params) and int(covered.max()) == 1
    # a parameter leaves (what .to() / a re-stack does): detected, re-adopted, values kept
    extra.weight.data = extra.weight.data.clone()
    assert not arena.intact()
    n_before = arena.flat_p.numel()
    arena.adopt()
    assert arena.intact() and torch.equal(lins[1].weight, want_w[8:16] * 2)
    # re-adoption re-packs: the parameters that stayed do not drag the old arena's span (and the hole the leaver left)
    # along -- the flat buffer a data-parallel step all-reduces stays the size of the parameters
    assert arena.flat_p.numel() <= n_before
    w2, b2 = grp._stacked()                       # the group adopted earlier is still one contiguous block
    for l, rows in zip(lins, range(0, 24, 8)):
        assert l.weight.data_ptr() == w2.data_ptr() + rows * 8 * 4
    lo2, hi2 = arena.flat_p.data_ptr(), arena.flat_p.data_ptr() + arena.flat_p.numel() * 4
    assert lo2 <= w2.data_ptr() < hi2
    # gradients: packed into the flat buffer, None -> inactive and zero
    for p in params:
        p.grad = torch.full_like(p, 3.0)
    extra.bias.grad = None
    arena.pack_grads()
    assert arena.active.tolist() == [1] * 8 + [0] * 0 or arena.active.tolist()[-1] == 0
    assert float(arena.grad_views[-1].abs().sum()) == 0.0
    assert float(arena.grad_views[0].mean()) == 3.0 and lins[0].weight.grad.data_ptr() == arena.grad_views[0].data_ptr()


def test_param_arena_readoption_keeps_aliased_parameters_tied():
    """Two Parameter objects over ONE range of storage (tied weights) must still alias each other after the arena is
    rebuilt, also when they already lived in the previous arena: re-adoption re-packs that arena's parameters as runs,
    and a run must not end between two parameters that overlap (adjacency-only runs copied them apart)."""
    from graphgps_amd.optim import ParamArena
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(6, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    tied = torch.nn.Parameter(a.weight.data)              # same storage, same offset, a second Parameter object
    params = list(a.parameters()) + [tied] + list(b.parameters()) + list(c.parameters())
    arena = ParamArena(params)
    assert tied.data_ptr() == a.weight.data_ptr()
    c.weight.data = c.weight.data.clone()                 # one parameter leaves: the next adopt() rebuilds the arena
    assert not arena.intact()
    want = a.weight.detach().clone()
    arena.adopt()
    assert arena.intact()
    assert tied.data_ptr() == a.weight.data_ptr(), "tied parameters were copied apart"
    assert torch.equal(a.weight, want) and torch.equal(tied, want)
    with torch.no_grad():
        a.weight.add_(1.0)
    assert torch.equal(tied, want + 1.0)
    lo, hi = arena.flat_p.data_ptr(), arena.flat_p.data_ptr() + arena.flat_p.numel() * 4
    assert all(lo <= p.data_ptr() < hi for p in params)


def _bcast_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hashlib
        import graphgps_amd as g
        from graphgps_amd.dp import broadcast_state
        # ranks seeded DIFFERENTLY: different initial weights and -- the silent one -- different random Performer
        # projection matrices (performer_layer.py:272-273); a code2-shaped model (CustomGatedGCN+Performer layers)
        torch.manual_seed(100 + rank)
        m = g.create_model(os.path.join(g.CONFIG_DIR, "code2_gps.yaml"), ["gt.layers", 2], 1, 1)
        for bn in [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm1d)]:
            bn.running_mean.add_(float(rank + 1))            # buffers that moved during some rank-local warm-up
            bn.num_batches_tracked.add_(rank + 3)

        def digest():
            h = hashlib.sha256()
            for k, t in sorted(list(m.named_parameters()) + list(m.named_buffers()), key=lambda kv: kv[0]):
                h.update(k.encode()); h.update(t.detach().cpu().contiguous().numpy().tobytes())
            return h.hexdigest()
        before = digest()
        proj = [k for k, _ in m.named_buffers() if k.endswith("projection_matrix")]
        assert proj, "the model under test must hold a random Performer projection buffer"
        nbytes = broadcast_state(m)
        after = digest()
        # buffers drift per rank while training (running statistics); dp.broadcast_buffers (eval_epoch, before a
        # checkpoint) brings the BUFFERS back to rank 0's and leaves the parameters alone
        from graphgps_amd.dp import broadcast_buffers

        def part(named):
            h = hashlib.sha256()
            for k, t in sorted(named, key=lambda kv: kv[0]):
                h.update(k.encode()); h.update(t.detach().cpu().contiguous().numpy().tobytes())
            return h.hexdigest()
        with torch.no_grad():
            for bn in [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm1d)]:
                bn.running_var.mul_(1.0 + 0.5 * rank)
            first = next(m.parameters())
            first.add_(float(rank))          # a parameter that differs: must NOT be touched by broadcast_buffers
        m.register_buffer("cpu_side_table", torch.full((3,), float(rank)))      # a second (dtype, device) bucket
        m.cpu_side_table = m.cpu_side_table.double()
        nb2 = broadcast_buffers(m)
        q.put((rank, before, after, nbytes, len(proj), part(m.named_buffers()), part(m.named_parameters()), nb2))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_broadcast_state_makes_differently_seeded_replicas_identical_world2_gloo():
    """dp.broadcast_state: ranks constructed under different seeds (different weights, different random Performer
    projection buffers, different BatchNorm running statistics) hold bit-identical parameters AND buffers afterwards
    (SURVEY.md section 8e)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, n0, k0, buf0, par0, nb0), (_, b1, a1, n1, k1, buf1, par1, nb1) = results
    assert b0 != b1, "the two ranks were supposed to start from different states"
    assert a0 == a1, "replicas differ after broadcast_state"
    assert a0 == b0, "rank 0's state is the one that must survive"
    assert n0 == n1 and n0 > 0 and k0 == 2
    assert buf0 == buf1, "buffers differ after broadcast_buffers"
    assert par0 != par1, "broadcast_buffers must leave the parameters alone"
    assert 0 < nb0 == nb1 < n0
