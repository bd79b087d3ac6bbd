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
        q.put((rank, worst, reducer.num_bytes))
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
    for rank, worst, nbytes in results:
        assert worst < 1e-6, (rank, worst)
        assert nbytes == 423_717 * 4 - (10 - 2) * 0 or nbytes > 0


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
    arena.adopt()
    assert arena.intact() and torch.equal(lins[1].weight, want_w[8:16] * 2)
    # gradients: packed into the flat buffer, None -> inactive and zero
    for p in params:
        p.grad = torch.full_like(p, 3.0)
    extra.bias.grad = None
    arena.pack_grads()
    assert arena.active.tolist() == [1] * 8 + [0] * 0 or arena.active.tolist()[-1] == 0
    assert float(arena.grad_views[-1].abs().sum()) == 0.0
    assert float(arena.grad_views[0].mean()) == 3.0 and lins[0].weight.grad.data_ptr() == arena.grad_views[0].data_ptr()
