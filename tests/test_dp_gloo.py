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
