"""Host side of the padded-batch path (no GPU): ``loader.BucketPadding`` (what it appends and where), the encoders'
BatchNorm over the real rows of a padded batch against ``nn.BatchNorm1d`` on the un-padded rows, and the loss slice
``TrainStep`` takes.  Reference loop: graphgps/train/custom_train.py:16-47 (every batch through the same step); reference
BatchNorms either side of the layers: graphgps/encoder/kernel_pos_encoder.py:44-47,92-94, graphgps/network/gps_model.py:27-46."""
import collections
import os

import pytest
import torch

from graphgps_amd.loader import BucketPadding
from graphgps_amd.synthetic import make_structure, model_batch


def _sizes(batch):
    return int(batch.x.shape[0]), int(batch.edge_index.shape[1]), int(batch.num_graphs)


@pytest.mark.parametrize("kind,nb", [("pcqm4m", 64), ("zinc", 32), ("code2", 3)])
def test_bucket_padding_appends_dead_graphs_and_self_loops_only(kind, nb):
    b = model_batch(kind, nb, seed=5)
    N, E, B = _sizes(b)
    pad = BucketPadding(node_step=64, edge_step=128, dead_graphs=4, tie_edges=False)
    pb = pad(b)
    Np, Ep, Bp = _sizes(pb)
    assert Np % 64 == 0 and Ep % 128 == 0 and Bp == B + 4
    assert N + 4 <= Np < N + 4 + 64 and E <= Ep < E + 128
    # real rows first and untouched, on every axis
    for k in b.keys():
        v, w = getattr(b, k), getattr(pb, k)
        if not torch.is_tensor(v):
            continue
        if k == "edge_index":
            assert torch.equal(w[:, :E], v)
        elif k == "ptr":
            assert torch.equal(w[:B + 1], v)
        else:
            assert torch.equal(w[:v.shape[0]], v), k
            assert k == "batch" or not w[v.shape[0]:].any(), k  # zeros behind them (ptr / batch / edge_index aside)
        assert getattr(b, k).shape == v.shape                    # the source batch is left alone
    assert pb.x.shape[0] == pb.batch.shape[0] == int(pb.ptr[-1])
    assert pb.edge_attr.shape[0] == Ep and pb.y.shape[0] == Bp
    # dead graphs: every one owns a node, their sizes differ by at most one, batch / ptr agree
    dead = pb.ptr[B + 1:] - pb.ptr[B:-1]
    assert dead.min() >= 1 and dead.max() - dead.min() <= 1 and int(dead.sum()) == Np - N
    assert torch.equal(pb.batch, torch.repeat_interleave(torch.arange(Bp), pb.ptr[1:] - pb.ptr[:-1]))
    # padding edges: self-loops on padding nodes only, spread round-robin
    ps, pd = pb.edge_index[0, E:], pb.edge_index[1, E:]
    assert torch.equal(ps, pd) and (ps >= N).all() and (ps < Np).all()
    if Ep > E:
        deg = torch.bincount(ps - N, minlength=Np - N)
        assert deg.max() - deg.min() <= 1
    assert pb.gps_counts.dtype == torch.int32 and pb.gps_counts.tolist() == [N, E, B]
    meta = vars(pb)["_gps_meta"]
    assert meta["b_real"] == B and meta["padded"] and meta["nmax"] == max(int((b.ptr[1:] - b.ptr[:-1]).max()), int(dead.max()))
    assert "b_real" not in (vars(b).get("_gps_meta") or {})     # the record of the source batch is not written to
    with pytest.raises(ValueError, match="padded already"):
        pad(pb)


def test_bucket_padding_ambiguous_axis_is_an_error_and_known_names_are_not():
    b = model_batch("pcqm4m", 8, seed=1)
    N, E, B = _sizes(b)
    b.mystery = torch.zeros(N, 3)
    b.table = torch.arange(5.0)                                  # on none of the axes: passes through
    b.scalar = torch.tensor(3.0)
    pad = BucketPadding(node_step=64, edge_step=64)
    pb = pad(b)
    assert pb.mystery.shape[0] % 64 == 0                         # first dim matches the node axis only
    assert pb.table is b.table and pb.scalar is b.scalar
    # make nodes == edges: an unknown tensor is ambiguous now, the named ones still resolve
    sizes, ei, bv, ptr, gen, _ = make_structure("P14", 4, 3)
    n = int(ptr[-1])
    from graphgps_amd.data import Batch
    c = Batch(x=torch.zeros(n, 2, dtype=torch.long), edge_index=ei[:, :n].contiguous(), edge_attr=torch.zeros(n, dtype=torch.long),
              batch=bv, ptr=ptr, y=torch.zeros(4))
    c.num_graphs = 4
    pc = pad(c)
    assert pc.x.shape[0] % 64 == 0 and pc.edge_attr.shape[0] % 64 == 0
    c.mystery = torch.zeros(n)
    with pytest.raises(ValueError, match="cannot tell which axis"):
        pad(c)


def test_bucket_padding_ties_the_edge_bucket_to_the_node_bucket():
    """P30 x 256 graphs, 50 shuffled batches: the auto-chosen 3 % steps are 256 nodes / 512 edges; with the edge bucket
    tied to the node bucket the stream falls into <= 4 shapes (so a second-sight capture policy replays >= 90 % of it),
    the average padding stays under 3.5 %."""
    pad = BucketPadding()
    shapes, waste = [], 0.0
    from graphgps_amd.data import Batch
    for s in range(50):
        sizes, ei, bv, ptr, gen, _ = make_structure("P30", 256, 2000 + s)
        n = int(ptr[-1])
        b = Batch(x=torch.zeros(n, 1, dtype=torch.long), edge_index=ei, batch=bv, ptr=ptr)
        b.num_graphs = 256
        pb = pad(b)
        shapes.append((pb.x.shape[0], pb.edge_index.shape[1]))
        waste += (pb.x.shape[0] - n) / n + (pb.edge_index.shape[1] - ei.shape[1]) / ei.shape[1]
    assert (pad.node_step, pad.edge_step) == (256, 512)
    distinct = collections.Counter(shapes)
    assert len(distinct) <= 4, distinct
    assert 1.0 - len(distinct) / 50 >= 0.9
    assert waste / 100 < 0.035, waste / 100


@pytest.mark.parametrize("momentum", [0.1, None])
def test_masked_batch_norm_equals_batchnorm1d_on_the_real_rows(momentum):
    """encoder/encoders.py masked_batch_norm (the RWSE raw-norm / BatchNorm1dNode of a padded batch) against
    nn.BatchNorm1d over the un-padded rows: outputs of the real rows, running statistics, gradients of the input's real
    rows and of (weight, bias) when the padding rows receive no gradient (what the step guarantees)."""
    from graphgps_amd.encoder.encoders import masked_batch_norm
    torch.manual_seed(0)
    R, P, C = 301, 19, 16
    x = torch.rand(R, C) * 3 - 0.5
    ref = torch.nn.BatchNorm1d(C, momentum=momentum)
    got = torch.nn.BatchNorm1d(C, momentum=momentum)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-1, 1)
        got.load_state_dict(ref.state_dict())
    gy = torch.randn(R, C)
    for step in range(3):
        xr = (x + step).clone().requires_grad_(True)
        xp = torch.cat([x + step, torch.full((P, C), 7.0)]).requires_grad_(True)       # padding far from the data
        yr = ref(xr)
        yp = masked_batch_norm(got, xp, torch.tensor([R], dtype=torch.int32))
        assert torch.isfinite(yp).all()
        torch.testing.assert_close(yp[:R], yr, rtol=1e-5, atol=1e-5)
        (yr * gy).sum().backward()
        (yp[:R] * gy).sum().backward()
        torch.testing.assert_close(xp.grad[:R], xr.grad, rtol=1e-4, atol=1e-5)
        assert not xp.grad[R:].any()
        torch.testing.assert_close(got.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(got.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(got.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(got.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
        ref.zero_grad(); got.zero_grad()
    # eval mode: the running statistics, padding or not
    got.eval(); ref.eval()
    from graphgps_amd.encoder.encoders import _batch_norm

    class B:
        gps_counts = torch.tensor([R, 0, 0], dtype=torch.int32)
    torch.testing.assert_close(_batch_norm(got, xp.detach(), B)[:R], ref(x + 2))


def test_rwse_encoder_ignores_padding_rows_in_training_mode():
    """The measured configuration's node encoder (Atom+RWSE: graphgps/encoder/composed_encoders.py:36-58 with the raw
    BatchNorm of kernel_pos_encoder.py:92-94) on a padded CPU batch: real rows and running statistics as on the
    un-padded batch."""
    import graphgps_amd as g
    torch.manual_seed(0)
    model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), ["gt.layers", 1], 9, 1)
    enc = model.encoder.node_encoder
    enc.train()
    import copy
    enc2 = copy.deepcopy(enc)
    b = model_batch("pcqm4m", 16, seed=3)
    N = b.x.shape[0]
    pb = BucketPadding(node_step=64, edge_step=64)(b)
    out = enc(b.clone()).x
    outp = enc2(pb).x
    assert outp.shape[0] == pb.batch.shape[0] and torch.isfinite(outp).all()
    torch.testing.assert_close(outp[:N], out, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(enc2.encoder2.raw_norm.running_var, enc.encoder2.raw_norm.running_var, rtol=1e-5, atol=1e-6)


def test_head_rows_and_padding_support_predicate():
    import graphgps_amd as g
    from graphgps_amd.train import _head_rows, _real_graphs_of, padding_supported
    t = torch.arange(10.0)
    assert torch.equal(_head_rows(t, 4), t[:4])
    assert [x.shape[0] for x in _head_rows([t.view(10, 1), t], 3)] == [3, 3]
    assert _head_rows({"y_arr": t.view(5, 2)}, 2)["y_arr"].shape == (2, 2)
    b = model_batch("pcqm4m", 8, seed=1)
    assert _real_graphs_of(b) is None
    assert _real_graphs_of(BucketPadding(node_step=64, edge_step=64)(b)) == 8
    # round 5: all three fused blocks count the real rows (GINE + Transformer: zinc; CustomGatedGCN + Performer: code2);
    # a layer outside the blocks -- no global model here -- still is not served
    cfgs = {"pcqm4m_gpsmedium_rwse.yaml": (9, 1, [], True), "zinc_gps_rwse.yaml": (1, 1, [], True),
            "code2_gps.yaml": (2, 5002, [], True),
            "zinc_gps_rwse.yaml#none": (1, 1, ["gt.layer_type", "GINE+None"], False)}
    for name, (din, dout, extra, want) in cfgs.items():
        model = g.create_model(os.path.join(g.CONFIG_DIR, name.split("#")[0]), ["gt.layers", 1] + extra, din, dout)
        assert padding_supported(model) is want, name


def _collate_graphs(items):
    from graphgps_amd.data import Batch
    return Batch.from_graph_list(items)


class _Molecules(torch.utils.data.Dataset):
    """Synthetic PCQM-like graphs as dicts (what Batch.from_graph_list collates)."""

    def __init__(self, n):
        from graphgps_amd.synthetic import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS, graph_sizes, molecule_edges
        gen = torch.Generator().manual_seed(11)
        self.items = []
        for k in graph_sizes("P14", n, gen):
            ei = molecule_edges(k, gen)
            self.items.append(dict(
                x=torch.stack([torch.randint(0, d, (k,), generator=gen) for d in ATOM_FEATURE_DIMS], 1),
                edge_index=ei,
                edge_attr=torch.stack([torch.randint(0, d, (ei.shape[1],), generator=gen) for d in BOND_FEATURE_DIMS], 1),
                pestat_RWSE=torch.rand(k, 16, generator=gen), y=torch.randn(1, generator=gen)))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_padding_in_the_dataloader_workers_collate():
    """BucketPadding.collate: the padding as part of a torch DataLoader's collate function, in its worker PROCESSES
    (shuffled, two workers): every emitted batch is padded (gps_counts, dead graphs), the shuffled epoch falls into a
    few shape buckets although (almost) no two raw batches share a shape; padding twice is refused."""
    import pickle
    from torch.utils.data import DataLoader
    ds = _Molecules(64 * 12)
    pad = BucketPadding(node_step=64, edge_step=128)               # fixed steps: every worker holds its own copy
    pickle.loads(pickle.dumps(pad.collate(_collate_graphs)))        # (spawn-safe)
    torch.manual_seed(0)
    dl = DataLoader(ds, batch_size=64, shuffle=True, num_workers=2, collate_fn=pad.collate(_collate_graphs))
    shapes, raw = [], set()
    for b in dl:
        n, e, g = b.gps_counts.tolist()
        assert g == 64 and b.num_graphs == 64 + pad.dead_graphs and b.y.shape[0] == b.num_graphs
        assert b.x.shape[0] % 64 == 0 and b.edge_index.shape[1] % 128 == 0
        assert vars(b)["_gps_meta"]["padded"] and vars(b)["_gps_meta"]["b_real"] == 64
        assert int(b.ptr[64]) == n and int(b.ptr[-1]) == b.x.shape[0]
        raw.add((n, e))
        shapes.append((b.x.shape[0], b.edge_index.shape[1]))
    assert len(shapes) == 12 and len(raw) >= 11
    assert len(set(shapes)) <= 4, set(shapes)
    with pytest.raises(ValueError, match="padded already"):
        pad(b)


def test_switch_interval_is_counted_and_restored():
    """loader._short_switch_interval: lowered by the first background loader, restored by the LAST one to end -- whatever
    the order in which overlapping loaders finish."""
    import sys
    from graphgps_amd import loader as L
    before = sys.getswitchinterval()
    try:
        sys.setswitchinterval(0.005)
        L._short_switch_interval(True)
        assert sys.getswitchinterval() == pytest.approx(2e-4, rel=0.05)
        L._short_switch_interval(True)                 # a second loader while the first is alive
        L._short_switch_interval(False)                # the first one ends: still short
        assert sys.getswitchinterval() == pytest.approx(2e-4, rel=0.05)
        L._short_switch_interval(False)                # the last one ends: restored
        assert sys.getswitchinterval() == pytest.approx(0.005, rel=0.05)
        L._short_switch_interval(False)                # an unmatched exit changes nothing
        assert sys.getswitchinterval() == pytest.approx(0.005, rel=0.05) and L._SWITCH["users"] == 0
        sys.setswitchinterval(1e-4)                    # a process that already runs shorter keeps its value
        L._short_switch_interval(True)
        assert sys.getswitchinterval() == pytest.approx(1e-4, rel=0.05)
        L._short_switch_interval(False)
        assert sys.getswitchinterval() == pytest.approx(1e-4, rel=0.05)
    finally:
        sys.setswitchinterval(before)


def test_background_loader_control_flow_without_a_gpu(monkeypatch):
    """DeviceLoader._iter_background with the device side stubbed out (streams, events, staging): order of the batches,
    a consumer that leaves early (worker stopped, source iterator closed), a worker exception surfaced on the consumer's
    thread -- and the switch interval back at its value after each of them."""
    import sys
    import threading
    from graphgps_amd import loader as L

    class Ev:
        pass

    class Stream:
        def wait_event(self, ev):
            assert isinstance(ev, Ev)

    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: Stream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: Stream())
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(L.DeviceLoader, "_hand_over", classmethod(lambda cls, b, s: None))
    staged = []

    def stage(self, host, copy_stream, ring=None):
        if host == "boom":
            raise RuntimeError("staging failed")
        staged.append((host, threading.current_thread().name))
        return ("dev-" + host, Ev())
    monkeypatch.setattr(L.DeviceLoader, "_stage", stage)
    before = sys.getswitchinterval()

    class Source:
        def __init__(self, items):
            self.items, self.closed = list(items), False

        def __iter__(self):
            return self

        def __next__(self):
            if not self.items:
                raise StopIteration
            return self.items.pop(0)

        def close(self):
            self.closed = True

    dl = L.DeviceLoader(Source(["a", "b", "c", "d"]), "cuda:0", depth=2, background=True)
    assert list(dl) == ["dev-a", "dev-b", "dev-c", "dev-d"]
    assert {t for _, t in staged} == {"gps-device-loader"}              # staged on the worker thread
    assert sys.getswitchinterval() == pytest.approx(before, rel=0.05) and L._SWITCH["users"] == 0
    src = Source([str(i) for i in range(50)])
    for i, b in enumerate(L.DeviceLoader(src, "cuda:0", depth=2, background=True)):
        assert L._SWITCH["users"] == 1 and sys.getswitchinterval() <= 2.1e-4
        if i == 2:
            break                                                        # early exit: GeneratorExit inside the loader
    import gc
    gc.collect()
    assert L._SWITCH["users"] == 0 and sys.getswitchinterval() == pytest.approx(before, rel=0.05)
    assert src.closed and len(src.items) > 40                            # stopped early, source closed
    with pytest.raises(RuntimeError, match="staging failed"):
        list(L.DeviceLoader(Source(["a", "boom", "c"]), "cuda:0", depth=2, background=True))
    assert L._SWITCH["users"] == 0 and sys.getswitchinterval() == pytest.approx(before, rel=0.05)


def test_pinned_ring_reuses_buffers_and_keeps_contents(monkeypatch):
    """loader._PinnedRing (round 5): the staging buffers are per (slot, tensor name), grown to a power of two, reused
    for every later batch that fits, and a slot is not written before the event of its previous use was waited on.
    Pinned allocation needs a device; here ``torch.empty`` is asked for pageable memory instead -- the logic under
    test (views, growth, reuse, the wait) does not depend on where the bytes live."""
    from graphgps_amd import loader as L
    real_empty = torch.empty
    allocs = []

    def empty(*a, pin_memory=False, **kw):
        allocs.append(a[0])
        return real_empty(*a, **kw)
    monkeypatch.setattr(torch, "empty", empty)
    waited = []

    class Ev:
        def __init__(self, tag):
            self.tag = tag

        def synchronize(self):
            waited.append(self.tag)

    ring = L._PinnedRing(3)
    gen = torch.Generator().manual_seed(3)
    for i in range(9):
        slot = ring.next_slot()
        n = [700, 40, 1500, 0, 900, 1500, 2, 5000, 100][i]
        src = {"x": torch.randint(0, 100, (n, 9), generator=gen),
               "pe": torch.randn(n, 16, generator=gen)[:, ::2],                 # non-contiguous source
               "ptr": torch.arange(n + 1, dtype=torch.int32)}
        for k, v in src.items():
            out = L._PinnedRing.stage(slot, k, v)
            assert out.shape == v.shape and out.dtype == v.dtype and out.is_contiguous() and torch.equal(out, v)
            assert out.untyped_storage().data_ptr() == slot["bufs"][k].untyped_storage().data_ptr()
        slot["ready"] = Ev(i)
    assert waited == [0, 1, 2, 3, 4, 5]                  # slot i % 3 waits for the batch that used it three batches ago
    assert all(c & (c - 1) == 0 and c >= 4096 for c in allocs)
    # 3 names x 3 slots first allocations + growth when a slot meets a bigger batch; never one allocation per tensor per batch
    assert 9 <= len(allocs) <= 9 + 8


def test_staging_thread_keeps_host_operators_on_one_thread():
    """loader._SerialHostOps: the intra-op thread count drops to 1 on the thread that enters it and nowhere else, and
    comes back on exit (the staging thread beside the launching thread: DESIGN section 6, round 5)."""
    import threading
    from graphgps_amd import loader as L
    if not L._openmp_runtimes():
        pytest.skip("no OpenMP runtime loaded")
    before = torch.get_num_threads()
    if before < 2:
        pytest.skip("single-threaded torch build / machine")
    seen = {}

    def worker():
        seen["worker_before"] = torch.get_num_threads()
        L._SerialHostOps().__enter__()
        seen["worker_inside"] = torch.get_num_threads()
        x = torch.arange(1 << 18)
        seen["sum"] = int(torch.cat([x, x.new_zeros(1 << 17)]).sum())       # operators above the parallel grain still work
    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert seen["worker_before"] == before and seen["worker_inside"] == 1
    assert seen["sum"] == (1 << 18) * ((1 << 18) - 1) // 2
    assert torch.get_num_threads() == before                                 # the main thread never noticed
    with L._SerialHostOps():
        assert torch.get_num_threads() == 1
    assert torch.get_num_threads() == before


class _PyGLikeBatch:
    """What a torch_geometric ``Batch`` does with ``num_graphs``: a property WITHOUT a setter over
    ``_num_graphs``, and a ``__setattr__`` that files an assignment to it as a key of ``_store`` without raising."""

    def __init__(self, **kw):
        object.__setattr__(self, "_store", {})
        object.__setattr__(self, "_num_graphs", None)
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_graphs(self):
        return self._num_graphs

    def __setattr__(self, k, v):
        if k.startswith("_"):
            object.__setattr__(self, k, v)
        else:
            self._store[k] = v              # 'num_graphs' lands here too: no setter, no exception

    def __getattr__(self, k):
        store = object.__getattribute__(self, "_store")
        if k in store:
            return store[k]
        raise AttributeError(k)

    def keys(self):
        return [k for k in self._store if k != "num_graphs"]

    def __copy__(self):
        out = _PyGLikeBatch.__new__(_PyGLikeBatch)
        object.__setattr__(out, "_store", dict(self._store))
        object.__setattr__(out, "_num_graphs", self._num_graphs)
        for k, v in vars(self).items():
            if k not in ("_store", "_num_graphs"):
                object.__setattr__(out, k, v)
        return out


def test_bucket_padding_sets_num_graphs_on_a_batch_whose_property_has_no_setter():
    """on a PyG ``Batch`` the plain assignment ``out.num_graphs = B + G`` is swallowed by ``_store`` and
    ``batch.num_graphs`` keeps answering B while ptr / batch / y describe B + G graphs; graph_index_of and the pooling
    head read ``num_graphs``."""
    b = model_batch("pcqm4m", 8, seed=2)
    N, E, B = _sizes(b)
    pg = _PyGLikeBatch(**{k: getattr(b, k) for k in b.keys() if torch.is_tensor(getattr(b, k))})
    pg._num_graphs = B
    assert pg.num_graphs == B
    pg.num_graphs = 99                      # the trap itself: no exception, no effect
    assert pg.num_graphs == B
    del pg._store["num_graphs"]
    pad = BucketPadding(node_step=64, edge_step=64, dead_graphs=4)
    out = pad(pg)
    assert out.num_graphs == B + 4 == out.ptr.numel() - 1 == out.y.shape[0]
    assert "num_graphs" not in out._store   # no stray key left behind
    assert pg.num_graphs == B               # the source batch is left alone
    assert out.gps_counts.tolist() == [N, E, B]


def test_bucket_padding_puts_a_node_level_target_on_the_node_axis():
    """``y`` of a node-level task has N rows; it is padded with the node padding, not with G graph rows."""
    b = model_batch("pcqm4m", 8, seed=3)
    N, E, B = _sizes(b)
    b.y = torch.arange(N, dtype=torch.float32)
    pb = BucketPadding(node_step=64, edge_step=64, dead_graphs=4)(b)
    assert pb.y.shape[0] == pb.x.shape[0] and torch.equal(pb.y[:N], b.y) and not pb.y[N:].any()
    b.y = torch.zeros(B + 1)                # on none of the axes: a target must not pass through silently mis-sized
    with pytest.raises(ValueError, match="matches none of the batch's axes"):
        BucketPadding(node_step=64, edge_step=64)(b)
