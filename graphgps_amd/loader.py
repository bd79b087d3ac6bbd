"""Host loader -> device batches, one batch ahead of the step.

The reference moves each batch with a blocking ``batch.to(torch.device(cfg.accelerator))`` at the top of
the iteration (``/root/reference/graphgps/train/custom_train.py:21-22``) and rebuilds the dense/padded
views inside every layer.  Here the next batch is staged while the current step runs: its tensors go
through pinned host memory and a non-blocking H2D copy on a dedicated copy stream, and the per-batch
graph index (CSR by target, CSC by source, ``ptr``, attention tile map: ``ops.build_graph_index``) is
built on that same stream right behind the copies.  The step's stream only waits on one event.

At PCQM4M sizes a batch is ~1.7 MB of int64 features/indices + fp32 RWSE (≈35 µs of PCIe) and the index
build is ≈30 µs, against a 9 ms step, so with one batch of look-ahead neither shows up in the step time.

``BucketPadding`` (round 4) is the other half of feeding the step from a real loader: a shuffled loader never emits the
same (nodes, edges) twice, a captured step replays on ONE shape, so host batches are padded up to a few shape buckets --
in the staging thread here, or in the DataLoader's worker processes (``BucketPadding.collate``) -- with the real row
counts travelling to the device as a tensor of the batch (``gps_counts``).
"""
from __future__ import annotations

import ctypes
import queue
import threading
from collections import deque
from typing import Iterable, Iterator

import torch

import os

from .ops import graph_index_of

# Held while a batch is staged (worker thread) and while a step is captured into a hipGraph (train.TrainStep): stream
# capture on one thread and allocations / copies / launches on another must not interleave on this runtime.
STAGE_LOCK = threading.RLock()
_BACKGROUND_DEFAULT = os.environ.get("GPS_LOADER_BACKGROUND", "1") != "0"


_SWITCH = {"users": 0, "saved": None, "lock": threading.Lock()}


def _short_switch_interval(enter: bool, interval: float = 2e-4) -> None:
    """The interpreter's thread switch interval, lowered while at least one background loader is alive and put back when
    the last one ends (counted: a train and an eval loader may overlap, and each restoring "its" old value would leave the
    process on the short interval for good)."""
    import sys
    with _SWITCH["lock"]:
        if enter:
            if _SWITCH["users"] == 0:
                _SWITCH["saved"] = sys.getswitchinterval()
                sys.setswitchinterval(min(_SWITCH["saved"], interval))
            _SWITCH["users"] += 1
        elif _SWITCH["users"] > 0:
            _SWITCH["users"] -= 1
            if _SWITCH["users"] == 0 and _SWITCH["saved"] is not None:
                sys.setswitchinterval(_SWITCH["saved"])
                _SWITCH["saved"] = None


def _openmp_runtimes():
    """The OpenMP runtime libraries loaded into this process (torch's intra-op pool lives in one of them), as ctypes
    handles -- found in /proc/self/maps, loaded once."""
    libs = _openmp_runtimes.__dict__.get("libs")
    if libs is None:
        import re
        paths = set()
        try:
            with open("/proc/self/maps") as fh:
                for line in fh:
                    m = re.search(r"(/\S*lib(?:g|i)?omp[^/\s]*\.so[^/\s]*)", line)
                    if m:
                        paths.add(m.group(1))
        except OSError:
            pass
        libs = []
        for path in sorted(paths):
            try:
                lib = ctypes.CDLL(path)
                lib.omp_get_max_threads.restype = ctypes.c_int
                lib.omp_set_num_threads.argtypes = [ctypes.c_int]
                libs.append(lib)
            except (OSError, AttributeError):
                pass
        _openmp_runtimes.libs = libs
    return libs


class _SerialHostOps:
    """While active, torch's CPU operators run single-threaded ON THE CALLING THREAD ONLY (OpenMP's thread count is a
    per-thread setting; ``torch.set_num_threads`` would change the whole process).  The staging side of the loader is a
    few dozen host operators on ~1 MB tensors (``cat``, ``new_zeros``, copies); each one above 32k elements fans out
    over the intra-op pool, whose workers then spin on every core for a while.  Alone that is harmless (0.4 ms per
    batch); beside the thread that launches the steps it is not: measured (round 5, tools/loader_stage_trace.py, 24
    pcqm4m batches through DeviceLoader + step_cached) the padding took 10-11 ms per batch and the loader-fed step
    17.1 ms, against 0.3 ms and 9.6 ms with these operators kept on one thread.  An optimisation: silently a no-op when
    no OpenMP runtime is found."""

    def __enter__(self):
        torch.get_num_threads()             # ATen's per-thread lazy initialisation runs first (it sets the count itself)
        self.prev = []
        for lib in _openmp_runtimes():
            try:
                self.prev.append((lib, int(lib.omp_get_max_threads())))
                lib.omp_set_num_threads(1)
            except Exception:
                pass
        return self

    def __exit__(self, *exc):
        for lib, n in self.prev:
            try:
                lib.omp_set_num_threads(max(n, 1))
            except Exception:
                pass
        return False


_NODE_KEYS = ("x", "batch", "node_depth", "node_is_attributed", "EigVecs", "EigVals")
_EDGE_KEYS = ("edge_attr",)
_GRAPH_KEYS = ("y", "y_arr")


def _bucket_step(n: int, frac: float) -> int:
    """Power of two nearest (in the log) to ``frac * n``, at least 64: the granularity of a padded axis."""
    import math
    return max(64, 1 << max(0, round(math.log2(max(frac * n, 1.0)))))


def _set_num_graphs(batch, n: int) -> None:
    """``batch.num_graphs = n`` that is checked.  A ``torch_geometric`` ``Batch`` exposes ``num_graphs`` as a property
    WITHOUT a setter over ``_num_graphs``; its ``__setattr__`` then files the assignment as a new key of ``_store`` and
    raises nothing, and ``batch.num_graphs`` keeps answering the old count while ``ptr`` / ``batch`` / ``y`` describe
    the padded one.  So: assign, read back, and if the answer is not ``n`` set the backing field and remove
    the stray store key."""
    try:
        batch.num_graphs = n
    except Exception:                       # a read-only property that does raise
        pass
    try:
        ok = int(batch.num_graphs) == n
    except Exception:
        ok = False
    if ok:
        return
    vars(batch)["_num_graphs"] = n
    store = vars(batch).get("_store")
    for holder in (store, getattr(store, "_mapping", None)):
        try:
            if holder is not None and "num_graphs" in holder:
                del holder["num_graphs"]
        except Exception:
            pass
    if int(batch.num_graphs) != n:
        raise ValueError(f"BucketPadding: could not set num_graphs = {n} on a {type(batch).__name__} "
                         f"(it answers {batch.num_graphs})")


class _PaddedCollate:
    """``collate_fn`` then ``pad``: a picklable callable (DataLoader workers under the spawn start method)."""

    def __init__(self, pad, collate_fn):
        self.pad, self.collate_fn = pad, collate_fn

    def __call__(self, items):
        return self.pad(self.collate_fn(items))


class BucketPadding:
    """Pads a HOST batch up to a shape bucket so that a captured step can be replayed on it (``TrainStep.step_cached``
    keys on shapes; a shuffled loader never emits the same (nodes, edges) twice: graphgps/train/custom_train.py:16-47
    runs every batch through the same Python, a hipGraph needs the same shapes).

    What is appended, always at the END of every axis so that real rows keep their indices (the dropout masks are
    counter hashes of the row index):
      * ``dead_graphs`` extra graphs, each with >= 1 padding node (features 0), that take the nodes up to the next
        multiple of ``node_step``;
      * padding edges up to the next multiple of ``edge_step``: self-loops on the padding nodes, round-robin -- they
        connect padding to padding only;
      * zero rows for every per-graph tensor (``y``).
    The batch additionally carries ``gps_counts`` = int32 [real nodes, real edges, real graphs] -- a TENSOR, staged
    to the device and copied into a captured step's static inputs like any other: the BatchNorm statistics, their
    gradients' ``1 / R`` and the zero gate on padding rows read the real counts from it (ops.GraphIndex.n_real /
    e_real, csrc/block_norm.hip ``rdev``, csrc/gemm_panel.hip ``m_dev``) -- and ``_gps_meta['b_real']``, the host-side
    number of real graphs: ``TrainStep`` takes the loss over ``pred[:b_real]`` only.  Padding never reaches a real row:
    message passing and attention stay inside a graph, statistics skip padding rows, and every backward tensor is exactly
    zero on them (the loss ignores the dead graphs; the BatchNorm backward gates them), so the row contractions of the
    weight gradients see zeros.

    ``node_step`` / ``edge_step``: 0 = chosen from the first batch (the power of two nearest to 3 % of its size).
    ``tie_edges``: the edge bucket is at least what the node bucket implies through the first batch's edges-per-node
    ratio -- nodes and edges of a batch of molecules move together, and two independent axes would multiply the number
    of live shapes (P30 x 256 graphs, 50 shuffled batches: 5 shapes untied, 3 tied, at 2.5 % average padding)."""

    def __init__(self, node_step: int = 0, edge_step: int = 0, dead_graphs: int = 8, frac: float = 0.03,
                 tie_edges: bool = True):
        self.node_step, self.edge_step = int(node_step), int(edge_step)
        self.dead_graphs = max(int(dead_graphs), 1)
        self.frac = float(frac)
        self.tie_edges = bool(tie_edges)
        self.edges_per_node = None          # of the first batch (tie_edges)

    @staticmethod
    def _kind(key, t, N, E, B):
        if key in _NODE_KEYS or key.startswith("pestat_") or key.startswith("pe_"):
            return "node"
        if key in _EDGE_KEYS:
            return "edge"
        hits = [k for k, n in (("node", N), ("edge", E), ("graph", B)) if t.dim() >= 1 and t.shape[0] == n]
        if key in _GRAPH_KEYS:
            # a target lives on the axis its leading dimension says (graph-level: B rows; node- / edge-level tasks: N / E
            # rows); the name only settles a tie (B == N happens with one-node graphs) and never overrides the shape
            if "graph" in hits:
                return "graph"
            if len(hits) == 1:
                return hits[0]
            raise ValueError(f"BucketPadding: target {key!r} {tuple(t.shape)} matches none of the batch's axes "
                             f"unambiguously (nodes {N}, edges {E}, graphs {B})")
        if len(hits) == 1:
            return hits[0]
        if not hits:            # lives on none of the three axes (a per-batch scalar, a lookup table): left as it is
            return None
        raise ValueError(f"BucketPadding: cannot tell which axis tensor {key!r} {tuple(t.shape)} lives on "
                         f"(nodes {N}, edges {E}, graphs {B})")

    def collate(self, collate_fn):
        """``collate_fn`` followed by the padding: hand it to the ``DataLoader`` (``collate_fn=pad.collate(collater)``) so
        that the padding runs in the loader's WORKER PROCESSES and ``pin_memory=True`` pins the padded batch -- the staging
        thread of ``DeviceLoader`` is then left with the H2D copies and the index build (measured, 256-graph batches of
        never-repeating shapes, round 4: 9.8 ms per step, against 16.0 ms with the padding on the staging thread and 33.8
        ms for the eager step on the un-padded stream.  Round 5 found what the staging-thread form was paying --
        ``_SerialHostOps`` -- and it now runs at the same 9.3 ms; the worker-process form still takes the padding's
        0.3 ms per batch off this process entirely).
        Fix ``node_step`` / ``edge_step`` in the constructor then: every worker process holds its own copy of this object,
        and steps chosen from 'the first batch' would be chosen per worker."""
        return _PaddedCollate(self, collate_fn)

    def __call__(self, batch):
        """A new host batch (new container, new tensors where rows were appended)."""
        ei = batch.edge_index
        if ei.is_cuda:
            raise ValueError("BucketPadding pads HOST batches (before the H2D copy)")
        N, E = int(batch.x.shape[0]), int(ei.shape[1])
        ptr = getattr(batch, "ptr", None)
        bv = getattr(batch, "batch", None)
        B = int(batch.num_graphs)
        if self.node_step <= 0:
            self.node_step = _bucket_step(N, self.frac)
        if self.edge_step <= 0:
            self.edge_step = _bucket_step(max(E, 1), self.frac)
        G = self.dead_graphs
        n_pad = -(-(N + G) // self.node_step) * self.node_step
        e_pad = -(-E // self.edge_step) * self.edge_step
        if self.tie_edges:
            if self.edges_per_node is None:
                self.edges_per_node = E / max(N, 1)
            e_pad = max(e_pad, -(-int(self.edges_per_node * n_pad) // self.edge_step) * self.edge_step)
        pn, pe = n_pad - N, e_pad - E                       # pn >= G: every dead graph owns a node
        sizes = torch.full((G,), pn // G, dtype=torch.long)
        sizes[: pn % G] += 1
        out = DeviceLoader._host_copy(batch)
        vars(out).pop("_gps_index", None)
        for k in DeviceLoader._keys(batch):
            v = getattr(batch, k, None)
            if not torch.is_tensor(v):
                continue
            if k == "edge_index":
                loops = N + torch.arange(pe, dtype=ei.dtype) % pn
                new = torch.cat([v, torch.stack([loops, loops])], dim=1)
            elif k == "ptr":
                new = torch.cat([v, v[-1] + torch.cumsum(sizes, 0).to(v.dtype)])
            elif k == "batch":
                new = torch.cat([v, B + torch.repeat_interleave(torch.arange(G, dtype=v.dtype), sizes)])
            elif k == "gps_counts":
                raise ValueError("BucketPadding: this batch is padded already")
            else:
                kind = self._kind(k, v, N, E, B)
                if kind is None:
                    continue
                rows = {"node": pn, "edge": pe, "graph": G}[kind]
                new = torch.cat([v, v.new_zeros((rows,) + tuple(v.shape[1:]))], dim=0)
            setattr(out, k, new)
        out.gps_counts = torch.tensor([N, E, B], dtype=torch.int32)
        _set_num_graphs(out, B + G)
        if torch.is_tensor(ptr) and ptr.numel() > 1:
            real_max = int((ptr[1:] - ptr[:-1]).max())
        elif torch.is_tensor(bv) and bv.numel():
            real_max = int(torch.bincount(bv).max())
        else:
            real_max = 0
        meta = dict(vars(batch).get("_gps_meta") or {})
        meta.update(nmax=max(real_max, int(sizes.max())), b_real=B, n_real=N, e_real=E, padded=True)
        vars(out)["_gps_meta"] = meta
        return out


_PINNED_RING = os.environ.get("GPS_LOADER_PINNED_RING", "1") != "0"      # 0: ``tensor.pin_memory()`` per tensor (A/B)


class _PinnedRing:
    """Pinned staging buffers that are REUSED: ``slots`` slots, one per batch in flight, each with one pinned byte buffer
    per tensor name (grown to the next power of two when a batch needs more), and the event behind the slot's last H2D
    copies.  ``tensor.pin_memory()`` per tensor per batch is an allocation from the pinned-memory cache plus a copy that
    fans out over the intra-op thread pool (see ``stage``): measured (round 5, tools/loader_stage_probe.py /
    loader_stage_trace.py, profiles/r05_loader_staging_thread.txt) the 24-batch pcqm4m stream ran at 16.7 ms per step with
    the pinning on the staging thread against 9.7 with batches that arrive pinned, although the pinning is 0.13 ms per
    batch when nothing else runs.  With the ring the steady state allocates nothing: one plain memcpy into the slot's
    buffer, the non-blocking copy out of it, and the slot is not written again before its event has completed (9.3 ms)."""

    def __init__(self, slots: int):
        self.slots = [{"bufs": {}, "ready": None} for _ in range(max(int(slots), 2))]
        self.i = 0

    def next_slot(self):
        slot = self.slots[self.i % len(self.slots)]
        self.i += 1
        if slot["ready"] is not None:
            slot["ready"].synchronize()     # the copies that read this slot's buffers are done (normally long ago)
            slot["ready"] = None
        return slot

    @staticmethod
    def stage(slot, key, v):
        """``v`` (a host tensor) copied into the slot's pinned buffer for ``key``: a pinned tensor of v's shape / dtype."""
        nbytes = v.numel() * v.element_size()
        buf = slot["bufs"].get(key)
        if buf is None or buf.numel() < nbytes:
            cap = max(4096, 1 << max(nbytes - 1, 1).bit_length())
            buf = slot["bufs"][key] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        out = buf[:nbytes].view(v.dtype).view(v.shape)
        if nbytes and v.is_contiguous():
            # one plain memcpy on THIS thread (ctypes drops the interpreter lock around it).  ``out.copy_(v)`` splits a
            # copy of more than 32k elements over the intra-op thread pool, whose workers then spin on every core for a
            # while: with the step being launched from the other thread that cost 7 ms per step (round 5,
            # tools/loader_stage_trace.py: this copy 7.7 ms per batch under load, 0.1 ms alone)
            ctypes.memmove(out.data_ptr(), v.data_ptr(), nbytes)
        else:
            out.copy_(v)
        return out


class DeviceLoader:
    """Iterate ``loader`` (host batches) and yield device batches with the graph index attached.

    ``depth`` batches are in flight ahead of the consumer (1 = stage the next batch while the current one
    is consumed).  ``background`` (default): the staging -- pulling the next host batch out of ``loader``, pinning,
    the H2D copies and the index build's launches -- runs on a worker thread, so none of its host time sits between two
    kernel launches of the step (the eager step is enqueue-bound: ~340 launches in ~10 ms; a millisecond of staging on
    the launching thread is a millisecond of idle GPU).  On a CPU ``device`` this is a pass-through: the product path
    has no CPU kernels, and the reference's loop is what the oracle runs."""

    def __init__(self, loader: Iterable, device, depth: int = 2, build_index: bool = True, background: bool = None,
                 pad: "BucketPadding" = None):
        self.loader = loader
        self.pad = pad                      # shape buckets (BucketPadding): applied to the host batch before it is pinned
        self.device = torch.device(device)
        self.depth = max(int(depth), 1)
        self.build_index = build_index
        self.background = _BACKGROUND_DEFAULT if background is None else bool(background)
        self._rings = []                    # idle _PinnedRing objects (one is taken per iteration, handed back at its end)

    def __len__(self) -> int:
        return len(self.loader)

    # -- one batch: pinned staging + async copies + index, all on the copy stream ----------------
    @staticmethod
    def _keys(batch):
        """Attribute names of the batch's tensors.  A ``torch_geometric`` ``Batch`` / ``Data`` keeps them in
        ``batch._store`` (NOT in ``__dict__``) and lists them through ``keys`` (a property in PyG 2.2, a method
        from 2.4 on); ``graphgps_amd.data.Batch`` has the same ``keys()``.  Anything else: public ``__dict__``."""
        ks = getattr(batch, "keys", None)
        if callable(ks):
            ks = ks()
        if ks is None:
            ks = [k for k in vars(batch) if not k.startswith("_")]
        return list(ks)

    @staticmethod
    def _host_copy(batch):
        """A shallow copy of the host batch (same tensors, new container): staging re-points attributes at device
        tensors and the model re-assigns ``batch.x`` / ``batch.edge_attr`` (gps_layer.py:174,231), which must not
        leak into a list-style loader that is iterated again next epoch."""
        if hasattr(batch, "shallow_copy"):
            return batch.shallow_copy()
        import copy
        return copy.copy(batch)             # PyG Data implements __copy__ as a per-store shallow copy

    def _stage(self, batch, copy_stream, ring=None):
        dev = self.device
        slot = ring.next_slot() if ring is not None else None
        already = bool((vars(batch).get("_gps_meta") or {}).get("padded"))     # padded by the DataLoader's collate
        if self.pad is not None and not already:
            try:
                batch = self.pad(batch)
            except ValueError as exc:
                # a tensor whose axis BucketPadding cannot tell (E == N, a pair-indexed operand ...): this loader's batches
                # go un-padded from here on -- slower (fewer replays), never wrong and never an aborted epoch
                import warnings
                warnings.warn(f"DeviceLoader: shape buckets switched off for this loader: {exc}")
                self.pad = None
                batch = self._host_copy(batch)
        else:
            batch = self._host_copy(batch)
        vars(batch).pop("_gps_index", None)
        # what the host can tell the kernels for free while ``ptr`` is still here: the longest graph of the batch
        # (from ``ptr``, or from a host-side ``batch`` vector when the collater emitted no ``ptr``: without the record
        # ops._host_max_graph_nodes would pay a synchronising device read on the copy stream)
        p, bv = getattr(batch, "ptr", None), getattr(batch, "batch", None)
        meta = vars(batch).get("_gps_meta")          # a padded batch (padded here or by the caller) brings its record
        if meta and meta.get("padded") and "nmax" in meta:
            pass
        elif torch.is_tensor(p) and not p.is_cuda and p.numel() > 1:
            vars(batch)["_gps_meta"] = {"nmax": int((p[1:] - p[:-1]).max())}
        elif p is None and torch.is_tensor(bv) and not bv.is_cuda and bv.numel():
            vars(batch)["_gps_meta"] = {"nmax": int(torch.bincount(bv).max())}
        with torch.cuda.stream(copy_stream):
            for k in self._keys(batch):
                v = getattr(batch, k, None)
                if torch.is_tensor(v) and v.device != dev:
                    if v.device.type == "cpu" and not v.is_pinned():
                        v = _PinnedRing.stage(slot, k, v) if slot is not None else v.pin_memory()
                    setattr(batch, k, v.to(dev, non_blocking=True))
            if self.build_index and hasattr(batch, "edge_index"):
                graph_index_of(batch)
            ready = torch.cuda.Event()
            ready.record(copy_stream)
            if slot is not None:
                slot["ready"] = ready
        return batch, ready

    @classmethod
    def _hand_over(cls, batch, stream) -> None:
        """The tensors were allocated on the copy stream: tell the caching allocator the consumer's
        stream uses them, so their blocks are not recycled while the step still reads them."""
        if os.environ.get("GPS_LOADER_RECORD_STREAM", "1") == "0":
            return
        for k in cls._keys(batch):
            v = getattr(batch, k, None)
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(stream)
        gi = vars(batch).get("_gps_index")
        if gi is not None:
            for t in vars(gi).values():
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(stream)

    def __iter__(self) -> Iterator:
        if self.device.type != "cuda":
            for batch in self.loader:
                batch = self._host_copy(batch)
                yield batch.to(self.device) or batch
            return
        copy_stream = torch.cuda.Stream(device=self.device)
        ring = None
        if _PINNED_RING:                    # depth staged + one being consumed + one being written
            ring = self._rings.pop() if self._rings else _PinnedRing(self.depth + 2)
        try:
            if self.background:
                yield from self._iter_background(copy_stream, ring)
            else:
                yield from self._iter_inline(copy_stream, ring)
        finally:
            if ring is not None and not getattr(ring, "abandoned", False):
                self._rings.append(ring)

    def _iter_inline(self, copy_stream, ring=None) -> Iterator:
        pending = deque()
        source = iter(self.loader)

        def fill():
            while len(pending) < self.depth:
                try:
                    host = next(source)
                except StopIteration:
                    return
                with _SerialHostOps():
                    pending.append(self._stage(host, copy_stream, ring))

        fill()
        while pending:
            batch, ready = pending.popleft()
            fill()                                  # next copy is queued before this batch is consumed
            stream = torch.cuda.current_stream(self.device)
            stream.wait_event(ready)
            self._hand_over(batch, stream)
            yield batch

    def _iter_background(self, copy_stream, ring=None) -> Iterator:
        """Staging on a worker thread, ``depth`` staged batches queued ahead of the consumer."""
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        END = object()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        # The iterator is created HERE, on the consumer's thread: a DataLoader draws its base seed from the
        # global CPU generator in __iter__, and drawn on the worker it would race whatever the training thread draws
        # from the same generator (shuffle order no longer a function of the seed alone).
        source = iter(self.loader)
        failure = []

        def worker():
            try:
                torch.cuda.set_device(self.device)
                _SerialHostOps().__enter__()       # this thread's host operators stay on this thread (never undone: the
                for host in source:                # setting is per thread and the thread ends with the iteration)
                    with STAGE_LOCK:
                        item = self._stage(host, copy_stream, ring)
                    if not put(item):
                        return
                put(END)
            except BaseException as exc:       # surfaced on the consumer's thread -- also when the consumer is leaving
                failure.append(exc)            # (`put` gives up once `stop` is set: the exception would be dropped)
                put(exc)

        # Two Python threads now share the interpreter lock: the staging thread (a few dozen small host ops per batch,
        # more with padding) and the caller's, which launches the steps.  With the default 5 ms switch interval the
        # launching thread can sit for milliseconds behind the staging thread every time it comes back from a call
        # that released the lock (a graph replay, a synchronising copy): measured 10.8 ms per replay call against 7.0
        # without a busy staging thread.  A short interval while the loader is alive keeps the hand-over prompt.
        # (Round 5: most of that measured gap was the intra-op pool's spinning workers, not the lock -- _SerialHostOps.
        # The short interval stays: it costs nothing and the lock hand-over is still on the launching thread's path.)
        _short_switch_interval(True)
        th = threading.Thread(target=worker, name="gps-device-loader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ready = item
                stream = torch.cuda.current_stream(self.device)
                stream.wait_event(ready)
                self._hand_over(batch, stream)
                yield batch
        finally:
            # Early exit (a `break` in the consumer, an exception): stop the worker, drain what it staged so that a
            # blocked `put` returns, and close the source iterator (a DataLoader's workers shut down with it).  The
            # worker is a daemon thread: if it is stuck inside the source's `next` for longer than the join below, it is
            # left behind and ends with the process.
            stop.set()
            _short_switch_interval(False)
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            th.join(timeout=5.0)
            if th.is_alive() and ring is not None:
                ring.abandoned = True          # a worker left behind may still write into it: never handed to another iteration
            close = getattr(source, "close", None) or getattr(source, "_shutdown_workers", None)
            if close is not None and not th.is_alive():
                try:
                    close()
                except Exception:
                    pass
            if failure and not isinstance(failure[0], (GeneratorExit, StopIteration)):
                import sys
                if sys.exc_info()[0] is None:      # nothing else is propagating: do not swallow the worker's error
                    raise failure[0]
