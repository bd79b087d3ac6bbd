"""Per-phase wall time of DeviceLoader's staging thread WHILE the replayed step runs on the main thread (the phases
alone are ~0.5 ms per batch; with the step running the loader-fed step loses ~7 ms -- tools/loader_stage_probe.py)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgps_amd as g
from graphgps_amd import loader as L
from graphgps_amd.loader import BucketPadding, DeviceLoader
from graphgps_amd.loss.losses import compute_loss
from graphgps_amd.ops import graph_index_of
from graphgps_amd.optim import FlatAdamW
from graphgps_amd.synthetic import model_batch
from graphgps_amd.train import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), None, 9, 1).to(dev)
model.train()
cfg = g.cfg
opt = FlatAdamW(model.parameters(), lr=cfg.optim.base_lr, weight_decay=cfg.optim.weight_decay,
                max_grad_norm=cfg.optim.clip_grad_norm_value if cfg.optim.clip_grad_norm else None)
NB = 24
host = [model_batch("pcqm4m", 256, seed=5000 + i, profile="P30") for i in range(NB)]
order2 = lambda seq: [seq[(7 * i + 3) % NB] for i in range(NB)]
pad = BucketPadding()
ts = TrainStep(model, opt, loss_fn=compute_loss)
T = collections.defaultdict(float)
MODE = os.environ.get("TRACE_PIN", "ring")        # ring | alloc | pageable


def stage(self, batch, copy_stream, ring=None):
    t0 = time.perf_counter()
    slot = ring.next_slot() if ring is not None and MODE == "ring" else None
    T["slot_wait"] += time.perf_counter() - t0; t0 = time.perf_counter()
    already = bool((vars(batch).get("_gps_meta") or {}).get("padded"))
    batch = self.pad(batch) if self.pad is not None and not already else self._host_copy(batch)
    vars(batch).pop("_gps_index", None)
    T["pad"] += time.perf_counter() - t0
    with torch.cuda.stream(copy_stream):
        for k in self._keys(batch):
            v = getattr(batch, k, None)
            if torch.is_tensor(v) and v.device != self.device:
                t0 = time.perf_counter(); pinned = v.is_pinned(); T["is_pinned"] += time.perf_counter() - t0
                if not pinned and MODE != "pageable":
                    t0 = time.perf_counter()
                    v = L._PinnedRing.stage(slot, k, v) if slot is not None else v.pin_memory()
                    T["pin"] += time.perf_counter() - t0
                t0 = time.perf_counter(); d = v.to(self.device, non_blocking=True); T["to"] += time.perf_counter() - t0
                setattr(batch, k, d)
        t0 = time.perf_counter(); graph_index_of(batch); T["index"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        ready = torch.cuda.Event(); ready.record(copy_stream)
        if slot is not None:
            slot["ready"] = ready
        T["event"] += time.perf_counter() - t0
    T["batches"] += 1
    return batch, ready


def run(seq, p, traced):
    M = collections.defaultdict(float)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = iter(DeviceLoader((q.shallow_copy() for q in seq), dev, pad=p))
    while True:
        t = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        M["wait_for_batch"] += time.perf_counter() - t; t = time.perf_counter()
        ts.step_cached(b, max_graphs=12)
        M["step_cached_call"] += time.perf_counter() - t
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(seq) * 1e3
    if traced:
        n = max(T["batches"], 1)
        print(f"  {ms:6.2f} ms/step | main thread per step: " + ", ".join(f"{k} {v / len(seq) * 1e3:.2f}" for k, v in M.items()))
        print("      staging thread per batch: " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in T.items() if k != "batches"))
    T.clear()
    return ms


run(host + host, pad, False)
pre = [pad(b) for b in host]
pre_pinned = [pad(b) for b in host]
for b in pre_pinned:
    for k, v in list(b.__dict__.items()):
        if torch.is_tensor(v):
            b.__dict__[k] = v.pin_memory()
L.DeviceLoader._stage = stage
for MODE in ("ring", "alloc", "pageable"):
    print(f"pin mode {MODE}: batches arrive padded, not pinned")
    run(order2(pre), None, True); run(order2(pre), None, True)
MODE = "ring"
print("ring, un-padded batches: padding on the staging thread")
run(order2(host), pad, True); run(order2(host), pad, True)
if os.environ.get("TRACE_ONE_THREAD") == "1":
    print("the same with torch.set_num_threads(1)")
    torch.set_num_threads(1)
    run(order2(host), pad, True); run(order2(host), pad, True)
print("batches arrive padded + pinned")
run(order2(pre_pinned), None, True); run(order2(pre_pinned), None, True)
sw = sys.getswitchinterval()
print("batches arrive padded, not pinned, ring, inline staging (no thread)")
os.environ["X"] = "1"
MODE = "ring"
L._BACKGROUND_DEFAULT = False
run(order2(pre), None, True)
