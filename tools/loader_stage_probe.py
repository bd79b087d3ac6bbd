"""Where the loader-fed step's host time goes when the padding sits on DeviceLoader's staging thread (bench.py's
`pcie_inclusive_bucketed`: 16.4 ms per step against 9.6 with batches that arrive padded + pinned).  Variants of the same
24-batch stream through DeviceLoader + TrainStep.step_cached, and the phases of one staging pass timed alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgps_amd as g
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


def run(seq, p):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in DeviceLoader((q.shallow_copy() for q in seq), dev, pad=p):
        ts.step_cached(b, max_graphs=12)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(seq) * 1e3


def pinned(b):
    for k, v in list(b.__dict__.items()):
        if torch.is_tensor(v):
            b.__dict__[k] = v.pin_memory()
    return b


run(host + host, pad)                               # meet the buckets, capture
res = {}
pre = [pad(b) for b in host]
pre_pinned = [pinned(pad(b)) for b in host]
for rep in range(2):
    res.setdefault("a_pad_and_pin_on_staging", []).append(run(order2(host), pad))
    res.setdefault("c_pin_only_on_staging", []).append(run(order2(pre), None))
    res.setdefault("b_arrive_padded_pinned", []).append(run(order2(pre_pinned), None))
for k, v in res.items():
    print(f"{k:28s} " + "  ".join(f"{x:6.2f}" for x in v) + " ms/step")

# phases of one staging pass, alone on the main thread (no step running, no second thread)
cs = torch.cuda.Stream(device=dev)
acc = {"pad": 0.0, "pin": 0.0, "h2d_issue": 0.0, "index_issue": 0.0}
for b in host:
    t = time.perf_counter(); pb = pad(b); acc["pad"] += time.perf_counter() - t
    t = time.perf_counter(); pinned(pb); acc["pin"] += time.perf_counter() - t
    with torch.cuda.stream(cs):
        t = time.perf_counter()
        for k, v in list(pb.__dict__.items()):
            if torch.is_tensor(v):
                pb.__dict__[k] = v.to(dev, non_blocking=True)
        acc["h2d_issue"] += time.perf_counter() - t
        t = time.perf_counter(); graph_index_of(pb); acc["index_issue"] += time.perf_counter() - t
    torch.cuda.synchronize()
print("staging phases alone (ms per batch): " + ", ".join(f"{k} {v / NB * 1e3:.3f}" for k, v in acc.items()))
# the replayed step alone on resident batches of the same buckets (no loader)
dev_batches = [b for b in DeviceLoader((q.shallow_copy() for q in pre_pinned), dev, background=False)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in dev_batches:
    ts.step_cached(b, max_graphs=12)
torch.cuda.synchronize()
print(f"step_cached on resident padded batches: {(time.perf_counter() - t0) / NB * 1e3:.2f} ms/step")
