#!/usr/bin/env python
"""Where a loader-fed step spends its HOST time (gpurun): the un-padded eager leg against the bucketed, replayed leg
(DeviceLoader(pad=BucketPadding()) + TrainStep.step_cached), per batch: time the consumer waits for the loader, the
loader's own pieces (padding, pinning, H2D + graph index enqueue) and the step call.

    python tools/loader_profile.py [n_batches]
"""
import os
import sys
import time
import threading
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import graphgps_amd as g                                        # noqa: E402
from graphgps_amd import loader as L                            # noqa: E402
from graphgps_amd.loss.losses import compute_loss              # noqa: E402
from graphgps_amd.optim import FlatAdamW                        # noqa: E402
from graphgps_amd.synthetic import model_batch                  # noqa: E402
from graphgps_amd.train import TrainStep                        # noqa: E402

ACC = defaultdict(float)
LOCK = threading.Lock()


def timed(name, fn):
    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            with LOCK:
                ACC[name] += time.perf_counter() - t
                ACC[name + "#"] += 1
    return wrapper


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), None, 9, 1).to(dev).train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
    host = [model_batch("pcqm4m", 256, seed=9000 + i) for i in range(n)]
    # instrument the loader's pieces
    L.BucketPadding.__call__ = timed("pad", L.BucketPadding.__call__)
    L.graph_index_of = timed("index_enqueue", L.graph_index_of)
    torch.Tensor.pin_memory = timed("pin_memory", torch.Tensor.pin_memory)
    L.DeviceLoader._stage = timed("stage_total", L.DeviceLoader._stage)
    torch._foreach_copy_ = timed("foreach_copy", torch._foreach_copy_)
    torch.cuda.CUDAGraph.replay = timed("graph_replay_call", torch.cuda.CUDAGraph.replay)
    for mode in ("eager", "bucketed", "bucketed-prepinned"):
        ts = TrainStep(model, opt, loss_fn=compute_loss)
        pad = None if mode == "eager" else L.BucketPadding()
        src = host
        if mode == "bucketed-prepinned":                # what DataLoader(pin_memory=True) would hand over: padding first,
            src = [pad(b) for b in host]                # pinned once, nothing left for the loader thread but the copies
            for b in src:
                for k, v in list(b.__dict__.items()):
                    if torch.is_tensor(v):
                        b.__dict__[k] = v.pin_memory()
            pad = None
        for rep in range(3):                            # pass 0 / 1 meet the buckets, pass 2 is reported
            ACC.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wait = call = 0.0
            it = iter(L.DeviceLoader((b.shallow_copy() for b in src), dev, pad=pad))
            while True:
                tw = time.perf_counter()
                try:
                    b = next(it)
                except StopIteration:
                    break
                tc = time.perf_counter()
                wait += tc - tw
                if mode == "eager":
                    ts._eager_triplet(b)
                else:
                    ts.step_cached(b, max_graphs=12)
                call += time.perf_counter() - tc
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        per = {k: round(v / n * 1e3, 3) for k, v in ACC.items() if not k.endswith("#")}
        cnt = {k[:-1]: int(v / n) for k, v in ACC.items() if k.endswith("#")}
        print(f"{mode:20s} {dt:7.2f} ms/step | consumer waits for the loader {wait / n * 1e3:6.2f} ms, step call "
              f"{call / n * 1e3:6.2f} ms | per batch (ms): {per} | calls per batch: {cnt}", flush=True)
        del ts


if __name__ == "__main__":
    main()
