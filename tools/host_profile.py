#!/usr/bin/env python
"""Where the HOST side of one eager training step goes (the step launches ~345 kernels; bench.py reports the total as
``host_enqueue_ms_per_step``): cProfile over 20 eager steps of the pcqm4m bench configuration, no synchronisation inside
the profiled region, top functions by own time and by cumulative time.

    python tools/host_profile.py [--steps 20] [--top 45]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphgps_amd as g  # noqa: E402
from graphgps_amd.loss.losses import compute_loss  # noqa: E402
from graphgps_amd.optim import FlatAdamW  # noqa: E402
from graphgps_amd.synthetic import model_batch  # noqa: E402
from graphgps_amd.train import TrainStep  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), None, 9, 1).to(dev).train()
    cfg = g.cfg
    batch = model_batch("pcqm4m", 256, seed=1234).to(dev)
    opt = FlatAdamW(model.parameters(), lr=cfg.optim.base_lr, weight_decay=cfg.optim.weight_decay,
                    max_grad_norm=cfg.optim.clip_grad_norm_value if cfg.optim.clip_grad_norm else None)
    ts = TrainStep(model, opt, loss_fn=compute_loss)
    for _ in range(5):
        ts.run_eager(batch.shallow_copy())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ts.run_eager(batch.shallow_copy())
    host = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"unprofiled: host enqueue {host:.2f} ms per step, step {total:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        ts.run_eager(batch.shallow_copy())
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(a.top)
        print(f"===== by {key} (totals over {a.steps} steps) =====")
        print("\n".join(s.getvalue().splitlines()[:a.top + 12]))
