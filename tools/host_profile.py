#!/usr/bin/env python
"""cProfile of the host side of the benchmark step (where do the ~17 ms of enqueue time go?)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graphgps_amd as g  # noqa: E402
from graphgps_amd.loss.losses import compute_loss  # noqa: E402
from graphgps_amd.synthetic import model_batch  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), None, 9, 1).to(dev).train()
batch = model_batch("pcqm4m", 256, seed=1234).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.0, fused=True)
step = bench.make_step(model, opt, None, batch, compute_loss, 1.0)
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    torch.cuda.synchronize()
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
