#!/usr/bin/env python
"""Where do the D2D copies / fills of one training step come from?  (torch.profiler, shapes + stacks)"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphgps_amd as g  # noqa: E402
from graphgps_amd.loss.losses import compute_loss  # noqa: E402
from graphgps_amd.optim import FlatAdamW  # noqa: E402
from graphgps_amd.synthetic import model_batch  # noqa: E402
from graphgps_amd.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = g.create_model(os.path.join(g.CONFIG_DIR, "pcqm4m_gpsmedium_rwse.yaml"), None, 9, 1).to(dev).train()
b = model_batch("pcqm4m", 256, seed=1234).to(dev)
opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
ts = TrainStep(model, opt, loss_fn=compute_loss)
for _ in range(3):
    ts(b.clone())
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts(b.clone())
    torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::copy_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone"]
cnt = Counter()
for ev in prof.events():
    if ev.name in want:
        stack = [s for s in (ev.stack or []) if "graphgps_amd" in s or "bench" in s or "tools/" in s]
        cnt[(ev.name, str(ev.input_shapes)[:80], stack[0][-90:] if stack else "?")] += 1
for (name, shapes, where), n in cnt.most_common(40):
    print(f"{n:4d}  {name:18s} {shapes:80s} {where}")
