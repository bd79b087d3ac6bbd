#!/usr/bin/env python
"""Plain PyTorch (no graphgps_amd code): eager steps on inputs that carry record_stream(other stream), then a
torch.cuda.graph capture of the same step.  Does the runtime alone reproduce the capture_end crash?"""
import sys

import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "record"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 1)).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
x0 = torch.randn(4096, 256, device=dev)
cs = torch.cuda.Stream(device=dev)


def step(x):
    opt.zero_grad(set_to_none=True)
    model(x).square().mean().backward()
    opt.step()


for _ in range(3):
    x = x0.clone()
    if mode == "record":
        x.record_stream(cs)
    elif mode == "alloc_on_cs":
        with torch.cuda.stream(cs):
            x = x0.clone()
        torch.cuda.current_stream().wait_stream(cs)
        x.record_stream(torch.cuda.current_stream())
    step(x)
    del x
torch.cuda.synchronize()
print(mode, "eager done", flush=True)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(x0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step(x0)
g.replay()
torch.cuda.synchronize()
print(mode, "CAPTURE OK", flush=True)
