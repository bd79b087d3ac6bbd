#!/usr/bin/env python
"""Host cost of a library GEMM at a shape it has not seen before (first call) against a repeated shape: the encoders'
GEMMs have the batch's node / edge count as M or K, so a loader whose shapes never repeat pays the first-call cost on every
step (33.8 ms per eager step on never-repeating shapes against ~10 ms on a fixed shape)."""
import time
import torch

dev = torch.device("cuda:0")
w = torch.randn(176, 364, device=dev)
b = torch.randn(364, device=dev)
x0 = torch.randn(9000, 176, device=dev)
torch.addmm(b, x0[:7000], w); torch.cuda.synchronize()
first, again, kdim = [], [], []
for i in range(16):
    M = 7401 + 37 * i
    x = x0[:M]
    g = torch.randn(M, 364, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter(); y = torch.addmm(b, x, w); t1 = time.perf_counter()
    y = torch.addmm(b, x, w); t2 = time.perf_counter()
    gw = x.t() @ g; t3 = time.perf_counter()        # K = M: the weight-gradient form
    gw = x.t() @ g; t4 = time.perf_counter()
    torch.cuda.synchronize()
    first.append(t1 - t); again.append(t2 - t1); kdim.append((t3 - t2, t4 - t3))
ms = lambda v: round(sum(v) / len(v) * 1e3, 3)
print("addmm [M,176]x[176,364], host time per call: first sight of M", ms(first), "ms, same M again", ms(again), "ms")
print("x^T g  [176,M]x[M,364],  host time per call: first sight of M", ms([a for a, _ in kdim]), "ms, same M again",
      ms([b_ for _, b_ in kdim]), "ms")
