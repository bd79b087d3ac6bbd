#!/usr/bin/env python
"""Probe: weight-gradient GEMM (long K, small output) as split-K batched GEMM via torch.bmm."""
import torch

N, E, d = 7569, 15348, 384
dev = torch.device("cuda:0")


def t(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def splitk(g, x, S):
    R = g.shape[0]
    c = R // S
    R0 = c * S
    part = torch.bmm(g[:R0].view(S, c, -1).transpose(1, 2), x[:R0].view(S, c, -1))   # [S, n, k]
    out = part.sum(0)
    if R0 < R:
        out.addmm_(g[R0:].t(), x[R0:])
    return out


for name, R, k, n in [("proj ABDE", N, d, 4 * d), ("C", E, d, d), ("in_proj", N, d, 3 * d),
                      ("out_proj", N, d, d), ("ff1", N, d, 2 * d), ("ff2", N, 2 * d, d)]:
    x = torch.randn(R, k, device=dev)
    g = torch.randn(R, n, device=dev)
    fl = 2.0 * R * k * n / 1e6
    ref = g.t().mm(x)
    base = t(lambda: g.t().mm(x))
    line = f"{name:10s} mm={base:.0f}us({fl/base:.0f}TF)"
    for S in (4, 8, 16, 32):
        err = (splitk(g, x, S) - ref).abs().max().item() / ref.abs().max().item()
        tt = t(lambda: splitk(g, x, S))
        line += f"  S={S}:{tt:.0f}us({fl/tt:.0f}TF,err{err:.0e})"
    print(line)
