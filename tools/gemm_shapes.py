#!/usr/bin/env python
"""Time the rocBLAS/hipBLASLt fp32 GEMMs of one GPS-medium layer at P30 sizes (fwd, dgrad, wgrad)."""
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
    return a.elapsed_time(b) / iters


tot = 0.0
for name, R, k, n in [("proj ABDE", N, d, 4 * d), ("C", E, d, d), ("in_proj", N, d, 3 * d),
                      ("out_proj", N, d, d), ("ff1", N, d, 2 * d), ("ff2", N, 2 * d, d)]:
    x = torch.randn(R, k, device=dev)
    w = torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    g = torch.randn(R, n, device=dev)
    fl = 2.0 * R * k * n
    f = t(lambda: torch.nn.functional.linear(x, w, b))
    dg = t(lambda: g.mm(w))
    wg = t(lambda: g.t().mm(x))
    tot += f + dg + wg
    print(f"{name:10s} R={R:6d} k={k:4d} n={n:5d}  fwd {f*1e3:6.1f}us {fl/f/1e9:6.1f}TF  "
          f"dgrad {dg*1e3:6.1f}us {fl/dg/1e9:6.1f}TF  wgrad {wg*1e3:6.1f}us {fl/wg/1e9:6.1f}TF")
print(f"sum per layer {tot*1e3:.0f} us  -> {tot*10:.2f} ms per 10-layer step")
