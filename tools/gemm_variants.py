#!/usr/bin/env python
"""Compare library call variants for the GPS-medium layer GEMMs (which BLAS backend / op form is fastest)."""
import sys
import torch

N, E, d = 7569, 15348, 384
dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    torch.backends.cuda.preferred_blas_library(sys.argv[1])
print("preferred blas:", torch.backends.cuda.preferred_blas_library())


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


for name, R, k, n in [("proj ABDE", N, d, 4 * d), ("C", E, d, d), ("in_proj", N, d, 3 * d),
                      ("out_proj", N, d, d), ("ff1", N, d, 2 * d), ("ff2", N, 2 * d, d)]:
    x = torch.randn(R, k, device=dev)
    w = torch.randn(n, k, device=dev)
    wt = w.t().contiguous()
    b = torch.randn(n, device=dev)
    g = torch.randn(R, n, device=dev)
    gt = g.t().contiguous()
    xt = x.t().contiguous()
    fl = 2.0 * R * k * n / 1e6
    res = {
        "fwd linear+bias": t(lambda: torch.nn.functional.linear(x, w, b)),
        "fwd mm(x,w.t())": t(lambda: x.mm(w.t())),
        "fwd mm(x,wt)": t(lambda: x.mm(wt)),
        "fwd addmm": t(lambda: torch.addmm(b, x, w.t())),
        "dgrad g.mm(w)": t(lambda: g.mm(w)),
        "wgrad g.t().mm(x)": t(lambda: g.t().mm(x)),
        "wgrad (x.t().mm(g))": t(lambda: x.t().mm(g)),
        "wgrad gt.mm(x) [pre-transposed g]": t(lambda: gt.mm(x)),
        "wgrad xt.mm(g) [pre-transposed x]": t(lambda: xt.mm(g)),
    }
    print(f"{name} R={R} k={k} n={n}: " + "  ".join(f"{kk}={v:.0f}us({fl/v:.0f}TF)" for kk, v in res.items()))
