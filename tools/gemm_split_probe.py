#!/usr/bin/env python
"""gps_gemm_nt (bf16-split MFMA GEMM) vs rocBLAS/hipBLASLt fp32 (torch.addmm): error vs fp64 and time at the
block's forward and input-gradient shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphgps_amd import lib as L_  # noqa: E402
from graphgps_amd.lib import check, current_stream, ptr  # noqa: E402

L = L_.load()
dev = torch.device("cuda:0")
N, E, d = 7569, 15348, 384


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


def gemm(x, w, bias=None, cin=None, out=None):
    R, K = x.shape
    M = w.shape[0]
    out = torch.empty(R, M, device=dev) if out is None else out
    check(L.gps_gemm_nt(ptr(x), x.stride(0), ptr(w), w.stride(0), R, M, K, ptr(bias), ptr(cin),
                        cin.stride(0) if cin is not None else 0, ptr(out), out.stride(0),
                        current_stream(dev)), "gps_gemm_nt")
    return out


if os.environ.get("PROBE_TUNE", "1") == "1":
    import graphgps_amd
    graphgps_amd.enable_gemm_tuning()
shapes = [("fwd 7d", N, d, 7 * d), ("fwd C", E, d, d), ("fwd out_proj", N, d, d), ("fwd ff1", N, d, 2 * d),
          ("fwd ff2", N, 2 * d, d), ("dgrad 7d", N, 7 * d, d), ("dgrad ff1", N, 2 * d, d),
          ("odd", 1000, 52, 100), ("tiny", 37, 64, 20)]
tot_lib = tot_mine = 0.0
for name, R, K, M in shapes:
    x = torch.randn(R, K, device=dev)
    w = torch.randn(M, K, device=dev) / K ** 0.5
    b = torch.randn(M, device=dev)
    c = torch.randn(R, M, device=dev)
    ref = x.double() @ w.double().t() + b.double() + c.double()
    mine = gemm(x, w, b, c)
    lib = torch.addmm(b, x, w.t()) + c
    sc = ref.abs().max().item()
    e_m = (mine.double() - ref).abs().max().item() / sc
    e_l = (lib.double() - ref).abs().max().item() / sc
    fl = 2.0 * R * K * M / 1e6
    t_lib = t(lambda: torch.addmm(b, x, w.t()))
    t_mine = t(lambda: gemm(x, w, b))
    if R > 2000:
        tot_lib += t_lib
        tot_mine += t_mine
    print(f"{name:13s} R={R:6d} K={K:5d} M={M:5d}: err {e_m:.1e} (lib {e_l:.1e})  lib {t_lib:6.1f}us "
          f"({fl/t_lib:5.1f} TF)  gps_gemm_nt {t_mine:6.1f}us ({fl/t_mine:5.1f} TF)")
print(f"sum over the block shapes: lib {tot_lib:.0f}us  gps_gemm_nt {tot_mine:.0f}us")
