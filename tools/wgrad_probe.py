#!/usr/bin/env python
"""gps_wgrad (HIP split-K MFMA weight+bias gradient) vs rocBLAS mm + colsum: error and time per layer shape."""
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


def wgrad(g, x):
    R, M = g.shape
    Nn = x.shape[1]
    gw = torch.empty(M, Nn, device=dev)
    gb = torch.empty(M, device=dev)
    ws = torch.empty(L.gps_wgrad_workspace_floats(R, M, Nn), device=dev)
    check(L.gps_wgrad(ptr(g), g.stride(0), ptr(x), x.stride(0), R, M, Nn, ptr(gw), ptr(gb), ptr(ws),
                      current_stream(dev)), "gps_wgrad")
    return gw, gb


for name, R, k, n in [("xcat 7d", N, d, 7 * d), ("proj ABDE", N, d, 4 * d), ("C", E, d, d), ("in_proj", N, d, 3 * d),
                      ("out_proj", N, d, d), ("ff1", N, d, 2 * d), ("ff2", N, 2 * d, d),
                      ("zinc d64", 738, 64, 64), ("odd", 1000, 52, 100)]:
    x = torch.randn(R, k, device=dev)
    g = torch.randn(R, n, device=dev)
    ref_w = (g.double().t() @ x.double())
    ref_b = g.double().sum(0)
    gw, gb = wgrad(g, x)
    ew = ((gw.double() - ref_w).abs().max() / ref_w.abs().max()).item()
    eb = ((gb.double() - ref_b).abs().max() / ref_b.abs().max()).item()
    lib_w = (g.t().mm(x).double() - ref_w).abs().max().item() / ref_w.abs().max().item()
    fl = 2.0 * R * k * n / 1e6
    t_lib = t(lambda: (g.t().mm(x), g.sum(0)))
    t_mine = t(lambda: wgrad(g, x))
    print(f"{name:10s} R={R:6d} M={n:5d} Nn={k:4d}: err_w={ew:.1e} (lib {lib_w:.1e}) err_b={eb:.1e}  "
          f"lib mm+sum {t_lib:6.1f}us  gps_wgrad {t_mine:6.1f}us ({fl/t_mine:5.1f} TF)")


# ---- grouped: the five problems of one GPS block in one launch -------------------------------
import ctypes
shapes = [("xcat", N, d, 7 * d), ("C", E, d, d), ("out_proj", N, d, d), ("ff1", N, d, 2 * d),
          ("ff2", N, 2 * d, d)]
pairs = [(torch.randn(R, n, device=dev), torch.randn(R, k, device=dev)) for _, R, k, n in shapes]


def grouped():
    n = len(pairs)
    probs = (L_.WgradProblem * n)()
    outs = []
    for q, (g, x) in zip(probs, pairs):
        R, M = g.shape
        Nn = x.shape[1]
        gw = torch.empty(M, Nn, device=dev)
        gb = torch.empty(M, device=dev)
        q.g, q.x, q.gw, q.gb = g.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr()
        q.ldg, q.ldx, q.R, q.M, q.Nn = g.stride(0), x.stride(0), R, M, Nn
        outs.append((gw, gb))
    ws = torch.empty(max(L.gps_wgrad_grouped_workspace_floats(n, probs), 4), device=dev)
    check(L.gps_wgrad_grouped(n, probs, ptr(ws), current_stream(dev)), "gps_wgrad_grouped")
    return outs


outs = grouped()
for (name, *_), (g, x), (gw, gb) in zip(shapes, pairs, outs):
    ref_w = g.double().t() @ x.double()
    ref_b = g.double().sum(0)
    print(f"grouped {name:9s} err_w={((gw.double() - ref_w).abs().max() / ref_w.abs().max()).item():.1e} "
          f"err_b={((gb.double() - ref_b).abs().max() / ref_b.abs().max()).item():.1e}")
fl = sum(2.0 * R * k * n for _, R, k, n in shapes) / 1e6
t_sep = t(lambda: [wgrad(g, x) for g, x in pairs])
t_grp = t(grouped)
print(f"block total: 5 separate launches {t_sep:6.1f}us ({fl/t_sep:5.1f} TF)   grouped {t_grp:6.1f}us "
      f"({fl/t_grp:5.1f} TF)")
