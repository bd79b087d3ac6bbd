#!/usr/bin/env python
"""The grouped weight-gradient launch of one GPS block (csrc/wgrad.hip, 5 problems at P30 x 256 graphs, d = 384) in its two
arithmetic forms -- three bf16 pieces / 6 products and two fp16 pieces / 3 products -- : error against fp64 and HIP-event
time of one hipGraph replay of 20 launches over rotating operands."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graphgps_amd import lib as L_  # noqa: E402
from graphgps_amd.gemm import absmax  # noqa: E402
from graphgps_amd.lib import check, current_stream, ptr  # noqa: E402

if __name__ == "__main__":
    L = L_.load()
    dev = torch.device("cuda:0")
    d, N, E = (int(v) for v in os.environ.get("GEMM_BENCH", "384,7569,15348").split(","))
    shapes = [(N, d, 7 * d), (E, d, d), (N, d, d), (N, d, 2 * d), (N, 2 * d, d)]       # (rows, in, out)
    nset = 3
    sets = []
    for _ in range(nset):
        pairs = [(torch.randn(r, n, device=dev) * 1e-4, torch.randn(r, k, device=dev)) for r, k, n in shapes]
        words = absmax([t for pr in pairs for t in pr])
        outs = [(torch.empty(g.shape[1], x.shape[1], device=dev), torch.empty(g.shape[1], device=dev)) for g, x in pairs]
        sets.append((pairs, words, outs))

    def problems(i, f16):
        pairs, words, outs = sets[i]
        probs = (L_.WgradProblem * len(pairs))()
        for j, (q, (g, x), (gw, gb)) in enumerate(zip(probs, pairs, outs)):
            q.g, q.x, q.gw, q.gb = g.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr()
            q.ldg, q.ldx, q.R, q.M, q.Nn = g.stride(0), x.stride(0), g.shape[0], g.shape[1], x.shape[1]
            if f16:
                q.g_amax, q.x_amax = words[2 * j].data_ptr(), words[2 * j + 1].data_ptr()
        return probs

    ws = torch.empty(max(L.gps_wgrad_grouped_workspace_floats(5, problems(0, False)), 4), device=dev)
    gf = sum(2.0 * r * k * n for r, k, n in shapes) / 1e9
    for f16 in (False, True):
        P = [problems(i, f16) for i in range(nset)]
        check(L.gps_wgrad_grouped(5, P[0], ptr(ws), current_stream(dev)), "gps_wgrad_grouped")
        errs = []
        for (g, x), (gw, gb) in zip(sets[0][0], sets[0][2]):
            ref = g.double().t() @ x.double()
            errs.append(float((gw.double() - ref).abs().max() / ref.abs().max()))
        tt = bench.time_kernel(lambda i: check(L.gps_wgrad_grouped(5, P[i], ptr(ws), current_stream(dev)), "g"), iters=20,
                               nsets=nset)
        print(f"grouped wgrad {'f16x3 ' if f16 else 'bf16x6'}: {tt * 1e3:7.1f} us  ({gf / tt / 1e3:6.1f} TF/s fp32-equivalent)  "
              f"max rel err {max(errs):.2e}", flush=True)
