#!/usr/bin/env python
"""Per-workgroup timeline of the ring GEMM (gps_gemm_panel_trace): where a launch's time goes at the block's shapes.
Shader-clock stamps: entry, first stage landed, k-loop done, stores done."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphgps_amd import lib as _lib  # noqa: E402
from graphgps_amd.gemm import gemm_panel, split_weights  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    L = _lib.load()
    Nn, E, d = 7569, 15348, 384
    for name, M, K, N in (("pq", Nn, d, 7 * d), ("out", Nn, d, d), ("C", E, d, d), ("dgrad_pq", Nn, 7 * d, d)):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        (img, _), = split_weights([w], tn=False)
        c = torch.empty(M, N, device=dev)
        for _ in range(3):
            gemm_panel(a, img, N, out=c)
        buf = torch.zeros(4 * 4096, dtype=torch.int64, device=dev)
        L.gps_gemm_panel_trace(buf.data_ptr())
        gemm_panel(a, img, N, out=c)
        torch.cuda.synchronize()
        L.gps_gemm_panel_trace(None)
        t = buf.view(-1, 4).cpu()
        t = t[t[:, 3] > 0].double()
        t0 = t[:, 0].min()
        q = lambda x: [round(float(v)) for v in torch.quantile(x, torch.tensor([0.1, 0.5, 0.9], dtype=torch.double))]
        print(f"{name}: {len(t)} workgroups; start {q(t[:, 0] - t0)}  prologue {q(t[:, 1] - t[:, 0])}  loop {q(t[:, 2] - t[:, 1])}  "
              f"epilogue {q(t[:, 3] - t[:, 2])}  end {q(t[:, 3] - t0)}  (shader cycles, p10/p50/p90); span {float(t[:, 3].max() - t0):.0f}")
