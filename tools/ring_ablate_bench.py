#!/usr/bin/env python
"""Times the fp16-form ring GEMM (k_gemm_ring16) of whatever library GPS_HIP_LIB names on the pcqm4m block's seven
projection shapes, rotating operands, one hipGraph of 40 launches each: one line per library.  With the ablation builds of
tools/micro/ring_ablate.sh (parts of the main loop compiled out; results are garbage, timing only) the differences say what
the loop spends its time on."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graphgps_amd.gemm import absmax, gemm_panel, split_weights  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    d, Nn, E = 384, 7569, 15348
    shapes = [("pq", Nn, d, 7 * d), ("out", Nn, d, d), ("C", E, d, d), ("ff1", Nn, d, 2 * d), ("ff2", Nn, 2 * d, d),
              ("dpq", Nn, 7 * d, d), ("df1", Nn, 2 * d, d)]
    res, tot = [], 0.0
    for name, M, K, N in shapes:
        nset = max(2, int(bench.ROTATE_BYTES // (4 * (M * K + M * N))) + 1)
        A = [torch.randn(M, K, device=dev) for _ in range(nset)]
        C = [torch.empty(M, N, device=dev) for _ in range(nset)]
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        (img16, _), = split_weights([w], tn=False, f16=True)
        words = absmax(A)
        th = bench.time_kernel(lambda i: gemm_panel(A[i], img16, N, bias=b, out=C[i], a_amax=words[i]), iters=40, nsets=nset)
        res.append(f"{name} {th * 1e3:6.1f}")
        tot += th
        del A, C
    tag = os.path.basename(os.environ.get("GPS_HIP_LIB", "default"))
    print(f"{tag:20s} sum {tot * 1e3:7.1f} us | " + " | ".join(res), flush=True)
