#!/usr/bin/env python
"""gps_gemm_panel against torch.addmm (rocBLAS / hipBLASLt, TunableOp-selected) on the GPS block's projection shapes at
P30 x 256 graphs, d = 384: HIP-event time of one hipGraph replay of 40 back-to-back launches over rotating operands."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graphgps_amd as g  # noqa: E402
from graphgps_amd.gemm import gemm_panel, split_weights  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    try:
        g.enable_gemm_tuning()
    except Exception as exc:
        print("TunableOp unavailable:", exc)
    # GEMM_BENCH="d,N,E" (default: the pcqm4m block; "256,25600,76800" = the code2-long layer)
    d, Nn, E = (int(v) for v in os.environ.get("GEMM_BENCH", "384,7569,15348").split(","))
    shapes = [("pq  x[N,d] W[7d,d]", Nn, d, 7 * d), ("out o[N,d] W[d,d]", Nn, d, d), ("C   e[E,d] W[d,d]", E, d, d),
              ("ff1 h[N,d] W[2d,d]", Nn, d, 2 * d), ("ff2 t[N,2d] W[d,2d]", Nn, 2 * d, d),
              ("dgrad g_pq[N,7d] wcat", Nn, 7 * d, d), ("dgrad g_f1[N,2d] W1", Nn, 2 * d, d)]
    from graphgps_amd.gemm import absmax
    print(f"{'shape':28s} {'GF':>6s} | torch us  TF/s | bf16x6 us TF/s | f16x3 us  TF/s  (+absmax us) | f16/bf16 | f16/torch")
    tot_t = tot_p = tot_h = tot_ha = 0.0
    for name, M, K, N in shapes:
        nset = max(2, int(bench.ROTATE_BYTES // (4 * (M * K + M * N))) + 1)
        A = [torch.randn(M, K, device=dev) for _ in range(nset)]
        C = [torch.empty(M, N, device=dev) for _ in range(nset)]
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        (img, _), = split_weights([w], tn=False, f16=False)
        (img16, _), = split_weights([w], tn=False, f16=True)
        words = absmax(A)
        for i in range(3):
            torch.addmm(b, A[0], w.t(), out=C[0])          # TunableOp picks its solution here
        tt = bench.time_kernel(lambda i: torch.addmm(b, A[i], w.t(), out=C[i]), iters=40, nsets=nset)
        tp = bench.time_kernel(lambda i: gemm_panel(A[i], img, N, bias=b, out=C[i]), iters=40, nsets=nset)
        th = bench.time_kernel(lambda i: gemm_panel(A[i], img16, N, bias=b, out=C[i], a_amax=words[i]), iters=40,
                               nsets=nset)
        wz = torch.zeros(nset, 512, dtype=torch.int32, device=dev)     # (words only ever rise: re-raising them costs the same pass)
        ta = bench.time_kernel(lambda i: absmax([A[i]], out=wz[i:i + 1]), iters=40, nsets=nset)
        gf = 2.0 * M * K * N / 1e9
        tot_t += tt
        tot_p += tp
        tot_h += th
        tot_ha += ta
        print(f"{name:28s} {gf:6.2f} | {tt*1e3:8.1f} {gf/tt/1e3:5.0f} | {tp*1e3:8.1f} {gf/tp/1e3:5.0f} | {th*1e3:8.1f} "
              f"{gf/th/1e3:5.0f}  (+{ta*1e3:5.1f}) | {tp/th:5.2f}x | {tt/th:5.2f}x", flush=True)
    print(f"sum: torch {tot_t*1e3:.1f} us, bf16x6 {tot_p*1e3:.1f} us, f16x3 {tot_h*1e3:.1f} us (+ absmax {tot_ha*1e3:.1f} us)")
    # the epilogue forms the block uses (panel only): in-place addend, ReLU + dropout, mask of a saved activation
    for name, M, K, N, kw in (("dgrad C  + addend (in place)", E, d, d, dict(addend=True)),
                              ("dgrad out + addend", Nn, d, d, dict(addend=True)),
                              ("ff1 relu + dropout", Nn, d, 2 * d, dict(epilogue=1, p_drop=0.1, seed=7)),
                              ("dgrad ff2 x mask", Nn, d, 2 * d, dict(epilogue=2, p_drop=0.1, seed=7, mask=True))):
        nset = max(2, int(bench.ROTATE_BYTES // (4 * (M * K + 2 * M * N))) + 1)
        A = [torch.randn(M, K, device=dev) for _ in range(nset)]
        C = [torch.randn(M, N, device=dev) for _ in range(nset)]
        S = [torch.randn(M, N, device=dev) for _ in range(nset)] if kw.get("mask") else None
        w = torch.randn(N, K, device=dev) / K ** 0.5
        (img, _), = split_weights([w], tn=False)
        def run(i):
            gemm_panel(A[i], img, N, addend=C[i] if kw.get("addend") else None, out=C[i], epilogue=kw.get("epilogue", 0),
                       mask_src=S[i] if S else None, p_drop=kw.get("p_drop", 0.0), seed=kw.get("seed", 0))
        tp = bench.time_kernel(run, iters=40, nsets=nset)
        print(f"{name:28s} {2.0 * M * K * N / 1e9:6.2f} | panel {tp*1e3:8.1f} us", flush=True)
    w5 = [torch.randn(7 * d, d, device=dev), torch.randn(d, d, device=dev), torch.randn(d, d, device=dev),
          torch.randn(2 * d, d, device=dev), torch.randn(d, 2 * d, device=dev)]
    ts = bench.time_kernel(lambda i: split_weights(w5), iters=20)
    print(f"split_weights (5 weights of one layer, both images): {ts*1e3:.1f} us")
