#!/usr/bin/env python
"""Probe: fp32-equivalent GEMM as ONE library bf16 GEMM over K-concatenated exact bf16 pieces.

x = hi + mid + lo exactly (3 bf16 pieces by truncation of the running remainder); the 6 piece products
hh, hm, mh, hl, lh, mm carry x.w to ~2^-24 relative.  Concatenating the pieces along K turns the six
products into one [M, 6K] x [6K, N] bf16 GEMM with fp32 accumulation/output (hipBLASLt), which is the
question this probe answers for the block's shapes: is 6x the flops on the 16x faster pipe quicker than the
fp32 library GEMM, and what does producing the pieces cost?  (csrc/wgrad.hip already does the split inside
its own kernel; csrc/gemm_split.hip does it on the fly for NT GEMMs and only reaches library parity because
the split is VALU work inside the K loop.)  Not on the product path.
"""
import sys
import time

import torch


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def kcat_a(x):
    h, m, l = split3(x)
    return torch.cat([h, h, m, h, l, m], dim=1).contiguous()


def kcat_b(w):
    h, m, l = split3(w)
    return torch.cat([h, m, h, l, h, m], dim=1).contiguous()


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(7569, 384, 2688), (7569, 2688, 384), (15348, 384, 384), (7569, 384, 768), (7569, 768, 384)]
    print(f"{'M':>6} {'K':>5} {'N':>5} | fp32 us | bf16x6 us | split(A) us | rel err vs fp64")
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        ref = (x.double() @ w.double().t())
        t32 = timeit(lambda: torch.mm(x, w.t()))
        xa, wb = kcat_a(x), kcat_b(w)
        try:
            f = lambda: torch.mm(xa, wb.t(), out_dtype=torch.float32)
            out = f()
        except Exception as exc:            # older torch: no out_dtype on mm
            print("torch.mm(out_dtype=) unavailable:", type(exc).__name__, exc)
            f = lambda: torch.mm(xa, wb.t())
            out = f().float()
        t16 = timeit(f)
        tsp = timeit(lambda: kcat_a(x))
        e32 = float(((torch.mm(x, w.t()).double() - ref).abs().max() / ref.abs().max()))
        e16 = float(((out.double() - ref).abs().max() / ref.abs().max()))
        print(f"{M:6d} {K:5d} {N:5d} | {t32:7.1f} | {t16:9.1f} | {tsp:11.1f} | fp32 {e32:.2e}  bf16x6 {e16:.2e}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
