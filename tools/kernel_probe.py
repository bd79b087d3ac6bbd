#!/usr/bin/env python
"""Launch only the hand-written kernels at the benchmark's layer shape (what bench.py's
``kernels`` section times), for rocprofv3 --pmc / --kernel-trace passes:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o probe -- python tools/kernel_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    res, shape = bench.kernel_rooflines(dev, sys.argv[1] if len(sys.argv) > 1 else "P30", 256,
                                        only=("gatedgcn_fwd", "gatedgcn_bwd", "gatedgcn_bwd_unfolded", "seg_attn_fwd", "seg_attn_bwd", "wgrad_grouped"))
    print(shape, {k: round(v["ms"] * 1e3, 2) for k, v in res.items()})
    if os.environ.get("GPS_PROBE_GEMM", "1") != "0":
        # the ring GEMM at two of the block's shapes: k_gemm_ring<2, 0, false> = x[N,d] W[7d,d] (14 column panels of
        # 128 rows), k_gemm_ring<1, 0, true> = t[N,2d] W[d,2d] + residual (2 column panels of 64 rows)
        from graphgps_amd.gemm import absmax, gemm_panel, split_weights
        Nn, d = 7569, 384
        for K, N, cin in ((d, 7 * d, False), (2 * d, d, True)):
            a = torch.randn(Nn, K, device=dev)
            w = torch.randn(N, K, device=dev) / K ** 0.5
            (img, _), = split_weights([w], tn=False)        # the default form: fp16 pieces, k_gemm_ring16 (round 4)
            rec = absmax([a])[0] if getattr(img, "amax", None) is not None else None
            c = torch.zeros(Nn, N, device=dev)
            for _ in range(20):
                gemm_panel(a, img, N, bias=None, addend=c if cin else None, out=c, a_amax=rec)
        torch.cuda.synchronize()
