#!/usr/bin/env python
"""FAVOR+ forward + backward (csrc/favor.hip) at the code2-long layer shape (32 graphs of 600-1000 nodes, 4 heads x 64,
m = 266 random features) for rocprofv3 --kernel-trace / --pmc passes:

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -d out -o probe -- python tools/favor_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from graphgps_amd.ops import build_graph_index, favor_attention
    from graphgps_amd.synthetic import make_structure
    from oracle.gps_oracle import gaussian_orthogonal_random_matrix      # (test infrastructure: the fixed projection)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # FAVOR_PROFILE / FAVOR_GRAPHS: another batch shape (CODE2_REAL = the dataset's own sizes, ~125 nodes per graph)
    sizes, ei, bvec, ptr, gen, _ = make_structure(os.environ.get("FAVOR_PROFILE", "CODE2_LONG"),
                                                  int(os.environ.get("FAVOR_GRAPHS", "32")), 1234)
    N, H, dh, m = int(ptr[-1]), 4, 64, 266
    gi = build_graph_index(ei.to(dev), N, len(ptr) - 1, batch_vec=bvec.to(dev), ptr_vec=ptr.to(dev))
    proj = gaussian_orthogonal_random_matrix(m, dh).to(dev)
    qkv = (torch.randn(N, 3 * H * dh, generator=gen) * 0.7).to(dev).requires_grad_(True)
    w = torch.randn(N, H * dh, generator=gen).to(dev)
    iters = int(os.environ.get("FAVOR_ITERS", "6"))
    for _ in range(iters):
        qkv.grad = None
        out = favor_attention(qkv, proj, gi, H)
        (out * w).sum().backward()
    torch.cuda.synchronize()
    flops_fwd = 8.0 * N * m * (dh * H) + 2.0 * N * m * H
    print(f"N={N} H={H} m={m}: fwd {flops_fwd / 1e9:.2f} GFLOP, fwd+bwd ~{3 * flops_fwd / 1e9:.2f} GFLOP per layer")
