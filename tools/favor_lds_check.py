#!/usr/bin/env python
"""FAVOR+ with the projection staged in LDS (GPS_FAVOR_LDS=1) against the one-wavefront-per-tile launches (=0): the same
arithmetic in the same order, so outputs and gradients must be BIT-IDENTICAL -- code2-long shape and a ragged small one."""
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
    bad = 0
    for profile, nb, m in (("CODE2_LONG", 32, 266), ("CODE2_LONG", 3, 266), ("CODE2_REAL", 37, 266), ("ZINC", 40, 100)):
        sizes, ei, bvec, ptr, gen, _ = make_structure(profile, nb, 1234)
        N, H, dh = int(ptr[-1]), 4, 64
        gi = build_graph_index(ei.to(dev), N, len(ptr) - 1, batch_vec=bvec.to(dev), ptr_vec=ptr.to(dev))
        proj = gaussian_orthogonal_random_matrix(m, dh).to(dev)
        qkv0 = (torch.randn(N, 3 * H * dh, generator=gen) * 0.7).to(dev)
        w = torch.randn(N, H * dh, generator=gen).to(dev)
        res = {}
        for mode, (lds, lc) in {"plain": ("0", "0"), "lds": ("1", "0"), "lc": ("1", "1")}.items():
            os.environ["GPS_FAVOR_LDS"], os.environ["GPS_FAVOR_LC"] = lds, lc
            os.environ["GPS_FAVOR_CTX_LDS"] = "0" if mode == "plain" else "1"      # context kernels: per-wavefront / staged
            qkv = qkv0.clone().requires_grad_(True)
            out = favor_attention(qkv, proj, gi, H)
            (out * w).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (out.detach().clone(), qkv.grad.clone())
        inner = H * dh
        for mode in ("lds", "lc"):
            d = (res[mode][1] - res["plain"][1]).abs()
            parts = [float(d[:, i * inner:(i + 1) * inner].max()) for i in range(3)]
            rows = int((d.max(dim=1).values > 0).sum())
            print(f"   {mode:5s} vs plain: max |d_q| {parts[0]:.3e} |d_k| {parts[1]:.3e} |d_v| {parts[2]:.3e}, rows that differ {rows} of {N}, "
                  f"out identical {torch.equal(res[mode][0], res['plain'][0])}")
        res["0"], res["1"] = res["plain"], res["lc"]
        same_o, same_g = torch.equal(res["0"][0], res["1"][0]), torch.equal(res["0"][1], res["1"][1])
        print(f"{profile} x {nb} (N={N}, m={m}): out identical {same_o}, d_qkv identical {same_g}; "
              f"max|out| {float(res['1'][0].abs().max()):.3f}, finite {bool(torch.isfinite(res['1'][1]).all())}")
        bad += (not same_o) + (not same_g)
    sys.exit(1 if bad else 0)
