#!/usr/bin/env python
"""Launch-shape sweep of the GatedGCN kernels at the benchmark's layer shape (P30 x 256 graphs, d = 384):
hot (one operand set, Infinity-Cache resident) and rotating (> 512 MiB of operand sets, HBM-sourced) HIP-event
durations for every (GPS_GG_THREADS, GPS_GG_TARGET_WG) pair, one subprocess each (the library reads the
variables once).    python tools/gg_sweep.py            -> table on stdout"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import bench
    dev = torch.device("cuda:0")
    kr, shape = bench.kernel_rooflines(dev, os.environ.get("GG_PROFILE", "P30"), int(os.environ.get("GG_NB", "256")),
                                       d=int(os.environ.get("GG_D", "384")), H=16, only=("gatedgcn_fwd", "gatedgcn_bwd"))
    print("RESULT " + json.dumps({k: dict(hot=v["isolated_hot_ms"], rot=v["isolated_rotating_ms"],
                                           frac_rot=v["frac_isolated_rotating"]) for k, v in kr.items()}))


def copy_baseline():
    """What a plain device copy moving the same bytes achieves: torch's copy kernel over 58.5 MB -> 58.5 MB
    (117 MB of HBM traffic = one GatedGCN forward at P30 x 256, d = 384), hot and rotating; and a 1 GiB copy."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    for mb in (58.5, 1024.0):
        n = int(mb * 2 ** 20 / 4)
        nset = max(2, int(bench.ROTATE_BYTES // (8 * n)) + 1) if mb < 512 else 2
        a = [torch.randn(n, device=dev) for _ in range(nset)]
        b = [torch.empty(n, device=dev) for _ in range(nset)]
        hot = bench.time_kernel(lambda i: b[0].copy_(a[0]), nsets=1)
        rot = bench.time_kernel(lambda i: b[i].copy_(a[i]), nsets=nset)
        print(f"copy {mb:7.1f} MiB -> same: hot {hot*1e3:7.1f} us = {2*4*n/hot/1e9:6.2f} TB/s | rotating "
              f"{rot*1e3:7.1f} us = {2*4*n/rot/1e9:6.2f} TB/s", flush=True)


def main():
    if os.environ.get("GG_ONE"):
        return one()
    if os.environ.get("GG_COPY"):
        return copy_baseline()
    grid = [(768, 512), (768, 256), (768, 1024), (384, 512), (384, 1024), (384, 2048), (192, 2048), (192, 4096)]
    if os.environ.get("GG_GRID"):
        grid = [tuple(int(v) for v in item.split(":")) for item in os.environ["GG_GRID"].split(",")]
    print(f"{'threads':>8} {'target':>7} | fwd hot / rot us (frac rot) | bwd hot / rot us (frac rot)")
    for th, tg in grid:
        env = dict(os.environ, GG_ONE="1", GPS_GG_THREADS=str(th), GPS_GG_FWD_THREADS=str(th), GPS_GG_TARGET_WG=str(tg))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(th, tg, "FAILED", r.stderr[-400:])
            continue
        d = json.loads(line[0][7:])
        f, b = d["gatedgcn_fwd"], d["gatedgcn_bwd"]
        print(f"{th:8d} {tg:7d} | {f['hot']*1e3:6.1f} / {f['rot']*1e3:6.1f} ({f['frac_rot']:.3f}) | "
              f"{b['hot']*1e3:6.1f} / {b['rot']*1e3:6.1f} ({b['frac_rot']:.3f})", flush=True)


if __name__ == "__main__":
    main()
