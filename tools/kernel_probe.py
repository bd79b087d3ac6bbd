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
    res, shape = bench.kernel_rooflines(dev, sys.argv[1] if len(sys.argv) > 1 else "P30", 256)
    print(shape, {k: round(v["ms"] * 1e3, 2) for k, v in res.items()})
