#!/usr/bin/env python
"""One shape of gps_gemm_nt in a loop (for rocprofv3 --pmc passes).  usage: gemm_split_one.py R K M"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphgps_amd import lib as L_  # noqa: E402
from graphgps_amd.lib import check, current_stream, ptr  # noqa: E402

L = L_.load()
dev = torch.device("cuda:0")
R, K, M = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (7569, 384, 2688)
x = torch.randn(R, K, device=dev)
w = torch.randn(M, K, device=dev)
b = torch.randn(M, device=dev)
out = torch.empty(R, M, device=dev)
for _ in range(20):
    check(L.gps_gemm_nt(ptr(x), K, ptr(w), K, R, M, K, ptr(b), None, 0, ptr(out), M, current_stream(dev)))
torch.cuda.synchronize()
