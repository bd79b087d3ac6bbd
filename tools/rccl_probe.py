#!/usr/bin/env python
"""World-size-1 RCCL all-reduce of the gradient arena: what the collective itself costs on this box (the multi-GPU step
pays at least this between its two hipGraph halves)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
for mb in (77.7, 119.1):
    flat = torch.randn(int(mb * 1e6 / 4), device=dev)
    for op in (dist.ReduceOp.AVG, dist.ReduceOp.SUM):
        for _ in range(3):
            dist.all_reduce(flat, op=op)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat, op=op)
        torch.cuda.synchronize()
        print(f"{mb} MB all_reduce {op}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
dist.destroy_process_group()
