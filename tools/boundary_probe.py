"""Cost of a dependent kernel boundary on this stack: N trivial dependent launches, eager and as one replayed hipGraph.
usage: python tools/boundary_probe.py [N]"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")


def chain(x, n):
    for _ in range(n):
        x.add_(1.0)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for numel, name in ((64, "1 wavefront"), (256 * 256, "256 workgroups"), (4 << 20, "16 MB r+w")):
    x = torch.zeros(numel, device=dev)
    e = timed(lambda: chain(x, N))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(x, 3)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(x, N)
    r = timed(g.replay)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g1, stream=s):
            chain(x, 1)
    r1 = timed(g1.replay)
    print(f"{name:16s} eager {e / N:6.2f} us/kernel   graph {r / N:6.2f} us/kernel   (a 1-kernel graph replay: {r1:.1f} us)")
