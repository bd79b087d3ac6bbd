#!/usr/bin/env python
"""Times the eight norm / residual / dropout task-list launches of one CustomGatedGCN+Transformer layer
(graphgps_amd/layer/gps_block.py -> csrc/block_norm.hip) at the benchmark's layer shape, each over ROTATING buffer sets
(ten layers' worth of distinct tensors, so nothing is re-read out of the Infinity Cache that the real step would not find
there either).  Prints us per launch and the algorithmic GB/s.

    python tools/norm_probe.py [N E d]          GPS_NORM_BLOCKS selects the row blocks per task
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphgps_amd import gemm as _gemm      # noqa: E402
from graphgps_amd import norm as _norm      # noqa: E402


def main():
    N, E, d = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (7569, 15348, 384)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    SETS, REPS = 10, 20
    f32 = dict(dtype=torch.float32, device=dev)

    def bn():
        m = torch.nn.BatchNorm1d(d).to(dev)
        with torch.no_grad():
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.5, 0.5)
        return m

    class Owner:
        pass
    sets = []
    for _ in range(SETS):
        s = dict(n=[torch.randn(N, d, **f32) for _ in range(9)], e=[torch.randn(E, d, **f32) for _ in range(5)],
                 bn=[bn() for _ in range(5)], stats=torch.rand(10, d, **f32) + 0.5, gpar=torch.zeros(10, d, **f32),
                 rec=_gemm.amax_records(8, dev), owner=Owner())
        s["sync"] = _norm.sync_arena(s["owner"], dev)
        sets.append(s)
    p = 0.1

    def desc(s, i):
        return _norm.bn_desc(s["bn"][i], s["stats"][2 * i], s["stats"][2 * i + 1])

    def mid(s):
        n, e = s["n"], s["e"]
        _norm.fwd([_norm.fwd_task(_norm.BN_ACT, n[0], N, res=n[1], bn1=desc(s, 0), relu=True, p=p, seed=1, out=n[2],
                                  stats=desc(s, 2)),
                   _norm.fwd_task(_norm.BN_ACT, e[0], E, res=e[1], bn1=desc(s, 1), relu=True, p=p, seed=2, out=e[2],
                                  amax=s["rec"][0])], d, dev, s["sync"].site(3))

    def dual(s):
        n = s["n"]
        _norm.fwd([_norm.fwd_task(_norm.BN_DUAL, n[2], N, b=n[3], bn1=desc(s, 2), bn2=desc(s, 3), out=n[4],
                                  amax=s["rec"][1])], d, dev, None)

    def out(s):
        n = s["n"]
        _norm.fwd([_norm.fwd_task(_norm.BN_ACT, n[5], N, bn1=desc(s, 4), out=n[6], amax=s["rec"][2])], d, dev, None)

    def b1_tasks(s):
        n, e, g = s["n"], s["e"], s["gpar"]
        return [_norm.bwd_task(n[5], n[6], desc(s, 4), N, g[0], g[1], g_z=n[7], g_drop=n[8], p2=p, seed2=5,
                               amax_drop=s["rec"][3]),
                _norm.bwd_task(e[0], e[3], desc(s, 1), E, g[2], g[3], relu=True, p=p, seed=2, g_z=e[4])]

    def b1_partial(s):
        _norm.bwd_partial(b1_tasks(s), d, dev, s["sync"].site(5))

    def b1_apply(s):
        _norm.bwd_apply(b1_tasks(s), d, dev, None)

    def b3_tasks(s):
        n, g = s["n"], s["gpar"]
        return [_norm.bwd_task(n[2], n[7], desc(s, 2), N, g[4], g[5], z2=n[3], bn2=desc(s, 3), g_gamma2=g[6], g_beta2=g[7],
                               g_z=n[4], g_sum=n[6], g_drop=n[8], p2=p, seed2=3, cz=n[0], cbn=desc(s, 0), crelu=True, cp=p,
                               cseed=1, cg_gamma=g[8], cg_beta=g[9], amax_drop=s["rec"][4])]

    def b3_partial(s):
        _norm.bwd_partial(b3_tasks(s), d, dev, s["sync"].site(6))

    def b3_apply(s):
        _norm.bwd_apply(b3_tasks(s), d, dev, s["sync"].site(7))

    def bnx_apply(s):
        n, g = s["n"], s["gpar"]
        _norm.bwd_apply([_norm.bwd_task(n[0], n[4], desc(s, 0), N, g[8], g[9], relu=True, p=p, seed=1, g_z=n[7])], d, dev, None)

    nb, eb = N * d * 4, E * d * 4
    launches = [("fwd mid (x1 + stats | e1)", mid, 3 * nb + 3 * eb), ("fwd dual -> h", dual, 3 * nb), ("fwd norm2 -> out", out, 2 * nb),
                ("bwd b1 partial (norm2 | bn_e)", b1_partial, 2 * nb + 2 * eb), ("bwd b1 apply", b1_apply, 4 * nb + 3 * eb),
                ("bwd b3 partial (dual)", b3_partial, 3 * nb), ("bwd b3 apply (dual + chain)", b3_apply, 7 * nb),
                ("bwd bn_x apply", bnx_apply, 3 * nb)]
    total_us = total_b = 0.0
    side = torch.cuda.Stream(dev)
    for name, fn, nbytes in launches:
        with torch.cuda.stream(side):         # replayed: eagerly the ~20 us of host work per launch would be the measurement
            for s in sets:
                fn(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for s in sets:
                    fn(s)
            graph.replay()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(REPS):
                graph.replay()
            t1.record()
            torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / (REPS * SETS)
        total_us += us
        total_b += nbytes
        print(f"{name:34s} {us:7.2f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 1e3:6.0f} GB/s")
    print(f"{'sum':34s} {total_us:7.2f} us  {total_b / 1e6:7.1f} MB  {total_b / total_us / 1e3:6.0f} GB/s   "
          f"(GPS_NORM_BLOCKS={os.environ.get('GPS_NORM_BLOCKS', '-')})")


if __name__ == "__main__":
    main()
