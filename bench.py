#!/usr/bin/env python
"""bench.py -- PCQM4M GPS-medium training step on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic PCQM4M-shaped input
(profile P30, 256 graphs per GPU, inputs resident in HBM): per-batch graph index build ->
GPSModel forward (encoders, 10 x GPSLayer on the HIP kernels, head) -> L1 loss -> backward ->
gradient all-reduce (N > 1) -> grad-clip + AdamW step, i.e. the whole of the reference's
train_epoch body (graphgps/train/custom_train.py:22-39) minus logging.  Dropout is ON
(0.1 / 0.1 as in configs/GPS/pcqm4m-GPSmedium+RWSE.yaml).  Weak scaling: per-GPU work is fixed.

How the step is driven (graphgps_amd/train.py: TrainStep): flat-arena clip+AdamW; for N > 1 ONE RCCL
all-reduce of the flat gradient arena between [index + fwd + bwd + pack] and [clip + AdamW]; the step is
launched eagerly or replayed from hipGraph(s), whichever of the two measures faster on 8 untimed trial
steps each (eager measured before anything is captured) (--launch auto; every rank takes the same decision).  rocBLAS / hipBLASLt GEMM
solutions are picked per shape by TunableOp during the warm-up and frozen before the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the GatedGCN gather-gate-segment-reduce forward kernel (HBM-bound): algorithmic
                bytes (8*E*d + 20*N*d + CSR index bytes) / mean launch
                duration measured with HIP events on the launching stream
  kernels       the same measurement for every hand-written kernel (HBM GB/s or MFMA TFLOP/s)
  cpu_baseline  the CPU oracle (pure-torch restatement of the reference path) timed on the
                host cores of the same box, same batch, same step definition (rank 0, N = 1)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3     # v_mfma_f32_16x16x4_f32 dense peak


_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graphs-per-gpu", type=int, default=0, help="0 = the config's train.batch_size")
    ap.add_argument("--profile", default=None, help="synthetic size profile (pcqm4m: P30 | P14)")
    ap.add_argument("--workload", choices=("pcqm4m", "zinc", "code2"), default="pcqm4m",
                    help="pcqm4m = the BASELINE.json metric (default).  zinc (GINE+Transformer, 10L x 64d, "
                         "32 graphs) and code2 (CustomGatedGCN+Performer, 4L x 256d, 32 graphs of 600-1000 "
                         "nodes) are the other BASELINE configs: extra measurements, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="how the step is launched: replayed from hipGraph(s), eagerly, or (auto) "
                         "whichever of the two measures faster on a few untimed trial steps -- "
                         "replay removes the host cost, eager keeps the weight-gradient stream "
                         "overlapped, which one wins depends on the host")
    ap.add_argument("--no-graph", action="store_true", help="alias for --launch eager")
    ap.add_argument("--exchange", choices=("flat", "bucketed"), default="flat",
                    help="N>1 gradient exchange: one all-reduce of the flat gradient arena between "
                         "two hipGraphs (default), or eager per-layer buckets from autograd hooks")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="keep the BLAS libraries' default kernel heuristics (no TunableOp pass)")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true",
                    help="skip the secondary PCIe-inclusive measurement (host batches through DeviceLoader)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="torch threads for the CPU baseline (0 = min(usable cores, 32))")
    return ap.parse_args()


def make_step(model, opt, reducer, batch_dev, compute_loss, clip_value, salt=None):
    params = [p for p in model.parameters() if p.requires_grad]

    def step():
        if salt is not None:
            salt.add_(1)                 # device-side: every hipGraph replay draws new dropout masks
        b = batch_dev.shallow_copy()     # fresh batch object -> the graph index is rebuilt
        if reducer is None:              # single GPU: nothing to exchange, autograd owns .grad
            opt.zero_grad(set_to_none=True)
        else:
            reducer.zero_grad()
        pred, true = model(b)
        loss, _ = compute_loss(pred, true)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        torch.nn.utils.clip_grad_norm_(params, clip_value, foreach=True)
        opt.step()
        return loss
    return step


def time_kernel(fn, iters=50, warm=5):
    """Mean duration (ms) of ``fn`` (enqueues on torch's current stream, the stream the C-ABI
    launches on) measured with HIP events."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def kernel_rooflines(dev, profile, nb, d=384, H=16):
    """Layer-shaped micro-measurements of each hand-written kernel at the benchmark's sizes."""
    from graphgps_amd import lib as L_
    from graphgps_amd.lib import check, current_stream, ptr
    from graphgps_amd.ops import build_graph_index
    from graphgps_amd.synthetic import layer_batch
    L = L_.load()
    b = layer_batch(profile, nb, d, seed=1234).to(dev)
    N, E = b.x.shape[0], b.edge_index.shape[1]
    gi = build_graph_index(b.edge_index, N, nb, ptr_vec=b.ptr)
    st = current_stream(dev)
    f = lambda *s: torch.randn(*s, device=dev)
    proj, ce = f(N, 4 * d), f(E, d)
    xt, eh, ag, dn = f(N, d), f(E, d), f(N, d), f(N, d)
    gx, ge, gproj, gce = f(N, d), f(E, d), f(N, 4 * d), f(E, d)
    P, G, fs = proj.data_ptr(), gproj.data_ptr(), d * 4

    def gg_fwd():
        check(L.gps_gatedgcn_fwd(P, P + fs, P + 2 * fs, P + 3 * fs, 4 * d, ptr(ce), ptr(gi.rowptr_dst),
                                 ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(xt), ptr(eh),
                                 ptr(ag), ptr(dn), None, st))

    def gg_bwd():
        check(L.gps_gatedgcn_bwd(ptr(gx), ptr(ge), ptr(eh), P + fs, 4 * d, ptr(ag), ptr(dn),
                                 ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                                 ptr(gi.rowptr_src), ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, d,
                                 ptr(gce), G, G + fs, G + 2 * fs, G + 3 * fs, 4 * d, None, st))

    dh = d // H
    qkv, out, lse = f(N, 3 * d), f(N, d), f(H, N)
    dout, delta, dqkv = f(N, d), f(H, N), f(N, 3 * d)
    scale = dh ** -0.5

    def at_fwd(p=0.1):
        check(L.gps_seg_attn_fwd(ptr(qkv), 3 * d, ptr(gi.ptr), ptr(gi.tile_graph), ptr(gi.tile_row0),
                                 gi.max_tiles, N, H, dh, scale, p, 1234, ptr(out), ptr(lse), st))

    def at_bwd(p=0.1):
        check(L.gps_seg_attn_bwd(ptr(dout), ptr(qkv), 3 * d, ptr(out), ptr(lse), ptr(gi.ptr),
                                 ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh, scale,
                                 p, 1234, ptr(delta), ptr(dqkv), 3 * d, st))

    gg_fwd(); at_fwd()                     # produce valid saved tensors for the backward kernels
    sizes = (b.ptr[1:] - b.ptr[:-1]).double()
    s2 = float((sizes * sizes).sum())
    idx = 4 * (N + 1) + 8 * E
    res = {}
    t = time_kernel(gg_fwd)
    bytes_f = 8 * E * d + 20 * N * d + idx
    res["gatedgcn_fwd"] = dict(bound="hbm", ms=t, bytes=bytes_f, achieved=bytes_f / t / 1e6,
                               peak=HBM_PEAK_GBS, unit="GB/s")
    t = time_kernel(gg_bwd)
    bytes_b = 12 * E * d + 28 * N * d + 2 * idx
    res["gatedgcn_bwd"] = dict(bound="hbm", ms=t, bytes=bytes_b, achieved=bytes_b / t / 1e6,
                               peak=HBM_PEAK_GBS, unit="GB/s", launches=2)
    t = time_kernel(at_fwd)
    fl = 4 * s2 * d
    res["seg_attn_fwd"] = dict(bound="mfma", ms=t, flops=fl, achieved=fl / t / 1e9,
                               peak=MFMA_F32_PEAK_TF, unit="TFLOP/s")
    t = time_kernel(at_bwd)
    flb = 8 * s2 * d
    res["seg_attn_bwd"] = dict(bound="mfma", ms=t, flops=flb, achieved=flb / t / 1e9,
                               peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", launches=3)
    # the five weight(+bias) gradients of one block as ONE grouped split-K launch (csrc/wgrad.hip)
    shapes = [(N, d, 7 * d), (E, d, d), (N, d, d), (N, d, 2 * d), (N, 2 * d, d)]   # (rows, in, out)
    pairs = [(f(R, n), f(R, k)) for R, k, n in shapes]
    probs = (L_.WgradProblem * len(pairs))()
    keep = []
    for q, (g_, x_) in zip(probs, pairs):
        gw, gb = torch.empty(g_.shape[1], x_.shape[1], device=dev), torch.empty(g_.shape[1], device=dev)
        keep.append((gw, gb))
        q.g, q.x, q.gw, q.gb = g_.data_ptr(), x_.data_ptr(), gw.data_ptr(), gb.data_ptr()
        q.ldg, q.ldx, q.R, q.M, q.Nn = g_.stride(0), x_.stride(0), g_.shape[0], g_.shape[1], x_.shape[1]
    wws = torch.empty(max(L.gps_wgrad_grouped_workspace_floats(len(pairs), probs), 4), device=dev)
    t = time_kernel(lambda: check(L.gps_wgrad_grouped(len(pairs), probs, ptr(wws), st)))
    flw = sum(2.0 * R * k * n for R, k, n in shapes)
    res["wgrad_grouped"] = dict(bound="mfma", ms=t, flops=flw, achieved=flw / t / 1e9,
                                peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", launches=2,
                                note="fp32-equivalent flops against the fp32-input MFMA peak; the contraction "
                                     "itself runs on the bf16 pipe (exact 3-way split, 6 products)")
    for v in res.values():
        v["frac"] = v["achieved"] / v["peak"]
    return res, dict(N=N, E=E, d=d, H=H, sum_n2=s2)


def cpu_baseline(model, batch_cpu, compute_loss, clip_value, steps, threads=0):
    """CPU oracle (kind='port'): same batch, same step definition, on the host cores.  The op
    granularity of the reference path ([7.7k,384] GEMMs, gathers, elementwise) stops scaling
    well before 32 threads, so more threads than that only add barrier cost."""
    from oracle.gps_oracle import to_oracle_model
    cores = threads or min(usable_cores(), 32)
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} torch threads of {usable_cores()} usable cores")
    oracle = to_oracle_model(model).train()
    opt = torch.optim.AdamW(oracle.parameters(), lr=2e-4, weight_decay=0.0)
    params = list(oracle.parameters())

    def step():
        opt.zero_grad(set_to_none=True)
        pred, true = oracle(batch_cpu.clone())
        loss, _ = compute_loss(pred, true)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, clip_value)
        opt.step()

    tw = time.perf_counter()
    step()                                   # warm-up (allocator, thread pool)
    log(f"cpu baseline: warm-up step {time.perf_counter() - tw:.2f}s")
    t0 = time.perf_counter()
    done = 0
    while done < steps and (done == 0 or time.perf_counter() - t0 < 30.0):
        step()
        done += 1
    dt = (time.perf_counter() - t0) / done
    nb = int(batch_cpu.num_graphs)
    return dict(value=nb / dt, unit="graphs/s", cores=cores, kind="port", ms_per_step=dt * 1e3,
                sample=f"{done} full training step(s) of the same {nb}-graph batch after 1 warm-up, "
                       f"pure-torch CPU oracle of the reference path, torch threads={cores}")


def main():
    args = parse_args()
    # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract
    # is ONE JSON line on stdout, so everything before the final print goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GPS hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import graphgps_amd as g
    from graphgps_amd.dp import GradBucketReducer
    from graphgps_amd.loss.losses import compute_loss, subtoken_cross_entropy
    from graphgps_amd.synthetic import model_batch

    tunable = None
    if not args.no_gemm_tuning:                # rocBLAS/hipBLASLt solution selection per GEMM shape,
        try:                                   # timed at first use during the untimed warm-up
            tunable = g.enable_gemm_tuning()
        except Exception as exc:               # an optimisation, never a requirement
            log(f"TunableOp unavailable ({type(exc).__name__}: {exc}); library default heuristics")
            tunable = None
    torch.manual_seed(0)                       # identical initial weights on every rank
    WORKLOADS = {   # yaml, dim_in, dim_out, default graphs/GPU (the config's train.batch_size), label
        "pcqm4m": ("pcqm4m_gpsmedium_rwse.yaml", 9, 1, 256,
                   "pcqm4m-GPSmedium+RWSE (CustomGatedGCN+Transformer, 10L x 384d, 16 heads, dropout 0.1/0.1)"),
        "zinc": ("zinc_gps_rwse.yaml", 1, 1, 32, "zinc-GPS+RWSE (GINE+Transformer, 10L x 64d, 4 heads)"),
        "code2": ("code2_gps.yaml", 2, 5002, 32,
                  "ogbg-code2-GPS (CustomGatedGCN+Performer, 4L x 256d, 4 heads x 64, m=266)"),
    }
    yaml_name, dim_in, dim_out, default_nb, wl_label = WORKLOADS[args.workload]
    model = g.create_model(os.path.join(g.CONFIG_DIR, yaml_name), None, dim_in, dim_out)
    cfg = g.cfg
    model.train()
    if args.workload == "code2":
        compute_loss = subtoken_cross_entropy       # custom_train.py:24-25
    nb = args.graphs_per_gpu or default_nb
    if args.workload != "pcqm4m":
        args.no_kernel_roofline = True              # the kernel table is the PCQM4M layer shape
    batch_cpu = model_batch(args.workload, nb, seed=1234 + rank, profile=args.profile)
    cpu_ref_model = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import copy
        cpu_ref_model = copy.deepcopy(model)
    model.to(dev)
    batch_dev = batch_cpu.clone().to(dev)
    # GPS_BENCH_FORCE_REDUCER=1 exercises the RCCL exchange on a single rank too (validates the
    # plumbing on the 1-GPU box; not the default measurement)
    use_exchange = world > 1 or os.environ.get("GPS_BENCH_FORCE_REDUCER") == "1"
    if use_exchange and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(1000 + rank)             # dropout streams differ per rank
    # fresh batch object over the resident tensors -> the graph index is rebuilt every step, nothing is
    # copied (inputs already in HBM is the measurement contract)
    make_batch = batch_dev.shallow_copy
    reducer = exchange = None
    trial = {}
    if args.exchange == "bucketed" and use_exchange:
        # hook-driven per-layer buckets overlapped with backward (dp.GradBucketReducer), eager
        reducer = GradBucketReducer(model, force_collective=True)
        opt = torch.optim.AdamW(model.parameters(), lr=cfg.optim.base_lr,
                                weight_decay=cfg.optim.weight_decay, fused=True)
        step = make_step(model, opt, reducer, batch_dev, compute_loss,
                         cfg.optim.clip_grad_norm_value)
        graph_mode = "eager (bucketed all-reduce from autograd hooks)"
        allreduce_bytes = reducer.num_bytes
    else:
        # the product step (graphgps_amd/train.py): flat-arena clip+AdamW, one all-reduce of the
        # flat gradient, hipGraph replay of the compute on either side of it (the synthetic
        # batch has a fixed shape, which is what a capture needs; a loader would keep one graph
        # per shape bucket)
        from graphgps_amd.dp import FlatGradExchange
        from graphgps_amd.ops import enable_dropout_salt
        from graphgps_amd.optim import FlatAdamW
        from graphgps_amd.train import TrainStep
        opt = FlatAdamW(model.parameters(), lr=cfg.optim.base_lr,
                        weight_decay=cfg.optim.weight_decay,
                        max_grad_norm=cfg.optim.clip_grad_norm_value
                        if cfg.optim.clip_grad_norm else None)
        if use_exchange:
            exchange = FlatGradExchange(opt.arena, force_collective=True)
        launch = "eager" if args.no_graph else args.launch
        salt = None if launch == "eager" else enable_dropout_salt(dev)
        ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=exchange, salt=salt)

        def step():
            if ts.use_replay and ts.mode != "eager":
                return ts.replay()           # the captured step owns its (static) batch
            return ts.run_eager(make_batch())

        def trial_ms(n_warm=3, n=8):
            for _ in range(n_warm):
                step()
            barrier()
            tt = time.perf_counter()
            for _ in range(n):
                step()
            barrier()
            t = (time.perf_counter() - tt) / n * 1e3
            if world > 1:                    # every rank must take the same decision
                tv = torch.tensor([t], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tv, op=torch.distributed.ReduceOp.MAX)
                t = float(tv[0])
            return t

        if launch == "auto":                 # untimed trial, eager first: measured BEFORE anything is captured
            trial["eager"] = trial_ms()      # (a live hipGraph slows eager launches down by ~5 % here)
        if launch != "eager":
            try:
                ts.capture(make_batch)
                log("step captured: " + ts.mode)
            except Exception as exc:         # capture is an optimisation, never a requirement
                log(f"hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly")
                ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=exchange, salt=salt)
                launch = "eager"
            if world > 1:                    # one rank falling back must take every rank with it
                ok = torch.tensor([0.0 if launch == "eager" else 1.0], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                if float(ok) == 0.0 and launch != "eager":
                    log("another rank could not capture the step; running eagerly everywhere")
                    launch = "eager"
        if launch == "auto":
            ts.use_replay = True
            trial["graph"] = trial_ms()
            launch = min(trial, key=trial.get)
            log(f"launch-mode trial: eager {trial['eager']:.2f} ms, graph {trial['graph']:.2f} ms "
                f"-> {launch}")
        if launch == "eager" and ts.mode != "eager":
            # drop the captured graphs (and their private memory pool) before running eagerly
            ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=exchange, salt=salt)
        ts.use_replay = launch == "graph"
        graph_mode = ts.mode if launch == "graph" else "eager (2 HIP streams: main + weight-gradient)"
        allreduce_bytes = exchange.num_bytes if exchange is not None else 0
    log("model on device, starting warm-up")
    for i in range(args.warmup):
        loss = step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    if tunable is not None:
        try:
            tunable.tuning_enable(False)       # frozen: nothing is tuned inside the timed region
        except Exception as exc:
            log(f"could not freeze TunableOp ({type(exc).__name__}: {exc})")
    barrier()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    final_loss = float(loss.item())
    # host-side cost of enqueueing one step (GPU queue drained first): if this is close to
    # ms_per_step the step is launch-bound, not GPU-bound
    host_ms = []
    for _ in range(3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        host_ms.append((time.perf_counter() - th) * 1e3)
    torch.cuda.synchronize()
    host_enqueue_ms = min(host_ms)
    log(f"timed region done: {ms:.2f} ms/step")
    # secondary, never `value`: the same step fed from pinned HOST batches through the prefetching
    # DeviceLoader (H2D copies + graph index on a copy stream, one batch ahead) -- the PCIe-inclusive rate
    h2d_ms = None
    if not args.no_h2d_leg and reducer is None and world == 1:     # a 1-GPU side measurement only
        try:
            from graphgps_amd.loader import DeviceLoader
            pinned = batch_cpu.shallow_copy()
            for k, v in list(pinned.__dict__.items()):
                if torch.is_tensor(v):
                    pinned.__dict__[k] = v.pin_memory()
            for n_h in (3, min(args.steps, 20)):       # 3 untimed, then the measured pass
                torch.cuda.synchronize()
                th = time.perf_counter()
                for b in DeviceLoader((pinned.shallow_copy() for _ in range(n_h)), dev):
                    ts.run_eager(b)
                torch.cuda.synchronize()
                h2d_ms = (time.perf_counter() - th) / n_h * 1e3
            log(f"host-batch leg (H2D + index on the copy stream, eager launch): {h2d_ms:.2f} ms/step")
        except Exception as exc:       # never let the side measurement take the headline line with it
            h2d_ms = None
            log(f"host-batch leg skipped ({type(exc).__name__}: {exc})")

    if rank == 0:
        N, E = batch_dev.x.shape[0], batch_dev.edge_index.shape[1]
        out = {
            "metric": ("graphs/sec, PCQM4M GPS-medium training step (fwd+bwd+optimizer)"
                       if args.workload == "pcqm4m" else
                       f"graphs/sec, {args.workload} GPS training step (fwd+bwd+optimizer)"),
            "value": world * nb / (ms / 1e3), "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{wl_label}, synthetic profile "
                                   f"{args.profile or {'pcqm4m': 'P30', 'zinc': 'ZINC', 'code2': 'CODE2_LONG'}[args.workload]}, "
                                   f"{nb} graphs/GPU ({N} nodes, {E} directed edges on rank 0)",
                       "global_batch": world * nb, "parallelism": f"dp{world}",
                       "timed_region": "graph-index build + forward + L1 loss + backward + "
                                       "grad all-reduce + clip + AdamW"},
            "final_loss": final_loss,
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "pcie_inclusive_ms_per_step": h2d_ms,
            "launch_mode": graph_mode,
            "launch_trial_ms": trial or None,
            "grad_allreduce_bytes": allreduce_bytes,
            "optimizer": type(opt).__name__,
            "gemm_selection": "TunableOp (rocBLAS/hipBLASLt solutions timed during warm-up)"
                              if tunable is not None else "library default heuristics",
        }
        if not args.no_kernel_roofline:
            kr, shape = kernel_rooflines(dev, args.profile or "P30", nb)
            log("kernel rooflines done")
            k = kr["gatedgcn_fwd"]
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_gatedgcn_fwd.json")
            if os.path.exists(pmc):           # HBM bytes per launch from a rocprofv3 --pmc pass
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            out["roofline"] = {"kernel": "k_gatedgcn_fwd", "bound": "hbm", "achieved": k["achieved"],
                               "peak": k["peak"], "unit": "GB/s", "frac": k["frac"],
                               "traffic": traffic, "algorithmic_bytes": k["bytes"],
                               "launch_ms": k["ms"]}
            out["kernels"] = kr
            out["kernel_shape"] = shape
        if cpu_ref_model is not None:
            out["cpu_baseline"] = cpu_baseline(cpu_ref_model, batch_cpu, compute_loss,
                                               cfg.optim.clip_grad_norm_value, args.cpu_steps,
                                               args.cpu_threads)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)             # RCCL / the runtime may still print while tearing down
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
