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
replayed from hipGraph(s) -- the product mode, what train_epoch does per shape bucket -- unless the eagerly launched
step beats the replay by more than 3 % in an untimed, interleaved trial taken AFTER the capture with the capture alive,
i.e. under the conditions of the timed region (--launch auto; every rank takes the same decision; both figures, the
rule and the chosen form's host enqueue time are in the JSON line).  rocBLAS / hipBLASLt GEMM
solutions are picked per shape by TunableOp during the warm-up and frozen before the timed region.
`secondary` carries the zinc and code2 workloads (BASELINE.json configs[1] / [4]), a few steps each in child processes
after the timed region; `gemm_arith` states the arithmetic of the dense products behind `dtype: f32`.

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under
torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the hand-written kernel with the LARGEST share of the step's GPU time (mean launch duration x
                launches per step over the roctracer kernel records of a few extra steps after the timed region --
                the quantity `rocprofv3 --kernel-trace --stats` reports, committed under profiles/): its algorithmic
                flops (or bytes) per launch / its mean in-step launch duration, against the
                dense fp16 MFMA peak (or 8 TB/s); `traffic` = HBM bytes per launch from the committed PMC passes
                (profiles/pmc_static.json); `in_step_share` ranks the step's eight largest kernels.  The isolated
                HIP-event durations (hot: one buffer set, resident in the 256 MiB Infinity Cache; rotating:
                > 512 MiB of buffer sets, so every launch comes from HBM) are carried next to it
  roofline_step the algorithmic floor of the WHOLE step (dense products at the fp16 MFMA peak + the bytes that must
                cross HBM once at 6.3 TB/s) and floor / measured step
  dispatches_per_step   GPU activity records (kernels + copies) per step, from the same records
  kernels       the same three durations for every hand-written kernel (HBM GB/s or MFMA TFLOP/s)
  cpu_baseline  the CPU oracle (pure-torch restatement of the reference path) timed on the host cores of the
                same box, same batch, same step definition (rank 0, N = 1): all usable cores (the headline
                leg) and torch.set_num_threads(6) = the reference's default cfg.num_threads (main.py:125)
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
MFMA_BF16_PEAK_TF = 2500.0   # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md: 2495 measured)


HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy -- the byte rate the STEP floor is priced at

# in-step kernel name -> entry of `kernels` that carries its work (flops / bytes per launch)
_DOMINANT_TABLE = (("k_wgrad_direct", "wgrad_grouped"), ("k_wgrad_stream", "wgrad_grouped"), ("k_gatedgcn_fwd", "gatedgcn_fwd"), ("k_gatedgcn_bwd", "gatedgcn_bwd"),
                   ("k_sattn_fwd", "seg_attn_fwd"), ("k_attn_fwd", "seg_attn_fwd"), ("k_sattn_bwd", "seg_attn_bwd"),
                   ("k_attn_bwd", "seg_attn_bwd"))


def _per_step(c, layers):
    """Launches per step of a kernel from its record count.  The activity buffer drops records of long graph replays (a
    once-per-layer kernel shows 6.2 launches per step for 10), so the counts of per-layer kernels are rounded up to whole
    layers; kernels seen less than once per two layers are taken as recorded."""
    return -(-c // layers) * layers if c > layers / 2 else c


def dominant_roofline(in_step, kr, pmc_static, layers=10):
    """`roofline` of the JSON line = the hand-written kernel with the LARGEST share of the step's GPU time (mean launch
    duration x launches per step over the roctracer records), not the best one.  Kernels whose name serves several
    shapes (k_gemm_ring16 instantiations, the norm task lists) have no per-launch work figure: they are ranked in
    `in_step_share` beside it and the ring family carries its own entry (kernels.gemm_ring_in_step)."""
    import re
    per_step = lambda c: _per_step(c, layers)
    share = sorted(((t * per_step(c), k) for k, (t, c) in in_step.items()), reverse=True)
    total = sum(x for x, _ in share) or 1.0

    def short(name):
        m = re.search(r"(k_\w+(?:<[^>]*>)?|Cijk_\w{0,24}|\w+)\s*(?:\(|$)", name.replace("(anonymous namespace)::", ""))
        return (m.group(1) if m else name)[:64]
    ranked = [dict(kernel=short(k), ms_per_step=round(x, 4), share=round(x / total, 4)) for x, k in share[:8]]
    for x, name in share:
        ent = next((e for needle, e in _DOMINANT_TABLE if needle in name and e in kr), None)
        if ent is None:
            continue
        k = kr[ent]
        rec = pmc_static.get(ent, {})
        r = {"kernel": short(name),
             "kernels_entry": ent, "step_share": round(x / total, 4),
             "bound": "hbm" if k["bound"] == "hbm" else "mfma", "achieved": k["achieved"], "peak": k["peak"],
             "unit": k["unit"], "frac": k["frac"], "launch_ms": k["ms"], "launch_ms_source": k["ms_source"],
             "isolated_hot_ms": k["isolated_hot_ms"], "isolated_rotating_ms": k["isolated_rotating_ms"],
             "traffic": rec.get("hbm_bytes_per_launch"),
             "traffic_source": ("static: " + str(rec.get("source"))) if rec else None,
             "in_step_share": ranked}
        r["algorithmic_bytes" if k["bound"] == "hbm" else "algorithmic_flops"] = k.get("bytes", k.get("flops"))
        if k["bound"] != "hbm":
            r["peak_note"] = "dense fp16 / bf16 MFMA peak; flops = the piece products issued (kernels.%s.note)" % ent
        return r
    return None


def step_roofline(N, E, d, H, layers, n_params, ms, nprod):
    """Algorithmic floor of the whole step (SURVEY.md section 8d figures): every dense product of the GPS blocks
    (forward + input gradient + weight gradient = 3 x the forward flops, x `nprod` piece products) at the dense fp16 MFMA
    peak, plus the bytes that must cross HBM once -- GatedGCN gather / scatter forward + backward, the attention core's
    q|k|v / o / gradients, clip + AdamW's 28 B per parameter -- at the achievable copy rate.  Norm stages, encoders, head and
    launch gaps have no term: the floor is what the arithmetic needs, `frac` is floor / measured step."""
    shapes = [(N, d, 7 * d), (E, d, d), (N, d, d), (N, d, 2 * d), (N, 2 * d, d)]
    gemm_flops = 3 * layers * sum(2.0 * R * k * n for R, k, n in shapes)
    idx = 4 * (N + 1) + 8 * E
    sparse = layers * ((8 * E * d + 20 * N * d + idx) + (12 * E * d + 28 * N * d + 2 * idx))
    attn = layers * ((16 * N * d + 4 * H * N) + (32 * N * d + 12 * H * N))
    adamw = 28 * n_params
    mfma_ms = gemm_flops * nprod / (MFMA_BF16_PEAK_TF * 1e9)
    hbm_ms = (sparse + attn + adamw) / (HBM_ACHIEVABLE_GBS * 1e6)
    floor = mfma_ms + hbm_ms
    return {"floor_ms": floor, "frac": floor / ms, "mfma_ms": mfma_ms, "hbm_ms": hbm_ms,
            "gemm_flops_fp32_equiv": gemm_flops, "piece_products": nprod, "mfma_peak_tflops": MFMA_BF16_PEAK_TF,
            "sparse_bytes": sparse, "attention_bytes": attn, "adamw_bytes": adamw, "hbm_rate_gbs": HBM_ACHIEVABLE_GBS}


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
                    help="how the step is launched: replayed from hipGraph(s), eagerly, or (auto) replayed unless "
                         "eager beats the replay by > 3 %% in an interleaved trial taken after the capture")
    ap.add_argument("--no-graph", action="store_true", help="alias for --launch eager")
    ap.add_argument("--exchange", choices=("flat", "bucketed"), default="flat",
                    help="N>1 gradient exchange: one all-reduce of the flat gradient arena between "
                         "two hipGraphs (default), or eager per-layer buckets from autograd hooks")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="keep the BLAS libraries' default kernel heuristics (no TunableOp pass)")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true",
                    help="skip the secondary PCIe-inclusive measurement (host batches through DeviceLoader)")
    ap.add_argument("--no-bucketed-leg", action="store_true",
                    help="skip the third, secondary measurement (pcqm4m, 1 GPU, after the timed region): DIFFERENT host "
                         "batches (a shuffled loader's never-repeating shapes) through DeviceLoader + BucketPadding and "
                         "TrainStep.step_cached -- what train_epoch does -- beside the eager step on the same stream")
    ap.add_argument("--bucketed-leg", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (pcqm4m, 1 GPU, after the timed region): the zinc and code2 workloads "
                         "-- BASELINE.json configs[1] and configs[4] -- a few steps each, in child processes")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo ranks that sleep instead of stepping -- exercises the launcher / barrier / "
                         "max-over-ranks / JSON-line path of --gpus N on a CPU box (CPU test suite)")
    ap.add_argument("--cpu-steps", type=int, default=3)
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


def time_kernel(fn, iters=48, warm=6, nsets=1):
    """Mean duration (ms) of ``fn(i)`` (enqueues on torch's current stream, the stream the C-ABI
    launches on) measured with HIP events around ONE hipGraph replay of ``iters`` back-to-back launches: a
    ctypes call + hipLaunchKernel costs the host 10-25 us, so an eagerly launched loop of kernels shorter than
    that measures the host, not the kernel.  ``i`` cycles over ``nsets`` operand sets: 1 = the same buffers every
    launch (they stay in the 256 MiB Infinity Cache), > 1 = a rotation larger than the cache, so every launch
    streams from HBM.  Falls back to the eager loop when the launches cannot be captured."""
    for i in range(warm):
        fn(i % nsets)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    graph = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for i in range(iters):
                    fn(i % nsets)
        torch.cuda.current_stream().wait_stream(side)
        graph = g
    except Exception as exc:                      # noqa: BLE001 -- measurement fallback only
        log(f"time_kernel: capture failed ({type(exc).__name__}: {exc}); eager loop (host-bound below ~25 us)")
        torch.cuda.synchronize()
    if graph is not None:
        graph.replay()                            # one untimed replay (graph upload)
        torch.cuda.synchronize()
        t0.record()
        graph.replay()
        t1.record()
    else:
        t0.record()
        for i in range(iters):
            fn(i % nsets)
        t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


TRIAL_ROUNDS, TRIAL_STEPS = 2, 10   # launch-mode trial: 2 x (10 replayed + 10 eager) steps, interleaved
EAGER_MARGIN = 0.03                 # eager is only taken when it beats the replayed step by more than this
ROTATE_BYTES = 640 << 20      # > 2x the Infinity Cache: a rotation this large cannot be served from it


GEMM_ARITH = ("ring GEMMs + streaming weight gradients of the GPS blocks: fp32 storage, each operand value as 2 fp16 "
              "pieces (hi + lo, 22 significant bits) under a per-tensor power-of-two scale, 3 piece products "
              "(hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_f16, fp32 accumulation; max error vs fp64 <= the library "
              "fp32 GEMM's (tests/test_hip_ops.py::test_gemm_panel_split_products); GPS_GEMM_F16=0 / GPS_WGRAD_F16=0 select "
              "the exact 3 x bf16 / 6-product form; attention / FAVOR+ contractions: v_mfma_f32_16x16x4_f32 (fp32 in)")


def secondary_workloads(steps=10, warmup=5):
    """BASELINE.json configs[1] (zinc-GPS+RWSE) and configs[4] (ogbg-code2 GPS / Performer) made driver-visible: the same
    bench, ``steps`` timed steps each, run as CHILD processes after the headline's timed region (one model per process:
    INTEGRATION.md, process-wide state).  Never `value`; a failure here never takes the headline line with it."""
    import subprocess
    out = {}
    for wl, limit in (("zinc", 150), ("code2", 240)):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-kernel-roofline", "--no-h2d-leg", "--no-bucketed-leg", "--no-secondary"]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            out[wl] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"],
                       "warmup": d["warmup"], "workload": d["config"]["workload"], "launch_mode": d["launch_mode"],
                       "launch_trial_ms": d.get("launch_trial_ms"), "final_loss": d.get("final_loss"),
                       "wall_s": round(time.time() - t0, 1)}
            log(f"secondary {wl}: {d['ms_per_step']:.2f} ms/step = {d['value']:.0f} graphs/s ({d['launch_mode']})")
        except Exception as exc:            # noqa: BLE001 -- a side measurement
            out[wl] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
            log(f"secondary {wl} failed ({type(exc).__name__}: {exc})")
    return out


def eager_after_capture_probe(ts, make_batch, dev, n=20):
    """GPS_BENCH_LAUNCH_PROBE=1 (after the timed region): where VERDICT r4's '+13.6 % eager after the capture was dropped'
    comes from.  Eager steps timed (a) with the captured graph alive, (b) after the graph objects are destroyed,
    (c) after the caching allocator's free blocks went back too -- each with its host enqueue time per step."""
    import gc

    def leg(tag):
        for _ in range(3):
            ts.run_eager(make_batch())
        torch.cuda.synchronize(dev)
        host = []
        t0 = time.perf_counter()
        for _ in range(n):
            th = time.perf_counter()
            ts.run_eager(make_batch())
            host.append(time.perf_counter() - th)
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / n * 1e3
        res = {"ms_per_step": wall, "host_enqueue_ms_mean": sum(host) / n * 1e3,
               "reserved_MB": torch.cuda.memory_reserved(dev) / 2**20, "allocated_MB": torch.cuda.memory_allocated(dev) / 2**20}
        log(f"launch probe, eager {tag}: {wall:.2f} ms/step, host {res['host_enqueue_ms_mean']:.2f} ms, "
            f"reserved {res['reserved_MB']:.0f} MB")
        return res
    out = {"capture_alive": leg("with the captured graph alive")}
    for _ in range(3):
        ts.replay()
    torch.cuda.synchronize(dev)
    out["capture_alive_after_replays"] = leg("again, right after 3 replays")
    ts._g_fb = ts._g_fb2 = ts._g_up = ts._static_loss = ts._tick = None
    ts.mode = "eager"
    gc.collect()
    torch.cuda.synchronize(dev)
    out["graph_destroyed"] = leg("after the graph objects were destroyed")
    torch.cuda.empty_cache()
    out["cache_emptied"] = leg("after torch.cuda.empty_cache()")
    return out


def in_step_kernel_ms(step, n_steps=3):
    """{kernel name: (mean duration in ms, launches per step)} from the GPU activity records (roctracer via
    torch.profiler) of ``n_steps`` training steps -- the durations rocprofv3 --kernel-trace reports, taken in the
    step's real context (neighbouring kernels, the weight-gradient stream running beside it)."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
    out = {}
    for ev in prof.key_averages():
        tot = getattr(ev, "device_time_total", None)
        if tot is None:
            tot = getattr(ev, "cuda_time_total", 0.0)
        if ev.count and tot:
            out[ev.key] = (tot / ev.count / 1e3, ev.count / n_steps)
    return out


def _match(in_step, *needles):
    """(sum of the MEAN launch durations of the distinct kernels whose name contains one of ``needles`` -- one
    layer's worth when each of them runs once per layer --, launches per step seen).  Means, not totals: the
    activity buffer may drop records of the tail of a step."""
    ms = n = 0.0
    for k, (t, c) in in_step.items():
        if any(x in k for x in needles):
            ms += t
            n += c
    return (ms, n) if n else (None, 0)


def kernel_rooflines(dev, profile, nb, d=384, H=16, in_step=None, layers=10, only=None):
    """Layer-shaped measurements of each hand-written kernel at the benchmark's sizes: isolated (hot and
    rotating operand sets, HIP events) and -- when ``in_step`` holds the step's kernel records -- in-step."""
    from graphgps_amd import lib as L_
    from graphgps_amd.lib import check, current_stream, ptr
    from graphgps_amd.ops import build_graph_index
    from graphgps_amd.synthetic import layer_batch
    L = L_.load()
    b = layer_batch(profile, nb, d, seed=1234).to(dev)
    N, E = b.x.shape[0], b.edge_index.shape[1]
    gi = build_graph_index(b.edge_index, N, nb, ptr_vec=b.ptr)
    nmax_host = int((b.ptr[1:] - b.ptr[:-1]).max())
    st = current_stream(dev)
    f = lambda *s: torch.randn(*s, device=dev)
    fs = d * 4
    dh = d // H
    scale = dh ** -0.5
    sizes = (b.ptr[1:] - b.ptr[:-1]).double()
    s2 = float((sizes * sizes).sum())
    idx = 4 * (N + 1) + 8 * E

    def gg_set():
        return dict(proj=f(N, 4 * d), ce=f(E, d), xt=f(N, d), eh=f(E, d),
                    gx=f(N, d), ge=f(E, d), gproj=f(N, 4 * d), gce=f(E, d))
    set_bytes = 4 * (2 * N * 4 * d + 4 * E * d + 2 * N * d)
    n_rot = max(2, -(-ROTATE_BYTES // set_bytes))
    gsets = [gg_set() for _ in range(n_rot)]

    def gg_fwd(i=0):
        q = gsets[i]
        P = q["proj"].data_ptr()
        check(L.gps_gatedgcn_fwd(P, P + fs, P + 2 * fs, P + 3 * fs, 4 * d, ptr(q["ce"]), ptr(gi.rowptr_dst),
                                 ptr(gi.src_by_dst), ptr(gi.eid_by_dst), N, E, d, ptr(q["xt"]), ptr(q["eh"]),
                                 None, current_stream(dev)))

    def gg_bwd(i=0):
        q = gsets[i]
        P, G = q["proj"].data_ptr(), q["gproj"].data_ptr()
        check(L.gps_gatedgcn_bwd(ptr(q["gx"]), d, ptr(q["ge"]), ptr(q["eh"]), P, P + fs, 4 * d, ptr(q["xt"]),
                                 ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                                 ptr(gi.rowptr_src), ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, d,
                                 ptr(q["gce"]), G, G + fs, G + 2 * fs, G + 3 * fs, 4 * d, None, None, None,
                                 current_stream(dev)))

    # the form the fused block launches since round 6: bn_node_x / bn_edge_e backward applies evaluated in the loads
    # (gps_gatedgcn_bwd_bn): the same tensors cross HBM (g_x1 / g_e1 instead of g_x~ / g_e^) + x~ read once more
    import ctypes as _ct
    cols = [torch.rand(d, device=dev) + 0.5 for _ in range(8)] + [torch.randn(d, device=dev) for _ in range(4)]
    bnd = [L_.BnDesc(cols[4 * i].data_ptr(), cols[4 * i + 1].data_ptr(), cols[4 * i + 2].data_ptr(), cols[4 * i + 3].data_ptr(),
                     0, 0, 1e-5, 0.1) for i in range(2)]     # (weight, bias, mean, rstd, running stats unused)
    folds = [L_.BnBwdFold(_ct.addressof(bnd[i]), cols[8 + 2 * i].data_ptr(), cols[9 + 2 * i].data_ptr(), 0.1, 77 + i, 1, None)
             for i in range(2)]

    def gg_bwd_bn(i=0):
        q = gsets[i]
        P, G = q["proj"].data_ptr(), q["gproj"].data_ptr()
        check(L.gps_gatedgcn_bwd_bn(ptr(q["gx"]), d, ptr(q["ge"]), ptr(q["eh"]), P, P + fs, 4 * d, ptr(q["xt"]),
                                    ptr(gi.rowptr_dst), ptr(gi.src_by_dst), ptr(gi.eid_by_dst),
                                    ptr(gi.rowptr_src), ptr(gi.dst_by_src), ptr(gi.eid_by_src), N, E, d,
                                    ptr(q["gce"]), G, G + fs, G + 2 * fs, G + 3 * fs, 4 * d, None, None, None,
                                    _ct.byref(folds[0]), _ct.byref(folds[1]), current_stream(dev)))

    def at_set():
        return dict(qkv=f(N, 3 * d), out=f(N, d), lse=f(H, N), dout=f(N, d), delta=f(H, N), dqkv=f(N, 3 * d))
    a_bytes = 4 * (2 * N * 3 * d + 2 * N * d + 2 * H * N)
    a_rot = max(2, -(-ROTATE_BYTES // a_bytes))
    asets = [at_set() for _ in range(a_rot)]

    gi.nmax_host = nmax_host
    order = gi.attn_order(H)            # the balanced dispatch order the step's attention launches use (ops.GraphIndex)

    def at_fwd(i=0, p=0.1):
        q = asets[i]
        check(L.gps_seg_attn_fwd(ptr(q["qkv"]), 3 * d, ptr(gi.ptr), ptr(gi.tile_graph), ptr(gi.tile_row0),
                                 gi.max_tiles, N, H, dh, scale, p, 1234, ptr(q["out"]), ptr(q["lse"]), nb, nmax_host, None, ptr(order), current_stream(dev)))

    def at_bwd(i=0, p=0.1):
        q = asets[i]
        check(L.gps_seg_attn_bwd(ptr(q["dout"]), ptr(q["qkv"]), 3 * d, ptr(q["out"]), ptr(q["lse"]), ptr(gi.ptr),
                                 ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, H, dh, scale,
                                 p, 1234, ptr(q["delta"]), ptr(q["dqkv"]), 3 * d, nb, nmax_host, None, ptr(order), current_stream(dev)))

    for i in range(n_rot):                 # valid saved tensors (e_hat, x_tilde) for the backward kernel
        gg_fwd(i)
    for i in range(a_rot):
        at_fwd(i)
    res = {}

    def entry(name, fn, nsets, bound, work, launches, needles, note=None):
        if only is not None and name not in only:
            return
        hot = time_kernel(fn, nsets=1)
        rot = time_kernel(fn, nsets=nsets)
        ins, cnt = _match(in_step, *needles) if in_step else (None, 0)
        peak, unit, div = {"hbm": (HBM_PEAK_GBS, "GB/s", 1e6), "mfma": (MFMA_F32_PEAK_TF, "TFLOP/s", 1e9),
                           "mfma_bf16": (MFMA_BF16_PEAK_TF, "TFLOP/s", 1e9)}[bound]
        ms = ins if ins is not None else rot
        e = dict(bound=bound, ms=ms, ms_source="in-step (roctracer)" if ins is not None else "isolated, rotating",
                 in_step_ms=ins, isolated_hot_ms=hot, isolated_rotating_ms=rot,
                 achieved=work / ms / div, peak=peak, unit=unit, launches=launches,
                 in_step_launches_per_step=cnt or None)
        e["bytes" if bound == "hbm" else "flops"] = work
        e["frac"] = e["achieved"] / peak
        e["frac_isolated_hot"] = work / hot / div / peak
        e["frac_isolated_rotating"] = work / rot / div / peak
        if note:
            e["note"] = note
        res[name] = e

    entry("gatedgcn_fwd", gg_fwd, n_rot, "hbm", 8 * E * d + 20 * N * d + idx, 1, ("k_gatedgcn_fwd",))
    entry("gatedgcn_bwd", gg_bwd_bn, n_rot, "hbm", 12 * E * d + 32 * N * d + 2 * idx, 1, ("k_gatedgcn_bwd",),
          note="the folded form the step runs (gps_gatedgcn_bwd_bn): algorithmic bytes per SURVEY.md 8d (which counts num "
               "and den as read: 8*N*d that this kernel recomputes instead) + 4*N*d for x~ on the node fold")
    entry("gatedgcn_bwd_unfolded", gg_bwd, n_rot, "hbm", 12 * E * d + 28 * N * d + 2 * idx, 1, ("k_gatedgcn_bwd<4, false, 0>", "k_gatedgcn_bwd<4, false, 4>"),
          note="gps_gatedgcn_bwd behind two BatchNorm backward apply launches (rounds 1 - 5; GPS_GG_BN_FOLD=0)")
    # the attention core is HBM-bound at these graph sizes (7.8 flop/byte against a ridge of 19.6): the binding
    # roofline is Q/K/V read + O write (fwd), + dO read + dQ/dK/dV write (bwd); the MFMA figure rides along
    entry("seg_attn_fwd", at_fwd, a_rot, "hbm", 16 * N * d + 4 * H * N, 1, ("k_attn_fwd", "k_sattn_fwd"))
    if "seg_attn_fwd" in res:
        res["seg_attn_fwd"]["mfma_tflops"] = 4 * s2 * d / res["seg_attn_fwd"]["ms"] / 1e9
        res["seg_attn_fwd"]["mfma_frac"] = res["seg_attn_fwd"]["mfma_tflops"] / MFMA_F32_PEAK_TF
    entry("seg_attn_bwd", at_bwd, a_rot, "hbm", 32 * N * d + 12 * H * N, 2,
          ("k_attn_bwd", "k_sattn_bwd"))
    if "seg_attn_bwd" in res:
        res["seg_attn_bwd"]["mfma_tflops"] = 8 * s2 * d / res["seg_attn_bwd"]["ms"] / 1e9
        res["seg_attn_bwd"]["mfma_frac"] = res["seg_attn_bwd"]["mfma_tflops"] / MFMA_F32_PEAK_TF
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
    from graphgps_amd import gemm as _gemm
    f16 = bool(_gemm.F16)
    NPROD = 3 if f16 else 6         # piece products per fp32 product: 2 x fp16 / 3 products (round 4) or 3 x bf16 / 6
    if f16:                         # the operands' max|.| records, made once (in the step their producers make them)
        wrec = _gemm.absmax([t_ for pr in pairs for t_ in pr])
        keep.append(wrec)
        for j, q in enumerate(probs):
            q.g_amax, q.x_amax = wrec[2 * j].data_ptr(), wrec[2 * j + 1].data_ptr()
    SPLIT_NOTE = (("flops = the fp16 MFMA flops issued: 3 piece products per fp32 product (two fp16 pieces per value "
                   "under a per-tensor power-of-two scale: lo*hi, hi*lo, hi*hi), against the dense fp16 / bf16 MFMA peak; "
                   "fp32-equivalent rate = achieved / 3") if f16 else
                  ("flops = the bf16 MFMA flops issued: 6 piece products per fp32 product (exact 3-way split, "
                   "hh hm mh hl lh mm), against the dense bf16 MFMA peak; fp32-equivalent rate = achieved / 6"))
    entry("wgrad_grouped", lambda i=0: check(L.gps_wgrad_grouped(len(pairs), probs, ptr(wws), current_stream(dev))), 1,
          "mfma_bf16", NPROD * sum(2.0 * R * k * n for R, k, n in shapes), 2, ("k_wgrad",), note=SPLIT_NOTE)
    # the projection GEMMs of one block through the ring kernel (csrc/gemm_panel.hip): forward five, input-gradient five
    if _gemm.supported(d, d):
        ws_ = [f(n, k) for _, k, n in shapes]
        imgs = _gemm.split_weights(ws_)
        xs_ = [f(R, k) for R, k, _ in shapes]
        gs_ = [f(R, n) for R, _, n in shapes]
        ys_ = [torch.empty(R, n, device=dev) for R, _, n in shapes]
        dx_ = [torch.empty(R, k, device=dev) for R, k, _ in shapes]
        xrec = _gemm.absmax(xs_) if f16 else [None] * len(xs_)
        grec = _gemm.absmax(gs_) if f16 else [None] * len(gs_)
        def fwd5(i=0):
            for x_, (nt, _), y_, (_, _, n), r_ in zip(xs_, imgs, ys_, shapes, xrec):
                _gemm.gemm_panel(x_, nt, n, out=y_, a_amax=r_)
        def dgrad5(i=0):
            for g_, (_, tn), o_, (_, k, _), r_ in zip(gs_, imgs, dx_, shapes, grec):
                _gemm.gemm_panel(g_, tn, k, out=o_, a_amax=r_)
        fl = NPROD * sum(2.0 * R * k * n for R, k, n in shapes)
        entry("gemm_fwd_block", fwd5, 1, "mfma_bf16", fl, 5, (), note=SPLIT_NOTE + "; isolated only (one kernel name serves every shape)")
        entry("gemm_dgrad_block", dgrad5, 1, "mfma_bf16", fl, 5, (), note=SPLIT_NOTE + "; isolated only")
        if in_step and (only is None or "gemm_ring_in_step" in only):
            # the ring GEMM FAMILY inside the step: every k_gemm_ring16 instantiation's (mean duration x launches per step)
            # against the forward + input-gradient flops of all layers (one kernel name serves several shapes, so the
            # family is the finest grain the in-step records resolve)
            tot = sum(t * _per_step(c, layers) for k_, (t, c) in in_step.items() if "k_gemm_ring" in k_)
            cnt = sum(_per_step(c, layers) for k_, (t, c) in in_step.items() if "k_gemm_ring" in k_)
            if tot > 0:
                work = 2 * fl * layers
                res["gemm_ring_in_step"] = dict(
                    bound="mfma_bf16", ms=tot, ms_source="in-step (roctracer), all instantiations, per STEP",
                    in_step_ms=tot, achieved=work / tot / 1e9, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", launches=cnt,
                    in_step_launches_per_step=cnt, flops=work, frac=work / tot / 1e9 / MFMA_BF16_PEAK_TF,
                    note=SPLIT_NOTE + "; flops and ms are per step (all layers, forward + input gradient)")
    return res, dict(N=N, E=E, d=d, H=H, sum_n2=s2, rotation_sets=dict(gatedgcn=n_rot, attention=a_rot))


def favor_rooflines(dev, nb, d=256, H=4, in_step=None):
    """BASELINE configs[4] (ogbg-code2-GPS.yaml): the FAVOR+ kernels (csrc/favor.hip) at code2-long sizes --
    4 heads x dim_head 64, m = 266 random features -- against the fp32 MFMA peak.  Flops per layer (the
    reference's formulation): fwd 8*N*m*(64H) + 2*N*m*H, bwd 2x."""
    from graphgps_amd import lib as L_
    from graphgps_amd.lib import check, current_stream, ptr
    from graphgps_amd.ops import _nmax_dev, build_graph_index
    from graphgps_amd.synthetic import layer_batch
    L = L_.load()
    b = layer_batch("CODE2_LONG", nb, d, seed=1234).to(dev)
    N, E = b.x.shape[0], b.edge_index.shape[1]
    gi = build_graph_index(b.edge_index, N, nb, ptr_vec=b.ptr)
    st = current_stream(dev)
    dh, m = 64, 266
    inner = H * dh
    f32 = dict(dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    from graphgps_amd.layer.performer_layer import gaussian_orthogonal_random_matrix
    proj = gaussian_orthogonal_random_matrix(m, dh).to(dev)
    qkv = torch.randn(N, 3 * inner, **f32) * 0.7
    out = torch.empty(N, inner, **f32)
    cbuf, ksum = torch.empty(nb * H, 272, dh, **f32), torch.empty(nb * H, 272, **f32)
    kmax = torch.empty(nb * H, dtype=torch.int64, device=dev)
    mq, D = torch.empty(H, N, **f32), torch.empty(H, N, **f32)
    nmax = _nmax_dev(gi)
    g_out = torch.randn(N, inner, **f32)
    gD, g_ctx, g_ksum = torch.empty(H, N, **f32), torch.empty_like(cbuf), torch.empty_like(ksum)
    gm_part = torch.empty(max(gi.max_tiles * H, 1), **f32)
    d_qkv = torch.empty_like(qkv)
    wsf = int(L.gps_favor_workspace_floats(N, nb, H))       # partial context records of the row slices
    fws = torch.empty(max(wsf, 1), **f32)

    def fwd(i=0):
        check(L.gps_favor_fwd(ptr(qkv), 3 * inner, ptr(proj), m, ptr(gi.ptr), ptr(nmax), ptr(gi.tile_graph),
                              ptr(gi.tile_row0), gi.max_tiles, N, nb, H, dh, ptr(out), ptr(cbuf), ptr(ksum),
                              ptr(kmax), ptr(mq), ptr(D), ptr(fws) if wsf else None, wsf, current_stream(dev)))

    def bwd(i=0):
        check(L.gps_favor_bwd(ptr(g_out), ptr(qkv), 3 * inner, ptr(proj), m, ptr(out), ptr(gi.ptr), ptr(nmax),
                              ptr(gi.tile_graph), ptr(gi.tile_row0), gi.max_tiles, N, nb, H, dh, ptr(cbuf),
                              ptr(ksum), ptr(kmax), ptr(mq), ptr(D), ptr(gD), ptr(g_ctx), ptr(g_ksum),
                              ptr(gm_part), ptr(d_qkv), 3 * inner, ptr(fws) if wsf else None, wsf, current_stream(dev)))

    fwd()
    flf = 8.0 * N * m * inner + 2.0 * N * m * H
    res = {}
    for name, fn, fl, is_bwd in (("favor_fwd", fwd, flf, False), ("favor_bwd", bwd, 2 * flf, True)):
        t = time_kernel(fn, iters=20, warm=3)
        e = dict(bound="mfma", isolated_hot_ms=t, flops=fl, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s")
        if in_step:
            import re
            names = [k for k in in_step if "favor" in k.lower() and ("bwd" in k.lower()) == is_bwd]
            short = lambda k: (re.search(r"k_favor\w*", k) or re.search(r"\w+", k)).group(0)
            e["in_step_kernels"] = {short(k): dict(ms=in_step[k][0], per_step=in_step[k][1]) for k in names}
            tot = sum(in_step[k][0] for k in names)      # each kernel runs once per layer: sum of mean durations
            if tot:
                e["in_step_ms"] = tot
        e["ms"] = e.get("in_step_ms", t)
        t = e["ms"]
        e["achieved"] = fl / t / 1e9
        e["frac"] = e["achieved"] / MFMA_F32_PEAK_TF
        res[name] = e
    if in_step:     # in-step: all FAVOR kernels of a step / (layers x (fwd + bwd)) against 3x the forward flops
        tot = sum(t for k, (t, c) in in_step.items() if "favor" in k.lower())
        if tot:
            res["favor_in_step"] = dict(bound="mfma", ms_per_layer_fwd_bwd=tot, flops=3 * flf,
                                        achieved=3 * flf / tot / 1e9, peak=MFMA_F32_PEAK_TF,
                                        unit="TFLOP/s", frac=3 * flf / tot / 1e9 / MFMA_F32_PEAK_TF)
    return res, dict(N=N, E=E, d=d, H=H, dim_head=dh, nb_features=m)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, batch_cpu, compute_loss, clip_value, steps, threads=0):
    """CPU oracle (kind='port'): same batch, same step definition, on the host cores -- two legs:
    all usable cores (capped at 32: the op granularity of the reference path -- [7.7k,384] GEMMs, gathers,
    elementwise -- stops scaling well before that) and torch.set_num_threads(6), the reference's default
    ``cfg.num_threads`` (main.py:125).  The headline leg is the faster one."""
    from oracle.gps_oracle import to_oracle_model
    oracle = to_oracle_model(model).train()
    opt = torch.optim.AdamW(oracle.parameters(), lr=2e-4, weight_decay=0.0)
    params = list(oracle.parameters())
    nb = int(batch_cpu.num_graphs)

    def step():
        opt.zero_grad(set_to_none=True)
        pred, true = oracle(batch_cpu.clone())
        loss, _ = compute_loss(pred, true)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, clip_value)
        opt.step()

    def leg(cores, budget_s):
        torch.set_num_threads(cores)
        tw = time.perf_counter()
        step()                                   # warm-up (allocator, thread pool)
        log(f"cpu baseline [{cores} threads]: warm-up step {time.perf_counter() - tw:.2f}s")
        t0 = time.perf_counter()
        done = 0
        while done < steps and (done == 0 or time.perf_counter() - t0 < budget_s):
            step()
            done += 1
        dt = (time.perf_counter() - t0) / done
        return dict(value=nb / dt, unit="graphs/s", cores=cores, ms_per_step=dt * 1e3, steps=done)

    all_cores = threads or min(usable_cores(), 32)
    legs = [leg(all_cores, 12.0)]
    if all_cores != 6 and usable_cores() >= 6:
        legs.append(leg(6, 12.0))
    best = max(legs, key=lambda l: l["value"])
    return dict(value=best["value"], unit="graphs/s", cores=best["cores"], kind="port",
                ms_per_step=best["ms_per_step"], cpu_model=cpu_model_name(), usable_cores=usable_cores(),
                legs=legs,
                sample=f"{best['steps']} full training step(s) of the same {nb}-graph batch after 1 warm-up per leg, "
                       f"pure-torch CPU oracle of the reference path; legs: "
                       + ", ".join(f"{l['cores']} threads = {l['value']:.1f} graphs/s" for l in legs))


def bucketed_loader_leg(model, opt, loss_fn, nb, profile, dev, n_batches=24):
    """Secondary, never `value`: the loader-fed step on batches that DIFFER in shape like a shuffled loader's
    (custom_train.py:16-47 feeds every batch of the epoch through the same step): `n_batches` synthetic batches of
    different structure through DeviceLoader + BucketPadding -- padded up to shape buckets, H2D copies + graph index
    staged ahead on a copy stream -- and TrainStep.step_cached, which replays a captured step per bucket.  Two forms:
    the padding on DeviceLoader's staging thread (host batches arrive un-padded and un-pinned, as from a plain
    DataLoader), and batches that arrive padded and pinned (BucketPadding.collate in the DataLoader's worker processes +
    pin_memory=True).  Untimed passes meet the buckets (each captured at its first sight, after one eager step), the timed pass runs
    the same batches in another order.  For scale, the eager step on the same un-padded stream (every batch a new shape:
    every first sight of a row count costs host time in the libraries) is timed too.  Runs after the timed region: it cannot move `value`."""
    try:
        from graphgps_amd.loader import BucketPadding, DeviceLoader
        from graphgps_amd.synthetic import model_batch
        from graphgps_amd.train import TrainStep, padding_supported
        if not padding_supported(model):
            return {"skipped": "model not served by the padded path"}
        host = [model_batch("pcqm4m", nb, seed=5000 + i, profile=profile) for i in range(n_batches)]
        raw_shapes = len({(b.x.shape[0], b.edge_index.shape[1]) for b in host})
        order2 = lambda seq: [seq[(7 * i + 3) % n_batches] for i in range(n_batches)]

        def run(ts, seq, pad, cached):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in DeviceLoader((q.shallow_copy() for q in seq), dev, pad=pad):
                ts.step_cached(b, max_graphs=12) if cached else ts._eager_triplet(b)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / len(seq) * 1e3
        out = {"batches": n_batches, "distinct_raw_shapes": raw_shapes}
        # (0) eager on the un-padded stream
        tse = TrainStep(model, opt, loss_fn=loss_fn)
        run(tse, host, None, False)
        out["eager_unpadded_ms_per_step"] = run(tse, order2(host), None, False)
        # (a) padding on the staging thread
        pad = BucketPadding()
        tsb = TrainStep(model, opt, loss_fn=loss_fn)
        run(tsb, host + host, pad, True)
        out["staging_thread_padding_ms_per_step"] = run(tsb, order2(host), pad, True)
        # (b) batches arrive padded and pinned
        pre = [pad(b) for b in host]
        real = sum(int(b.__dict__["_gps_meta"]["n_real"]) + int(b.__dict__["_gps_meta"]["e_real"]) for b in pre)
        rows = sum(b.x.shape[0] + b.edge_index.shape[1] for b in pre)
        for b in pre:
            for k, v in list(b.__dict__.items()):
                if torch.is_tensor(v):
                    b.__dict__[k] = v.pin_memory()
        run(tsb, pre, None, True)
        cache = tsb.__dict__.get("_shape_cache", {})
        before = {k for k, v in cache.items() if v}
        out["ms_per_step"] = run(tsb, order2(pre), None, True)
        out.update(shape_buckets=len(cache), failed_captures=sum(1 for v in cache.values() if v is False),
                   replayed_steps=sum(1 for b in pre if tsb._shape_key(b) in before),
                   padding_fraction=rows / max(real, 1) - 1.0, node_step=pad.node_step, edge_step=pad.edge_step)
        log(f"bucketed loader leg: {out['ms_per_step']:.2f} ms/step with batches that arrive padded + pinned, "
            f"{out['staging_thread_padding_ms_per_step']:.2f} with the padding on the staging thread, "
            f"{out['eager_unpadded_ms_per_step']:.2f} eager on the un-padded stream ({raw_shapes} raw shapes in "
            f"{len(cache)} buckets, padding {out['padding_fraction'] * 100:.1f} %)")
        torch.cuda.synchronize()
        del tsb, tse
        return out
    except Exception as exc:        # a side measurement never takes the headline line with it
        log(f"bucketed loader leg skipped ({type(exc).__name__}: {exc})")
        return {"skipped": f"{type(exc).__name__}: {exc}"}


def timed_region(step, steps, barrier, world, dev):
    """EXACTLY ``steps`` steps between two barriers (torch.distributed.barrier + device synchronize), the MAX of the
    elapsed wall time over the ranks: (seconds, last step's return value).  Shared by the real run and --dry-run."""
    loss = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, loss


def dry_run(args, real_stdout):
    """--dry-run: the launcher / rank / barrier / max-over-ranks / one-JSON-line path of ``--gpus N`` WITHOUT a GPU -- gloo
    on CPU, the step replaced by a rank-dependent sleep (rank r sleeps 2 (r + 1) ms, so the max over ranks is checkable).
    Exists so that the first real 8-GPU run is not the first run of this path (tests/test_cpu_boundary.py drives it with
    2 ranks under torch.distributed.run and through the self-respawn).  Nothing of the product is measured."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(real_stdout, 1)
        respawn_under_launcher(args.gpus, check_devices=False)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend="gloo")

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def step():
        time.sleep(2e-3 * (rank + 1))
        return torch.zeros(())
    for _ in range(args.warmup):
        step()
    elapsed, _ = timed_region(step, args.steps, barrier, world, dev)
    ms = elapsed / args.steps * 1e3
    nb = args.graphs_per_gpu or 256
    if rank == 0:
        out = {"metric": "dry-run of the launcher path (no GPU work, nothing of the product measured)",
               "value": world * nb / (ms / 1e3), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "none", "data": "none (dry run: sleeps)",
               "config": {"workload": "dry-run", "global_batch": world * nb, "parallelism": f"dp{world}"}}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        torch.distributed.destroy_process_group()


def respawn_under_launcher(n, check_devices=True):
    """`python bench.py --gpus N` (N > 1) without a launcher environment: re-execute under torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1 -- the command the driver itself uses."""
    import socket
    import subprocess
    if check_devices and torch.cuda.device_count() < n:
        raise SystemExit(f"--gpus {n}: only {torch.cuda.device_count()} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("re-executing under the launcher: " + " ".join(cmd))
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse_args()
    # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract
    # is ONE JSON line on stdout, so everything before the final print goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, real_stdout)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GPS hot path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(real_stdout, 1)                # the ranks inherit the real stdout for the JSON line
        respawn_under_launcher(args.gpus)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # GPS_BENCH_SHARE_GPU=1 + GPS_BENCH_BACKEND=gloo (testing only, never a measurement): every rank on device 0 with the
    # collectives on gloo -- RCCL refuses two ranks on one device -- so that the N > 1 control flow of this file (matched
    # collectives, the two-graph step around the all-reduce, one JSON line) can be run on a 1-GPU box
    share_gpu = os.environ.get("GPS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("GPS_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

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
    if args.workload == "zinc":
        args.no_kernel_roofline = True              # launch-latency-bound sizes: numerics config, no roofline
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

    if world > 1:
        # replicas: rank 0's parameters and buffers everywhere (dp.broadcast_state: BatchNorm running statistics, the
        # Performer's random projection buffers) -- identical seeds above already make them equal; this makes it a property
        # of the run instead of a property of the seeding
        from graphgps_amd.dp import broadcast_state
        bcast_bytes = broadcast_state(model)
        log(f"rank {rank}: replica state broadcast from rank 0 ({bcast_bytes / 1e6:.1f} MB)")
    torch.manual_seed(1000 + rank)             # dropout streams differ per rank
    # fresh batch object over the resident tensors -> the graph index is rebuilt every step, nothing is
    # copied (inputs already in HBM is the measurement contract)
    make_batch = batch_dev.shallow_copy
    reducer = exchange = None
    trial = {}
    h2d_ms = None
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
        # N > 1: the backward is cut in the middle of the layer stack, the gradients of the upper half travel while the
        # lower half is differentiated (train.TrainStep backward_split)
        layers = getattr(model, "layers", None)
        cut_at = layers[len(layers) // 2 - 1] if (exchange is not None and layers is not None and len(layers) >= 2
                                                   and os.environ.get("GPS_DP_SPLIT", "1") != "0") else None
        ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=exchange, salt=salt, backward_split=cut_at)

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

        if launch == "auto":                 # reference figure only (never decides): eager before anything is captured
            trial["eager_before_capture"] = trial_ms()
        # secondary, never `value`: the same step fed from pinned HOST batches through the prefetching DeviceLoader (H2D
        # copies + graph index staged by a worker thread on a copy stream) -- the PCIe-inclusive rate, i.e. what
        # train_epoch sees.  Eager launches, so it is measured here, before a captured graph is alive.
        if not args.no_h2d_leg and world == 1:
            try:
                from graphgps_amd.loader import DeviceLoader
                pinned = batch_cpu.shallow_copy()
                for k, v in list(pinned.__dict__.items()):
                    if torch.is_tensor(v):
                        pinned.__dict__[k] = v.pin_memory()
                for n_h in (4, min(max(args.steps, 8), 24)):       # a few untimed, then the measured pass
                    torch.cuda.synchronize()
                    th = time.perf_counter()
                    for b in DeviceLoader((pinned.shallow_copy() for _ in range(n_h)), dev):
                        ts.run_eager(b)
                    torch.cuda.synchronize()
                    h2d_ms = (time.perf_counter() - th) / n_h * 1e3
                log(f"host-batch leg (H2D + index staged on a worker thread / copy stream, eager launch): {h2d_ms:.2f} ms/step")
            except Exception as exc:       # never let the side measurement take the headline line with it
                h2d_ms = None
                log(f"host-batch leg skipped ({type(exc).__name__}: {exc})")
        if launch != "eager":
            try:
                ts.capture(make_batch)
                log("step captured: " + ts.mode)
                if os.environ.get("GPS_BENCH_SNAPSHOT"):     # who owns which page (the linear-capture fault)
                    torch.cuda.synchronize()
                    segs = [{"address": q["address"], "size": q["total_size"], "type": q.get("segment_type"),
                             "pool": str(q.get("segment_pool_id")),
                             "blocks": [{"address": b_.get("address"), "size": b_["size"], "state": b_["state"]}
                                        for b_ in q.get("blocks", [])][:2000]}
                            for q in torch.cuda.memory_snapshot()]
                    maps = []
                    with open("/proc/self/maps") as f:
                        for line in f:
                            p_ = line.split()
                            lo, hi = (int(x, 16) for x in p_[0].split("-"))
                            maps.append({"lo": lo, "hi": hi, "perm": p_[1], "name": p_[-1] if len(p_) > 5 else ""})
                    json.dump({"segments": segs, "maps": maps}, open(os.environ["GPS_BENCH_SNAPSHOT"], "w"))
            except Exception as exc:         # capture is an optimisation, never a requirement
                log(f"hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly")
                ts = TrainStep(model, opt, loss_fn=compute_loss, exchange=exchange, salt=salt, backward_split=cut_at)
                launch = "eager"
            if world > 1:                    # one rank falling back must take every rank with it
                ok = torch.tensor([0.0 if launch == "eager" else 1.0], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                if float(ok) == 0.0 and launch != "eager":
                    log("another rank could not capture the step; running eagerly everywhere")
                    launch = "eager"
        if launch == "auto":
            # Both launch forms measured under the conditions of the timed region: AFTER the capture, with the captured
            # graph alive (it stays alive whichever form wins -- nothing is dropped between the trial and the timed
            # loop), interleaved replay / eager / replay / eager so that clock and cache drift hit both alike.  Replay is
            # the product mode (train_epoch replays per shape bucket; its host cost is one graph launch per step, which
            # is what 8 ranks sharing a host need), so eager has to win by more than 3 % to be taken, and the choice is
            # re-checked against a few steps right before the timed region.
            per = {"graph": [], "eager": []}
            for _ in range(TRIAL_ROUNDS):
                for mode in ("graph", "eager"):
                    ts.use_replay = mode == "graph"
                    per[mode].append(trial_ms(n_warm=2, n=TRIAL_STEPS))
            trial["graph"] = sum(per["graph"]) / len(per["graph"])
            trial["eager"] = sum(per["eager"]) / len(per["eager"])
            trial["rounds"] = per
            launch = "eager" if trial["eager"] < (1.0 - EAGER_MARGIN) * trial["graph"] else "graph"
            log(f"launch-mode trial ({TRIAL_ROUNDS} x {TRIAL_STEPS} steps each, interleaved, capture alive): "
                f"graph {trial['graph']:.2f} ms, eager {trial['eager']:.2f} ms "
                f"(eager before the capture: {trial['eager_before_capture']:.2f}) -> {launch}")
            if launch == "eager":            # the re-check: the chosen form once more, right before the timed region
                ts.use_replay = False
                again = trial_ms(n_warm=1, n=TRIAL_STEPS)
                trial["eager_recheck"] = again
                if again >= (1.0 - EAGER_MARGIN) * trial["graph"]:
                    log(f"re-check: eager {again:.2f} ms no longer beats replay {trial['graph']:.2f} ms by "
                        f"{EAGER_MARGIN:.0%} -> graph")
                    launch = "graph"
        ts.use_replay = launch == "graph"
        from graphgps_amd import fused as _fused
        two = _fused._BLOCK_SIDE_ENABLED if args.workload == "pcqm4m" else _fused._SIDE_ENABLED
        graph_mode = ts.mode if launch == "graph" else (
            "eager (2 HIP streams: main + weight-gradient)" if two else "eager (one HIP stream)")
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
    log("warm-up done")
    elapsed, loss = timed_region(step, args.steps, barrier, world, dev)
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
    # (the host-batch leg -- pcie_inclusive_ms_per_step -- was measured before the capture, see host_batch_leg above)
    bucketed = None
    if not args.no_bucketed_leg and args.workload == "pcqm4m" and world == 1 and reducer is None:
        bucketed = bucketed_loader_leg(model, opt, compute_loss, nb, args.profile, dev)

    # the in-step kernel records: a few more steps under the tracer.  On EVERY rank -- with N > 1 a step contains the
    # gradient all-reduce, and a collective that only rank 0 enters never returns (found by reading the N > 1 path in round 5:
    # no 8-GPU run has exercised it yet); rank 0 alone uses the records
    in_step = None
    if not args.no_kernel_roofline:
        try:
            in_step = in_step_kernel_ms(step)
        except Exception as exc:
            log(f"in-step kernel records unavailable ({type(exc).__name__}: {exc}); isolated timings only")
        barrier()
    secondary = None
    if rank == 0 and world == 1 and args.workload == "pcqm4m" and not args.no_secondary:
        secondary = secondary_workloads()

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
            "dropout": "on (config values; masks from the kernels' counter hash -- parity of this configuration is "
                       "element-wise: the oracle with the blocks' masks injected, 1 / 2 layers and the 10-layer model, "
                       "tests/test_hip_layer.py::test_fused_block_with_dropout_on_vs_masked_oracle, "
                       "::test_full_model_with_dropout_on_vs_masked_oracle)",
            "config": {"workload": f"{wl_label}, synthetic profile "
                                   f"{args.profile or {'pcqm4m': 'P30', 'zinc': 'ZINC', 'code2': 'CODE2_LONG'}[args.workload]}, "
                                   f"{nb} graphs/GPU ({N} nodes, {E} directed edges on rank 0)",
                       "global_batch": world * nb, "parallelism": f"dp{world}",
                       "timed_region": "graph-index build + forward + L1 loss + backward + "
                                       "grad all-reduce + clip + AdamW"},
            "final_loss": final_loss,
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "pcie_inclusive_ms_per_step": h2d_ms,
            "pcie_inclusive_bucketed": bucketed,
            "launch_mode": graph_mode,
            "launch_trial_ms": trial or None,
            "launch_rule": (f"replay unless eager beats it by > {EAGER_MARGIN:.0%}; both forms timed after the capture, "
                            f"capture alive, {TRIAL_ROUNDS} x {TRIAL_STEPS} steps each, interleaved; "
                            "host_enqueue_ms_per_step is the chosen form's"),
            "gemm_arith": GEMM_ARITH,
            "collective_backend": (None if world == 1 and not use_exchange else
                                   ("RCCL (backend nccl)" if backend == "nccl" else f"{backend} -- TEST RUN, not a measurement"
                                    + (", all ranks on one GPU" if share_gpu else ""))),
            "secondary": secondary,
            "grad_allreduce_bytes": allreduce_bytes,
            "optimizer": type(opt).__name__,
            "gemm_selection": "TunableOp (rocBLAS/hipBLASLt solutions timed during warm-up)"
                              if tunable is not None else "library default heuristics",
        }
        if not args.no_kernel_roofline:
            if in_step is not None:
                out["dispatches_per_step"] = round(sum(_per_step(c, int(cfg.gt.layers)) for _, c in in_step.values()), 1)
                out["in_step_kernel_ms"] = {k: dict(ms=round(v[0], 5), per_step=v[1]) for k, v in
                                            sorted(in_step.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:24]}
            if args.workload == "pcqm4m":
                layers_n = int(cfg.gt.layers)
                kr, shape = kernel_rooflines(dev, args.profile or "P30", nb, in_step=in_step, layers=layers_n)
                pmc_static = {}
                pmc = os.path.join(ROOT, "profiles", "pmc_static.json")
                if os.path.exists(pmc):           # HBM bytes per launch from separate rocprofv3 --pmc passes (counters and
                    pmc_static = json.load(open(pmc))     # timing never share a run): read from the committed file
                dom = dominant_roofline(in_step, kr, pmc_static, layers_n) if in_step else None
                if dom is None:                   # no in-step records: the largest kernel of the last committed profile
                    dom = dominant_roofline({"k_wgrad_direct": (kr["wgrad_grouped"]["ms"], layers_n)}, kr, pmc_static, layers_n)
                out["roofline"] = dom
                from graphgps_amd import gemm as _gemm
                out["roofline_step"] = step_roofline(shape["N"], shape["E"], shape["d"], shape["H"], layers_n,
                                                     sum(p.numel() for p in model.parameters()), ms,
                                                     3 if _gemm.F16 else 6)
            else:
                kr, shape = favor_rooflines(dev, nb, in_step=in_step)
                k = kr.get("favor_in_step") or kr["favor_fwd"]
                out["roofline"] = {"kernel": "k_favor_* (FAVOR+ forward + backward of one layer)", "bound": "mfma",
                                   "achieved": k["achieved"], "peak": k["peak"], "unit": "TFLOP/s",
                                   "frac": k["frac"], "traffic": None}
            log("kernel rooflines done")
            out["kernels"] = kr
            out["kernel_shape"] = shape
        if os.environ.get("GPS_BENCH_LAUNCH_PROBE") == "1" and reducer is None and world == 1 and ts.mode != "eager":
            out["launch_probe"] = eager_after_capture_probe(ts, make_batch, dev)
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
