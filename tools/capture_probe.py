#!/usr/bin/env python
"""Which ingredient of the host-batch path makes a LATER hipGraph capture of the step die in capture_end?
One mode per process:  python tools/capture_probe.py <mode>   (see MODES)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ("keep_graph", "resident", "loader", "loader_noindex", "manual_keep", "manual_record", "record_only", "copystream_only",
         "event_only", "pinned_only")

if __name__ == "__main__":
    mode = sys.argv[1]
    import graphgps_amd as g
    from graphgps_amd.loader import DeviceLoader
    from graphgps_amd.loss.losses import compute_loss
    from graphgps_amd.optim import FlatAdamW
    from graphgps_amd.synthetic import model_batch
    from graphgps_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = g.create_model(os.path.join(g.CONFIG_DIR, "zinc_gps_rwse.yaml"),
                       ["gt.layers", 2, "gt.layer_type", "CustomGatedGCN+Transformer"], 1, 1).to(dev).train()
    opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    ts = TrainStep(m, opt, loss_fn=compute_loss)
    host = model_batch("zinc", 16, seed=7)
    res = host.clone().to(dev)
    keep = []
    cs = torch.cuda.Stream(device=dev)
    if mode == "keep_graph":       # the minimal reproducer: ONE eager step whose batch object (-> batch.x -> graph) stays alive
        held = res.shallow_copy()
        pred, true = m(held)
        compute_loss(pred, true)[0].backward()
        keep.append(held)
    elif mode == "resident":
        for _ in range(3):
            ts.run_eager(res.shallow_copy())
    elif mode in ("loader", "loader_noindex"):
        for b in DeviceLoader([host.clone() for _ in range(3)], dev, background=False, build_index=mode == "loader"):
            ts.run_eager(b)
    elif mode in ("manual_keep", "manual_record"):
        for _ in range(3):
            b = host.clone()
            with torch.cuda.stream(cs):
                for k, v in list(b.__dict__.items()):
                    if torch.is_tensor(v):
                        b.__dict__[k] = v.pin_memory().to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
            torch.cuda.current_stream(dev).wait_event(ev)
            if mode == "manual_record":
                for v in b.__dict__.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(dev))
            else:
                keep.append(b)
            ts.run_eager(b)
    elif mode == "record_only":
        for _ in range(3):
            b = res.clone()
            for v in b.__dict__.values():
                if torch.is_tensor(v):
                    v.record_stream(cs)
            ts.run_eager(b)
    elif mode == "copystream_only":
        for _ in range(3):
            with torch.cuda.stream(cs):
                t = torch.zeros(1000, device=dev)
            keep.append(t)
            ts.run_eager(res.shallow_copy())
    elif mode == "event_only":
        for _ in range(3):
            ev = torch.cuda.Event()
            ev.record(cs)
            torch.cuda.current_stream(dev).wait_event(ev)
            ts.run_eager(res.shallow_copy())
    elif mode in ("fresh_clone", "fresh_then_resident", "fresh_gc", "fresh_meta"):
        for _ in range(3):
            b = res.clone()
            if mode == "fresh_meta":
                b.__dict__["_gps_meta"] = res.__dict__.get("_gps_meta")
            ts.run_eager(b)
            del b
        if mode == "fresh_then_resident":
            ts.run_eager(res.shallow_copy())
        if mode == "fresh_gc":
            import gc
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.empty_cache()
    elif mode == "pinned_only":
        for _ in range(3):
            keep.append(torch.randn(1000).pin_memory().to(dev, non_blocking=True))
            ts.run_eager(res.shallow_copy())
    torch.cuda.synchronize()
    print(mode, "eager done", flush=True)
    pre = os.environ.get("GPS_PROBE_PRE", "")
    if "gc" in pre:
        import gc
        del keep
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if "host" in pre and hasattr(torch._C, "_host_emptyCache"):
        torch._C._host_emptyCache()
        print("host cache emptied", flush=True)
    if "sleep" in pre:
        import time
        time.sleep(1.0)
        torch.cuda.synchronize()
    ts.capture(res.shallow_copy, warmup=1)
    for _ in range(3):
        ts.replay()
    torch.cuda.synchronize()
    print(mode, "CAPTURE OK", flush=True)
