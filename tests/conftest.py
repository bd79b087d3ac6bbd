import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _usable_cores() -> int:
    """Host cores this process may actually use: affinity mask AND cgroup CPU quota.  torch sizes its intra-op pool by the
    visible cores; on a GPU box that shows every core of the host but grants a 16-core quota, the CPU oracle legs of the
    parity tests ran oversubscribed (the round-4 suite took > 9 minutes of mostly throttled CPU time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


torch.set_num_threads(max(1, min(torch.get_num_threads(), _usable_cores())))


@pytest.fixture(autouse=True)
def _zero_dropout_salt():
    """The host model of the kernels' dropout hash (ops.attn_dropout_keep_mask, helpers.row_mask) assumes the
    device-resident salt is 0; a test that captured a step (TrainStep.capture / step_cached install the salt and advance it
    per step) would otherwise change the masks of every masked-oracle test that runs after it in the same process."""
    yield
    if torch.cuda.is_available():
        from graphgps_amd import ops
        if ops._dropout_salt is not None:
            ops._dropout_salt.zero_()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


GRAPHORMER_GOLDEN = "graphormer_encoder_layer"      # encoder + GraphormerLayer fixture (own structure)
SIGNNET_GOLDEN = "signnet_encoder"                   # SignNet encoder fixture (own structure)
SAN_GOLDEN = "san_layers"                            # SANLayer / SAN2Layer fixture (own structure)
LAPPE_GOLDEN = "lappe_encoder"                       # LapPE encoder fixture (own structure)
CUSTOM_GNN_GOLDEN = "custom_gnn_layers"              # GatedGCNLayer(batch) / GINEConvLayer fixtures
AUX_GOLDEN = "aux_modules"                           # encoders / heads either side of the layers


def golden_names():
    """The GPSLayer fixtures (one layer, one batch each)."""
    return sorted(f[:-3] for f in os.listdir(GOLDEN_DIR)
                  if f.endswith(".pt") and f[:-3] not in (GRAPHORMER_GOLDEN, SIGNNET_GOLDEN, SAN_GOLDEN, LAPPE_GOLDEN, CUSTOM_GNN_GOLDEN, AUX_GOLDEN))


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)


class Tol:
    """Parity bar from BASELINE.json north_star: 1e-5 in fp32 on unit-variance activations;
    gradients relative to their max magnitude."""
    ACT = 1e-5
    GRAD_REL = 1e-5


def assert_close(a, b, tol, what, rel_to_max=False):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item() if a.numel() else 0.0
    # relative to max|ref| but never below 1: gradients that are mathematically zero (e.g. of a
    # bias feeding a BatchNorm) are pure rounding noise on both sides
    scale = max(b.abs().max().item(), 1.0) if (rel_to_max and b.numel()) else 1.0
    assert err / scale <= tol, f"{what}: max|d|={err:.3e} (scale {scale:.3e}) > {tol:.1e}"
    return err / scale


def assert_close_kink_tolerant(a, b, tol, what, max_outlier_row_frac=1e-3, min_scale=1.0,
                               min_allowed_rows=3, outlier_cap=5e-2):
    """Gradient parity at full BASELINE sizes.  With ~2e7 ReLU pre-activations per layer, a
    handful lie within fp32 rounding of 0 and land on different sides of the kink on the GPU
    and on the CPU; each such flip changes the gradient of ONE row (all its channels) by
    O(upstream grad) while moving the forward output by ~1e-7 (measured on MI355X: 2 rows of
    7,447 at P30/d=384, every other element within 7e-7).  So: all rows but a vanishing
    fraction (< max_outlier_row_frac, at least 3) must meet the relative tolerance; the report
    says how many did not.  Forward outputs never use this helper."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), min_scale)
    if a.dim() == 1:            # vectors: every element is its own "row"
        a, b = a.view(-1, 1), b.view(-1, 1)
    err = ((a - b).abs() / scale).view(a.shape[0], -1)
    bad_rows = (err > tol).any(dim=1)
    rows = int(bad_rows.sum())
    allowed = max(min_allowed_rows, int(max_outlier_row_frac * a.shape[0]))
    assert err.max().item() <= outlier_cap, f"{what}: outlier of {err.max().item():.2e} is not a kink-sized error"
    clean_max = err[~bad_rows].max().item() if rows < a.shape[0] else float("nan")
    assert rows <= allowed, (
        f"{what}: {rows} of {a.shape[0]} rows exceed {tol:.0e} (allowed {allowed}; max rel err "
        f"{err.max().item():.2e})")
    return clean_max, int((err > tol).sum()), rows


def fp32_grade(got, cpu32, ref64, min_scale=1.0):
    """Three-way error figures of one tensor: (rms|got - fp64|, rms|cpu32 - fp64|, max|got - fp64|, max|cpu32 - fp64|),
    all divided by max(max|fp64|, min_scale)."""
    g, c, r = (t.detach().double().cpu() for t in (got, cpu32, ref64))
    scale = max(r.abs().max().item() if r.numel() else 0.0, min_scale)
    eg, ec = (g - r).abs(), (c - r).abs()
    rms = lambda e: float((e * e).mean().sqrt()) / scale if e.numel() else 0.0
    mx = lambda e: float(e.max()) / scale if e.numel() else 0.0
    return rms(eg), rms(ec), mx(eg), mx(ec)


def assert_fp32_grade(got, cpu32, ref64, what, floor=1e-6, factor=3.0, min_scale=1.0):
    """The tolerance story of the deep comparisons (north_star: 1e-5 fp32): the CPU oracle and the HIP path are
    two different fp32 evaluation orders of the same function, and past a few BN-normalised layers -- or once a
    ReLU pre-activation sits within rounding of 0 -- they differ from each other by more than 1e-5 without either
    being wrong.  What CAN be asserted: against the fp64 evaluation of the same oracle, the HIP result is no
    further away than the reference's own fp32 arithmetic is.  RMS error (a kink flip is a rare, equally likely
    event on both sides; the max would compare two different random flips): rms|got - fp64| <= max(floor,
    factor * rms|cpu32 - fp64|), both relative to max|fp64|.  Returns the four figures of ``fp32_grade``."""
    rg, rc, mg, mc = fp32_grade(got, cpu32, ref64, min_scale)
    assert rg <= max(floor, factor * rc), (
        f"{what}: rms error vs fp64 {rg:.3e} (max {mg:.3e}) exceeds {factor} x the fp32 CPU oracle's own "
        f"{rc:.3e} (max {mc:.3e})")
    return rg, rc, mg, mc
