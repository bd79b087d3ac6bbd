import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_names():
    return sorted(f[:-3] for f in os.listdir(GOLDEN_DIR) if f.endswith(".pt"))


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
    scale = max(b.abs().max().item(), 1e-30) if (rel_to_max and b.numel()) else 1.0
    assert err / scale <= tol, f"{what}: max|d|={err:.3e} (scale {scale:.3e}) > {tol:.1e}"
    return err / scale
