import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    """numpy.load with allow_pickle=False (default): fixtures are plain arrays."""
    return np.load(os.path.join(GOLDEN, name))


def sub_sd(z, prefix):
    """Extract a state_dict stored under ``prefix`` from an npz as CPU tensors."""
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden_blocks():
    return load_npz("blocks.npz")


@pytest.fixture(scope="session")
def golden_pqmf():
    return load_npz("pqmf.npz")


def rel_err(a, b):
    """max|a-b| / max|b|  (scale-relative; the tolerance metric used across the parity tests)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def snr_db(x, ref):
    x = x.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float(10 * torch.log10(ref.pow(2).sum() / (x - ref).pow(2).sum().clamp_min(1e-300)))
