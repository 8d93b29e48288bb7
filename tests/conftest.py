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


class _RoundedConvF:
    """Stand-in for ``torch.nn.functional`` inside the oracle: every convolution sees its activation and weight
    rounded to ``dtype`` (what feeding a 16-bit MFMA costs at the very least); everything else stays fp32."""

    def __init__(self, dtype):
        import torch.nn.functional as F
        self._F, self._dt = F, dtype

    def __getattr__(self, name):
        return getattr(self._F, name)

    def _r(self, v):
        return v.to(self._dt).to(torch.float32)

    def conv1d(self, x, w, *a, **k):
        return self._F.conv1d(self._r(x), self._r(w), *a, **k)

    def conv_transpose1d(self, x, w, *a, **k):
        return self._F.conv_transpose1d(self._r(x), self._r(w), *a, **k)


def rounding_floor_of(oracle_module, call, dtype):
    """Error a reduced-precision path cannot go below: ``call()`` (a CPU oracle evaluation) with ONLY the MFMA
    operands (conv inputs and weights) rounded to ``dtype``, against the same call in plain fp32.  Returns
    (scale-relative max error, relative L2)."""
    ref = call()
    saved = oracle_module.F
    oracle_module.F = _RoundedConvF(dtype)
    try:
        got = call()
    finally:
        oracle_module.F = saved
    return rel_err(got, ref), float((got.double() - ref.double()).norm() / ref.double().norm())


def rounding_floor(sd, cfg, x, t, dtype, **extra):
    """rounding_floor_of for the DiffusionAttnUnet1D oracle."""
    from oracle import unet_oracle as U
    fn = U.make_model_fn(sd, U.make_config(**cfg))
    return rounding_floor_of(U, lambda: fn(x, t, **extra), dtype)


def bf16_bars(floor):
    """(max-rel bar, L2 bar) a bf16 result is held to, from its operand-rounding floor: north_star's 1e-2 where the
    format allows it, else 3 x the floor (16-bit storage of every activation on top of the operand rounding)."""
    return max(1e-2, 3 * floor[0]), max(1e-2, 3 * floor[1])
