"""Where the reduced-precision bars of the GPU tests come from (CPU, seconds).

north_star asks 1e-2 per denoiser call "in bf16".  Rounding ONLY the MFMA operands (conv activations + weights) to
bf16 inside the fp32 CPU oracle - no 16-bit storage of GroupNorm outputs, no 16-bit residual stream, exact fp32
accumulation - already costs about that much on these random-init models, so no bf16 implementation can be held to
1e-2 on the scale-relative max metric (conftest.rel_err: max|a-b| / max|b|).  IEEE half (the reference's own autocast
dtype, diffusion/sampling.py:30) has three more mantissa bits and sits an order of magnitude below the bound; it is
the dtype bench.py headlines.  The GPU tests hold
    f16 : <= 1e-2 (north_star's bound as written),
    bf16: <= max(1e-2, 3 x the floor computed here for the same model and input), never a bare constant.
"""
import json

import pytest
import torch

from conftest import load_npz, sub_sd, rounding_floor

MODELS = ["tiny_d3.npz", "tiny_d5.npz", "tiny_d4_noattn_nobias.npz"]


@pytest.mark.parametrize("name", MODELS)
def test_bf16_operand_rounding_alone_is_at_the_1e2_bound(name):
    z = load_npz(name)
    cfg = json.loads(bytes(z["cfg"]).decode())
    sd = sub_sd(z, "sd/")
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    mx, l2 = rounding_floor(sd, cfg, x, t, torch.bfloat16)
    hx, hl2 = rounding_floor(sd, cfg, x, t, torch.float16)
    print(f"{name}: bf16 floor max {mx:.2e} L2 {l2:.2e}; f16 floor max {hx:.2e} L2 {hl2:.2e}")
    assert mx >= 8e-3, "bf16 operand rounding alone already sits at north_star's 1e-2 (measured 1.0-1.9e-2)"
    assert l2 >= 2e-3
    assert hx <= 3e-3 and hl2 <= 3e-3, "f16 operand rounding leaves most of the 1e-2 as margin"
    assert mx >= 4 * hx                       # three mantissa bits = 8x; at least 4x shows up end to end


def test_floor_helper_is_exact_for_fp32():
    z = load_npz("tiny_d3.npz")
    cfg = json.loads(bytes(z["cfg"]).decode())
    mx, l2 = rounding_floor(sub_sd(z, "sd/"), cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.float32)
    assert mx == 0.0 and l2 == 0.0
