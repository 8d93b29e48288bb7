"""SURVEY 8(f4), SoundStream-XL decoder on the CPU: oracle vs fixtures generated from the reference
(tools/make_goldens.py tier S), constructor / state_dict parity, host-side plan validation, weight fold."""
import json

import pytest
import torch
import torch.nn.functional as F

from conftest import load_npz, sub_sd, rel_err

from audio_diffusion_amd import _lib
from audio_diffusion_amd.blocks import fold_up_weight, fold_down_causal_weight
from audio_diffusion_amd.soundstream import SoundStreamXLDecoder, SoundStreamXLEncoder, _SSPlan
from oracle import soundstream_oracle as O

CASES = ["sstream_tiny.npz", "sstream_s8.npz"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_goldens(name):
    z = load_npz(name)
    cfg = json.loads(bytes(z["cfg"]).decode())
    y = O.decoder_forward(sub_sd(z, "sd/"), cfg["strides"], torch.from_numpy(z["z"]))
    assert y.shape == z["y"].shape
    assert rel_err(y, torch.from_numpy(z["y"])) <= 1e-5


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_ctor_parity_and_plan(name, dtype):
    z = load_npz(name)
    cfg = json.loads(bytes(z["cfg"]).decode())
    torch.manual_seed(int(z["seed"]))
    m = SoundStreamXLDecoder(**cfg)
    ref_sd = sub_sd(z, "sd/")
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:
        assert torch.equal(sd[k], ref_sd[k]), k          # same seed => same init as the reference, bit for bit
    B, _, L = z["z"].shape
    plan = _SSPlan(m, B, L, dtype, torch.device("cpu"))
    plan.validate()
    kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
    n_blocks = len(cfg["strides"])
    assert kinds.count(_lib.OP_CONV) == 2 + n_blocks * 13 and kinds.count(_lib.OP_DEPTH_TO_SPACE) == n_blocks
    with pytest.raises(RuntimeError):
        m(torch.from_numpy(z["z"]))                      # CPU tensors: loud failure, no fallback


@pytest.mark.parametrize("f", [2, 3, 5, 8])
def test_causal_transpose_fold_matches_torch(f):
    """CausalConvTranspose1d(k=2f, stride f)[..., :-f] == 3-tap conv (taps x[j-1], x[j], 0) producing f phases."""
    torch.manual_seed(f)
    c, co, L = 5, 4, 11
    wt, b, x = torch.randn(c, co, 2 * f, dtype=torch.float64), torch.randn(co, dtype=torch.float64), torch.randn(2, c, L, dtype=torch.float64)
    want = F.conv_transpose1d(x, wt, b, stride=f)[..., :-f]
    w3, b3 = fold_up_weight(wt, b, f, pad=0)
    zz = F.conv1d(x, w3, b3, padding=1)                                  # [2, f*co, L], channel = s*co + o
    got = zz.view(2, f, co, L).permute(0, 2, 3, 1).reshape(2, co, L * f)
    assert (got - want).abs().max() < 1e-12
    assert not w3[:, :, 2].any()                                         # causal: nothing reads x[j+1]


@pytest.mark.parametrize("name", ["sstream_enc_tiny.npz", "sstream_enc_s8.npz"])
def test_encoder_oracle_ctor_and_plan(name):
    z = load_npz(name)
    cfg = json.loads(bytes(z["cfg"]).decode())
    ref_sd = sub_sd(z, "sd/")
    y = O.encoder_forward(ref_sd, cfg["strides"], torch.from_numpy(z["x"]))
    assert y.shape == z["y"].shape and rel_err(y, torch.from_numpy(z["y"])) <= 1e-5
    torch.manual_seed(int(z["seed"]))
    m = SoundStreamXLEncoder(**cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:
        assert torch.equal(sd[k], ref_sd[k]), k
    B, _, T = z["x"].shape
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        plan = _SSPlan(m, B, T, dtype, torch.device("cpu"))
        plan.validate()
        kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
        assert kinds.count(_lib.OP_CONV) == 2 + len(cfg["strides"]) * 13 and kinds.count(_lib.OP_SPACE_TO_DEPTH) == len(cfg["strides"])
    with pytest.raises(RuntimeError):
        m(torch.from_numpy(z["x"]))                      # CPU tensors: loud failure, no fallback


@pytest.mark.parametrize("f", [2, 3, 5, 8])
def test_strided_causal_fold_matches_torch(f):
    """CausalConv1d(k=2f, stride f) == causal 3-tap conv over the phase-folded input."""
    torch.manual_seed(f)
    c, co, L = 5, 4, 11 * f
    w, x = torch.randn(co, c, 2 * f, dtype=torch.float64), torch.randn(2, c, L, dtype=torch.float64)
    want = F.conv1d(F.pad(x, [2 * f - 1, 0]), w, stride=f)
    xf = x.view(2, c, L // f, f).permute(0, 3, 1, 2).reshape(2, f * c, L // f)          # x'[p*C + c][m] = x[c][m*f + p]
    got = F.conv1d(F.pad(xf, [2, 0]), fold_down_causal_weight(w, f))
    assert got.shape == want.shape and (got - want).abs().max() < 1e-12


def test_ctor_errors():
    with pytest.raises(NotImplementedError):
        SoundStreamXLDecoder(capacity=16)
