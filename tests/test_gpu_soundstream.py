"""SURVEY 8(f4), SoundStream-XL decoder on the GPU against the reference's goldens (tests/golden/sstream_*.npz) and,
at the default architecture, against the CPU oracle; plus the dilated causal conv kernel through the C ABI.
Tolerances: fp32 1e-4 (max error / max |ref|), f16 1e-2 (north_star's reduced-precision bound); bf16 bars derived per test from the
operand-rounding floor (conftest.bf16_bars)."""
import ctypes
import json

import pytest
import torch
import torch.nn.functional as F

from conftest import load_npz, sub_sd, rel_err, rounding_floor_of, bf16_bars
from oracle import soundstream_oracle as O
from test_gpu_kdiff import to_cpl, from_cpl

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["sstream_tiny.npz", "sstream_s8.npz"])
def test_decoder_vs_reference_golden(name):
    from audio_diffusion_amd.soundstream import SoundStreamXLDecoder
    z = load_npz(name)
    m = SoundStreamXLDecoder(**json.loads(bytes(z["cfg"]).decode()))
    m.load_state_dict(sub_sd(z, "sd/"), strict=True)
    m = m.to(DEV).eval()
    lat, ref = torch.from_numpy(z["z"]).to(DEV), torch.from_numpy(z["y"])
    m.compute_dtype = torch.float32
    y = m(lat)
    assert y.shape == ref.shape and rel_err(y, ref) <= 1e-4
    assert torch.equal(y, m(lat)), "reruns must be bit-identical"
    assert torch.equal(m(lat[1:2].contiguous())[0], y[1]), "a sample's result must not depend on its batch"
    m.compute_dtype = torch.bfloat16
    yb = m(lat).cpu()
    cfg = json.loads(bytes(z["cfg"]).decode())
    sd = sub_sd(z, "sd/")
    bl2 = bf16_bars(rounding_floor_of(O, lambda: O.decoder_forward(sd, cfg["strides"], lat.cpu()), torch.bfloat16))[1]
    assert float((yb - ref).norm() / ref.norm()) <= bl2
    m.compute_dtype = torch.float16
    assert rel_err(m(lat), ref) <= 1e-2


@pytest.mark.parametrize("name", ["sstream_enc_tiny.npz", "sstream_enc_s8.npz"])
def test_encoder_vs_reference_golden(name):
    from audio_diffusion_amd.soundstream import SoundStreamXLEncoder
    z = load_npz(name)
    m = SoundStreamXLEncoder(**json.loads(bytes(z["cfg"]).decode()))
    m.load_state_dict(sub_sd(z, "sd/"), strict=True)
    m = m.to(DEV).eval()
    x, ref = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"])
    m.compute_dtype = torch.float32
    y = m(x)
    assert y.shape == ref.shape and rel_err(y, ref) <= 1e-4
    assert torch.equal(y, m(x)) and torch.equal(m(x[1:2].contiguous())[0], y[1])
    m.compute_dtype = torch.bfloat16
    cfg = json.loads(bytes(z["cfg"]).decode())
    sd = sub_sd(z, "sd/")
    bl2 = bf16_bars(rounding_floor_of(O, lambda: O.encoder_forward(sd, cfg["strides"], x.cpu()), torch.bfloat16))[1]
    assert float((m(x).cpu() - ref).norm() / ref.norm()) <= bl2
    m.compute_dtype = torch.float16
    assert rel_err(m(x), ref) <= 1e-2
    with pytest.raises(NotImplementedError):
        m(x[..., :-1])                                   # length not a multiple of the total stride


def test_default_autoencoder_roundtrip_shapes_vs_oracle():
    """Default encoder (x1280 down) then default decoder on 3 latent frames, each half against the CPU oracle."""
    from audio_diffusion_amd.soundstream import SoundStreamXLDecoder, SoundStreamXLEncoder
    from oracle import soundstream_oracle as O
    torch.manual_seed(5)
    enc, dec = SoundStreamXLEncoder().eval(), SoundStreamXLDecoder().eval()
    x = torch.randn(1, 2, 3 * 1280)
    want_z = O.encoder_forward(enc.state_dict(), enc.strides, x)
    want_y = O.decoder_forward(dec.state_dict(), dec.strides, want_z)
    enc, dec = enc.to(DEV), dec.to(DEV)
    enc.compute_dtype = dec.compute_dtype = torch.float32
    zz = enc(x.to(DEV))
    assert zz.shape == (1, 128, 3) and rel_err(zz, want_z) <= 1e-4
    assert rel_err(dec(zz), want_y) <= 1e-4


def test_default_architecture_vs_oracle():
    """The reference's default decoder (8.9 M parameters, x1280 upsampling) on 6 latent frames."""
    from audio_diffusion_amd.soundstream import SoundStreamXLDecoder
    from oracle import soundstream_oracle as O
    torch.manual_seed(4)
    m = SoundStreamXLDecoder().eval()
    lat = torch.randn(2, 128, 6)
    want = O.decoder_forward(m.state_dict(), m.strides, lat)
    m = m.to(DEV)
    m.compute_dtype = torch.float32
    got = m(lat.to(DEV))
    assert got.shape == (2, 2, 6 * 1280) and rel_err(got, want) <= 1e-4
    m.compute_dtype = torch.bfloat16
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    bl2 = bf16_bars(rounding_floor_of(O, lambda: O.decoder_forward(sd, m.strides, lat), torch.bfloat16))[1]
    assert float((m(lat.to(DEV)).cpu() - want).norm() / want.norm()) <= bl2
    m.compute_dtype = torch.float16
    assert rel_err(m(lat.to(DEV)), want) <= 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,Co,L,dil", [(32, 32, 300, 1), (32, 32, 1000, 9), (64, 64, 70, 3), (64, 128, 700, 9),
                                         (128, 128, 3000, 3), (256, 256, 40, 9), (128, 128, 200, 1)])
def test_dilated_causal_conv_elu(dtype, C, Co, L, dil):
    """adk_run_plan with ADK_FLAG_CAUSAL | ADK_FLAG_ELU and a residual, all tile shapes (simple, k-split, pipelined)."""
    from audio_diffusion_amd import _lib
    lib = _lib.load()
    torch.manual_seed(C + L + dil)
    B, P, cb = 2, 54, 16 if dtype == torch.float32 else 32
    adt = _lib.adk_dtype(dtype)
    x = torch.randn(B, C, L, device=DEV)
    w = torch.randn(Co, C, 7, device=DEV) * (C * 7) ** -0.5
    bias, res = torch.randn(Co, device=DEV), torch.randn(B, Co, L, device=DEV)
    bn = lib.adk_conv_cotile(adt, Co, L)
    cop = (Co + bn - 1) // bn * bn
    wp = torch.zeros(cop, C, 7, device=DEV)
    wp[:Co] = w
    wp = wp.view(cop, C // cb, cb, 7).permute(1, 3, 0, 2).contiguous().to(dtype)
    bp = torch.zeros(cop, device=DEV)
    bp[:Co] = bias
    xs, rs = to_cpl(x, P, cb, dtype), to_cpl(res, P, cb, dtype)
    out = torch.full_like(rs, 5.0)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, adt, _lib.FLAG_CAUSAL | _lib.FLAG_ELU
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps, op.i0 = B, L, P, C, Co, cop, 7, dil
    op.x, op.w, op.bias, op.residual, op.y = xs.data_ptr(), wp.data_ptr(), bp.data_ptr(), rs.data_ptr(), out.data_ptr()
    rc = lib.adk_run_plan(ctypes.byref(op), 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.adk_last_error()
    xin, wq = from_cpl(xs, C, P), wp.float().permute(2, 0, 3, 1).reshape(cop, C, 7)[:Co]
    want = F.elu(F.conv1d(F.pad(xin, [6 * dil, 0]), wq, bias, dilation=dil) + from_cpl(rs, Co, P))
    assert rel_err(from_cpl(out, Co, P), want) <= (1e-4 if dtype == torch.float32 else 2e-2)
    assert not out[:, :, :P].any() and not out[:, :, -P:].any(), "halo rows must be zero"
    op.P = 6 * dil - 1                                   # too little left halo for the causal padding: rejected
    assert lib.adk_validate_plan(ctypes.byref(op), 1) != 0
