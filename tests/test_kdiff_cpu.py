"""SURVEY 8(f4) on the CPU: the k-diffusion oracle against fixtures generated from the reference
(tools/make_goldens.py tier K), constructor / state_dict parity of AudioDenoiserModel, host-side plan validation."""
import json

import pytest
import torch

from conftest import load_npz, sub_sd, rel_err

import audio_diffusion_amd as A
from audio_diffusion_amd import _lib
from audio_diffusion_amd.kdiffusion import _KPlan
from oracle import kdiff_oracle as K

CASES = ["kdiff_tiny.npz", "kdiff_attn128.npz", "kdiff_cond.npz"]


def unpack(z):
    cfg = json.loads(bytes(z["cfg"]).decode())
    extra = {k[len("extra/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("extra/")}
    return cfg, extra, torch.from_numpy(z["x"]), torch.from_numpy(z["sigma"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_goldens(name):
    z = load_npz(name)
    cfg, extra, x, sigma = unpack(z)
    sd = sub_sd(z, "sd/")
    f = K.make_model_fn(sd, K.make_config(**cfg))
    assert rel_err(f(x, sigma, **extra), torch.from_numpy(z["y"])) <= 1e-5
    assert rel_err(f(x, sigma, log_sigma=False, **extra), torch.from_numpy(z["y_nolog"])) <= 1e-5
    d = K.make_model_fn(sd, K.make_config(**cfg), sigma_data=float(z["sigma_data"]))
    assert rel_err(d(x, sigma, **extra), torch.from_numpy(z["denoised"])) <= 1e-5


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_ctor_parity_and_plan(name, dtype):
    z = load_npz(name)
    cfg, extra, x, sigma = unpack(z)
    torch.manual_seed(int(z["seed"]))
    m = A.AudioDenoiserModel(**cfg)
    ref_sd = sub_sd(z, "sd/")
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:                      # same seed => same init as the reference, bit for bit
        assert sd[k].shape == ref_sd[k].shape, k
        if k.startswith("proj_out."):
            assert not sd[k].any()    # zero init (decoders/diffusion_decoder.py:281-282); the fixture re-draws it
        elif k.startswith("mapping.") and k.endswith(".weight"):
            # nn.init.orthogonal_ (blocks.py:296-297) goes through LAPACK QR: same draws, last bits depend on the CPU
            assert torch.allclose(sd[k], ref_sd[k], rtol=0, atol=1e-5), k
        else:
            assert torch.equal(sd[k], ref_sd[k]), k
    m.load_state_dict(ref_sd, strict=True)
    cond_len = extra["unet_cond"].shape[2] if "unet_cond" in extra else 0
    for scaled in (False, True):
        plan = _KPlan(m, x.shape[0], x.shape[2], dtype, torch.device("cpu"), cond_len=cond_len,
                      has_mapping_cond="mapping_cond" in extra, scaled=scaled)
        plan.validate()
        kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
        n_ada = sum(isinstance(mod, A.blocks.AdaGN) for mod in m.modules())
        assert kinds.count(_lib.OP_GN_APPLY) == n_ada
        # statistics ride on the producing conv's epilogue; only the concat inputs of the up path need a stats pass
        assert kinds.count(_lib.OP_GN_STATS) == len(cfg["depths"]) - 1
        assert kinds.count(_lib.OP_LINEAR) == 3 + ("mapping_cond" in extra) and kinds.count(_lib.OP_FOURIER) == 1
    with pytest.raises(RuntimeError):
        m(x, sigma, **extra)          # CPU tensors: loud failure, no fallback


def test_errors():
    kw = dict(c_in=2, feats_in=32, depths=[1, 1], channels=[32, 32], self_attn_depths=[False, False], strides=[2])
    m = A.AudioDenoiserModel(**kw)
    with pytest.raises(RuntimeError, match="multiple of its stride"):
        m.check_input(torch.zeros(1, 2, 33))
    with pytest.raises(RuntimeError, match="channel mismatch"):
        m.check_input(torch.zeros(1, 3, 32))
    with pytest.raises(NotImplementedError):          # head dim 96 (c=96 -> 1 head): only 32 and 64 are built
        A.AudioDenoiserModel(c_in=2, feats_in=32, depths=[1], channels=[96], self_attn_depths=[True], strides=[])
    with pytest.raises(NotImplementedError):
        A.Denoiser(m).loss()
    c_skip, c_out, c_in = A.Denoiser(m, sigma_data=0.5).get_scalings(torch.tensor([0.5]))
    assert abs(float(c_skip) - 0.5) < 1e-7 and abs(float(c_out) - 0.25 / 0.5 ** 0.5) < 1e-6
    assert abs(float(c_in) - 1 / 0.5 ** 0.5) < 1e-6


def test_karras_samplers_match_reference_goldens():
    """get_sigmas_* / sample_heun / sample_lms (host loops) around the ORACLE denoiser on the CPU reproduce what the
    reference's samplers gave around the reference's Denoiser (fixture kdiff_tiny.npz, batch 1)."""
    z = load_npz("kdiff_tiny.npz")
    cfg, _, _, _ = unpack(z)
    den = K.make_model_fn(sub_sd(z, "sd/"), K.make_config(**cfg), sigma_data=float(z["sigma_data"]))
    sig_k, sig_e = A.get_sigmas_karras(6, 1e-2, 3.0, rho=9.), A.get_sigmas_exponential(5, 1e-2, 3.0)
    assert torch.equal(sig_k, torch.from_numpy(z["samp/sigmas_karras"]))
    assert torch.equal(sig_e, torch.from_numpy(z["samp/sigmas_exp"]))
    x1 = torch.from_numpy(z["samp/x"])
    assert rel_err(A.sample_heun(den, x1, sig_k), torch.from_numpy(z["samp/heun"])) <= 1e-5
    assert rel_err(A.sample_heun(den, x1, sig_e, second_order=False), torch.from_numpy(z["samp/euler"])) <= 1e-5
    assert rel_err(A.sample_lms(den, x1, sig_k, order=3), torch.from_numpy(z["samp/lms"])) <= 1e-5
    seen = []
    A.sample_heun(den, x1, sig_e, callback=lambda d: seen.append(int(d["i"])), s_churn=1.0)
    assert seen == [0, 1, 2, 3, 4]
