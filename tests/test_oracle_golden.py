"""CPU: pin the oracle (oracle/) against golden vectors captured from the imported reference
(tools/make_goldens.py).  Tolerances: blocks <= 1e-6 scale-relative (same ATen kernels, different
graph shape), DDIM tables bit-exact."""
import json

import numpy as np
import pytest
import torch

from conftest import load_npz, sub_sd, rel_err
from oracle import unet_oracle as U
from oracle import pqmf_oracle as Q
from oracle import sampling_oracle as S


@pytest.mark.parametrize("tag,last", [("rcb_diff", False), ("rcb_same", False), ("rcb_last", True)])
def test_res_conv_block(golden_blocks, tag, last):
    z = golden_blocks
    sd = sub_sd(z, f"{tag}/sd/")
    y = U.res_conv_block(sd, "", torch.from_numpy(z[f"{tag}/x"]), is_last=last)
    assert rel_err(y, torch.from_numpy(z[f"{tag}/y"])) <= 1e-6


@pytest.mark.parametrize("s", [8, 64, 256])
def test_self_attention(golden_blocks, s):
    z = golden_blocks
    sd = sub_sd(z, f"attn{s}/sd/")
    y = U.self_attention(sd, "", torch.from_numpy(z[f"attn{s}/x"]))
    assert rel_err(y, torch.from_numpy(z[f"attn{s}/y"])) <= 1e-6


@pytest.mark.parametrize("kern", ["linear", "cubic", "lanczos3"])
@pytest.mark.parametrize("L", [8, 16, 4096])
def test_resamplers(golden_blocks, kern, L):
    z = golden_blocks
    for d in ("down", "up"):
        key = f"{d}_{kern}_{L}/x"
        if key not in z.files:
            continue
        x = torch.from_numpy(z[key])
        taps = torch.from_numpy(z[f"{d}_{kern}/kernel"])
        y = U.downsample_fir(x, taps) if d == "down" else U.upsample_fir(x, taps)
        ref = torch.from_numpy(z[f"{d}_{kern}_{L}/y"])
        assert y.shape == ref.shape
        assert rel_err(y, ref) <= 1e-6
    # the stored buffers are the table (down) and 2*table (up)
    assert np.array_equal(z[f"down_{kern}/kernel"], np.asarray(U.RESAMPLE_KERNELS[kern], np.float32))
    assert np.array_equal(z[f"up_{kern}/kernel"], 2 * np.asarray(U.RESAMPLE_KERNELS[kern], np.float32))


def test_fourier_and_skip(golden_blocks):
    z = golden_blocks
    e = U.fourier_features(torch.from_numpy(z["fourier/weight"]), torch.from_numpy(z["fourier/t"]))
    assert torch.equal(e, torch.from_numpy(z["fourier/y"]))
    x = torch.from_numpy(z["skip/x"])
    assert torch.equal(torch.cat([x, x], 1), torch.from_numpy(z["skip/y"]))


@pytest.mark.parametrize("steps", [5, 10, 100])
def test_schedule_bit_exact(golden_blocks, steps):
    z = golden_blocks
    t, a, s = S.schedule(steps)
    assert np.array_equal(t.numpy(), z[f"sched{steps}/t"])
    assert np.array_equal(a.numpy(), z[f"sched{steps}/alpha"])
    assert np.array_equal(s.numpy(), z[f"sched{steps}/sigma"])
    if steps == 10:
        assert np.array_equal(S.alpha_sigma_to_t(a, s).numpy(), z["sched10/t_back"])


PQMF_CASES = [(1, 70, 32), (2, 70, 32), (2, 70, 4), (2, 70, 16), (2, 100, 32)]


@pytest.mark.parametrize("ch,att,M", PQMF_CASES)
def test_pqmf_design_and_transform(golden_pqmf, ch, att, M):
    z = golden_pqmf
    tag = f"pqmf_c{ch}_a{att}_m{M}"
    d = Q.design(att, M)
    # prototype design goes through scipy (version-dependent last digits): tight but not bitwise
    assert d["hk"].shape == tuple(z[f"{tag}/hk"].shape)
    assert rel_err(d["h"], torch.from_numpy(z[f"{tag}/h"])) <= 1e-6
    assert rel_err(d["hk"], torch.from_numpy(z[f"{tag}/hk"])) <= 1e-6
    assert rel_err(d["forward_w"], torch.from_numpy(z[f"{tag}/fw"])) <= 1e-6
    assert rel_err(d["inverse_w"], torch.from_numpy(z[f"{tag}/iw"])) <= 1e-6
    x = torch.from_numpy(z[f"{tag}/x"])
    fw, iw = torch.from_numpy(z[f"{tag}/fw"]), torch.from_numpy(z[f"{tag}/iw"])
    y = Q.analysis(x, fw, M)
    assert y.shape == tuple(z[f"{tag}/y"].shape) == (x.shape[0], ch * M, x.shape[2] // M)
    assert rel_err(y, torch.from_numpy(z[f"{tag}/y"])) <= 1e-6
    # anchor that does not go through cached_conv: reference polyphase_forward + reverse_half
    assert rel_err(y[:, :M], torch.from_numpy(z[f"{tag}/y_polyphase_ch0"])) <= 1e-5
    r = Q.synthesis(y, iw, M)
    assert r.shape == x.shape                                   # reference test/pqmf_test.py:9-26
    assert rel_err(r, torch.from_numpy(z[f"{tag}/z"])) <= 1e-6
    # near-perfect reconstruction delayed by M samples
    err = (r[..., M:] - x[..., :-M])[..., 8 * M:-8 * M]
    assert float(err.pow(2).mean().sqrt() / x.pow(2).mean().sqrt()) < 10 ** (-40 / 20)


TINY = ["tiny_d3.npz", "tiny_d5.npz", "tiny_d3_pqmf4.npz", "tiny_d4_noattn_nobias.npz",
        # learned resamplers (strides 2/4/1 and 3/8), kernel_size 7 and 3
        "tiny_learned_k7.npz", "tiny_learned_s3_k3.npz", "tiny_d3_k7_fir.npz"]


@pytest.mark.parametrize("name", TINY)
def test_tiny_models(name):
    z = load_npz(name)
    cfg = U.make_config(**json.loads(bytes(z["cfg"]).decode()))
    sd = sub_sd(z, "sd/")
    model = U.make_model_fn(sd, cfg)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    y = model(x, t)
    # whole-model forward: oneDNN picks different fp32 summation orders on different host CPUs (seen: 1.08e-6)
    assert rel_err(y, torch.from_numpy(z["y"])) <= 5e-6
    trace = []
    pred = S.ddim_sample(model, x, int(z["steps"]), 0, trace=trace)
    assert rel_err(torch.stack(trace), torch.from_numpy(z["v_steps"])) <= 1e-5
    assert rel_err(pred, torch.from_numpy(z["pred"])) <= 1e-5


def test_attention_level_rule():
    # n_attn_layers+2 levels, never level 1 (diffusion_decoder.py:85,96)
    assert U.attn_levels(U.make_config(n_attn_layers=4)) == list(range(9, 15))
    assert U.attn_levels(U.make_config()) == list(range(7, 15))
    assert U.attn_levels(U.make_config(depth=3, n_attn_layers=1, c_mults=[32, 32, 64])) == [2, 3]
    assert U.attn_levels(U.make_config(n_attn_layers=0)) == []


@pytest.mark.parametrize("name", ["tiny_d3_cond.npz", "tiny_d3_cond_samelen.npz", "tiny_d3_cond_aug0.npz"])
def test_conditioning_path(name):
    """cond interpolated to the signal length, optional noise-aug level embedding
    (decoders/diffusion_decoder.py:148-169), forward + 4-step DDIM with cond forwarded through **extra_args."""
    import json
    from oracle import unet_oracle as U, sampling_oracle as S
    z = load_npz(name)
    cfg = U.make_config(**json.loads(bytes(z["cfg"]).decode()))
    model = U.make_model_fn(sub_sd(z, "sd/"), cfg)
    x, t, cond = (torch.from_numpy(z[k]) for k in ("x", "t", "cond"))
    extra = dict(cond=cond)
    if float(z["aug_scale"]) >= 0:
        extra["cond_aug_scale"] = float(z["aug_scale"])
    assert rel_err(model(x, t, **extra), torch.from_numpy(z["y"])) <= 5e-6   # host-CPU dependent, as above
    assert rel_err(S.ddim_sample(model, x, int(z["steps"]), 0, **extra), torch.from_numpy(z["pred"])) <= 1e-5


# ---- fixtures of tools/make_goldens.py tier G -----------------------------------------------------
def _tiny_d3_oracle():
    import json
    from oracle import unet_oracle as U
    z = load_npz("tiny_d3.npz")
    sd = sub_sd(z, "sd/")
    g = load_npz("glue.npz")
    assert torch.equal(sd["net.0.main.0.weight"][:2, :2], torch.from_numpy(g["model_sd_check"]))    # same model
    return U.make_model_fn(sd, U.make_config(**json.loads(bytes(z["cfg"]).decode()))), g


def test_crash_schedule_vs_reference_fixture():
    """train_uncond.py:35-38, outputs recorded from the reference's own function definitions."""
    import audio_diffusion_amd as A
    from oracle import sampling_oracle as S
    g = load_npz("glue.npz")
    for steps in (5, 6, 10, 100, 250):
        want = torch.from_numpy(g[f"crash/t{steps}"])
        t = torch.linspace(1, 0, steps + 1)[:-1]
        # sin / atan2 of the host's libm: the last bit depends on the CPU the fixture was recorded on (2 ulp allowed);
        # the oracle and the product's host function must agree with each other exactly on any one host
        close = lambda a: torch.allclose(a, want, rtol=0, atol=2.4e-7)
        assert close(S.crash_schedule(t))
        assert close(A.get_crash_schedule(t))                                  # the product's host function
        assert torch.equal(A.get_crash_schedule(t), S.crash_schedule(t))
        assert torch.equal(A.step_coefficients(steps, 0, A.get_crash_schedule)[:, 0], A.get_crash_schedule(t))


def test_crash_sampler_and_diffae_oracle_vs_reference_fixture():
    """The reference's train_uncond.py sample() (crash-warped DDIM) and diffusion/inference.py sample() (encode once,
    decode per step) on the tiny_d3 model, against the oracle's restatements."""
    from oracle import sampling_oracle as S
    model, g = _tiny_d3_oracle()
    x = torch.from_numpy(g["crash/x"])
    assert rel_err(S.ddim_sample(model, x, 6, 0, crash=True), torch.from_numpy(g["crash/pred6"])) <= 1e-5

    class Toy:
        def encode(self, signal):
            return signal.mean(dim=2, keepdim=True)

        def decode(self, xx, tt, z_sem):
            return model(xx + 0.1 * z_sem, tt)

    got = S.diffae_sample(Toy(), torch.from_numpy(g["diffae/signal"]), 4, 0, torch.from_numpy(g["diffae/noise"]))
    assert rel_err(got, torch.from_numpy(g["diffae/pred4"])) <= 1e-5


@pytest.mark.parametrize("M", [4, 16, 32])
def test_base_pqmf_paths_and_cached_bank_cross_check(M):
    """pqmf.py:95-233 (PQMF, polyphase_* / classic_*): the oracle's restatements against the reference's outputs, and
    the cached bank the denoiser uses against them: CachedPQMF.forward == PQMF.forward exactly, CachedPQMF.inverse ==
    PQMF.inverse delayed by M samples exactly.  Neither base path touches cached_conv, so this pins the cached bank's
    sample values everywhere except the first M output samples of the inverse."""
    from oracle import pqmf_oracle as Q
    g = load_npz("glue.npz")
    d = Q.design(70, M)
    for tag in ("poly", "classic"):
        k = f"pqmf_base_m{M}_{tag}"
        x, y, z, hk = (torch.from_numpy(g[f"{k}/{n}"]) for n in ("x", "y", "z", "hk"))
        assert rel_err(d["hk"], hk) <= 1e-6
        assert rel_err(Q.base_forward(x, hk, tag == "poly"), y) <= 1e-6
        assert rel_err(Q.base_inverse(y, hk, tag == "poly"), z) <= 1e-6
    x, y, z = (torch.from_numpy(g[f"pqmf_base_m{M}_poly/{n}"]) for n in ("x", "y", "z"))
    assert rel_err(Q.analysis(x, d["forward_w"], M), y) <= 1e-6
    zc = Q.synthesis(y, d["inverse_w"], M)
    assert rel_err(zc[..., M:], z[..., :-M]) <= 1e-6
    assert rel_err(torch.from_numpy(g[f"pqmf_base_m{M}_classic/z"]), z) <= 1e-5          # the two base paths agree
