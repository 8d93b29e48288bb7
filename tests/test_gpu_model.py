"""GPU parity of the whole denoiser and sampler against golden vectors captured from the reference.

Bars (north_star): per-denoiser-call output within rtol 1e-4 in fp32 (scale-relative max error,
conftest.rel_err) and final eta=0 audio at SNR >= 60 dB in fp32.

Reduced precision: north_star asks 1e-2.  f16 (the reference's own autocast dtype, diffusion/sampling.py:30, and
the dtype bench.py headlines) is held to exactly that.  bf16 cannot be: rounding ONLY the MFMA operands to bf16 in
the CPU oracle already costs 1.0-1.9e-2 (tests/test_dtype_floor_cpu.py derives it), so every bf16 bar below is
max(1e-2, 3 x that floor computed in the test for the same model and input) - never a bare constant.
"""
import json

import pytest
import torch

from conftest import load_npz, sub_sd, rel_err, snr_db, rounding_floor

pytestmark = pytest.mark.gpu

TINY = ["tiny_d3.npz", "tiny_d5.npz", "tiny_d3_pqmf4.npz", "tiny_d4_noattn_nobias.npz",
        # learned resamplers (strides 2/4/1 and 3/8), kernel_size 7 and 3
        "tiny_learned_k7.npz", "tiny_learned_s3_k3.npz", "tiny_d3_k7_fir.npz"]


def build(z, dev):
    import audio_diffusion_amd as A
    cfg = json.loads(bytes(z["cfg"]).decode())
    m = A.DiffusionAttnUnet1D(**cfg)
    missing = m.load_state_dict(sub_sd(z, "sd/"), strict=True)
    return m.to(dev).eval()


@pytest.mark.parametrize("name", TINY)
def test_tiny_forward_fp32(name):
    dev = torch.device("cuda:0")
    z = load_npz(name)
    m = build(z, dev)
    m.compute_dtype = torch.float32
    y = m(torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev))
    assert y.dtype == torch.float32 and y.shape == tuple(z["y"].shape)
    assert rel_err(y, torch.from_numpy(z["y"])) <= 1e-4
    # deterministic: a second call gives the same bits
    y2 = m(torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev))
    assert torch.equal(y, y2)


@pytest.mark.parametrize("name", ["tiny_d3.npz", "tiny_d3_pqmf4.npz", "tiny_d3_k7_fir.npz"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_last_block_joint_conv_matches_two_launch_form(name, dtype):
    """The last ResConvBlock's second conv and its 1x1 skip conv as ONE conv over the joint buffer [h1 | x]
    (unet.py last_block_joint; reference blocks/blocks.py:66-76, is_last) against the two-launch form of the same plan:
    same operands, only the fp32 summation order differs (skip terms join the accumulator instead of being added as an
    f32 residual)."""
    dev = torch.device("cuda:0")
    z = load_npz(name)
    x, t = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev)
    outs = []
    for joint in (True, False):
        m = build(z, dev)
        m.compute_dtype = dtype
        m.fold_last_skip = joint
        outs.append(m(x, t))
        plan = next(iter(m._plans.values()))
        n_conv = sum(1 for i in range(plan.n_ops) if plan.ops[i].kind == 1)
        outs.append(n_conv)
    assert outs[1] == outs[3] - 1, "the joint form must save exactly one conv launch"
    assert rel_err(outs[0], outs[2]) <= 2e-6
    assert rel_err(outs[0], torch.from_numpy(z["y"])) <= (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("name", TINY)
def test_tiny_forward_bf16(name):
    dev = torch.device("cuda:0")
    z = load_npz(name)
    m = build(z, dev)
    m.compute_dtype = torch.bfloat16
    y = m(torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev))
    ref = torch.from_numpy(z["y"])
    cfg = json.loads(bytes(z["cfg"]).decode())
    fmax, fl2 = rounding_floor(sub_sd(z, "sd/"), cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.bfloat16)
    l2 = float((y.cpu().double() - ref.double()).norm() / ref.double().norm())
    print(f"{name}: bf16 max-rel {rel_err(y, ref):.2e} (operand-rounding floor {fmax:.2e}), L2 {l2:.2e} (floor {fl2:.2e})")
    assert rel_err(y, ref) <= max(1e-2, 3 * fmax)
    assert l2 <= max(1e-2, 3 * fl2)


@pytest.mark.parametrize("name", TINY)
def test_tiny_forward_f16(name):
    """IEEE half (ADK_F16), the dtype of the reference's own autocast region (diffusion/sampling.py:30): three more
    mantissa bits than bf16, so the north_star's reduced-precision bound of 1e-2 per denoiser call holds with margin."""
    dev = torch.device("cuda:0")
    z = load_npz(name)
    m = build(z, dev)
    m.compute_dtype = torch.float16
    x, t = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev)
    y = m(x, t)
    ref = torch.from_numpy(z["y"])
    assert y.dtype == torch.float32 and torch.isfinite(y).all()
    assert rel_err(y, ref) <= 1e-2                       # measured 1.8e-3 .. 3.5e-3 (profiles/r01/v9_dtype_error.txt)
    assert float((y.cpu().double() - ref.double()).norm() / ref.double().norm()) <= 5e-3
    assert torch.equal(y, m(x, t))                       # deterministic
    m.compute_dtype = None                               # under the reference's autocast the dtype follows the region
    with torch.autocast("cuda", dtype=torch.float16):
        assert torch.equal(m(x, t), y)


@pytest.mark.parametrize("name", TINY)
@pytest.mark.parametrize("graph", [False, True])
def test_tiny_sample_fp32(name, graph):
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z = load_npz(name)
    m = build(z, dev)
    m.compute_dtype = torch.float32
    m.use_graph = graph
    x = torch.from_numpy(z["x"]).to(dev)
    x_before = x.clone()
    pred = A.sample(m, x, int(z["steps"]), 0)
    assert torch.equal(x, x_before), "sample() must not modify its input"
    ref = torch.from_numpy(z["pred"])
    assert snr_db(pred, ref) >= 60.0
    assert rel_err(pred, ref) <= 1e-4
    pred2 = A.sample(m, x, int(z["steps"]), 0)      # cached sampler / graph replay: same bits
    assert torch.equal(pred, pred2)


def test_generic_sampler_matches_fused_and_eta_runs():
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z = load_npz("tiny_d3.npz")
    m = build(z, dev)
    m.compute_dtype = torch.float32
    x = torch.from_numpy(z["x"]).to(dev)
    fused = A.sample(m, x, 5, 0)
    generic = A.sample(lambda xx, tt: m(xx, tt), x, 5, 0)          # any callable works
    assert torch.equal(fused, generic)
    torch.manual_seed(0)
    a = A.sample(m, x, 5, 0.5)
    torch.manual_seed(0)
    b = A.sample(m, x, 5, 0.5)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert not torch.equal(a, fused)


def test_input_validation():
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z = load_npz("tiny_d3.npz")
    m = build(z, dev)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 8, device=dev), torch.zeros(1, device=dev))       # too short for reflect pad
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 1024, device=dev), torch.zeros(1, device=dev))    # wrong channel count
    with pytest.raises(RuntimeError):
        m.cpu()(torch.zeros(1, 2, 1024), torch.zeros(1))                      # no CPU path


def test_full_model_first_call_and_10_steps():
    """Config 1 of BASELINE.json: the 221M-parameter uncond model, seeds 0/1, B=1, 10-step DDIM."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z = load_npz("full_seed0_steps10.npz")
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
    torch.manual_seed(1)
    x = torch.randn(1, 2, 65536).to(dev)
    m.compute_dtype = torch.float32
    v0 = m(x, torch.ones(1, device=dev))
    ref_v0 = torch.from_numpy(z["v0"])
    assert rel_err(v0, ref_v0) <= 1e-4
    pred = A.sample(m, x, 10, 0)
    ref = torch.from_numpy(z["pred"])
    assert snr_db(pred, ref) >= 60.0
    m.compute_dtype = torch.float16                      # the reference's autocast dtype and bench.py's headline dtype
    v0h = m(x, torch.ones(1, device=dev))
    l2h = float((v0h.cpu().double() - ref_v0.double()).norm() / ref_v0.double().norm())
    assert rel_err(v0h, ref_v0) <= 5e-3 and l2h <= 5e-3  # north_star: 1e-2; measured 1.9e-3
    predh = A.sample(m, x, 10, 0)
    assert torch.isfinite(predh).all() and snr_db(predh, ref) >= 60.0      # measured 79 dB: the fp32 bar, in half
    # the DROP-IN DEFAULT: sample(model, x, steps, eta) with nothing configured runs the reference's autocast dtype
    # (diffusion/sampling.py:30-31) and meets north_star's 1e-2 per denoiser call.  A 1-step loop returns
    # pred = x * alpha_0 - v * sigma_0 with alpha_0 = cos(pi/2) ~ -4e-8, sigma_0 = 1: minus the first call's v.
    m.compute_dtype = None
    assert m.resolve_dtype(in_sampler=True) == torch.float16
    v_default = -A.sample(m, x, 1, 0)
    assert rel_err(v_default, ref_v0) <= 1e-2
    pred_default = A.sample(m, x, 10, 0)
    assert torch.equal(pred_default, predh) and snr_db(pred_default, ref) >= 60.0
    with torch.autocast("cuda", dtype=torch.bfloat16):                      # an explicit autocast region still decides
        assert m.resolve_dtype(in_sampler=True) == torch.bfloat16
    # bf16: bar derived from the operand-rounding floor of THIS model and input (CPU oracle, two calls)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    fmax, fl2 = rounding_floor(sd, dict(n_attn_layers=4), x.cpu(), torch.ones(1), torch.bfloat16)
    m.compute_dtype = torch.bfloat16
    v0b = m(x, torch.ones(1, device=dev))
    l2b = float((v0b.cpu().double() - ref_v0.double()).norm() / ref_v0.double().norm())
    predb = A.sample(m, x, 10, 0)
    print(f"full model bf16: max-rel {rel_err(v0b, ref_v0):.2e} (floor {fmax:.2e}), L2 {l2b:.2e} (floor {fl2:.2e}), "
          f"10-step SNR {snr_db(predb, ref):.1f} dB; f16: {rel_err(v0h, ref_v0):.2e}, {snr_db(predh, ref):.1f} dB")
    assert rel_err(v0b, ref_v0) <= max(1e-2, 3 * fmax) and l2b <= max(1e-2, 3 * fl2)
    assert snr_db(predb, ref) >= 50.0                    # measured 61 dB; 60 dB is the fp32 bar


def test_full_model_calls_identical_at_batch_16():
    """The configs[1] model at the bench batch (every long level cycles 8 generations of blocks through the GPU): five
    denoiser calls on the same input must agree bit for bit - a whole-sample difference here is how a missing MFMA wait
    state in the fused conv + GroupNorm kernel showed (tests/test_gpu_kernels.py::test_fused_conv_reruns_identical_...)."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
    m.compute_dtype = torch.float16
    x = torch.randn(16, 2, 65536, device=dev)
    t = torch.full((16,), 0.5, device=dev)
    with torch.no_grad():
        ref = m(x, t).clone()
        for _ in range(4):
            assert torch.equal(m(x, t), ref)
    assert torch.isfinite(ref).all()
    plan = next(iter(m._plans.values()))
    assert plan.gn_error() == 0


def test_crash_schedule_sampler_and_int16_output_stage():
    """the training demos' sampler (train_uncond.py:45-86: crash-warped timesteps) against the output of
    the reference's own function definitions (tests/golden/glue.npz, tools/make_goldens.py tier G), and the int16
    output stage (train_uncond.py:182-187) bit-exact against torch."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z, g = load_npz("tiny_d3.npz"), load_npz("glue.npz")
    m = build(z, dev)                                        # glue.npz was recorded on this very model (seed 0)
    m.compute_dtype = torch.float32
    x = torch.from_numpy(g["crash/x"])
    want = torch.from_numpy(g["crash/pred6"])
    got = A.sample_crash(m, x.to(dev), 6, 0)
    assert rel_err(got, want) <= 1e-4 and snr_db(got, want) >= 60.0
    assert not torch.allclose(got, A.sample(m, x.to(dev), 6, 0), atol=1e-3)      # the warp matters

    fakes = torch.randn(3, 2, 1001, device=dev) * 0.8                                   # odd length: scalar tail
    fakes[0, 0, :4] = torch.tensor([1.5, -1.5, 1.0, -1.0], device=dev)
    pcm = A.audio_to_int16(fakes)
    ref = fakes.cpu().permute(1, 0, 2).reshape(2, -1).clamp(-1, 1).mul(32767).to(torch.int16)
    assert pcm.dtype == torch.int16 and torch.equal(pcm.cpu(), ref)
    big = torch.randn(4, 2, 65536, device=dev)
    assert torch.equal(A.audio_to_int16(big).cpu(), big.cpu().permute(1, 0, 2).reshape(2, -1).clamp(-1, 1).mul(32767).to(torch.int16))


def test_diffae_sampler_variant(monkeypatch):
    """diffusion/inference.py:14-57: encode once, start from fresh noise, decode(x, t, z_sem) per step - against the
    output of the reference's own sample() on a toy encode/decode object (glue.npz).  The reference's start noise is
    a CPU randn_like under seed 11; the fixture carries that draw and the test feeds it through randn_like."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z, g = load_npz("tiny_d3.npz"), load_npz("glue.npz")
    m = build(z, dev)
    m.compute_dtype = torch.float32

    class DiffAE:
        def encode(self, signal):
            return signal.mean(dim=2, keepdim=True)

        def decode(self, x, t, z_sem):
            return m(x + 0.1 * z_sem, t)

    sig = torch.from_numpy(g["diffae/signal"]).to(dev)
    noise = torch.from_numpy(g["diffae/noise"]).to(dev)
    calls = []

    def fixed_randn_like(t, *a, **k):
        calls.append(tuple(t.shape))
        return noise.clone()
    monkeypatch.setattr(torch, "randn_like", fixed_randn_like)
    got = A.sample_diffae(DiffAE(), sig, 4, 0)
    monkeypatch.undo()
    assert calls == [tuple(sig.shape)]                       # exactly the one draw of inference.py:26 (eta = 0)
    want = torch.from_numpy(g["diffae/pred4"])
    assert rel_err(got, want) <= 1e-4 and snr_db(got, want) >= 60.0


@pytest.mark.parametrize("name", ["tiny_d3_cond.npz", "tiny_d3_cond_samelen.npz", "tiny_d3_cond_aug0.npz"])
def test_conditioning_path_vs_reference_golden(name):
    """cond (linear interpolation to the signal length inside pack_input), optional noise-aug level
    embedding; forward and DDIM (fused captured loop when cond is constant) vs goldens captured from the reference."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    z = load_npz(name)
    m = build(z, dev)
    x, t, cond = (torch.from_numpy(z[k]).to(dev) for k in ("x", "t", "cond"))
    extra = dict(cond=cond)
    if float(z["aug_scale"]) >= 0:
        extra["cond_aug_scale"] = float(z["aug_scale"])
    m.compute_dtype = torch.float32
    y = m(x, t, **extra)
    assert rel_err(y, torch.from_numpy(z["y"])) <= 1e-4
    pred = A.sample(m, x, int(z["steps"]), 0, **extra)
    assert rel_err(pred, torch.from_numpy(z["pred"])) <= 1e-4 and snr_db(pred, torch.from_numpy(z["pred"])) >= 60.0
    m.compute_dtype = torch.bfloat16
    yb = m(x, t, **extra).cpu()
    ref = torch.from_numpy(z["y"])
    cfg = json.loads(bytes(z["cfg"]).decode())
    cpu_extra = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in extra.items()}
    floor = rounding_floor(sub_sd(z, "sd/"), cfg, x.cpu(), t.cpu(), torch.bfloat16, **cpu_extra)
    assert float((yb - ref).norm() / ref.norm()) <= max(1e-2, 3 * floor[1])
    m.compute_dtype = torch.float16
    assert rel_err(m(x, t, **extra), ref) <= 1e-2                  # north_star's bound in the headline dtype
    with pytest.raises(RuntimeError):
        m(x, t)                                            # cond missing: the reference fails on the channel count
    with pytest.raises(RuntimeError):
        m(x, t, cond=cond[:, :1])


def test_cond_noise_aug_draws_noise():
    """cond_noise_aug with a non-zero level: the result moves with the RNG draw and is reproducible under a seed."""
    dev = torch.device("cuda:0")
    z = load_npz("tiny_d3_cond_aug0.npz")
    m = build(z, dev)
    m.compute_dtype = torch.float32
    x, t, cond = (torch.from_numpy(z[k]).to(dev) for k in ("x", "t", "cond"))
    torch.manual_seed(5)
    a = m(x, t, cond=cond, cond_aug_scale=0.3)
    torch.manual_seed(5)
    b = m(x, t, cond=cond, cond_aug_scale=0.3)
    c = m(x, t, cond=cond, cond_aug_scale=0.3)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(c).all()
    d = m(x, t, cond=cond)                                 # Sobol-drawn level per sample (reference :155-156)
    assert torch.isfinite(d).all()


def test_flagged_fused_launch_in_forward_falls_back_then_raises():
    """The same rule for a direct model(x, t) call: forward() reads the error word after its plan has run, never returns
    output computed past a flagged launch, falls back ONCE to the unfused plan with a warning, clears the (sticky) word
    where it handles it, and raises when no fallback is left.  The error is injected by writing the word (the kernels
    of a flagged plan stop waiting for statistics, so the flagged call's own output is garbage by construction)."""
    import audio_diffusion_amd as A
    from audio_diffusion_amd.unet import FusedLaunchError
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(depth=4, c_mults=[32, 64, 64, 128], n_attn_layers=1).to(dev).eval()
    m.compute_dtype = torch.float32
    torch.manual_seed(1)
    x = torch.randn(2, 2, 4096, device=dev)
    t = torch.tensor([0.3, 0.8], device=dev)
    good = m(x, t)
    plan = next(iter(m._plans.values()))
    assert plan.fuse_gn and plan._gn_state is not None and plan.gn_error() == 0
    off = plan._gn_state.off
    plan.arena[off + 4:off + 8].view(torch.int32).fill_(0x60002001)
    with pytest.warns(UserWarning, match="fuse_gn = False"):
        again = m(x, t)
    assert m.fuse_gn is False and rel_err(again, good) <= 1e-5
    assert m.fused_error() == 0
    # the word was cleared where it was handled: a fresh fused plan of the same model runs clean again
    m.fuse_gn = True
    m._plans = {}
    assert torch.equal(m(x, t), good) and m.fused_error() == 0
    # a plan whose model has fusion switched off meanwhile has nothing to fall back to: the flagged call raises
    plan = next(iter(m._plans.values()))
    plan.arena[plan._gn_state.off + 4:plan._gn_state.off + 8].view(torch.int32).fill_(0x71c0)
    m.fuse_gn = False                                # (the cached plan is still the fused one)
    with pytest.raises(FusedLaunchError):
        m(x, t)
    assert m.fused_error() == 0


def test_flagged_fused_launch_falls_back_to_separate_groupnorm_passes():
    """A fused conv + GroupNorm launch that gives up waiting (another process holding the GPU's block slots) leaves an
    error word; sample() must not return audio computed past it: it re-runs ONCE with GroupNorm as passes of its own and
    warns.  The error is injected by writing the word."""
    import audio_diffusion_amd as A
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(depth=4, c_mults=[32, 64, 64, 128], n_attn_layers=1).to(dev).eval()
    m.compute_dtype = torch.float32
    torch.manual_seed(1)
    x = torch.randn(2, 2, 4096, device=dev)
    good = A.sample(m, x, 4, 0)
    smp = [p for p in m._plans.values() if hasattr(p, "plan")][0]
    assert smp.plan._gn_state is not None and smp.plan.gn_error() == 0
    off = smp.plan._gn_state.off
    smp.plan.arena[off + 4:off + 8].view(torch.int32).fill_(0x60001000)
    with pytest.warns(UserWarning, match="fuse_gn = False"):
        again = A.sample(m, x, 4, 0)
    assert m.fuse_gn is False
    assert rel_err(again, good) <= 1e-5
    assert not any(getattr(p, "plan", p).fuse_gn for p in m._plans.values())
