"""BASELINE.json configs 3 (long context), 4 (PQMF-banded) and 5 (batch sharding) on the GPU.

The CPU oracle needs minutes for the full-size long-context shapes, so each config is covered twice:
  * against the oracle (oracle/unet_oracle.py, pinned to the reference's goldens by tests/test_oracle_golden.py)
    at a size it finishes in seconds (one sample, or a shorter clip of the SAME architecture), fp32 1e-4;
  * at BASELINE's full size through size-independent properties: bit-identical reruns, bit-identical
    results when a sample is computed alone or inside a batch (what makes multi-GPU sharding exact),
    finite output, PQMF analysis->synthesis reconstruction (delay M, >= 40 dB).
"""
import pytest
import torch

from conftest import rel_err, snr_db, rounding_floor, bf16_bars

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bars(m, kw, x, t):
    """bf16 (max-rel, L2) bars from the operand-rounding floor of this model on this input (conftest.bf16_bars)."""
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return bf16_bars(rounding_floor(sd, kw, x, t, torch.bfloat16))


def make(seed=0, **kw):
    import audio_diffusion_amd as A
    from oracle import unet_oracle as U
    torch.manual_seed(seed)
    m = A.DiffusionAttnUnet1D(**kw).eval()
    ref = U.make_model_fn(m.state_dict(), U.make_config(**kw))
    return m.to(DEV), ref


def test_long_attention_small_channels_vs_oracle():
    """Attention at s = 4096 / 2048 / 1024 inside a model the CPU oracle runs in seconds (C = 32..64)."""
    kw = dict(depth=4, c_mults=[32, 32, 64, 64], n_attn_layers=2)
    m, ref = make(**kw)
    torch.manual_seed(3)
    x, t = torch.randn(2, 2, 8192), torch.tensor([0.3, 0.9])
    want = ref(x, t)
    m.compute_dtype = torch.float32
    assert rel_err(m(x.to(DEV), t.to(DEV)), want) <= 1e-4
    m.compute_dtype = torch.bfloat16                      # MFMA flash kernel, 64-query tiles
    got = m(x.to(DEV), t.to(DEV)).cpu()
    assert float((got - want).norm() / want.norm()) <= bars(m, kw, x, t)[1]
    m.compute_dtype = torch.float16                       # same kernel on v_mfma_f32_32x32x16_f16
    got = m(x.to(DEV), t.to(DEV)).cpu()
    assert rel_err(got, want) <= 1e-2 and float((got - want).norm() / want.norm()) <= 5e-3


def test_config3a_architecture_short_clip_vs_oracle():
    """Config 3a's model (default ctor: n_attn_layers=6, 234M parameters) on one 2^16-sample clip."""
    m, ref = make()
    torch.manual_seed(5)
    x, t = torch.randn(1, 2, 65536), torch.tensor([0.6])
    want = ref(x, t)
    m.compute_dtype = torch.float32
    assert rel_err(m(x.to(DEV), t.to(DEV)), want) <= 1e-4
    m.compute_dtype = torch.bfloat16
    got = m(x.to(DEV), t.to(DEV)).cpu()
    bmax, bl2 = bars(m, {}, x, t)
    assert float((got - want).norm() / want.norm()) <= bl2 and rel_err(got, want) <= bmax
    m.compute_dtype = torch.float16                       # the reference's autocast dtype: the north_star's 1e-2 holds
    got = m(x.to(DEV), t.to(DEV)).cpu()
    assert rel_err(got, want) <= 1e-2 and float((got - want).norm() / want.norm()) <= 5e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_config3a_full_size_properties(dtype):
    """Config 3a at full size: B=4 x 2 x 2^18, attention up to s = 4096.  f16 is the headline dtype (the reference's
    autocast dtype, what sample() uses by default and bench.py times); bf16 is the opt-in secondary record."""
    m, _ = make()
    m.compute_dtype = dtype
    torch.manual_seed(6)
    x = torch.randn(4, 2, 262144, device=DEV)
    t = torch.full((4,), 0.5, device=DEV)
    y = m(x, t)
    assert y.shape == x.shape and torch.isfinite(y).all()
    assert torch.equal(y, m(x, t)), "reruns must be bit-identical"
    alone = m(x[2:3].contiguous(), t[2:3])
    assert torch.equal(alone[0], y[2]), "a sample's result must not depend on its batch"
    assert m.fused_error() == 0


def test_config3b_vs_oracle_and_properties():
    """Config 3b (k_audio_uncond_long_1 shape mapped on DiffusionAttnUnet1D: 16 PQMF bands, 8 levels)."""
    kw = dict(io_channels=2, pqmf_bands=16, depth=8, c_mults=[256, 256, 512, 512, 512, 512, 512, 512], n_attn_layers=1)
    m, ref = make(**kw)
    torch.manual_seed(7)
    x = torch.randn(4, 2, 262144)
    t = torch.tensor([0.2, 0.4, 0.6, 0.8])
    want = ref(x[:1], t[:1])                              # one sample on the CPU
    m.compute_dtype = torch.float32
    y32 = m(x.to(DEV), t.to(DEV))
    assert rel_err(y32[:1], want) <= 1e-4
    m.compute_dtype = torch.bfloat16
    y16 = m(x.to(DEV), t.to(DEV))
    assert torch.isfinite(y16).all()
    assert float((y16[:1].cpu() - want).norm() / want.norm()) <= bars(m, kw, x[:1], t[:1])[1]
    assert torch.equal(y16, m(x.to(DEV), t.to(DEV)))
    m.compute_dtype = torch.float16
    assert rel_err(m(x.to(DEV), t.to(DEV))[:1], want) <= 1e-2


def test_config3c_learned_strides_vs_oracle():
    """Config 3c: 3b with the strides of k_audio_uncond_long_1.json through Downsample1d_2 / Upsample1d_2."""
    kw = dict(io_channels=2, pqmf_bands=16, depth=8, c_mults=[256, 256, 512, 512, 512, 512, 512, 512], n_attn_layers=1,
              learned_resample=True, strides=[1, 4, 4, 4, 2, 2, 2, 2])
    m, ref = make(**kw)
    torch.manual_seed(9)
    x = torch.randn(2, 2, 262144)
    t = torch.tensor([0.3, 0.8])
    want = ref(x[:1], t[:1])                              # one sample on the CPU
    m.compute_dtype = torch.float32
    assert rel_err(m(x.to(DEV), t.to(DEV))[:1], want) <= 1e-4
    m.compute_dtype = torch.bfloat16
    y16 = m(x.to(DEV), t.to(DEV))
    assert float((y16[:1].cpu() - want).norm() / want.norm()) <= bars(m, kw, x[:1], t[:1])[1]
    assert torch.equal(y16, m(x.to(DEV), t.to(DEV)))
    m.compute_dtype = torch.float16
    assert rel_err(m(x.to(DEV), t.to(DEV))[:1], want) <= 1e-2


def test_config4_pqmf32_vs_oracle_and_roundtrip():
    """Config 4: 2ch x 2^17 split into 32 bands, U-Net, synthesis; plus the filterbank's own round trip."""
    kw = dict(io_channels=2, pqmf_bands=32, depth=10, c_mults=[256, 256] + [512] * 8, n_attn_layers=2)
    m, ref = make(**kw)
    torch.manual_seed(8)
    x = torch.randn(2, 2, 131072)
    t = torch.tensor([0.35, 0.7])
    want = ref(x, t)
    m.compute_dtype = torch.float32
    assert rel_err(m(x.to(DEV), t.to(DEV)), want) <= 1e-4
    m.compute_dtype = torch.bfloat16
    got = m(x.to(DEV), t.to(DEV)).cpu()
    assert float((got - want).norm() / want.norm()) <= bars(m, kw, x, t)[1]
    m.compute_dtype = torch.float16
    assert rel_err(m(x.to(DEV), t.to(DEV)), want) <= 1e-2

    # filterbank alone at BASELINE's standalone size [16, 2, 2^17]: near-perfect reconstruction delayed by M
    xx = torch.randn(16, 2, 131072, device=DEV)
    bands = m.pqmf(xx)
    assert bands.shape == (16, 64, 4096)
    back = m.pqmf.inverse(bands)
    M = 32
    assert snr_db(back[..., M:-M], xx[..., : -2 * M]) >= 40.0
    # linearity (size-independent property of the analysis bank)
    a, b = torch.randn(2, 16, 2, 131072, device=DEV)
    assert rel_err(m.pqmf(a + 2 * b), m.pqmf(a) + 2 * m.pqmf(b)) <= 1e-5


def test_config5_sharding_is_bit_exact():
    """Config 5's premise: a sample sampled inside any batch slice equals the same sample of the whole batch."""
    import audio_diffusion_amd as A
    kw = dict(depth=6, c_mults=[32, 32, 64, 64, 128, 128], n_attn_layers=2)
    m, _ = make(**kw)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(8, 2, 8192, generator=g).to(DEV)
    whole = A.sample(m, noise, 6, 0)
    for lo, hi in [(0, 4), (4, 8), (5, 6)]:
        part = A.sample(m, noise[lo:hi].contiguous(), 6, 0)
        assert torch.equal(part, whole[lo:hi])


def test_config5_full_model_shard_of_32():
    """configs[4] as the 8-GPU run executes it: the full 221M-parameter uncond model on one rank's shard of 32 samples
    (global batch 256 / 8), in the benchmarked dtype.  Size-independent properties: finite, bit-identical rerun, and
    sample k computed alone == sample k inside the shard (so the all-gathered batch does not depend on N)."""
    import audio_diffusion_amd as A
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4).eval().to(DEV)
    m.compute_dtype = torch.float16
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(256, 2, 65536, generator=g)
    lo, hi = A.distributed.shard_bounds(256, 3, 8)        # rank 3 of 8
    assert hi - lo == 32
    x = noise[lo:hi].to(DEV)
    t = torch.full((32,), 0.7, device=DEV)
    y = m(x, t)
    assert y.shape == x.shape and torch.isfinite(y).all()
    assert torch.equal(y, m(x, t)), "reruns must be bit-identical"
    alone = m(x[17:18].contiguous(), t[17:18])
    assert torch.equal(alone[0], y[17]), "a sample's result must not depend on its shard"
    pred = A.sample(m, x, 3, 0)                           # a short captured loop on the full shard
    assert torch.isfinite(pred).all()
    assert torch.equal(A.sample(m, x[17:18].contiguous(), 3, 0)[0], pred[17])


def test_config5_two_gpu_shard_of_128_soak():
    """configs[4] at N = 2: 128 samples per GPU.  Thousands of ticketed fused conv + GroupNorm launches over grids of up
    to 36 864 blocks: with exactly one block per tile the LAST sample's tiles could be left in the queue of a lagging XCD
    behind blocks waiting for them (a bounded wait expired about once per 10^4 launches at this batch); the launches
    now carry spare blocks (gn_spare).  sample() raises if any launch flags its error word."""
    import audio_diffusion_amd as A
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4).eval().to(DEV)
    m.compute_dtype = torch.float16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(128, 2, 65536, generator=g).to(DEV)
    pred = A.sample(m, x, 25, 0)
    assert torch.isfinite(pred).all()
    plan = [p for p in m._plans.values() if hasattr(p, "plan")][0].plan
    assert plan.gn_error() == 0
    assert torch.equal(A.sample(m, x[77:78].contiguous(), 25, 0)[0], pred[77]), "a sample's result must not depend on its shard"


def test_config5_rccl_backend_single_rank():
    """configs[4]'s collective path through the REAL backend: `distributed.init(backend="nccl")` (= RCCL on ROCm), the
    all-gather of the output audio, the max-over-ranks all-reduce and the barrier that bench.py uses at N > 1 - with the one
    rank a one-GPU box allows (RCCL refuses two ranks on one device).  A single-rank communicator moves no bytes over
    xGMI; what this covers is that the library loads, the communicator initialises on this image, and every collective
    call of distributed.py runs on device tensors and returns the 1-rank result bit for bit."""
    import socket
    import os
    import audio_diffusion_amd as A
    from audio_diffusion_amd import distributed as D
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    kw = dict(depth=4, c_mults=[32, 64, 64, 128], n_attn_layers=1)
    m, _ = make(**kw)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(4, 2, 4096, generator=g).to(DEV)
    want = A.sample(m, noise, 4, 0)
    D.init(0, 1, backend="nccl")
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        got = D.sample_sharded(m, noise, 4, 0, sampler=A.sample)
        assert got.is_cuda and torch.equal(got, want)
        assert torch.equal(D.all_gather_audio(want[:3].contiguous()), want[:3])
        assert D.max_over_ranks(3.5, 2, torch.device(DEV)) == 3.5        # world > 1 forces the all-reduce call
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        D.shutdown()
