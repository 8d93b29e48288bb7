#!/usr/bin/env python3
"""Denoiser-call timings for the BASELINE.json configs that are not bench.py's headline line:
config 2 in fp32, config 3a / 3b (long context), config 4 (PQMF-banded) and the PQMF kernels alone.

usage: python tools/bench_configs.py [--reps 5] [--only 3a,4] [--dtype f16|bf16]
Prints one JSON line per config (ms per denoiser call, algorithmic TFLOP/s, audio-s/s at 100 DDIM steps).
Synthetic inputs, random-init weights (seed 0)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import _lib

SR = 48000
CONFIGS = {
    "2-fp32": (dict(n_attn_layers=4), 16, 65536, torch.float32),
    "2-bf16": (dict(n_attn_layers=4), 16, 65536, torch.bfloat16),
    "3a": (dict(), 4, 262144, torch.bfloat16),
    "3b": (dict(io_channels=2, pqmf_bands=16, depth=8, c_mults=[256, 256, 512, 512, 512, 512, 512, 512],
                n_attn_layers=1), 4, 262144, torch.bfloat16),
    # 3c: 3b with the JSON's own strides through the learned resamplers (needs f2)
    "3c": (dict(io_channels=2, pqmf_bands=16, depth=8, c_mults=[256, 256, 512, 512, 512, 512, 512, 512], n_attn_layers=1,
                learned_resample=True, strides=[1, 4, 4, 4, 2, 2, 2, 2]), 4, 262144, torch.bfloat16),
    "4": (dict(io_channels=2, pqmf_bands=32, depth=10, c_mults=[256, 256] + [512] * 8, n_attn_layers=2),
          16, 131072, torch.bfloat16),
    # model_configs/k_audio_uncond_long_1.json (the k-diffusion consumer of config 3b): AudioDenoiserModel behind the
    # Karras Denoiser, input = 16 PQMF bands x 2 channels of a 524288-sample clip
    "k-audio": (dict(c_in=32, feats_in=128, depths=[2, 2, 2, 3, 3, 3, 4, 4], channels=[256, 256, 512, 512, 512, 512, 512, 512],
                     self_attn_depths=[False] * 5 + [True] * 3, strides=[4, 4, 4, 2, 2, 2, 2]), 4, 524288, torch.bfloat16),
    # autoencoders/models.py:262-296 latent diffusion denoiser (SURVEY 8 f1's consumer): 32-channel latents of length 4096
    # (a 5.2 M-sample clip through the x1280 SoundStream encoder), conditioned on 32 x 256 second-stage latents
    "latent-f1": (dict(io_channels=32, cond_dim=32, n_attn_layers=0, c_mults=[512] * 10, depth=10), 16, 4096, torch.bfloat16),
    # autoencoders/soundstream.py default decoder: 16 clips of 52 latent frames -> 2 x 66560 samples
    "soundstream-dec": (dict(), 16, 66560, torch.bfloat16),
}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="", help="f16 | bf16: run every 16-bit config in this dtype (default: as listed, bf16)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = [n for n in CONFIGS if not a.only or n in a.only.split(",")]
    for name in names:
        kw, B, T, dtype = CONFIGS[name]
        if a.dtype and dtype != torch.float32:
            dtype = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
        torch.manual_seed(0)
        if name == "soundstream-dec":
            from audio_diffusion_amd.soundstream import SoundStreamXLDecoder
            m = SoundStreamXLDecoder(**kw).to(dev).eval()
            x = torch.randn(B, 128, T // m.upsampling(), device=dev)
            plan = m.plan_for(B, x.shape[2], dtype, dev)
            out = torch.empty(B, 2, T, device=dev)
            plan.bind_named(input=x.data_ptr(), output=out.data_ptr())
        elif name == "k-audio":
            m = A.AudioDenoiserModel(**kw).to(dev).eval()
            x = torch.randn(B, 32, T // 16, device=dev)
            t = torch.full((B,), 0.5, device=dev)
            sc = [s.contiguous() for s in A.Denoiser(m, 0.25).get_scalings(t)]
            plan = m.plan_for(B, T // 16, dtype, dev, scaled=True)
            out = torch.empty_like(x)
            plan.bind_named(input=x.data_ptr(), t=t.data_ptr(), output=out.data_ptr(), cskip=sc[0].data_ptr(),
                            cout=sc[1].data_ptr(), cin=sc[2].data_ptr())
        else:
            m = A.DiffusionAttnUnet1D(**kw).to(dev).eval()
            m.compute_dtype = dtype
            x = torch.randn(B, kw.get("io_channels", 2), T, device=dev)
            t = torch.full((B,), 0.5, device=dev)
            cond = torch.randn(B, kw["cond_dim"], T // 16, device=dev) if kw.get("cond_dim") else None
            plan = m.plan_for(B, T, dtype, dev, cond_len=cond.shape[2] if cond is not None else 0)
            out = torch.empty_like(x)
            plan.bind(x.data_ptr(), t.data_ptr(), out.data_ptr(), cond.data_ptr() if cond is not None else 0)
        st = torch.cuda.current_stream(dev)
        ms = timed(lambda: plan.run(st.cuda_stream), a.reps)
        flops = sum(mt.get("flops", 0.0) for mt in plan.meta)
        attn = sum(mt.get("attn_flops", 0.0) for mt in plan.meta)
        rec = {"config": name, "model": kw, "batch": B, "T": T, "dtype": str(dtype).split(".")[-1],
               "params_M": sum(p.numel() for p in m.parameters()) / 1e6,
               "ms_per_denoiser_call": ms, "conv_flop_per_call": flops, "attn_flop_per_call": attn,
               "algorithmic_tflops": (flops + attn) / (ms * 1e-3) / 1e12,
               "audio_s_per_s_at_100_steps": B * T / SR / (ms * 1e-3 * 100), "arena_GB": plan.peak / 1e9,
               "n_ops": plan.n_ops}
        if name == "k-audio":          # whole Denoiser call as the samplers issue it (input copies + one hipGraph replay)
            den = A.Denoiser(m, 0.25)
            m.compute_dtype = dtype
            rec["ms_per_denoiser_call_graph"] = timed(lambda: den(x, t), max(a.reps, 20))
            m.use_graph = False
            rec["ms_per_denoiser_call_eager_api"] = timed(lambda: den(x, t), max(a.reps, 20))
            sig = A.get_sigmas_karras(50, 1e-2, 3.0, rho=9., device=dev)
            m.use_graph = True
            rec["heun_50_sigmas_ms"] = timed(lambda: A.sample_heun(den, x, sig), 2)
            rec["note"] = "heun: 99 denoiser calls (2 per step, Euler last); 524288-sample clips at 12 kHz"
        print(json.dumps(rec), flush=True)
        if name == "4":                                    # the filterbank kernels alone, [16, 2, 2^17], 32 bands
            xx = torch.randn(16, 2, 131072, device=dev)
            ms_a = timed(lambda: m.pqmf(xx), 20)
            bands = m.pqmf(xx)
            ms_s = timed(lambda: m.pqmf.inverse(bands), 20)
            nbytes = 2 * xx.numel() * 4                    # read once + write once, fp32
            print(json.dumps({"config": "4-pqmf-kernels", "shape": [16, 2, 131072], "bands": 32,
                              "analysis_ms": ms_a, "synthesis_ms": ms_s,
                              "analysis_GBps_algorithmic": nbytes / ms_a / 1e6,
                              "synthesis_GBps_algorithmic": nbytes / ms_s / 1e6,
                              "note": "timings include the output allocation of the Python wrapper"}), flush=True)
        del m, plan
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
