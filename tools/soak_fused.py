#!/usr/bin/env python3
"""Soak of the ticketed fused conv + GroupNorm launches at the per-GPU batches bench.py runs (16 at N = 1; 128 / 64 / 32
at N = 2 / 4 / 8): many 100-step samples, failing on the first flagged error word (the fallback's warning is an error here).
usage: python tools/soak_fused.py [--batches 16,32,64,128] [--samples 4]"""
import argparse, os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="16,32,64,128")
ap.add_argument("--samples", type=int, default=4)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
warnings.simplefilter("error")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
for B in [int(v) for v in a.batches.split(",")]:
    m._plans = {}
    torch.cuda.empty_cache()
    x = torch.randn(B, 2, 65536, device=dev)
    ref = None
    for it in range(a.samples):
        t0 = time.perf_counter()
        out = A.sample(m, x, 100, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(out).all()
        if ref is None:
            ref = out
        assert torch.equal(out, ref), "reruns must be bit-identical"
        print(f"B={B} sample {it}: {dt:.2f} s, {B * 65536 / 48000 / dt:.2f} audio-s/s, clean, bit-identical", flush=True)
print("soak clean")
