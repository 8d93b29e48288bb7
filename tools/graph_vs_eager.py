#!/usr/bin/env python3
"""Whole sample() calls with the captured step graph against the eager step loop (same kernels, same bits), and the bare
plan back to back.  usage: python tools/graph_vs_eager.py [--batch 16] [--steps 100] [--dtype f16]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
x = torch.randn(a.batch, 2, 65536, device=dev)
outs = {}
for mode in ("graph", "eager", "graph", "eager"):
    m.use_graph = mode == "graph"
    m._plans = {k: v for k, v in m._plans.items() if not (isinstance(k, tuple) and k and k[0] == "sampler")}
    A.sample(m, x, a.steps, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = A.sample(m, x, a.steps, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {dt * 1e3 / a.steps:.3f} ms per step, {a.batch * 65536 / 48000 / dt:.2f} audio-s/s", flush=True)
    outs[mode] = out
print("graph == eager bitwise:", torch.equal(outs["graph"], outs["eager"]))
