#!/usr/bin/env python3
"""Measured error of the reduced-precision paths (bf16, f16) against the golden vectors captured from the reference
(tests/golden): scale-relative max error and relative L2 error per denoiser call.  usage: python tools/dtype_error.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_diffusion_amd as A

dev = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def errs(y, ref):
    y, ref = y.cpu().double(), ref.double()
    return float((y - ref).abs().max() / ref.abs().max()), float((y - ref).norm() / ref.norm())


for name in ("tiny_d3.npz", "tiny_d4_noattn_nobias.npz", "tiny_d5.npz", "tiny_d3_pqmf4.npz"):
    z = np.load(os.path.join(G, name))
    cfg = json.loads(bytes(z["cfg"]).decode())
    m = A.DiffusionAttnUnet1D(**cfg)
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
    m = m.to(dev).eval()
    x, t, ref = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["t"]).to(dev), torch.from_numpy(z["y"])
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        m.compute_dtype = dt
        print(f"{name:28s} {str(dt):16s} max-rel %.2e  L2-rel %.2e" % errs(m(x, t), ref))

z = np.load(os.path.join(G, "full_seed0_steps10.npz"))
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
torch.manual_seed(1)
x = torch.randn(1, 2, 65536).to(dev)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    m.compute_dtype = dt
    v0 = m(x, torch.ones(1, device=dev))
    pred = A.sample(m, x, 10, 0)
    ref = torch.from_numpy(z["pred"]).double()
    snr = float(10 * torch.log10(ref.pow(2).sum() / (pred.cpu().double() - ref).pow(2).sum()))
    print(f"{'full 221M, first call':28s} {str(dt):16s} max-rel %.2e  L2-rel %.2e" % errs(v0, torch.from_numpy(z["v0"])),
          f" 10-step DDIM SNR {snr:.1f} dB")
