#!/usr/bin/env python3
"""One denoiser call of the configs[1] model, repeated: how often does the output differ from the first, and where?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = torch.float16
if os.environ.get("NOFUSE"):
    m.fuse_gn = False
x = torch.randn(B, 2, 65536, device=dev)
t = torch.full((B,), 0.5, device=dev)
ref, bad = None, 0
with torch.no_grad():
    for it in range(n):
        y = m(x, t)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone(); continue
        d = (y != ref)
        if d.any():
            bad += 1
            idx = d.nonzero()
            print(f"call {it}: {idx.shape[0]} elements differ, samples {sorted(set(idx[:,0].tolist()))}, max abs diff {(y-ref).abs().max().item():.3e}, "
                  f"positions {int(idx[:,2].min())}..{int(idx[:,2].max())}")
print(f"{bad} of {n-1} calls differ")
