#!/usr/bin/env python3
"""Times adk_attn1d alone: config 3a's longest attention (B=4, 16 heads, s=4096) and shorter ones, per dtype.
usage: python tools/attn_bench.py [--dtypes f16,bf16,f32] [--s 4096,2048,1024,256]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_amd import _lib, layout

ap = argparse.ArgumentParser()
ap.add_argument("--dtypes", default="f16,bf16,f32")
ap.add_argument("--s", default="4096,2048,1024,256")
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--heads", type=int, default=16)
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for dt in a.dtypes.split(","):
    dtype = DT[dt]
    for s in [int(v) for v in a.s.split(",")]:
        Cc, P = a.heads * 32, 2
        qkv = layout.to_cpl(torch.randn(a.B, 3 * Cc, s, device=dev), dtype, P)
        y = torch.empty(Cc // layout.chunk(dtype), a.B, s + 2 * P, layout.chunk(dtype), dtype=dtype, device=dev)
        reps = 3 if dt == "f32" else 20
        for it in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                assert lib.adk_attn1d(_lib.adk_dtype(dtype), a.B, s, P, Cc, qkv.data_ptr(), y.data_ptr(), st) == 0
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 4.0 * s * s * 32 * a.heads * a.B
        print(f"{dt:5s} B={a.B} heads={a.heads} s={s:5d}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
