#!/usr/bin/env python3
"""Per-op HIP-event timing of one denoiser call (eager, op by op), aggregated by kind / level.
usage: python tools/profile_ops.py [--dtype bf16|f16|fp32] [--batch 16] [--reps 3]"""
import argparse
import ctypes
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import _lib

NAMES = {1: "conv", 2: "gn_apply", 3: "gn_stats", 4: "fir_down", 5: "fir_up", 6: "attn", 7: "pack", 8: "unpack",
         9: "pqmf_a", 10: "pqmf_s", 14: "s2d", 15: "d2s", 16: "linear", 17: "fourier"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--T", type=int, default=65536)
    ap.add_argument("--model", default="uncond", help="uncond (config 2) | 3a (long context) | k-audio (k_audio_uncond_long_1.json, B=4, 32 x 32768)")
    ap.add_argument("--no-fuse-gn", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[a.dtype]
    torch.manual_seed(0)
    if a.model == "k-audio":
        a.batch = 4 if a.batch == 16 else a.batch
        m = A.AudioDenoiserModel(c_in=32, feats_in=128, depths=[2, 2, 2, 3, 3, 3, 4, 4],
                                 channels=[256, 256, 512, 512, 512, 512, 512, 512],
                                 self_attn_depths=[False] * 5 + [True] * 3, strides=[4, 4, 4, 2, 2, 2, 2]).to(dev).eval()
        plan = m.plan_for(a.batch, 32768, dtype, dev)
        x = torch.randn(a.batch, 32, 32768, device=dev)
        t = torch.ones(a.batch, device=dev)
        out = torch.empty_like(x)
        plan.bind_named(input=x.data_ptr(), t=t.data_ptr(), output=out.data_ptr())
    else:
        if a.model == "3a":                       # BASELINE config 3a: default ctor (attention up to s = 4096), B = 4 x 2^18
            a.batch, a.T = (4 if a.batch == 16 else a.batch), (262144 if a.T == 65536 else a.T)
            m = A.DiffusionAttnUnet1D().to(dev).eval()
        else:
            m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
        m.fuse_gn = not a.no_fuse_gn
        plan = m.plan_for(a.batch, a.T, dtype, dev)
        x = torch.randn(a.batch, 2, a.T, device=dev)
        t = torch.ones(a.batch, device=dev)
        out = torch.empty_like(x)
        plan.bind(x.data_ptr(), t.data_ptr(), out.data_ptr())
    st = torch.cuda.current_stream(dev)
    sz = ctypes.sizeof(_lib.AdkOp)
    ms = [0.0] * plan.n_ops
    for r in range(a.reps + 1):
        evs = []
        for i in range(plan.n_ops):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            _lib.run_ops(ctypes.cast(ctypes.byref(plan.ops, i * sz), ctypes.POINTER(_lib.AdkOp)), 1, st.cuda_stream)
            e1.record(st)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        if r:
            for i, (e0, e1) in enumerate(evs):
                ms[i] += e0.elapsed_time(e1) / a.reps
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for i, meta in enumerate(plan.meta):
        k = NAMES.get(meta["kind"], str(meta["kind"]))
        if meta["kind"] == 1:
            fz = ("+gn" + ("+skip" if meta.get("fused_skip") else ("+res" if plan.ops[i].residual else ""))) if meta.get("fused_gn") else ""
            key = (k, f"k{meta['taps']}{fz}", f"L{meta['L']}", f"{meta['cin']}->{meta['cout']}", f"tile{meta['cotile']}x{meta['postile']}")
        else:
            key = (k, f"L{meta.get('L', 0)}", f"C{meta.get('c', 0)}")
        e = agg[key]
        e[0] += 1; e[1] += ms[i]; e[2] += meta.get("flops", 0.0); e[3] += meta.get("bytes", 0.0)
    total = sum(ms)
    print(f"total {total:.3f} ms over {plan.n_ops} ops (event-timed, includes launch gaps)")
    bykind = defaultdict(float)
    for key, e in agg.items():
        bykind[key[0] + (key[1][:2] if key[0] == "conv" else "")] += e[1]
    for k, v in sorted(bykind.items(), key=lambda kv: -kv[1]):
        print(f"  {k:12s} {v:8.3f} ms  {100 * v / total:5.1f}%")
    print("-- detail (sorted by time)")
    for key, e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        rate = f"{e[2] / e[1] / 1e9:8.1f} TF/s" if e[2] else (f"{e[3] / e[1] / 1e9:8.2f} TB/s" if e[3] else "")
        print(f"  {' '.join(key):55s} n={e[0]:3d} {e[1]:8.3f} ms  avg {1e3 * e[1] / e[0]:8.1f} us  {rate}")


if __name__ == "__main__":
    main()
