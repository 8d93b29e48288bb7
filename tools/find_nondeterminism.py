#!/usr/bin/env python3
"""Run a denoiser plan twice, op by op, and report the first op after which the arena differs between the two runs
(a data race or an uninitialised read shows up as run-to-run different bits).
usage: python tools/find_nondeterminism.py [--config 3a|2] [--dtype bf16]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3a")
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[a.dtype]
torch.manual_seed(0)
if a.config == "3a":
    m, B, T = A.DiffusionAttnUnet1D(), 4, 262144
else:
    m, B, T = A.DiffusionAttnUnet1D(n_attn_layers=4), 16, 65536
m = m.to(dev).eval()
plan = m.plan_for(B, T, dtype, dev)
torch.manual_seed(6)
x = torch.randn(B, 2, T, device=dev)
t = torch.full((B,), 0.5, device=dev)
out = torch.empty_like(x)
plan.bind(x.data_ptr(), t.data_ptr(), out.data_ptr())
st = torch.cuda.current_stream(dev)
sz = ctypes.sizeof(_lib.AdkOp)
view = plan.arena[: plan.arena.numel() // 8 * 8].view(torch.int64)


def run_once():
    sums = []
    plan.arena.zero_()
    out.zero_()
    for i in range(plan.n_ops):
        ops = ctypes.cast(ctypes.byref(plan.ops, i * sz), ctypes.POINTER(_lib.AdkOp))
        _lib.run_ops(ops, 1, st.cuda_stream)
        sums.append(int(view.sum().item()) ^ int(out.view(torch.int32).sum().item()))
    return sums


r1 = run_once()
for rep in range(3):
    r2 = run_once()
    bad = [i for i in range(plan.n_ops) if r1[i] != r2[i]]
    if bad:
        i = bad[0]
        print(f"rerun {rep}: first differing op {i} of {plan.n_ops}: {plan.meta[i]}")
        if i:
            print(f"   previous op: {plan.meta[i - 1]}")
        break
else:
    print(f"{plan.n_ops} ops: all arena checksums identical over 4 runs")

if bad:
    i = bad[0]
    snaps = []
    for rep in range(2):
        plan.arena.zero_(); out.zero_()
        for k in range(i + 1):
            ops = ctypes.cast(ctypes.byref(plan.ops, k * sz), ctypes.POINTER(_lib.AdkOp))
            _lib.run_ops(ops, 1, st.cuda_stream)
        torch.cuda.synchronize()
        snaps.append(plan.arena.clone())
    diff = (snaps[0] != snaps[1]).nonzero().flatten()
    base = plan.arena.data_ptr()
    op = plan.ops[i]
    print(f"   {diff.numel()} differing bytes; y at arena+{(op.y or 0) - base}, partials_out at arena+{(op.partials_out or 0) - base}")
    if diff.numel():
        d = diff.cpu()
        print("   first offsets:", d[:16].tolist(), " last:", d[-4:].tolist())
        if op.partials_out:
            po = op.partials_out - base
            inside = ((d >= po) & (d < po + 8 * 1024 * 1024)).sum().item()
            print(f"   of which at/after partials_out: {inside}")
        if op.partials_out:
            import numpy as np
            for s_ in snaps:
                pv = s_[po:po + 8 * 4096].cpu().numpy().view(np.float32).reshape(-1, 2)
                idx = sorted(set(((d[(d >= po)] - po) // 8).tolist()))[:4]
                print("   partial pairs", idx, [pv[k].tolist() for k in idx])
