#!/usr/bin/env python3
"""Experiment: per-launch time of short-level kernels when the SAME kernel repeats (warm instruction cache) vs inside the
plan (a different kernel every launch).  usage: python tools/tiny_kernel_exp.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
plan = m.plan_for(16, 65536, torch.float16, dev)
x = torch.randn(16, 2, 65536, device=dev); t = torch.ones(16, device=dev); out = torch.empty_like(x)
plan.bind(x.data_ptr(), t.data_ptr(), out.data_ptr())
st = torch.cuda.current_stream(dev)
plan.run(st.cuda_stream)
torch.cuda.synchronize()
sz = C.sizeof(_lib.AdkOp)
NAMES = {1: "conv", 2: "gn_apply", 4: "fir_down", 5: "fir_up", 6: "attn"}


def timed_range(first, count, reps):
    ops = C.cast(C.byref(plan.ops, first * sz), C.POINTER(_lib.AdkOp))
    for _ in range(3):
        _lib.run_ops(ops, count, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        _lib.run_ops(ops, count, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


# the deepest level (L = 8): ops with meta L == 8
idx = [i for i, mt in enumerate(plan.meta) if mt.get("L") == 8]
lo, hi = min(idx), max(idx)
chain = timed_range(lo, hi - lo + 1, 50)
print(f"L=8 section: {hi - lo + 1} ops, {chain:.1f} us per pass = {chain / (hi - lo + 1):.2f} us per op (different kernel each launch)")
seen = set()
for i in range(lo, hi + 1):
    mt = plan.meta[i]
    key = (mt["kind"], mt.get("taps"), mt.get("cin"), mt.get("cout"))
    if key in seen:
        continue
    seen.add(key)
    one = timed_range(i, 1, 300)
    print(f"  op {i} {NAMES.get(mt['kind'], mt['kind'])} {key[1:]}: {one:.2f} us per launch when repeated back-to-back")


# every short level as ONE back-to-back range (no events between the launches), and the whole L <= 256 section
whole = 0.0
for Lq in (256, 128, 64, 32, 16, 8):
    idx = [i for i, mt in enumerate(plan.meta) if mt.get("L") == Lq]
    # a level's ops form two contiguous runs (down path and up path) around the deeper levels
    runs, start = [], idx[0]
    for a_, b_ in zip(idx, idx[1:] + [None]):
        if b_ is None or b_ != a_ + 1:
            runs.append((start, a_))
            start = b_
    t = sum(timed_range(lo_, hi_ - lo_ + 1, 50) for lo_, hi_ in runs)
    n = sum(hi_ - lo_ + 1 for lo_, hi_ in runs)
    whole += t
    print(f"L={Lq}: {n} ops in {len(runs)} runs, {t:.1f} us back to back = {t / n:.2f} us per op")
idx = [i for i, mt in enumerate(plan.meta) if 0 < mt.get("L", 0) <= 256]
lo, hi = min(idx), max(idx)
sec = timed_range(lo, hi - lo + 1, 30)
print(f"L <= 256 section as one range: {hi - lo + 1} ops, {sec:.1f} us per call (sum of the level ranges: {whole:.1f} us)")
allt = timed_range(0, plan.n_ops, 10)
print(f"whole call, back to back: {plan.n_ops} ops, {allt / 1e3:.3f} ms")
