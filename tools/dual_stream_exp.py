#!/usr/bin/env python3
"""Experiment: run the batch as two half-batch chains on two HIP streams, offset in time, so that one half's
latency-bound short levels overlap the other half's MFMA-bound long levels.  usage: python tools/dual_stream_exp.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import sampling as S

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = torch.bfloat16
B, T, steps = 16, 65536, 100
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 2, T, generator=g).to(dev)
coef = S.step_coefficients(steps, 0)


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


dt, ref = timed(lambda: A.sample(m, x, steps, 0))
print(f"single stream: {dt * 1e3:.1f} ms  ({B * T / 48000 / dt:.2f} audio-s/s)")

for nsplit, delay_frac in [(4, 0.0), (4, 0.15), (4, 0.25), (8, 0.0), (8, 0.125), (16, 0.0625)]:
    hb = B // nsplit
    smps = [S._FusedSampler(m, x[i * hb:(i + 1) * hb], steps, torch.bfloat16) for i in range(nsplit)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    for i, smp in enumerate(smps):                      # capture each half's step graph
        smp.run(x[i * hb:(i + 1) * hb], coef)
    torch.cuda.synchronize()
    step_ms = dt * 1e3 / steps

    def run():
        cur = torch.cuda.current_stream(dev)
        for i, (smp, st) in enumerate(zip(smps, streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                smp.x.copy_(x[i * hb:(i + 1) * hb])
                smp.step.zero_()
                if i and delay_frac:
                    torch.cuda._sleep(int(i * delay_frac * step_ms * 1e-3 * 2.0e9))
        for _ in range(steps):
            for smp, st in zip(smps, streams):
                with torch.cuda.stream(st):
                    smp.graph.replay()
        for st in streams:
            cur.wait_stream(st)
        return torch.cat([smp.pred for smp in smps])

    d2, out = timed(run)
    print(f"{nsplit} streams, delay {delay_frac}: {d2 * 1e3:.1f} ms  ({B * T / 48000 / d2:.2f} audio-s/s)  "
          f"bit-equal to single: {torch.equal(out, ref)}")
    del smps
