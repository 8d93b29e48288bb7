#!/usr/bin/env python3
"""Experiment: does a power-capped MFMA conv overlap with an HBM-bound GroupNorm pass of ANOTHER tensor?
Serial (one stream) vs concurrent (two streams) time for N x (conv, gn_apply) on independent tensors.
usage: python tools/overlap_exp.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_amd import _lib, layout

lib = _lib.load()
dev = torch.device("cuda:0")
dtype, adk, cb, P = torch.bfloat16, 1, 32, 2


def make(Cc, L, B=16):
    x = layout.to_cpl(torch.randn(B, Cc, L, device=dev), dtype, P)
    w = (torch.randn(Cc // cb, 5, Cc, cb, device=dev) / (Cc * 5) ** 0.5).to(dtype)
    y = torch.empty_like(x)
    conv = _lib.AdkOp()
    conv.kind, conv.dtype, conv.B, conv.L, conv.P, conv.Cin, conv.Cout, conv.CoutPad, conv.taps = _lib.OP_CONV, adk, B, L, P, Cc, Cc, Cc, 5
    conv.x, conv.w, conv.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    g = layout.to_cpl(torch.randn(B, Cc, L, device=dev), dtype, P)
    st = _lib.AdkOp()
    st.kind, st.dtype, st.B, st.L, st.P, st.Cin = _lib.OP_GN_STATS, adk, B, L, P, Cc
    n = lib.adk_partials_count(C.byref(st))
    part = torch.zeros(B, n, 2, device=dev)
    st.x, st.partials_out = g.data_ptr(), part.data_ptr()
    _lib.run_ops(C.pointer(st), 1, torch.cuda.current_stream().cuda_stream)
    gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    gn = _lib.AdkOp()
    gn.kind, gn.dtype, gn.flags, gn.B, gn.L, gn.P, gn.Cin, gn.Cout, gn.f0 = _lib.OP_GN_APPLY, adk, _lib.FLAG_GELU, B, L, P, Cc, Cc, 1e-5
    gn.x, gn.y, gn.partials_in, gn.n_partials_in, gn.gamma, gn.beta = g.data_ptr(), g.data_ptr(), part.data_ptr(), n, gam.data_ptr(), bet.data_ptr()
    return conv, gn, (x, w, y, g, part, gam, bet)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N = 40
for Cc, L in [(256, 16384), (512, 4096), (128, 65536)]:
    conv, gn, keep = make(Cc, L)
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def serial():
        for _ in range(N):
            _lib.run_ops(C.pointer(conv), 1, main.cuda_stream)
            _lib.run_ops(C.pointer(gn), 1, main.cuda_stream)

    def only(op):
        def f():
            for _ in range(N):
                _lib.run_ops(C.pointer(op), 1, main.cuda_stream)
        return f

    def concurrent():
        side.wait_stream(main)
        for _ in range(N):
            _lib.run_ops(C.pointer(conv), 1, main.cuda_stream)
            _lib.run_ops(C.pointer(gn), 1, side.cuda_stream)
        main.wait_stream(side)

    tc, tg, ts, tp = timed(only(conv)), timed(only(gn)), timed(serial), timed(concurrent)
    print(f"C={Cc} L={L}: conv {tc / N * 1e3:.1f} us, gn {tg / N * 1e3:.1f} us, serial pair {ts / N * 1e3:.1f} us, "
          f"two streams {tp / N * 1e3:.1f} us  ({100 * (1 - tp / ts):+.1f} % vs serial)", flush=True)
