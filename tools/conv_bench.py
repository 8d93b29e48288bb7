#!/usr/bin/env python3
"""Isolated conv1d micro-benchmark through the C ABI (for rocprofv3 --pmc runs and A/B timing).
usage: python tools/conv_bench.py --B 16 --L 4096 --cin 512 --cout 512 --taps 5 [--dtype bf16] [--reps 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_amd import _lib, layout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--L", type=int, default=4096)
    ap.add_argument("--cin", type=int, default=512)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--taps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (same instruction stream, far less switching power)")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    dtype, adk = (torch.bfloat16, 1) if a.dtype == "bf16" else (torch.float32, 0)
    cb = layout.chunk(dtype)
    P = 2
    x = layout.to_cpl(torch.randn(a.B, a.cin, a.L, device=dev), dtype, P)
    bn = lib.adk_conv_cotile(adk, a.cout, a.L)
    cp = (a.cout + bn - 1) // bn * bn
    w = (torch.randn(a.cin // cb, a.taps, cp, cb, device=dev) / (a.cin * a.taps) ** 0.5).to(dtype)
    bias = torch.randn(cp, device=dev)
    if a.zeros:
        x.zero_(); w.zero_()
    y = torch.empty((a.cout + cb - 1) // cb, a.B, a.L + 2 * P, cb, dtype=dtype, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cout, op.CoutPad, op.Cin, op.B = _lib.OP_CONV, adk, a.L, a.cout, cp, a.cin, a.B
    op.taps = a.taps
    part = torch.zeros(a.B, lib.adk_partials_count(C.byref(op)), 2, device=dev)
    st = torch.cuda.current_stream()

    def run():
        rc = lib.adk_conv1d(adk, 0, a.B, a.L, P, a.cin, a.cout, cp, a.taps, x.data_ptr(), w.data_ptr(), bias.data_ptr(),
                            None, y.data_ptr(), part.data_ptr(), C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.adk_last_error()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    fl = 2.0 * a.cin * a.cout * a.taps * a.L * a.B
    print(f"conv B={a.B} L={a.L} {a.cin}->{a.cout} k{a.taps} {a.dtype} tile {lib.adk_conv_cotile_used(C.byref(op))}x{lib.adk_conv_postile(C.byref(op))}: "
          f"{us:.1f} us  {fl / us / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
