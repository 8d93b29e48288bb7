#!/usr/bin/env python3
"""Smallest programs around the rocprofv3 --kernel-trace SIGSEGV: no sampler, no captured graph, no
arena teardown - plain denoiser calls.  usage (one case per profiled run):
    rocprofv3 --kernel-trace --stats -d OUT -o p -- python3 tools/rocprof_crash_probe.py --dtype fp32 --batch 16
cases that matter: fp32 B=16 (its fused skip-conv kernels keep 56-64 bytes per lane in scratch), fp32 B=2, f16 B=16 (no
kernel of that plan uses scratch)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--calls", type=int, default=3)
ap.add_argument("--no-fold-skip", action="store_true", help="keep the 1x1 skip convs as launches of their own (no scratch-using kernel left)")
ap.add_argument("--seq", default="", help="sequence of sample() calls, e.g. f16:g:20,fp32:g:20")
ap.add_argument("--window", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = {"fp32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
if a.no_fold_skip:
    m.fold_skip = False
if a.seq:
    # a sequence of sample() calls: "f16:g:20,fp32:g:20" = dtype : g (captured graph) | e (eager loop) : DDIM steps
    x = torch.randn(a.batch, 2, 65536, device=dev)
    import bench
    DT = {"fp32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
    for item in a.seq.split(","):
        if item.startswith("roofline:"):             # bench.py's per-launch event timing of one eager call
            bench.conv_roofline(m, a.batch, DT[item.split(":")[1]], dev)
            print(f"[probe] {item} done", file=sys.stderr, flush=True)
            continue
        if item.startswith("acc:"):                  # bench.py's accuracy object: fp32 + dtype forward on two samples
            m.compute_dtype = DT[item.split(":")[1]]
            bench.first_call_accuracy(m, x, DT[item.split(":")[1]], dev)
            print(f"[probe] {item} done", file=sys.stderr, flush=True)
            continue
        dt, mode, steps = item.split(":")
        m.compute_dtype = {"fp32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
        m.use_graph = mode == "g"
        if a.window:                                 # bound the host's run-ahead: synchronise every `window` replays
            from audio_diffusion_amd import sampling as S
            S._FusedSampler.replay_window = a.window
        y = A.sample(m, x, int(steps), 0)
        torch.cuda.synchronize()
        print(f"[probe] sample {item} done, finite={bool(torch.isfinite(y).all())}", file=sys.stderr, flush=True)
    print("[probe] finished", file=sys.stderr, flush=True)
    sys.exit(0)
x = torch.randn(a.batch, 2, 65536, device=dev)
t = torch.full((a.batch,), 0.5, device=dev)
for i in range(a.calls):
    y = m(x, t)
    torch.cuda.synchronize()
    print(f"[probe] call {i} done, finite={bool(torch.isfinite(y).all())}", file=sys.stderr, flush=True)
print("[probe] finished", file=sys.stderr, flush=True)
