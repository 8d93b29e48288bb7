#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes: a few denoiser calls of config 2 (B=16, bf16, 2x65536) + the fused
DDIM update (PMC_CONFIG=2, default), config 4's PQMF-banded model (B=16, 2x2^17; PMC_CONFIG=4), or both.  Run as
    rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 tools/pmc_call.py
(one counter group per run; see tools/pmc_report.py for the summary)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A

dev = torch.device("cuda:0")
calls = int(os.environ.get("PMC_CALLS", "2"))
which = os.environ.get("PMC_CONFIG", "2")          # "2" (bench.py's workload: the roofline figures), "4", or "both"
if which in ("2", "both"):
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
    m.compute_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[os.environ.get("PMC_DTYPE", "f16")]
    m.use_graph = False                      # eager launches: one profiler record per kernel
    x = torch.randn(16, 2, 65536, device=dev)
    A.sample(m, x, calls, 0)
    torch.cuda.synchronize()
    del m
if which in ("4", "both"):
    torch.manual_seed(0)
    m4 = A.DiffusionAttnUnet1D(io_channels=2, pqmf_bands=32, depth=10, c_mults=[256, 256] + [512] * 8,
                               n_attn_layers=2).to(dev).eval()
    m4.compute_dtype = torch.bfloat16
    x4 = torch.randn(16, 2, 131072, device=dev)
    for _ in range(calls):
        m4(x4, torch.full((16,), 0.5, device=dev))
    torch.cuda.synchronize()
