#!/usr/bin/env python3
"""Register-spill audit of the MFMA kernels (CPU only: hipcc cross-compiles to assembly).

The pipelined convolution / attention loops wait for their LDS-DMA stages with COUNTED `s_waitcnt vmcnt(N)`.  A register
the compiler keeps in scratch is reloaded with a VMEM instruction; inside such a loop the reload joins the in-order
queue those waits look at.  That cannot make a wait too weak (vmcnt(N) still guarantees everything older than the N most
recent entries), but it makes it stricter - the prefetched stages drain - and it adds a memory round trip to the loop.
This tool compiles every kernel unit to assembly and fails if any kernel that uses LDS-DMA has a scratch access between
its first and last MFMA (spills in prologues / epilogues are reported, not fatal).   usage: python tools/audit_spills.py [unit ...]"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio_diffusion_amd", "csrc")
UNITS = sys.argv[1:] or ["conv_mfma", "conv_mfma_ext", "conv_ksplit", "conv_mfma_gn_bf16", "conv_mfma_gn_f16", "conv_mfma_gn_f32",
                         "attention", "pqmf"]


def asm(unit, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
                    "--cuda-device-only", os.path.join(CSRC, unit + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return unit, open(out).read()


bad = 0
with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(4) as ex:
    for unit, s in ex.map(lambda u: asm(u, os.path.join(d, u + ".s")), UNITS):
        nk = soft = 0
        for name in re.findall(r"\n(_Z\w+):\s", s):
            a = s.index("\n" + name + ":")
            b = s.find(".end_amdhsa_kernel", a)
            if b < 0:
                continue
            k = s[a:b].splitlines()
            nk += 1
            sc = [i for i, l in enumerate(k) if "scratch_" in l]
            mf = [i for i, l in enumerate(k) if l.strip().startswith("v_mfma")]
            dma = any("global_load_lds" in l for l in k)
            inside = [i for i in sc if mf and mf[0] < i < mf[-1]]
            if inside and dma:
                bad += 1
                print(f"FATAL {unit}: {name}: {len(inside)} scratch accesses inside the MFMA loop of an LDS-DMA kernel")
            elif sc:
                soft += 1
        print(f"{unit}: {nk} kernels, {soft} with spills outside their MFMA loops")
sys.exit(1 if bad else 0)
