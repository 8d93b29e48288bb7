#!/usr/bin/env python3
"""Register-spill audit of the MFMA kernels (CPU only: hipcc cross-compiles to assembly).

The pipelined convolution / attention loops wait for their LDS-DMA stages with COUNTED `s_waitcnt vmcnt(N)`.  A register
the compiler keeps in scratch is reloaded with a VMEM instruction; inside such a loop the reload joins the in-order
queue those waits look at.  That cannot make a wait too weak (vmcnt(N) still guarantees everything older than the N most
recent entries), but it makes it stricter - the prefetched stages drain - and it adds a memory round trip to the loop.
This tool compiles every kernel unit to assembly and fails if any kernel that uses LDS-DMA has a scratch access on a
control-flow cycle that also holds an MFMA (spills in prologues / epilogues / between an unrolled loop and its tail are
reported, not fatal).   usage: python tools/audit_spills.py [unit ...]"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio_diffusion_amd", "csrc")
UNITS = sys.argv[1:] or ["conv_mfma", "conv_mfma_ext", "conv_ksplit", "conv_mfma_gn_bf16", "conv_mfma_gn_f16", "conv_mfma_gn_f32",
                         "attention", "pqmf"]


def asm(unit, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
                    "--cuda-device-only", os.path.join(CSRC, unit + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return unit, open(out).read()


def in_loop_lines(k):
    """Indices of the lines of kernel text `k` that lie on a cycle of its control-flow graph which also holds an MFMA
    (basic blocks split at labels and after branches; strongly connected components by Tarjan's algorithm)."""
    ins = [(i, l.split(";")[0].strip()) for i, l in enumerate(k)]        # labels carry trailing comments
    ins = [(i, l) for i, l in ins if l and not l.startswith(";") and not (l.startswith(".") and not l.endswith(":"))]
    blocks, cur = [], []
    for i, l in ins:
        if l.endswith(":") and cur:
            blocks.append(cur); cur = []
        cur.append((i, l))
        op = l.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)
    label = {blk[0][1][:-1]: n for n, blk in enumerate(blocks) if blk[0][1].endswith(":")}
    succ = []
    for n, blk in enumerate(blocks):
        op = blk[-1][1].split()
        e = []
        if op[0].startswith("s_cbranch") or op[0] == "s_branch":
            if op[-1] in label:
                e.append(label[op[-1]])
        if op[0] not in ("s_branch", "s_endpgm", "s_setpc_b64") and n + 1 < len(blocks):
            e.append(n + 1)
        succ.append(e)
    index, low, on, stack, comps, cnt = {}, {}, set(), [], [], [0]
    for root in range(len(blocks)):                      # iterative Tarjan
        if root in index:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = cnt[0]; cnt[0] += 1; stack.append(v); on.add(v)
            rec = False
            for j in range(pi, len(succ[v])):
                w = succ[v][j]
                if w not in index:
                    work.append((v, j + 1)); work.append((w, 0)); rec = True
                    break
                if w in on:
                    low[v] = min(low[v], index[w])
            if rec:
                continue
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop(); on.discard(w); comp.append(w)
                    if w == v:
                        break
                comps.append(comp)
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    out = set()
    for comp in comps:
        if len(comp) == 1 and comp[0] not in succ[comp[0]]:
            continue
        lines = [x for n in comp for x in blocks[n]]
        if any(l.startswith("v_mfma") for _, l in lines):
            out.update(i for i, _ in lines)
    return out


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
            dma = any("global_load_lds" in l for l in k)
            loop = in_loop_lines(k) if sc and dma else set()
            inside = [i for i in sc if i in loop]
            if inside:
                bad += 1
                print(f"FATAL {unit}: {name}: {len(inside)} scratch accesses inside the MFMA loop of an LDS-DMA kernel")
            elif sc:
                soft += 1
        print(f"{unit}: {nk} kernels, {soft} with spills outside their MFMA loops")
sys.exit(1 if bad else 0)
