#!/usr/bin/env python3
"""ISA lint of the hand-scheduled kernels: reads the gfx950 machine code that SHIPS (libadk.so, or given code objects),
not the source, and fails when an instruction stream breaks one of the three rules inline asm takes out of the
compiler's hands (cdna_hip_programming.md section 5.7: hipcc neither counts the memory operations of an asm statement nor
pads their hazards, and an asm load's destination "counts as written" when the statement retires).

  (a) LDS results in flight.  Every DS instruction with a VGPR destination (ds_read_*, ds_read_b64_tr_b16, ds_bpermute,
      ...) stays "pending" until an s_waitcnt lgkmcnt(N) that provably covers it: LDS operations return in order, so a
      read with k younger DS operations behind it is complete once lgkmcnt <= k; scalar-memory loads share the counter and
      return out of order, so they only retire at lgkmcnt(0) and never help an LDS read retire.  Between the read and
      that wait NO instruction may name a register of its destination - not as a source (stale fragment into an MFMA), not
      as a destination (the late LDS write-back lands in a re-used register), not in a scratch store (spilled garbage).
      The state is propagated over the kernel's control-flow graph (all paths, loops to a fixed point).
  (b) MFMA results in flight.  After a v_mfma the ISA requires software wait states before (i) another MFMA reads, as SrcC,
      registers that OVERLAP its destination without being exactly its destination, (ii) an MFMA reads them as A / B,
      (iii) any other instruction reads or overwrites them.  The compiler pads these itself - unless an asm statement
      made it lose sight of the MFMA as the producer (conv_mainloop16.h tells that story).  The counts are LLVM's gfx950
      numbers (GCNHazardRecognizer: passes + 2 / + 4 / + 4 for the matrix (XDL) shapes, passes / passes + 2 / passes + 2
      for the f32-input shapes; passes = issue cycles / 4; cdna_hip_programming.md 5.7 item 2: "8-pass XDL: 12 states").
      Checked along every path from the MFMA for as many wait states as the largest requirement.
  (c) Counted vmcnt loops.  A loop that feeds LDS by LDS-DMA (global_load_lds_*) and waits with s_waitcnt vmcnt(N > 0)
      counts its own DMA instructions; any other vector-memory instruction on that cycle (a scratch reload, a stray global
      load or store) joins the same in-order counter and silently changes what every counted wait means.

  (d) v_permlane16_swap / v_permlane32_swap (the 16-byte-store exchange of the conv epilogues): an operand written by a VALU
      instruction needs 2 wait states before the swap reads it (LLVM gfx950 rule; cdna_hip_programming.md T21).

usage: python tools/isa_lint.py [--so audio_diffusion_amd/libadk.so | file.elf ... | file.s ...] [--kernel REGEX] [-v]
exit status 0 = clean, 1 = findings, 2 = could not analyse (unknown control flow, state explosion).
A `.s` argument is hipcc -S output (used by the positive controls in tests/test_isa_lint_cpu.py)."""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile
from collections import defaultdict

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---------------------------------------------------------------------------------------------------------------------
# code objects out of the shipped library


def code_objects_of(so_path, outdir):
    """The gfx950 ELF images inside a host shared object / object file (section .hip_fatbin: clang offload bundles)."""
    fat = os.path.join(outdir, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.devnull],
                   check=True, stderr=subprocess.DEVNULL)
    d = open(fat, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = d.find(magic, pos)
        if i < 0:
            break
        (cnt,) = struct.unpack_from("<Q", d, i + 24)
        o = i + 32
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", d, o)
            o += 24
            triple = d[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                p = os.path.join(outdir, f"co{len(out)}.elf")
                open(p, "wb").write(d[i + off:i + off + size])
                out.append(p)
        pos = i + 24
    return out


def disassemble(elf):
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", elf], check=True,
                          capture_output=True, text=True).stdout


# ---------------------------------------------------------------------------------------------------------------------
# instruction model

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def vregs(text):
    """Set of ('v'|'a', n) named anywhere in an operand string."""
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), n) for n in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "op", "args", "ops", "target", "text", "line")

    def __init__(self, addr, op, args, target, text, line):
        self.addr, self.op, self.args, self.target, self.text, self.line = addr, op, args, target, text, line
        self.ops = [a.strip() for a in split_operands(args)]


def split_operands(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_objdump(text):
    """{kernel name: [Ins]} from llvm-objdump -d output (addresses in the trailing comment, branch targets as <sym+0xoff>)."""
    kernels, cur, base = {}, None, 0
    for n, line in enumerate(text.splitlines()):
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            base = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, cmt = line.partition("//")
        body = body.strip()
        if not body:
            continue
        ma = re.match(r"\s*([0-9A-Fa-f]+):", cmt)
        if not ma:
            continue
        addr = int(ma.group(1), 16)
        op, _, args = body.partition(" ")
        target = None
        if op.startswith("s_cbranch") or op == "s_branch":
            mt = re.search(r"<[^>+]+(?:\+0x([0-9a-f]+))?>", cmt)
            if mt:
                target = base + (int(mt.group(1), 16) if mt.group(1) else 0)
        cur.append(Ins(addr, op, args.strip(), target, body, n + 1))
    return kernels


def parse_asm(text):
    """{kernel name: [Ins]} from hipcc -S output (labels; synthetic addresses = instruction index * 4)."""
    kernels = {}
    for name in re.findall(r"\n(_Z\w+):\s", text):
        a = text.index("\n" + name + ":")
        b = text.find(".Lfunc_end", a)
        if b < 0 or ".amdhsa_kernel " + name not in text:
            continue                                    # a device function, not a kernel
        lines = text[a:b].splitlines()[2:]
        ins, labels, pending = [], {}, []
        for n, raw in enumerate(lines):
            l = raw.split(";")[0].strip()
            if not l or l.startswith(";"):
                continue
            if l.endswith(":"):
                pending.append(l[:-1])
                continue
            if l.startswith("."):
                continue
            addr = 4 * len(ins)
            for lb in pending:
                labels[lb] = addr
            pending = []
            op, _, args = l.partition(" ")
            if "\t" in op:
                op, _, rest = op.partition("\t")
                args = (rest + " " + args).strip()
            ins.append(Ins(addr, op, args.strip(), None, l, n))
        for i in ins:
            if i.op.startswith("s_cbranch") or i.op == "s_branch":
                i.target = labels.get(i.args.strip())
        kernels[name] = ins
    return kernels


STORE = re.compile(r"^(global_store|buffer_store|scratch_store|flat_store|ds_write|ds_gws|global_load_lds|buffer_load.*\blds\b|"
                   r"s_|v_cmp_|v_cmpx_|global_atomic|buffer_atomic|ds_add|ds_sub|ds_min|ds_max|ds_and|ds_or|ds_xor|ds_inc|ds_dec)")
BOTH = ("v_swap_b32", "v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_permlane16_swap_b32_e32", "v_permlane32_swap_b32_e32",
        "v_permlane16_swap_b32_e64", "v_permlane32_swap_b32_e64")


def defs_uses(i):
    """(VGPR/AGPR registers written, read) - by operand position: destination first unless the opcode has none."""
    if not i.ops:
        return set(), set()
    if i.op in BOTH:
        r = vregs(i.args)
        return r, r
    if STORE.match(i.op) and not (i.op.startswith(("global_atomic", "buffer_atomic")) and (" sc0" in i.args or " glc" in i.args)) \
            and not i.op.startswith(("ds_add_rtn", "ds_sub_rtn")):
        return set(), vregs(i.args)
    d = vregs(i.ops[0])
    u = set()
    for o in i.ops[1:]:
        u |= vregs(o)
    if i.op.startswith(("v_mac_", "v_fmac_", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_dot8c", "v_movrel", "v_readlane", "v_writelane")):
        u |= d                                          # accumulate-into-destination forms read it too
    return d, u


def is_ds_result(i):
    """A DS instruction that returns data to VGPRs (counts in lgkmcnt, returns in order)."""
    return i.op.startswith("ds_") and not STORE.match(i.op) and bool(i.ops) and bool(vregs(i.ops[0]))


def is_ds(i):
    return i.op.startswith("ds_")


def is_smem(i):
    return i.op.startswith(("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_scratch_load", "s_atc_probe",
                            "s_sendmsg", "s_atomic", "s_buffer_atomic"))


def is_vmem(i):
    return i.op.startswith(("global_", "buffer_", "scratch_", "flat_", "tbuffer_")) and not i.op.startswith("buffer_wbl2") \
        and not i.op.startswith("buffer_inv")


def lgkm_of_wait(i):
    """N of s_waitcnt lgkmcnt(N), None if the instruction does not wait on that counter."""
    if i.op != "s_waitcnt":
        return None
    m = re.search(r"lgkmcnt\((\d+)\)", i.args)
    if m:
        return int(m.group(1))
    if re.fullmatch(r"\s*(0|0x0)\s*", i.args):
        return 0
    return None


def vm_of_wait(i):
    if i.op != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", i.args)
    return int(m.group(1)) if m else None


def wait_states(i):
    if i.op == "s_nop":
        try:
            return int(i.args, 0) + 1
        except ValueError:
            return 1
    return 1


# MFMA issue passes (4 cycles each) on gfx950 and whether LLVM treats the opcode as XDL (matrix) or SGEMM-class (f32 input)
def mfma_info(op):
    if not op.startswith(("v_mfma_", "v_smfmac_")):
        return None
    m = re.match(r"v_s?mfmac?_(?:scale_)?(\w+?)_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return (16, True)
    M, K = int(m.group(2)), int(m.group(4))
    rest = op[m.end():]
    f32in = rest.startswith(("f32", "_f32")) or re.search(r"x\d+_?f32$", op) is not None or op.endswith("xf32")
    f64 = "f64" in op
    if f64:
        return (16 if M == 16 else 4, False)
    if f32in:                                              # v_mfma_f32_32x32x2_f32: 64 cycles, 16x16x4_f32: 32, 4x4x1: 8
        return ({32: 16, 16: 8, 4: 2}.get(M, 16), False)
    if M == 32:
        passes = 8 if K <= 16 else 16                      # 32x32x16 (16-bit) 32 cycles; the 64-deep f8f6f4 forms 64
    elif M == 16:
        passes = 4 if K <= 32 else 8
    else:
        passes = 2
    return (passes, True)


def hazard_need(passes, xdl, kind):
    """Wait states LLVM's gfx950 hazard recogniser requires after an MFMA of `passes` before `kind` =
    'srcc' (MFMA SrcC overlapping, not identical), 'srcab' (MFMA A/B), 'other' (VALU / memory read, or any overwrite)."""
    if xdl:
        if kind == "srcc":
            return passes + 2
        return passes + 3 + (1 if passes != 2 else 0)
    if kind == "srcc":
        return passes
    return passes + 2


# ---------------------------------------------------------------------------------------------------------------------
# control-flow graph


class CFG:
    def __init__(self, ins):
        self.ins = ins
        index = {i.addr: n for n, i in enumerate(ins)}
        leaders = {0}
        self.unknown = []
        for n, i in enumerate(ins):
            if i.op.startswith("s_cbranch") or i.op == "s_branch":
                if i.target in index:
                    leaders.add(index[i.target])
                else:
                    self.unknown.append(i)
                if n + 1 < len(ins):
                    leaders.add(n + 1)
            elif i.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
                if i.op in ("s_setpc_b64", "s_swappc_b64"):
                    self.unknown.append(i)
                if n + 1 < len(ins):
                    leaders.add(n + 1)
        starts = sorted(leaders)
        self.blocks = [(s, (starts[k + 1] if k + 1 < len(starts) else len(ins))) for k, s in enumerate(starts)]
        bidx = {s: k for k, (s, _) in enumerate(self.blocks)}
        self.succ = []
        for k, (s, e) in enumerate(self.blocks):
            last = ins[e - 1]
            out = []
            if last.op.startswith("s_cbranch") or last.op == "s_branch":
                if last.target in index:
                    out.append(bidx[index[last.target]])
            if last.op not in ("s_branch", "s_endpgm", "s_setpc_b64") and e < len(ins):
                out.append(bidx[e])
            self.succ.append(out)

    def cycles(self):
        """Strongly connected components with at least one edge (Tarjan, iterative): lists of block indices."""
        n = len(self.blocks)
        index, low, on, stack, comps, cnt = {}, {}, set(), [], [], [0]
        for root in range(n):
            if root in index:
                continue
            work = [(root, 0)]
            while work:
                v, pi = work.pop()
                if pi == 0:
                    index[v] = low[v] = cnt[0]
                    cnt[0] += 1
                    stack.append(v)
                    on.add(v)
                rec = False
                for j in range(pi, len(self.succ[v])):
                    w = self.succ[v][j]
                    if w not in index:
                        work.append((v, j + 1))
                        work.append((w, 0))
                        rec = True
                        break
                    if w in on:
                        low[v] = min(low[v], index[w])
                if rec:
                    continue
                if low[v] == index[v]:
                    comp = []
                    while True:
                        w = stack.pop()
                        on.discard(w)
                        comp.append(w)
                        if w == v:
                            break
                    if len(comp) > 1 or comp[0] in self.succ[comp[0]]:
                        comps.append(comp)
                if work:
                    u = work[-1][0]
                    low[u] = min(low[u], low[v])
        return comps


# ---------------------------------------------------------------------------------------------------------------------
# rule (a): LDS results in flight

MAX_STATES = 200000
MAX_QUEUE = 96


def retire(queue, n):
    """Queue (oldest first) of ('L'|'S', frozenset regs, addr) after s_waitcnt lgkmcnt(n)."""
    if n == 0:
        return ()
    younger_lds = 0
    keep = []
    for e in reversed(queue):
        if e[0] == "L":
            if younger_lds >= n:
                continue                                # complete: n <= number of younger LDS operations
            younger_lds += 1
            keep.append(e)
        else:
            keep.append(e)                              # scalar memory: out of order, only lgkmcnt(0) retires it
    keep.reverse()
    return tuple(keep)


def lint_lds(name, cfg, findings):
    ins = cfg.ins
    seen = [set() for _ in cfg.blocks]
    work = [(0, ())]
    nstates = 0
    reported = set()
    while work:
        b, q = work.pop()
        key = tuple((k, r) for k, r, _ in q)
        if key in seen[b]:
            continue
        seen[b].add(key)
        nstates += 1
        if nstates > MAX_STATES:
            findings.append(("fatal", name, None, "rule (a): state explosion - lint incomplete"))
            return
        s, e = cfg.blocks[b]
        for n in range(s, e):
            i = ins[n]
            w = lgkm_of_wait(i)
            if w is not None:
                q = retire(q, w)
                continue
            if q:
                named = vregs(i.args)
                if named:
                    for kind, regs, addr in q:
                        if kind == "L" and regs and (regs & named) and (addr, i.addr) not in reported:
                            reported.add((addr, i.addr))
                            rr = sorted(regs & named)
                            findings.append(("a", name, i, f"names {fmt_regs(rr)} while the LDS read at {addr:#x} that writes "
                                                           f"them is still in flight ({len(q)} operations pending)"))
            if is_ds(i):
                q = q + (("L", frozenset(vregs(i.ops[0])) if is_ds_result(i) else frozenset(), i.addr),)
                while q and not q[0][1]:                 # the oldest entries, if they return nothing (stores), bind no register
                    q = q[1:]                            # and - being older than every read - count for none of them
            elif is_smem(i):
                pass                                     # out-of-order returns: never help an LDS read retire, name no VGPR
            elif i.op.startswith("flat_"):
                findings.append(("fatal", name, i, "rule (a): flat_ access (counts in both vmcnt and lgkmcnt, out of order): not modelled"))
                return
            if len(q) > MAX_QUEUE:
                findings.append(("fatal", name, i, "rule (a): more than %d LGKM operations pending - lint incomplete" % MAX_QUEUE))
                return
        for t in cfg.succ[b]:
            work.append((t, q))


def fmt_regs(rr):
    return ", ".join(f"{k}{n}" for k, n in rr[:6]) + (" ..." if len(rr) > 6 else "")


# ---------------------------------------------------------------------------------------------------------------------
# rule (b): MFMA results in flight


def lint_mfma(name, cfg, findings):
    ins = cfg.ins
    starts = {s: k for k, (s, _) in enumerate(cfg.blocks)}
    block_of = {}
    for k, (s, e) in enumerate(cfg.blocks):
        for n in range(s, e):
            block_of[n] = k
    reported = set()
    for n0, p in enumerate(ins):
        info = mfma_info(p.op)
        if not info:
            continue
        passes, xdl = info
        dst = vregs(p.ops[0])
        maxneed = max(hazard_need(passes, xdl, k) for k in ("srcc", "srcab", "other"))
        # walk every path forward for maxneed wait states
        work = [(n0 + 1, 0)]
        visited = set()
        while work:
            n, ws = work.pop()
            while True:
                if ws >= maxneed or n >= len(ins) or (n, ws) in visited:
                    break
                visited.add((n, ws))
                c = ins[n]
                cd, cu = defs_uses(c)
                cinfo = mfma_info(c.op)
                need, what = 0, ""
                if cinfo:
                    srcc = vregs(c.ops[3]) if len(c.ops) > 3 else set()
                    srcab = (vregs(c.ops[1]) | vregs(c.ops[2])) if len(c.ops) > 2 else set()
                    if srcc & dst and srcc != dst:
                        need, what = hazard_need(passes, xdl, "srcc"), "reads, as SrcC, registers that overlap (without being) the destination"
                    if srcab & dst:
                        need, what = max(need, hazard_need(passes, xdl, "srcab")), "reads, as A / B, the destination"
                elif (cu | cd) & dst:
                    need, what = hazard_need(passes, xdl, "other"), ("reads" if cu & dst else "overwrites") + " the destination"
                if need > ws and (p.addr, c.addr) not in reported:
                    reported.add((p.addr, c.addr))
                    findings.append(("b", name, c, f"{what} of the {p.op} at {p.addr:#x} after {ws} wait states; the ISA requires {need}"))
                if cd & dst and not cinfo:
                    dst = dst - cd                       # overwritten: later readers see the new producer
                    if not dst:
                        break
                if cinfo and cd == dst:
                    break                                # the accumulate chain's next link is the new producer
                ws += wait_states(c)
                if c.op.startswith("s_cbranch") or c.op == "s_branch":
                    k = block_of[n]
                    for t in cfg.succ[k]:
                        work.append((cfg.blocks[t][0], ws))
                    break
                if c.op in ("s_endpgm", "s_setpc_b64"):
                    break
                n += 1


# ---------------------------------------------------------------------------------------------------------------------
# rule (c): counted vmcnt loops


def lint_vmcnt_loops(name, cfg, findings):
    ins = cfg.ins
    for comp in cfg.cycles():
        body = [ins[n] for k in comp for n in range(*cfg.blocks[k])]
        dma = [i for i in body if i.op.startswith("global_load_lds") or (i.op.startswith("buffer_load") and " lds" in i.args)]
        counted = [i for i in body if (vm_of_wait(i) or 0) > 0]
        if not dma or not counted:
            continue
        for i in body:
            if is_vmem(i) and i not in dma:
                findings.append(("c", name, i, f"vector-memory instruction on a loop that counts its LDS-DMA with s_waitcnt vmcnt(N > 0) "
                                               f"({len(dma)} DMA instructions, {len(counted)} counted waits)"))


# ---------------------------------------------------------------------------------------------------------------------
# rule (d): VALU write -> v_permlane*_swap read


def lint_permlane(name, cfg, findings):
    ins = cfg.ins
    for s, e in cfg.blocks:
        for n in range(s, e):
            c = ins[n]
            if not c.op.startswith(("v_permlane16_swap", "v_permlane32_swap")):
                continue
            ops = vregs(c.args)
            ws, m = 0, n - 1
            while m >= s and ws < 2:                     # (within the block: a swap two instructions into a block is not checked
                p = ins[m]                               #  across the edge - the compiler's own pads are per block as well)
                if p.op.startswith("v_") and not mfma_info(p.op) and (defs_uses(p)[0] & ops):
                    findings.append(("d", name, c, f"reads {fmt_regs(sorted(defs_uses(p)[0] & ops))} {ws} wait states after the VALU write "
                                                   f"at {p.addr:#x}; the ISA requires 2"))
                    break
                ws += wait_states(p)
                m -= 1


# ---------------------------------------------------------------------------------------------------------------------


def lint_kernels(kernels, pattern=None, rules="abcd", verbose=False):
    findings, stats = [], defaultdict(int)
    for name, ins in kernels.items():
        if pattern and not re.search(pattern, name):
            continue
        if not ins or not any(i.op == "s_endpgm" for i in ins):
            continue
        cfg = CFG(ins)
        stats["kernels"] += 1
        stats["instructions"] += len(ins)
        stats["ds_reads"] += sum(1 for i in ins if is_ds_result(i))
        stats["mfma"] += sum(1 for i in ins if mfma_info(i.op))
        if cfg.unknown:
            findings.append(("fatal", name, cfg.unknown[0], "indirect or unresolved control flow: the kernel cannot be analysed"))
            continue
        before = len(findings)
        if "a" in rules:
            lint_lds(name, cfg, findings)
        if "b" in rules:
            lint_mfma(name, cfg, findings)
        if "c" in rules:
            lint_vmcnt_loops(name, cfg, findings)
        if "d" in rules:
            lint_permlane(name, cfg, findings)
        if verbose:
            print(f"  {name}: {len(ins)} instructions, {len(cfg.blocks)} blocks, {len(findings) - before} findings", file=sys.stderr)
    return findings, stats


def _lint_file(arg):
    """One input file -> (findings, stats): disassemble / parse / lint in a worker process."""
    path, pattern, rules, verbose = arg
    ks = parse_asm(open(path).read()) if path.endswith(".s") else parse_objdump(disassemble(path))
    return lint_kernels(ks, pattern, rules, verbose)


def lint_files(paths, tmp, pattern=None, rules="abcd", verbose=False, jobs=1):
    """Lint every kernel of the given inputs (.so / .o: the gfx950 code objects inside; .elf; hipcc -S text)."""
    files = []
    for p in paths:
        files += code_objects_of(p, tmp) if p.endswith((".so", ".o")) else [p]
    files.sort(key=lambda f: -os.path.getsize(f))
    findings, stats = [], defaultdict(int)
    args = [(f, pattern, rules, verbose) for f in files]
    if jobs > 1 and len(files) > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(min(jobs, len(files))) as ex:
            results = list(ex.map(_lint_file, args))
    else:
        results = [_lint_file(x) for x in args]
    for f, st in results:
        findings += f
        for k, v in st.items():
            stats[k] += v
    return findings, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("inputs", nargs="*")
    ap.add_argument("--so", default=None)
    ap.add_argument("--kernel", default=None, help="regex: lint only the kernels whose mangled name matches")
    ap.add_argument("--rules", default="abcd")
    ap.add_argument("--max-print", type=int, default=40)
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("-j", "--jobs", type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args()
    inputs = list(a.inputs)
    if a.so or not inputs:
        inputs.append(a.so or os.path.join(ROOT, "audio_diffusion_amd", "libadk.so"))
    with tempfile.TemporaryDirectory() as tmp:
        findings, stats = lint_files(inputs, tmp, a.kernel, a.rules, a.verbose, a.jobs)
    by = defaultdict(int)
    for rule, name, i, msg in findings[:a.max_print]:
        where = f"{i.addr:#x}: {i.text}" if i is not None else ""
        print(f"[{rule}] {name}\n      {where}\n      {msg}")
    for rule, *_ in findings:
        by[rule] += 1
    print(f"isa_lint: {stats['kernels']} kernels, {stats['instructions']} instructions, {stats['ds_reads']} LDS reads, "
          f"{stats['mfma']} MFMAs; findings: " + (", ".join(f"rule ({k}) {v}" for k, v in sorted(by.items())) or "none"))
    if by.get("fatal"):
        return 2
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
