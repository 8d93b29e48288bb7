"""tools/isa_lint.py: the ISA lint of the hand-scheduled kernels (CPU only: it reads machine code, runs nothing).

  * every rule fires on a small hand-written instruction stream that breaks it and stays quiet on the repaired stream;
  * the library that SHIPS (audio_diffusion_amd/libadk.so, every gfx950 kernel in it) is clean;
  * positive controls on real code: the fused conv + GroupNorm unit compiled with every counted fragment wait one too
    weak (-DADK_LINT_CONTROL_WEAK_WAIT) must be red under rule (a); the same unit with the compiler's own s_nop pads
    deleted must be red under rules (b) and (d), and with every pad shortened by ONE wait state still red under (b) -
    which pins the lint's wait-state table to the compiler's (exact, not merely sufficient)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint as L  # noqa: E402

SO = os.path.join(ROOT, "audio_diffusion_amd", "libadk.so")
CSRC = os.path.join(ROOT, "audio_diffusion_amd", "csrc")


def kernel(body):
    return f"\t.amdhsa_kernel _Z1kv\n\n_Z1kv:\n; %bb.0:\n{body}\n\ts_endpgm\n.Lfunc_end0:\n"


def rules_hit(body):
    findings, _ = L.lint_kernels(L.parse_asm(kernel(body)))
    return sorted({f[0] for f in findings})


def test_rule_a_lds_result_in_flight():
    # B fragment consumed with one DS operation too many still behind it
    bad = """
\tds_read_b128 v[10:13], v1
\tds_read_b128 v[14:17], v1 offset:1024
\tds_read_b128 v[18:21], v1 offset:2048
\ts_waitcnt lgkmcnt(2)
\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[14:17], v[100:103]"""
    assert rules_hit(bad) == ["a"]
    assert rules_hit(bad.replace("lgkmcnt(2)", "lgkmcnt(1)")) == []
    # a scalar load shares the counter but returns out of order: it never helps an LDS read retire
    smem = """
\tds_read_b128 v[10:13], v1
\ts_load_dwordx2 s[4:5], s[0:1], 0x0
\ts_waitcnt lgkmcnt(1)
\tv_add_f32_e32 v2, v10, v10"""
    assert rules_hit(smem) == ["a"]
    assert rules_hit(smem.replace("lgkmcnt(1)", "lgkmcnt(0)")) == []
    # a pending destination handed to another value (the late write-back would land in it), and spilled
    assert rules_hit("\tds_read_b128 v[10:13], v1\n\tv_mov_b32_e32 v11, 0\n\ts_waitcnt lgkmcnt(0)") == ["a"]
    assert rules_hit("\tds_read_b128 v[10:13], v1\n\tscratch_store_dwordx4 off, v[10:13], off\n\ts_waitcnt lgkmcnt(0)") == ["a"]
    # across a loop back edge: the read issued at the bottom of the body is consumed at its top without a wait
    loop = """
\tds_read_b128 v[10:13], v1
.LBB0_1:
\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[20:23], v[100:103]
\ts_waitcnt lgkmcnt(0)
\tds_read_b128 v[10:13], v1 offset:1024
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt lgkmcnt(0)"""
    assert rules_hit(loop) == ["a"]
    assert rules_hit(loop.replace(".LBB0_1:\n", ".LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n")) == []
    # stores ahead of the read do not count as "younger"; stores behind it do
    assert rules_hit("\tds_write_b32 v3, v4\n\tds_read_b32 v10, v1\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v2, v10") == ["a"]
    assert rules_hit("\tds_read_b32 v10, v1\n\tds_write_b32 v3, v4\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v2, v10") == []


def test_rule_b_mfma_wait_states():
    # SrcC overlapping, not identical: 4-pass shape needs 6 states
    bad = """
\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[14:17], v[100:103]
\ts_nop 3
\tv_mfma_f32_16x16x32_f16 v[102:105], v[10:13], v[14:17], v[102:105]"""
    assert rules_hit(bad) == ["b"]
    assert rules_hit(bad.replace("s_nop 3", "s_nop 5")) == []
    # the accumulate chain itself (identical registers) needs none
    assert rules_hit("\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[14:17], v[100:103]\n"
                     "\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[18:21], v[100:103]") == []
    # a VALU read of an 8-pass result: 12 states ("8-pass XDL: 12 states", cdna_hip_programming.md 5.7)
    valu = "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n\ts_nop 10\n\tv_add_f32_e32 v40, v3, v3"
    assert rules_hit(valu) == ["b"]
    assert rules_hit(valu.replace("s_nop 10", "s_nop 11")) == []
    # the exact-f32 shape (16 passes, not an XDL op): 18 states before a store reads it
    f32 = "\tv_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]\n\ts_nop 15\n\ts_nop 0\n\tglobal_store_dword v[30:31], v5, off"
    assert rules_hit(f32) == ["b"]
    assert rules_hit(f32.replace("s_nop 0", "s_nop 1")) == []
    # along a branch: the taken path is short of wait states, the fall-through is padded
    br = """
\tv_mfma_f32_16x16x32_f16 v[100:103], v[10:13], v[14:17], v[100:103]
\ts_cbranch_scc1 .LBB0_2
\ts_nop 7
.LBB0_2:
\tv_add_f32_e32 v40, v101, v101"""
    assert rules_hit(br) == ["b"]


def test_rule_c_counted_vmcnt_loop():
    loop = """
.LBB0_1:
\tglobal_load_lds_dwordx4 v[2:3], off
\ts_waitcnt vmcnt(2)
\ts_barrier
%s
\ts_cbranch_scc1 .LBB0_1"""
    assert rules_hit(loop % "\tscratch_load_dword v9, off, off offset:4\n\ts_waitcnt vmcnt(0)") == ["c"]
    assert rules_hit(loop % "\tv_add_f32_e32 v9, v9, v9") == []
    assert rules_hit(loop.replace("vmcnt(2)", "vmcnt(0)") % "\tglobal_load_dword v9, v[4:5], off") == []      # nothing is counted there


def test_rule_d_permlane_swap_operand():
    bad = "\tv_cvt_pk_bf16_f32 v4, v8, v9\n\ts_nop 0\n\tv_permlane32_swap_b32_e32 v4, v5"
    assert rules_hit(bad) == ["d"]
    assert rules_hit(bad.replace("s_nop 0", "s_nop 1")) == []


def test_shipped_library_is_clean():
    assert os.path.exists(SO), "build libadk.so first (make -C audio_diffusion_amd/csrc)"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), "--so", SO], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) kernels, (\d+) instructions, (\d+) LDS reads, (\d+) MFMAs; findings: none", r.stdout)
    assert m and int(m.group(1)) > 400 and int(m.group(3)) > 10000 and int(m.group(4)) > 10000, r.stdout[-500:]


@pytest.fixture(scope="module")
def fused_unit_asm(tmp_path_factory):
    """conv_mfma_gn_f16.hip -> assembly, as shipped and with the weak-wait control switch (two hipcc runs, in parallel)."""
    d = tmp_path_factory.mktemp("lintasm")
    base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
            "--cuda-device-only", os.path.join(CSRC, "conv_mfma_gn_f16.hip")]
    procs = [subprocess.Popen(base + ["-o", str(d / "clean.s")], stderr=subprocess.DEVNULL),
             subprocess.Popen(base + ["-DADK_LINT_CONTROL_WEAK_WAIT", "-o", str(d / "weak.s")], stderr=subprocess.DEVNULL)]
    assert all(p.wait() == 0 for p in procs)
    return d


def count(findings):
    out = {}
    for f in findings:
        out[f[0]] = out.get(f[0], 0) + 1
    return out


def test_positive_controls_on_the_fused_conv_unit(fused_unit_asm):
    d = fused_unit_asm
    only = "conv_mfma_gn_kernelIDF16_Li5ELi4E"             # the 128 x 256 tile on the hand-scheduled 16x16x32 loop, 5 taps
    clean_text = open(d / "clean.s").read()
    clean, st = L.lint_kernels(L.parse_asm(clean_text), only)
    assert clean == [] and st["kernels"] == 5 and st["ds_reads"] > 500
    weak, _ = L.lint_kernels(L.parse_asm(open(d / "weak.s").read()), only)
    assert count(weak).get("a", 0) >= 100 and set(count(weak)) == {"a"}, count(weak)
    # the compiler's own pads deleted: MFMA wait states and the permlane-swap operand rule must both fire
    nonop = "\n".join(l for l in clean_text.splitlines() if not re.match(r"\s*s_nop\b", l))
    c = count(L.lint_kernels(L.parse_asm(nonop))[0])       # (all tiles: the 32x32x16 loops of the smaller ones carry the MFMA pads)
    assert c.get("b", 0) >= 10 and c.get("d", 0) >= 10 and "a" not in c, c
    # every pad ONE wait state shorter: still red under (b) -> the table is the compiler's, not a looser one
    shorter = re.sub(r"(?m)^(\s*)s_nop (\d+)\s*$", lambda m: "" if m.group(2) == "0" else f"{m.group(1)}s_nop {int(m.group(2)) - 1}", clean_text)
    c = count(L.lint_kernels(L.parse_asm(shorter))[0])
    assert c.get("b", 0) >= 1, c
