#!/usr/bin/env python3
"""Per-block timeline of the fused conv + GroupNorm kernel (experiment build only): block start, ticket, end of the K
loop, statistics published, sample complete (wait over), stores issued, block end, and the CU it ran on.
    make -C audio_diffusion_amd/csrc exp-ts-gn        # -> .exp/libadk_TSGN.so
    ADK_LIB=$PWD/.exp/libadk_TSGN.so python tools/gn_block_timeline.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_amd import _lib, layout
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")


def run(cin, L, B=16, res=False):
    dtype, adk, P, taps, cb = torch.float16, 2, 2, 5, 32
    x = layout.to_cpl(torch.randn(B, cin, L, device=dev), dtype, P)
    w = (torch.randn(cin // cb, taps, cin, cb, device=dev) / (cin * taps) ** 0.5).to(dtype)
    bias, gamma, beta = torch.randn(cin, device=dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    y = torch.empty(cin // cb, B, L + 2 * P, cb, dtype=dtype, device=dev)
    r = layout.to_cpl(torch.randn(B, cin, L, device=dev), dtype, P) if res else None
    state = torch.zeros(4100, dtype=torch.int32, device=dev)
    ws = torch.zeros(B * 512 * 2, dtype=torch.int64, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, adk, _lib.FLAG_FUSE_GN | _lib.FLAG_GELU
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps, op.f0, op.i1 = B, L, P, cin, cin, cin, taps, 1e-5, 3
    op.x, op.w, op.bias, op.gamma, op.beta, op.y = x.data_ptr(), w.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr()
    op.residual = r.data_ptr() if res else None
    op.aux0, op.aux1 = state.data_ptr(), ws.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(5):
        assert lib.adk_step_advance(state.data_ptr(), st) == 0
        if it == 4:
            e0.record()
        assert lib.adk_run_plan(C.byref(op), 1, st) == 0, lib.adk_last_error()
        if it == 4:
            e1.record()
    torch.cuda.synchronize()
    assert int(state[1].item()) == 0
    bm = 256 if L >= 2048 else (128 if L > 256 else 64)      # conv_tile() of conv_mfma.hip
    nblk = B * ((L + bm - 1) // bm) * (cin // 128)
    n = min(nblk, 16384)
    buf = np.zeros(16384 * 9, dtype=np.uint64)
    raw.adk_exp_ts_gn.argtypes = [C.c_void_p, C.c_size_t]
    assert raw.adk_exp_ts_gn(buf.ctypes.data, buf.nbytes) == 0
    ts = buf[:16384 * 8].reshape(-1, 8)[:n]
    kstart_raw = buf[16384 * 8:][:n]
    t0 = ts[:, 0].min()
    us = lambda v: (v - t0).astype(np.float64) / 100.0
    start, kend, end, pub, waited, stored = us(ts[:, 0]), us(ts[:, 1]), us(ts[:, 2]), us(ts[:, 4]), us(ts[:, 5]), us(ts[:, 6])
    kstart = us(kstart_raw)
    ticket = ts[:, 7].astype(np.int64)
    hw = ts[:, 3]
    xcc = (hw >> np.uint64(32)).astype(np.int64); hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    npb = nblk // B
    print(f"C={cin} L={L} res={res}: {nblk} blocks ({npb} per sample), event time {e0.elapsed_time(e1) * 1e3:.1f} us, last block end {end.max():.1f} us, CUs {len(set(cuid.tolist()))}")
    m = lambda v: f"{np.mean(v):6.2f} (p10 {np.percentile(v, 10):5.2f}, p90 {np.percentile(v, 90):5.2f})"
    print("  ticket + prologue start :", m(kstart - start))
    print("  K loop                  :", m(kend - kstart))
    print("  tile statistics         :", m(pub - kend))
    print("  wait for the sample     :", m(waited - pub))
    print("  normalise + stores      :", m(stored - waited))
    print("  drain                   :", m(end - stored))
    print("  block lifetime          :", m(end - start))
    tot = np.zeros(3); T = end.max()
    for c in list(set(cuid.tolist()))[:64]:
        sel = cuid == c
        ev = sorted([(s, 1) for s in kstart[sel]] + [(e, -1) for e in kend[sel]])
        cur, last = 0, 0.0
        for tme, d in ev:
            tot[min(cur, 2)] += tme - last; last = tme; cur += d
        tot[0] += T - last
    print("  time share with 0/1/2 blocks of a CU in their K loop:", np.round(tot / tot.sum(), 3))
    sel = np.where(cuid == cuid[0])[0]
    o = sel[np.argsort(start[sel])][:10]
    for nm, v in (("start", start), ("K end", kend), ("published", pub), ("wait over", waited), ("end", end)):
        print(f"  one CU, first 10 blocks, {nm:10s}:", np.round(v[o], 1))
    print("  their tickets / samples:", ticket[o], ticket[o] // npb)
    for s in sorted({0, min(1, B - 1), min(2, B - 1), B - 1}):
        q = ticket // npb == s
        print(f"  sample {s}: starts {start[q].min():.1f}..{start[q].max():.1f}, published {pub[q].min():.1f}..{pub[q].max():.1f}, ends {end[q].min():.1f}..{end[q].max():.1f}")


if "--solo" in sys.argv:      # K loop of a block that has its CU to itself (256 blocks) / shares it with one other (512)
    run(128, 65536, B=1)
    run(128, 65536, B=2)
    run(256, 16384, B=2)
    run(256, 16384, B=4)
    run(512, 4096, B=4)
    run(512, 4096, B=8)
elif "--mid" in sys.argv:     # one-generation launches of the middle levels (C = 512, 128 x 128 / 128 x 64 tiles)
    run(512, 1024)
    run(512, 512)
    run(512, 256)
else:
    run(128, 65536)
    run(128, 65536, res=True)
    run(256, 16384)
    run(512, 4096)
