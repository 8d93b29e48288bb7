#!/usr/bin/env python3
"""Per-block timeline of the pipelined conv kernel (experiment build only): start, end of the K loop, end of the
epilogue's bias wait / store issue / reduction, block end and the CU it ran on, for the three heavy shapes of config 2.
Build the instrumented library first (it is NOT the shipped one):
    make -C audio_diffusion_amd/csrc exp-ts        # -> .exp/libadk_TS.so  (-DADK_ABL_TS)
    ADK_LIB=$PWD/.exp/libadk_TS.so python tools/conv_block_timeline.py
Prints, per shape: K-loop and epilogue time per block, the split of the epilogue, and the share of time a CU has 0 / 1 / 2
blocks inside their K loops."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_diffusion_amd import _lib, layout
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
def run(cin, L, B=16):
    dtype, adk, P, taps = torch.bfloat16, 1, 2, 5
    cb = 32
    x = layout.to_cpl(torch.randn(B, cin, L, device=dev), dtype, P)
    cp = cin
    w = (torch.randn(cin // cb, taps, cp, cb, device=dev) / (cin * taps) ** 0.5).to(dtype)
    bias = torch.randn(cp, device=dev)
    y = torch.empty(cin // cb, B, L + 2 * P, cb, dtype=dtype, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cout, op.CoutPad, op.Cin, op.B, op.taps = _lib.OP_CONV, adk, L, cin, cp, cin, B, taps
    part = torch.zeros(B, lib.adk_partials_count(C.byref(op)), 2, device=dev)
    st = torch.cuda.current_stream()
    for _ in range(5):
        rc = lib.adk_conv1d(adk, 0, B, L, P, cin, cin, cp, taps, x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), part.data_ptr(), C.c_void_p(st.cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    nblk = B * (L // 256) * (cin // 128)
    n = min(nblk, 16384)
    buf = np.zeros(16384 * 8, dtype=np.uint64)
    raw.adk_exp_ts.argtypes = [C.c_void_p, C.c_size_t]
    assert raw.adk_exp_ts(buf.ctypes.data, buf.nbytes) == 0
    ts = buf.reshape(-1, 8)[:n]
    t0 = ts[:, 0].min()
    start, kend, end = [(ts[:, i] - t0).astype(np.float64) / 100.0 for i in range(3)]   # us (100 MHz)
    hw = ts[:, 3]
    xcc = (hw >> np.uint64(32)).astype(np.int64); hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"C={cin} L={L}: {nblk} blocks, total {end.max():.1f} us; distinct CUs {len(set(cuid.tolist()))}")
    print(f"  per block: K-loop {np.mean(kend - start):.2f} us (p10 {np.percentile(kend-start,10):.2f}, p90 {np.percentile(kend-start,90):.2f}), epilogue+drain {np.mean(end - kend):.2f} us (p10 {np.percentile(end-kend,10):.2f}, p90 {np.percentile(end-kend,90):.2f})")
    tb, tst, tsum = [(ts[:, i] - t0).astype(np.float64) / 100.0 for i in (4, 5, 6)]
    print(f"  epilogue split: bias wait {np.mean(tb - kend):.2f}  compute+store issue {np.mean(tst - tb):.2f}  halo+block_sum {np.mean(tsum - tst):.2f}  drain {np.mean(end - tsum):.2f} us")
    # per CU: fraction of time with 0/1/2 blocks in K-loop
    tot = np.zeros(3); T = end.max()
    for c in list(set(cuid.tolist()))[:64]:
        sel = cuid == c
        ev = sorted([(s, 1) for s in start[sel]] + [(e, -1) for e in kend[sel]])
        cur, last = 0, 0.0
        for tme, d in ev:
            tot[min(cur, 2)] += tme - last; last = tme; cur += d
        tot[0] += T - last
    print("  time share with 0/1/2 blocks of a CU in their K-loop:", np.round(tot / tot.sum(), 3))
    # lockstep: histogram of start times modulo the mean block period
    order = np.argsort(start)
    print("  start times of the first 12 blocks on one CU:", np.round(np.sort(start[cuid == cuid[0]])[:12], 1))
    print("  K-loop end of those:", np.round(np.sort(kend[cuid == cuid[0]])[:12], 1))
    print("  end of those:", np.round(np.sort(end[cuid == cuid[0]])[:12], 1))
run(128, 65536)
run(256, 16384)
run(512, 4096)
