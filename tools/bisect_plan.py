#!/usr/bin/env python3
"""Which op of the denoiser plan gives run-to-run different results?  Runs the plan one op at a time (synchronising after
each), checksums the whole arena + output after every op, and does it again: the first op whose checksum differs between
passes is the culprit.  usage: python tools/bisect_plan.py [passes] [B] [chunk]   (chunk: ops per launch call, default 1)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_diffusion_amd as A
from audio_diffusion_amd import _lib
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.DiffusionAttnUnet1D(n_attn_layers=4).to(dev).eval()
m.compute_dtype = torch.float16
x = torch.randn(B, 2, 65536, device=dev)
t = torch.full((B,), 0.5, device=dev)
plan = m.plan_for(B, 65536, torch.float16, dev)
out = torch.empty_like(x)
plan.bind(x.data_ptr(), t.data_ptr(), out.data_ptr())
st = torch.cuda.current_stream().cuda_stream
lib = _lib.load()
n = plan.n_ops
arr_t = type(plan.ops)
ref = None
for p in range(passes):
    sums = []
    for i in range(0, n, chunk):
        k = min(chunk, n - i)
        sub = (_lib.AdkOp * k)(*[plan.ops[i + j] for j in range(k)])
        _lib.check(lib.adk_run_plan(sub, k, C.c_void_p(st)), "run")
        torch.cuda.synchronize()
        op = plan.ops[i + k - 1]
        base = plan.arena.data_ptr()
        y = op.y or 0
        if (op.kind == _lib.OP_CONV or i + k - 1 == 1) and base <= y < base + plan.arena.numel():
            planes = (op.Cout + 31) // 32 if op.kind == _lib.OP_CONV else 1
            nb = planes * op.B * ((op.L or 65536) + 2 * (op.P or 2)) * 64
            off = (y - base) // 8 * 8
            sums.append(int(plan.arena[off: off + nb // 8 * 8].view(torch.int64).sum().item()))
        else:
            sums.append(int(out.view(torch.int32).sum().item()) if i + k == n else 0)
    err = plan.gn_error()
    if p == 0:
        print(f"pass 0 (warm-up: the arena now holds a whole forward's leftovers): {n} ops, error word {err:#x}")
        continue
    if ref is None:
        ref = sums
        continue
    first = next((i for i in range(len(sums)) if sums[i] != ref[i]), None)
    if first is None:
        print(f"pass {p}: identical, error word {err:#x}")
    else:
        op = plan.ops[first * chunk]
        print(f"pass {p}: first difference after op group {first} (op {first * chunk}): kind {op.kind} flags {op.flags:#x} B {op.B} L {op.L} Cin {op.Cin} Cout {op.Cout} "
              f"taps {op.taps} i1 {op.i1} i2 {op.i2}; error word {err:#x}; differing groups: {sum(1 for i in range(len(sums)) if sums[i] != ref[i])}")
