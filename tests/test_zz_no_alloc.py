"""adk_group_norm_noalloc: the ONE entry point of libadk that takes no workspace and allocates nothing (the GroupNorm
statistics partials live in a file-scope device array inside the library, include/adk.h).  Called through the C ABI and
checked bit for bit against the existing pair adk_gn_stats + adk_gn_apply on a caller-owned partials buffer, on two
calls in a row, the second with a smaller input (the array keeps the first call's contents), and against torch's fp32
group_norm.  Collected last (file name) so a failure here hides no other test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

DT = {"f32": (torch.float32, 0, 1e-5), "bf16": (torch.bfloat16, 1, 1.5e-2), "f16": (torch.float16, 2, 2e-3)}


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_case(lib, _lib, layout, dev, dt, B, Cc, L, gelu, res, seed, check_mem=True):
    dtype, adk, tol = DT[dt]
    P = 2
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, L, generator=g) * 1.7 + 0.3
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    r = torch.randn(B, Cc, L, generator=g) if res else None
    xd = layout.to_cpl(x.to(dev), dtype, P)
    rd = layout.to_cpl(r.to(dev), dtype, P) if res else None
    cpad = (Cc + 31) // 32 * 32
    gd, bd = torch.zeros(cpad, device=dev), torch.zeros(cpad, device=dev)
    gd[:Cc], bd[:Cc] = gamma.to(dev), beta.to(dev)
    flags = _lib.FLAG_GELU if gelu else 0
    # existing entry points, caller-owned workspace
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cin = _lib.OP_GN_STATS, adk, L, Cc
    n = lib.adk_partials_count(C.byref(op))
    part = torch.zeros(B, n, 2, device=dev)
    want = torch.full_like(xd, 3.0)
    assert lib.adk_gn_stats(adk, B, L, P, Cc, xd.data_ptr(), part.data_ptr(), sp()) == 0, lib.adk_last_error()
    assert lib.adk_gn_apply(adk, flags, B, L, P, Cc, 1e-5, xd.data_ptr(), part.data_ptr(), n, gd.data_ptr(), bd.data_ptr(),
                            rd.data_ptr() if res else None, want.data_ptr(), None, sp()) == 0, lib.adk_last_error()
    # the allocation-free entry point
    got = torch.full_like(xd, 7.0)
    free_before = torch.cuda.mem_get_info(dev)[0]
    rc = lib.adk_group_norm_noalloc(adk, flags, B, L, P, Cc, 1e-5, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                    rd.data_ptr() if res else None, got.data_ptr(), sp())
    assert rc == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    if check_mem:                                                # (the first call on a device is the set-up: the
        assert torch.cuda.mem_get_info(dev)[0] >= free_before    # runtime then loads the code object that holds the array)
    assert torch.equal(got, want), "no-alloc GroupNorm differs from adk_gn_stats + adk_gn_apply"
    xq = x.to(dtype).float()
    ref = F.group_norm(xq, 1, gamma, beta, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.to(dtype).float()
    cb = layout.chunk(dtype)
    full = layout.from_cpl(got, got.shape[0] * cb).cpu()
    assert rel_err(full[:, :Cc, P:P + L], ref) <= tol
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_no_alloc_group_norm_two_calls_in_a_row(dt):
    from audio_diffusion_amd import _lib, layout
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    run_case(lib, _lib, layout, dev, dt, B=1, Cc=32, L=64, gelu=False, res=False, seed=0, check_mem=False)   # set-up call
    # then: the larger input (many partials), and smaller ones that reuse the front of the same array; none of these
    # calls may take device memory
    run_case(lib, _lib, layout, dev, dt, B=3, Cc=128, L=20000, gelu=True, res=True, seed=1)
    run_case(lib, _lib, layout, dev, dt, B=2, Cc=96, L=300, gelu=False, res=False, seed=2)
    run_case(lib, _lib, layout, dev, dt, B=1, Cc=32, L=8, gelu=True, res=False, seed=3)


def test_no_alloc_rejects_oversize_and_bad_arguments():
    from audio_diffusion_amd import _lib
    lib = _lib.load()
    p = C.c_void_p(256)                                        # never dereferenced: validation fails first
    # B * partials beyond the built-in array: 300 samples x (512/16 f32 planes x 65536/1024 row blocks = 2048) pairs
    rc = lib.adk_group_norm_noalloc(0, 0, 300, 65536, 2, 512, 1e-5, p, p, p, None, p, sp())
    assert rc == -1 and b"built-in scratch" in lib.adk_last_error()
    assert lib.adk_group_norm_noalloc(1, 0, 2, 64, 2, 64, 1e-5, None, p, p, None, p, sp()) == -1     # null x
    assert lib.adk_group_norm_noalloc(1, 0, 2, 64, 2, 64, 0.0, p, p, p, None, p, sp()) == -1        # eps
