"""GPU parity tests, one libadk entry point at a time, called through the C ABI (ctypes on raw device
pointers).  Reference = plain PyTorch fp32 ops on the CPU (the same ATen kernels the reference runs) or
the golden vectors captured from the imported reference.

Tolerances (scale-relative max error, conftest.rel_err): f32 kernels 1e-5 (north_star budget: 1e-4 per
whole denoiser call), bf16 kernels 1.5e-2 on a single op (inputs/outputs rounded to 8 mantissa bits), f16 kernels
2e-3 (11 bits).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, sub_sd

pytestmark = pytest.mark.gpu

DT = {"f32": (torch.float32, 0, 1e-5), "bf16": (torch.bfloat16, 1, 1.5e-2), "f16": (torch.float16, 2, 2e-3)}


@pytest.fixture(scope="module")
def env():
    from audio_diffusion_amd import _lib, layout
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    lib = _lib.load()
    return lib, _lib, layout, torch.device("cuda:0")


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_w(w, cin_alloc, cout_pad, cb, dtype, dev):
    co, ci, k = w.shape
    wp = torch.zeros(cout_pad, cin_alloc, k)
    wp[:co, :ci] = w
    return wp.view(cout_pad, cin_alloc // cb, cb, k).permute(1, 3, 0, 2).contiguous().to(dev, dtype)


def quant(x, dtype):
    return x.to(dtype).float()


CONV_CASES = [
    # B, L, Cin, Cout, taps, bias, residual
    (2, 256, 32, 128, 5, True, False),
    (2, 200, 64, 96, 5, True, False),      # ragged L (masked tile), Cout not a tile multiple
    (3, 64, 128, 128, 5, True, True),      # BM=64 variant
    (2, 8, 64, 128, 5, True, False),       # BM=32 variant, L < halo-extended tile
    (2, 24, 96, 64, 1, False, True),       # 1x1, no bias, residual
    (2, 384, 256, 2, 5, True, False),      # narrow-output variant (co tile 32)
    (1, 1024, 32, 32, 1, True, False),
    (8, 16384, 32, 128, 5, True, False),   # wide (256-position) tile variant
    (9, 16500, 64, 256, 5, True, True),    # wide tile, ragged last tile, 2 co tiles
    # short levels: K split over the 4 waves (conv_ksplit.hip): 32x32, 32x64 and 64x64 tiles
    (16, 8, 512, 512, 5, True, False),     # deepest level of the uncond model
    (1, 16, 32, 64, 5, True, True),        # one channel chunk: three waves contribute zeros
    (3, 40, 160, 96, 1, False, True),      # 5 chunks (not a multiple of 4 waves), ragged 64-position tile
    (2, 128, 1024, 512, 5, True, False),   # concat input (32 chunks), 64x64 tiles
    (2, 250, 96, 1536, 1, True, False),    # qkv projection shape, ragged last tile (pipelined 128x64 tile)
    (2, 100, 64, 128, 1, True, True),      # 1x1 at 64 < L <= 128: pipelined, where k=5 uses the k-split kernel
    (2, 100, 64, 128, 5, True, True),
    (3, 200, 128, 256, 5, True, False),    # L <= 256: pipelined 128x64 tile, ragged
    # kernel_size = 7 models and the 3-tap convolutions of the learned resamplers, halo P = 3
    (2, 2100, 64, 128, 7, True, True),
    (2, 300, 96, 160, 7, True, False),
    (4, 24, 128, 128, 7, True, False),
    (2, 4100, 64, 256, 3, True, False),
    (3, 100, 256, 128, 3, True, True),
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d(env, dt, case):
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    B, L, Cin, Cout, taps, has_bias, has_res = case
    P = max(2, taps // 2)
    cb = layout.chunk(dtype)
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, taps, generator=g) / (Cin * taps) ** 0.5
    bias = torch.randn(Cout, generator=g) if has_bias else None
    res = torch.randn(B, Cout, L, generator=g) if has_res else None
    xq, wq = quant(x, dtype), quant(w, dtype)
    ref = F.conv1d(xq, wq, bias, padding=taps // 2)
    if has_res:
        ref = ref + quant(res, dtype)
    bn = lib.adk_conv_cotile(adk, Cout, L)
    cout_pad = (Cout + bn - 1) // bn * bn
    cin_alloc = (Cin + cb - 1) // cb * cb
    xd = layout.to_cpl(x.to(dev), dtype, P)
    wd = pack_w(w, cin_alloc, cout_pad, cb, dtype, dev)
    bd = None
    if has_bias:
        bd = torch.zeros(cout_pad, device=dev)
        bd[:Cout] = bias.to(dev)
    rd = layout.to_cpl(res.to(dev), dtype, P) if has_res else None
    planes_out = (Cout + cb - 1) // cb
    yd = torch.full((planes_out, B, L + 2 * P, cb), 7.0, dtype=dtype, device=dev)   # poison incl. halo
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cout, op.CoutPad, op.Cin, op.B = _lib.OP_CONV, adk, L, Cout, cout_pad, cin_alloc, B
    op.taps = taps
    npart = lib.adk_partials_count(C.byref(op))
    part = torch.zeros(B, npart, 2, device=dev)
    rc = lib.adk_conv1d(adk, 0, B, L, P, cin_alloc, Cout, cout_pad, taps, xd.data_ptr(), wd.data_ptr(),
                        bd.data_ptr() if bd is not None else None, rd.data_ptr() if rd is not None else None,
                        yd.data_ptr(), part.data_ptr(), sp())
    assert rc == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    full = layout.from_cpl(yd, planes_out * cb).cpu()
    y = full[:, :Cout, P:P + L]
    assert rel_err(y, ref) <= tol
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0, "halo rows not zero"
    assert float(full[:, Cout:].abs().max() if full.shape[1] > Cout else 0.0) == 0.0, "pad channels not zero"
    # (sum, sumsq) partials describe the stored tensor (pre-rounding values in the bf16 path)
    s = part.double().sum(1).cpu()
    ref64 = ref.double()
    scale = ref64.abs().sum((1, 2))
    assert float(((s[:, 0] - ref64.sum((1, 2))).abs() / scale).max()) <= 1e-4
    assert float(((s[:, 1] - ref64.pow(2).sum((1, 2))).abs() / ref64.pow(2).sum((1, 2))).max()) <= 1e-4


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("Cin,Cout,L", [(32, 128, 65536), (128, 128, 32768), (256, 256, 8192)])
def test_conv1d_bits_and_partials_reproducible(env, dt, Cin, Cout, L):
    """A full-size launch (4096 / 2048 / 1024 blocks, every CU busy) repeated 4 times: output AND (sum, sumsq) partials
    must be bit-identical.  Regression test: an epilogue variant once produced run-to-run different sumsq partials in
    1-4 of 4096 blocks while the output stayed identical (see pack_bf16x2 in adk_common.h)."""
    lib, _lib, layout, dev = env
    dtype, adk, _ = DT[dt]
    B, P, taps, cb = 16, 2, 5, 32
    torch.manual_seed(L)
    xd = layout.to_cpl(torch.randn(B, Cin, L, device=dev), dtype, P)
    wd = (torch.randn(Cin // cb, taps, Cout, cb, device=dev) / (Cin * taps) ** 0.5).to(dtype)
    bd = torch.randn(Cout, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cout, op.CoutPad, op.Cin, op.B, op.taps = _lib.OP_CONV, adk, L, Cout, Cout, Cin, B, taps
    npart = lib.adk_partials_count(C.byref(op))
    outs = []
    for _ in range(4):
        yd = torch.zeros(Cout // cb, B, L + 2 * P, cb, dtype=dtype, device=dev)
        part = torch.zeros(B, npart, 2, device=dev)
        rc = lib.adk_conv1d(adk, 0, B, L, P, Cin, Cout, Cout, taps, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None,
                            yd.data_ptr(), part.data_ptr(), sp())
        assert rc == 0, lib.adk_last_error()
        torch.cuda.synchronize()
        outs.append((yd, part))
    for yd, part in outs[1:]:
        assert torch.equal(yd.view(torch.int16), outs[0][0].view(torch.int16)), "output bits differ between runs"
        assert torch.equal(part.view(torch.int32), outs[0][1].view(torch.int32)), "partials differ between runs"


def test_conv1d_rejects_bad_shapes(env):
    lib, _lib, layout, dev = env
    buf = torch.zeros(1 << 16, device=dev)
    p = buf.data_ptr()
    assert lib.adk_conv1d(0, 0, 1, 64, 2, 20, 32, 32, 5, p, p, None, None, p, None, sp()) == -1   # Cin not chunked
    assert b"Cin" in lib.adk_last_error()
    assert lib.adk_conv1d(0, 0, 1, 64, 1, 32, 32, 32, 5, p, p, None, None, p, None, sp()) == -1   # halo < taps/2
    assert lib.adk_conv1d(0, 0, 1, 64, 2, 32, 32, 32, 4, p, p, None, None, p, None, sp()) == -1   # taps
    assert lib.adk_conv1d(0, 0, 1, 64, 2, 32, 64, 64, 5, p, p, None, None, p, None, sp()) == -1   # CoutPad % 128
    assert lib.adk_conv1d(0, 0, 1, 64, 2, 32, 32, 32, 5, None, p, None, None, p, None, sp()) == -1


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,C,L,gelu,res", [(2, 64, 1500, True, True), (3, 96, 40, False, False), (1, 32, 4096, True, False)])
def test_gn_apply_and_stats(env, dt, B, C, L, gelu, res):
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    P = 2
    cb = layout.chunk(dtype)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, L, generator=g) * 1.7 + 0.3
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    r = torch.randn(B, C, L, generator=g) if res else None
    xq = quant(x, dtype)
    ref = F.group_norm(xq, 1, gamma, beta, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    if res:
        ref = ref + quant(r, dtype)
    xd = layout.to_cpl(x.to(dev), dtype, P)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cin = _lib.OP_GN_STATS, adk, L, C
    n = lib.adk_partials_count(C_byref(op))
    part = torch.zeros(B, n, 2, device=dev)
    assert lib.adk_gn_stats(adk, B, L, P, C, xd.data_ptr(), part.data_ptr(), sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    s = part.double().sum(1).cpu()
    assert torch.allclose(s[:, 0], xq.double().sum((1, 2)), rtol=1e-5, atol=1e-2)
    assert torch.allclose(s[:, 1], xq.double().pow(2).sum((1, 2)), rtol=1e-5)
    cpad = (C + 31) // 32 * 32
    gd, bd = torch.zeros(cpad, device=dev), torch.zeros(cpad, device=dev)
    gd[:C], bd[:C] = gamma.to(dev), beta.to(dev)
    rd = layout.to_cpl(r.to(dev), dtype, P) if res else None
    yd = torch.full_like(xd, 3.0)
    pout = torch.zeros(B, n, 2, device=dev)
    rc = lib.adk_gn_apply(adk, _lib.FLAG_GELU if gelu else 0, B, L, P, C, 1e-5, xd.data_ptr(), part.data_ptr(), n,
                          gd.data_ptr(), bd.data_ptr(), rd.data_ptr() if res else None, yd.data_ptr(), pout.data_ptr(), sp())
    assert rc == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    full = layout.from_cpl(yd, yd.shape[0] * cb).cpu()
    assert rel_err(full[:, :C, P:P + L], ref) <= tol
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0
    so = pout.double().sum(1).cpu()
    assert torch.allclose(so[:, 0], ref.double().sum((1, 2)), rtol=5e-3, atol=5e-3 * ref.abs().sum((1, 2)).max().item())
    assert torch.allclose(so[:, 1], ref.double().pow(2).sum((1, 2)), rtol=5e-3)
    # in place (y aliases x) gives the same bits
    rc = lib.adk_gn_apply(adk, _lib.FLAG_GELU if gelu else 0, B, L, P, C, 1e-5, xd.data_ptr(), part.data_ptr(), n,
                          gd.data_ptr(), bd.data_ptr(), rd.data_ptr() if res else None, xd.data_ptr(), None, sp())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(xd[:, :, P:P + L], yd[:, :, P:P + L])


def C_byref(op):
    return C.byref(op)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("kern", ["linear", "cubic", "lanczos3"])
@pytest.mark.parametrize("L", [8, 16, 4096])
def test_fir_resamplers_vs_golden(env, golden_blocks, dt, kern, L):
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    z = golden_blocks
    P = 2
    for d in ("down", "up"):
        if f"{d}_{kern}_{L}/x" not in z.files:
            continue
        x = torch.from_numpy(z[f"{d}_{kern}_{L}/x"])
        ref = torch.from_numpy(z[f"{d}_{kern}_{L}/y"])
        taps = torch.from_numpy(z[f"{d}_{kern}/kernel"]).to(dev)
        B, Cc, _ = x.shape
        xd = layout.to_cpl(x.to(dev), dtype, P)
        Lo = ref.shape[2]
        yd = torch.full((xd.shape[0], B, Lo + 2 * P, xd.shape[3]), 5.0, dtype=dtype, device=dev)
        fn = lib.adk_fir_down2 if d == "down" else lib.adk_fir_up2
        assert fn(adk, B, L, P, Cc, taps.numel(), taps.data_ptr(), xd.data_ptr(), yd.data_ptr(), sp()) == 0, lib.adk_last_error()
        torch.cuda.synchronize()
        full = layout.from_cpl(yd, Cc).cpu()
        if dt != "f32":      # compare against the stencil of the rounded input
            from oracle import unet_oracle as U
            xq = quant(x, dtype)
            ref = U.downsample_fir(xq, taps.cpu()) if d == "down" else U.upsample_fir(xq, taps.cpu())
        assert rel_err(full[:, :, P:P + Lo], ref) <= tol
        assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + Lo:].abs().max()) == 0.0


def test_fir_rejects_short_input(env):
    lib, _lib, layout, dev = env
    buf = torch.zeros(4096, device=dev)
    p = buf.data_ptr()
    # reference: F.pad(reflect) raises when pad >= length (cubic down: pad 3; up: pad 2)
    assert lib.adk_fir_down2(0, 1, 2, 2, 16, 8, p, p, p, sp()) == -1
    assert b"reflect" in lib.adk_last_error()
    assert lib.adk_fir_up2(0, 1, 2, 2, 16, 8, p, p, p, sp()) == -1
    assert lib.adk_fir_down2(0, 1, 16, 2, 16, 6, p, p, p, sp()) == -1


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("s", [8, 64, 256])
def test_attention_vs_golden(env, golden_blocks, dt, s):
    """Whole SelfAttention1d = gn_stats + gn_apply + conv1x1 + attn1d + conv1x1(+residual) vs the reference module."""
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    z = golden_blocks
    sd = sub_sd(z, f"attn{s}/sd/")
    x = torch.from_numpy(z[f"attn{s}/x"])
    ref = torch.from_numpy(z[f"attn{s}/y"])
    B, Cc, L = x.shape
    P, cb = 2, layout.chunk(dtype)
    xd = layout.to_cpl(x.to(dev), dtype, P)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.L, op.Cin = _lib.OP_GN_STATS, adk, L, Cc
    n = lib.adk_partials_count(C.byref(op))
    part = torch.zeros(B, n, 2, device=dev)
    assert lib.adk_gn_stats(adk, B, L, P, Cc, xd.data_ptr(), part.data_ptr(), sp()) == 0
    gd, bd = sd["norm.weight"].to(dev), sd["norm.bias"].to(dev)
    nd = torch.empty_like(xd)
    assert lib.adk_gn_apply(adk, 0, B, L, P, Cc, 1e-5, xd.data_ptr(), part.data_ptr(), n, gd.data_ptr(), bd.data_ptr(),
                            None, nd.data_ptr(), None, sp()) == 0
    wq = pack_w(sd["qkv_proj.weight"], Cc, 3 * Cc if 3 * Cc % 128 == 0 else (3 * Cc + 127) // 128 * 128, cb, dtype, dev)
    cp = wq.shape[2]
    bq = torch.zeros(cp, device=dev)
    bq[:3 * Cc] = sd["qkv_proj.bias"].to(dev)
    qkv = torch.empty(3 * Cc // cb, B, L + 2 * P, cb, dtype=dtype, device=dev)
    assert lib.adk_conv1d(adk, 0, B, L, P, Cc, 3 * Cc, cp, 1, nd.data_ptr(), wq.data_ptr(), bq.data_ptr(), None,
                          qkv.data_ptr(), None, sp()) == 0, lib.adk_last_error()
    yd = torch.empty_like(xd)
    assert lib.adk_attn1d(adk, B, L, P, Cc, qkv.data_ptr(), yd.data_ptr(), sp()) == 0, lib.adk_last_error()
    wo = pack_w(sd["out_proj.weight"], Cc, 128, cb, dtype, dev)
    bo = torch.zeros(128, device=dev)
    bo[:Cc] = sd["out_proj.bias"].to(dev)
    od = torch.empty_like(xd)
    assert lib.adk_conv1d(adk, 0, B, L, P, Cc, Cc, 128, 1, yd.data_ptr(), wo.data_ptr(), bo.data_ptr(), xd.data_ptr(),
                          od.data_ptr(), None, sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    y = layout.from_cpl(od, Cc).cpu()[:, :, P:P + L]
    assert rel_err(y, ref) <= (1e-5 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("s", [8, 64, 256])
def test_attention_block_fused_qkv_vs_golden(env, golden_blocks, dt, s):
    """ADK_FLAG_QKV: SelfAttention1d's norm -> qkv_proj -> attention core as ONE launch (+ the out_proj conv with the
    residual) vs the reference module's golden output; and bit-reproducible across two launches."""
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    z = golden_blocks
    sd = sub_sd(z, f"attn{s}/sd/")
    x = torch.from_numpy(z[f"attn{s}/x"])
    ref = torch.from_numpy(z[f"attn{s}/y"])
    B, Cc, L = x.shape
    P, cb = 2, layout.chunk(dtype)
    xd = layout.to_cpl(x.to(dev), dtype, P)
    sop = _lib.AdkOp()
    sop.kind, sop.dtype, sop.L, sop.Cin = _lib.OP_GN_STATS, adk, L, Cc
    n = lib.adk_partials_count(C.byref(sop))
    part = torch.zeros(B, n, 2, device=dev)
    assert lib.adk_gn_stats(adk, B, L, P, Cc, xd.data_ptr(), part.data_ptr(), sp()) == 0
    w, bq = sd["qkv_proj.weight"].float(), sd["qkv_proj.bias"].float()
    g, bt = sd["norm.weight"].float(), sd["norm.bias"].float()
    cout_pad = (3 * Cc + 127) // 128 * 128
    wg = w * g[None, :, None]
    wd = pack_w(wg, Cc, cout_pad, cb, dtype, dev)
    u, vb = torch.zeros(cout_pad), torch.zeros(cout_pad)
    u[:3 * Cc] = wg.to(dtype).float().sum((1, 2))
    vb[:3 * Cc] = bq + w[:, :, 0] @ bt
    u, vb = u.to(dev), vb.to(dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_ATTN, adk, _lib.FLAG_QKV
    op.B, op.L, op.P, op.Cin, op.CoutPad, op.i2, op.f0, op.n_partials_in = B, L, P, Cc, cout_pad, Cc, 1e-5, n
    op.x, op.w, op.bias, op.gamma, op.partials_in = xd.data_ptr(), wd.data_ptr(), vb.data_ptr(), u.data_ptr(), part.data_ptr()
    outs = []
    for _ in range(2):
        yd = torch.full_like(xd, 3.0)
        op.y = yd.data_ptr()
        assert lib.adk_run_plan(C.byref(op), 1, sp()) == 0, lib.adk_last_error()
        torch.cuda.synchronize()
        outs.append(yd)
    assert torch.equal(outs[0].view(torch.uint8), outs[1].view(torch.uint8))
    yd = outs[0]
    assert float(yd[:, :, :P].abs().max()) == 0.0 and float(yd[:, :, P + L:].abs().max()) == 0.0, "halo rows not zero"
    wo = pack_w(sd["out_proj.weight"], Cc, 128, cb, dtype, dev)
    bo = torch.zeros(128, device=dev)
    bo[:Cc] = sd["out_proj.bias"].to(dev)
    od = torch.empty_like(xd)
    assert lib.adk_conv1d(adk, 0, B, L, P, Cc, Cc, 128, 1, yd.data_ptr(), wo.data_ptr(), bo.data_ptr(), xd.data_ptr(),
                          od.data_ptr(), None, sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    y = layout.from_cpl(od, Cc).cpu()[:, :, P:P + L]
    assert rel_err(y, ref) <= 2e-2


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,heads,s", [(2, 2, 8), (3, 1, 100), (1, 3, 257), (2, 2, 1024), (1, 2, 4096), (1, 1, 1000)])
def test_attention_core(env, dt, B, heads, s):
    """adk_attn1d alone (softmax(q k^T / sqrt(32)) v per head) vs an fp32 CPU reference on the same (rounded)
    inputs: ragged key tiles, sequences from 8 to 4096 (the long-context config), both query-tile widths of
    the MFMA kernels (16-bit, and from s = 128 the exact-fp32 matrix-instruction kernel of the f32 path).  Scores are spread (|q.k| up to ~8) so the online-softmax rescale branch is exercised,
    and one key row per head is spiked to force a late running-max jump."""
    lib, _lib, layout, dev = env
    dtype, adk, _ = DT[dt]
    Cc, P, cb = heads * 32, 2, layout.chunk(dtype)
    g = torch.Generator().manual_seed(s * 7 + heads)
    qkv = torch.randn(B, 3 * Cc, s, generator=g) * 1.3
    qkv[:, Cc:2 * Cc, (3 * s) // 4] *= 4.0                      # a late key that dominates many rows
    qkvq = quant(qkv, dtype)
    q, k, v = [t.view(B, heads, 32, s).transpose(2, 3) for t in qkvq.split(Cc, dim=1)]
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).float().transpose(2, 3).reshape(B, Cc, s)
    qd = layout.to_cpl(qkv.to(dev), dtype, P)
    yd = torch.full((Cc // cb, B, s + 2 * P, cb), 5.0, dtype=dtype, device=dev)
    assert lib.adk_attn1d(adk, B, s, P, Cc, qd.data_ptr(), yd.data_ptr(), sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    full = layout.from_cpl(yd, Cc).cpu()
    y = full[:, :, P:P + s]
    assert rel_err(y, ref) <= (2e-5 if dt == "f32" else 1.5e-2)
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + s:].abs().max()) == 0.0, "halo rows not zero"


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,L,Cin,Cout", [(2, 8, 64, 192), (3, 64, 512, 1536), (2, 100, 96, 288), (2, 256, 512, 1536),
                                          (2, 1000, 64, 192), (1, 4096, 128, 384)])
def test_conv_with_folded_input_groupnorm(env, dt, B, L, Cin, Cout):
    """ADK_FLAG_PRE_NORM: conv1x1(GroupNorm(1, Cin)(x)) (SelfAttention1d's norm -> qkv_proj, blocks/blocks.py:107,115) as ONE
    conv on the raw x with the norm folded into weights / epilogue, vs the fp32 composition of the ATen ops; the k-split
    kernel (L <= 64) and the pipelined tiles (64, 128 and 256 positions).  x gets a mean of about one standard deviation."""
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    P, cb = 2, layout.chunk(dtype)
    g = torch.Generator().manual_seed(L + Cin)
    x = torch.randn(B, Cin, L, generator=g) * 1.5 + 1.2
    w = torch.randn(Cout, Cin, 1, generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    gamma, beta = torch.randn(Cin, generator=g) * 0.5 + 1.0, torch.randn(Cin, generator=g) * 0.2
    xq = quant(x, dtype)
    ref = F.conv1d(F.group_norm(xq, 1, gamma, beta, 1e-5), w, bias)
    cin_alloc = (Cin + cb - 1) // cb * cb
    bn = lib.adk_conv_cotile(adk, Cout, L)
    cout_pad = (Cout + bn - 1) // bn * bn
    wg = w * gamma[None, :, None]
    wd = pack_w(wg, cin_alloc, cout_pad, cb, dtype, dev)
    u, vb = torch.zeros(cout_pad), torch.zeros(cout_pad)
    u[:Cout] = wg.to(dtype).float().sum((1, 2))
    vb[:Cout] = bias + w[:, :, 0] @ beta
    u, vb = u.to(dev), vb.to(dev)
    xd = layout.to_cpl(x.to(dev), dtype, P)
    sop = _lib.AdkOp()
    sop.kind, sop.dtype, sop.L, sop.Cin = _lib.OP_GN_STATS, adk, L, Cin
    n = lib.adk_partials_count(C.byref(sop))
    part = torch.zeros(B, n, 2, device=dev)
    assert lib.adk_gn_stats(adk, B, L, P, Cin, xd.data_ptr(), part.data_ptr(), sp()) == 0
    planes = (Cout + cb - 1) // cb
    yd = torch.full((planes, B, L + 2 * P, cb), 3.0, dtype=dtype, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, adk, _lib.FLAG_PRE_NORM
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps = B, L, P, cin_alloc, Cout, cout_pad, 1
    op.i2, op.f0, op.n_partials_in = Cin, 1e-5, n
    op.x, op.w, op.bias, op.gamma, op.partials_in, op.y = (xd.data_ptr(), wd.data_ptr(), vb.data_ptr(), u.data_ptr(),
                                                          part.data_ptr(), yd.data_ptr())
    assert lib.adk_run_plan(C.byref(op), 1, sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    full = layout.from_cpl(yd, planes * cb).cpu()
    assert rel_err(full[:, :Cout, P:P + L], ref) <= (2e-5 if dt == "f32" else tol)
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0, "halo rows not zero"


FUSED_CASES = [
    # B, L, Cin, Cout, taps, residual, stats
    (3, 300, 64, 128, 5, False, False),     # 128x128 tiles, ragged, one round
    (2, 200, 32, 96, 5, True, True),        # 128x64 tile, Cout not a tile multiple (pad channels stay zero)
    (20, 16384, 32, 128, 5, True, False),   # 1280 tiles of 128x256: more tiles than block slots, 64 tiles per sample
    (2, 65536, 32, 128, 5, False, False),   # 256 tiles per sample (level 1 of the uncond model)
    (5, 4100, 64, 256, 3, True, True),      # two co tiles per position tile, ragged last tile, 3 taps
    (2, 1000, 96, 160, 7, True, False),     # 7 taps, halo 3
    # short levels: the k-split kernel (32x32, 32x64 and 64x64 tiles), grid of 256 blocks in co-tile-major order
    (16, 8, 512, 512, 5, True, True),       # deepest level of the uncond model
    (16, 64, 512, 512, 5, False, False),
    (16, 128, 1024, 512, 5, True, False),   # concat input, 64x64 tiles (one block per CU)
    (3, 40, 64, 96, 3, True, True),         # ragged 64-position tile, Cout not a tile multiple
    (40, 64, 128, 512, 5, True, False),     # 640 blocks > resident slots: tiles handed out by ticket
    # residual = an int: ResidualBlock's 1x1 skip conv over that many input channels, folded into the launch
    # (ADK_FLAG_SKIP_CONV, pipelined kernel)
    (3, 300, 64, 128, 5, 64, False),        # 128x128 tiles
    (2, 200, 32, 96, 5, 32, True),          # 128x64 tile, pad channels, output statistics
    (2, 65536, 32, 128, 5, 64, False),      # 128x256 tiles, 256 per sample
    (5, 4100, 64, 256, 3, 128, True),       # two co tiles, 3 taps, more skip chunks than main chunks
    (20, 4096, 128, 128, 5, 256, False),    # the up path's 2c -> c block: 8 skip chunks
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", FUSED_CASES, ids=[f"case{i}" for i in range(len(FUSED_CASES))])
def test_conv_groupnorm_gelu_fused(env, dt, case):
    """ADK_FLAG_FUSE_GN: conv -> GroupNorm(1, C) -> GELU (-> + residual) in one launch (the tiles of a sample exchange
    their statistics inside the launch) against the fp32 CPU composition of the same ATen ops, run twice with the
    forward counter advanced in between (bit-identical), and once more WITHOUT advancing it (must flag, not hang)."""
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    B, L, Cin, Cout, taps, res, stats = case
    P = taps // 2
    cb = layout.chunk(dtype)
    g = torch.Generator().manual_seed(100 + L + Cout)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, taps, generator=g) / (Cin * taps) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    gamma, beta = torch.randn(Cout, generator=g) * 0.5 + 1.0, torch.randn(Cout, generator=g) * 0.2
    skip_c = res if (res is not True and res is not False) else 0
    r = torch.randn(B, skip_c or Cout, L, generator=g) if res else None
    w2 = torch.randn(Cout, skip_c, 1, generator=g) / skip_c ** 0.5 if skip_c else None
    xq, wq = quant(x, dtype), quant(w, dtype)
    ref = F.gelu(F.group_norm(F.conv1d(xq, wq, bias, padding=P), 1, gamma, beta, 1e-5))
    if skip_c:
        ref = ref + F.conv1d(quant(r, dtype), quant(w2, dtype))
    elif res:
        ref = ref + quant(r, dtype)
    bn = lib.adk_conv_cotile(adk, Cout, L)
    cout_pad = (Cout + bn - 1) // bn * bn
    xd = layout.to_cpl(x.to(dev), dtype, P)
    wd = pack_w(w, Cin, cout_pad, cb, dtype, dev)
    bd, gd, btd = (torch.zeros(cout_pad, device=dev) for _ in range(3))
    bd[:Cout], gd[:Cout], btd[:Cout] = bias.to(dev), gamma.to(dev), beta.to(dev)
    rd = layout.to_cpl(r.to(dev), dtype, P) if res else None
    planes = (Cout + cb - 1) // cb
    state = torch.zeros(4096 + 4, dtype=torch.int32, device=dev)
    ws = torch.full((B * 512 * 2,), -1, dtype=torch.int64, device=dev)          # any contents
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, adk, _lib.FLAG_FUSE_GN | _lib.FLAG_GELU
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps = B, L, P, Cin, Cout, cout_pad, taps
    op.i1, op.f0 = 7, 1e-5
    assert lib.adk_conv_gn_fusable(C.byref(op)) == 1
    npart = ((L + lib.adk_conv_postile(C.byref(op)) - 1) // lib.adk_conv_postile(C.byref(op))) * \
        (cout_pad // lib.adk_conv_cotile_used(C.byref(op)))
    pout = torch.zeros(B, npart, 2, device=dev) if stats else None
    op.x, op.w, op.bias, op.gamma, op.beta = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), btd.data_ptr()
    op.residual = rd.data_ptr() if res else None
    if skip_c:
        w2d = pack_w(w2, skip_c, cout_pad, cb, dtype, dev)
        op.flags |= _lib.FLAG_SKIP_CONV
        op.w2, op.i2 = w2d.data_ptr(), skip_c
    op.partials_out = pout.data_ptr() if stats else None
    op.aux0, op.aux1 = state.data_ptr(), ws.data_ptr()
    outs = []
    for _ in range(2):
        yd = torch.full((planes, B, L + 2 * P, cb), 3.0, dtype=dtype, device=dev)
        op.y = yd.data_ptr()
        assert lib.adk_step_advance(state.data_ptr(), sp()) == 0
        assert lib.adk_run_plan(C.byref(op), 1, sp()) == 0, lib.adk_last_error()
        torch.cuda.synchronize()
        assert int(state[1].item()) == 0, "a tile waited in vain for its sample's statistics"
        outs.append(yd)
    assert torch.equal(outs[0].view(torch.uint8), outs[1].view(torch.uint8)), "reruns must be bit-identical"
    full = layout.from_cpl(outs[0], planes * cb).cpu()
    assert rel_err(full[:, :Cout, P:P + L], ref) <= (2e-5 if dt == "f32" else tol)
    assert float(full[:, Cout:].abs().max() if planes * cb > Cout else 0.0) == 0.0, "pad channels must stay zero"
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0, "halo rows not zero"
    if stats:
        so = pout.double().sum(1).cpu()
        assert torch.allclose(so[:, 0], ref.double().sum((1, 2)), rtol=5e-3, atol=5e-3 * ref.abs().sum((1, 2)).max().item())
        assert torch.allclose(so[:, 1], ref.double().pow(2).sum((1, 2)), rtol=5e-3)
    # the same op again without advancing the forward counter: where tiles go by ticket (every grid but a short-level
    # one that fits the GPU) every ticket is then out of range -> flagged, nothing hangs
    assert lib.adk_run_plan(C.byref(op), 1, sp()) == 0
    torch.cuda.synchronize()
    if L > 128 or B * npart > 256:
        assert int(state[1].item()) != 0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("cin", [32, 96, 128])
def test_fused_conv_reruns_identical_with_blocks_cycling(env, dt, cin):
    """B = 16 at L = 65536: 4096 tiles on 512 resident slots, so blocks start beside blocks that are in any phase of their
    life.  Odd channel-chunk counts (Cin = 32: the model's first conv; 96) run the tail of the 16x16x32 K loop, which is
    unrolled over chunk pairs.  A build whose inline-asm waits also named the accumulators gave different SAMPLES from run
    to run exactly here (missing MFMA wait states, conv_mainloop16.h) while every two-sample test passed.  Twelve launches
    must agree bit for bit, and with the fp32 composition of the ATen ops."""
    lib, _lib, layout, dev = env
    dtype, adk, tol = DT[dt]
    B, L, Cout, taps, P = 16, 65536, 128, 5, 2
    cb = layout.chunk(dtype)
    g = torch.Generator().manual_seed(7 + cin)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(Cout, cin, taps, generator=g) / (cin * taps) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.3
    gamma, beta = torch.randn(Cout, generator=g) * 0.5 + 1.0, torch.randn(Cout, generator=g) * 0.2
    xd = layout.to_cpl(x.to(dev), dtype, P)
    wd = pack_w(w, cin, Cout, cb, dtype, dev)
    bd, gd, btd = bias.to(dev), gamma.to(dev), beta.to(dev)
    state = torch.zeros(4096 + 4, dtype=torch.int32, device=dev)
    ws = torch.zeros(B * 512 * 2, dtype=torch.int64, device=dev)
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, adk, _lib.FLAG_FUSE_GN | _lib.FLAG_GELU
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps, op.i1, op.f0 = B, L, P, cin, Cout, Cout, taps, 5, 1e-5
    op.x, op.w, op.bias, op.gamma, op.beta = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), btd.data_ptr()
    op.aux0, op.aux1 = state.data_ptr(), ws.data_ptr()
    first = None
    for it in range(12):
        yd = torch.full((Cout // cb, B, L + 2 * P, cb), 3.0, dtype=dtype, device=dev)
        op.y = yd.data_ptr()
        assert lib.adk_step_advance(state.data_ptr(), sp()) == 0
        assert lib.adk_run_plan(C.byref(op), 1, sp()) == 0, lib.adk_last_error()
        torch.cuda.synchronize()
        assert int(state[1].item()) == 0
        if first is None:
            first = yd
        else:
            assert torch.equal(first.view(torch.int16), yd.view(torch.int16)), f"launch {it} differs from launch 0"
    ref = F.gelu(F.group_norm(F.conv1d(quant(x[:2], dtype), quant(w, dtype), bias, padding=P), 1, gamma, beta, 1e-5))
    got = layout.from_cpl(first, Cout)[:, :, P:P + L].float().cpu()
    assert rel_err(got[:2], ref) <= tol
    # every sample sees the same weights: per-sample error against its own reference for two more samples, spread out
    for b in (7, 15):
        rb = F.gelu(F.group_norm(F.conv1d(quant(x[b:b + 1], dtype), quant(w, dtype), bias, padding=P), 1, gamma, beta, 1e-5))
        assert rel_err(got[b:b + 1], rb) <= tol


def test_conv_groupnorm_fusion_argument_checks(env):
    lib, _lib, layout, dev = env
    op = _lib.AdkOp()
    op.kind, op.dtype, op.flags = _lib.OP_CONV, 1, _lib.FLAG_FUSE_GN
    op.B, op.L, op.P, op.Cin, op.Cout, op.CoutPad, op.taps = 2, 64, 2, 32, 128, 128, 5
    assert lib.adk_conv_gn_fusable(C.byref(op)) == 1          # short level: k-split kernel
    op.Cout = op.CoutPad = 32
    op.L = 4096
    assert lib.adk_conv_gn_fusable(C.byref(op)) == 0          # narrow-output kernel (Cout <= 32 above the short levels)
    op.Cout = op.CoutPad = 128
    op.L = 262144                                             # 1024 tiles per sample: more than the GPU holds at once
    assert lib.adk_conv_gn_fusable(C.byref(op)) == 0
    op.L = 4096
    assert lib.adk_conv_gn_fusable(C.byref(op)) == 1
    op.x = op.w = op.y = 256
    assert lib.adk_validate_plan(C.byref(op), 1) == -1        # gamma / beta / state / workspace missing


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_attention_bits_reproducible(env, dt):
    """The MFMA attention kernel packs its probabilities with the one-instruction conversion right after packed-f32
    row sums of the same registers - the instruction mix that made the conv epilogue's (sum, sumsq) partials differ
    from run to run.  Config 3a's largest launch (B=4, 16 heads, s=4096): repeated launches must give
    identical bits."""
    lib, _lib, layout, dev = env
    dtype, adk, _ = DT[dt]
    B, Cc, s, P = 4, 512, 4096, 2
    cb = layout.chunk(dtype)
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, 3 * Cc, s, generator=g) * 1.3
    qd = layout.to_cpl(qkv.to(dev), dtype, P)
    outs = []
    for _ in range(6):
        yd = torch.full((Cc // cb, B, s + 2 * P, cb), 5.0, dtype=dtype, device=dev)
        assert lib.adk_attn1d(adk, B, s, P, Cc, qd.data_ptr(), yd.data_ptr(), sp()) == 0, lib.adk_last_error()
        outs.append(yd)
    torch.cuda.synchronize()
    for y in outs[1:]:
        assert torch.equal(y.view(torch.int16), outs[0].view(torch.int16))


PQMF_CASES = [(1, 70, 32), (2, 70, 32), (2, 70, 4), (2, 70, 16), (2, 100, 32)]


@pytest.mark.parametrize("ch,att,M", PQMF_CASES)
def test_pqmf_vs_golden(env, golden_pqmf, ch, att, M):
    lib, _lib, layout, dev = env
    z = golden_pqmf
    tag = f"pqmf_c{ch}_a{att}_m{M}"
    x = torch.from_numpy(z[f"{tag}/x"]).to(dev)
    fw = torch.from_numpy(z[f"{tag}/fw"]).to(dev)
    iw = torch.from_numpy(z[f"{tag}/iw"]).to(dev)
    B, _, T = x.shape
    N = T // M
    fwT = fw[:, 0, :].t().contiguous()
    iwT = iw.permute(1, 2, 0).contiguous()
    y = torch.empty(B, ch * M, N, device=dev)
    assert lib.adk_pqmf_analysis(B, ch, T, M, fwT.shape[0], x.data_ptr(), fwT.data_ptr(), y.data_ptr(), sp()) == 0, lib.adk_last_error()
    r = torch.empty(B, ch, T, device=dev)
    assert lib.adk_pqmf_synthesis(B, ch, N, M, iwT.shape[1], y.data_ptr(), iwT.data_ptr(), r.data_ptr(), sp()) == 0, lib.adk_last_error()
    torch.cuda.synchronize()
    assert rel_err(y, torch.from_numpy(z[f"{tag}/y"])) <= 1e-5
    assert rel_err(r, torch.from_numpy(z[f"{tag}/z"])) <= 1e-5


@pytest.mark.parametrize("M", [4, 16, 32])
def test_pqmf_vs_reference_base_class(env, M):
    """the product's CachedPQMF (libadk kernels) against outputs of the reference's base-class PQMF
    (polyphase path, diffusion/pqmf.py:95-137, 212-233 - no cached_conv involved; tests/golden/glue.npz):
    forward equal, inverse equal after its M-sample delay."""
    from conftest import load_npz
    from audio_diffusion_amd import CachedPQMF
    lib, _lib, layout, dev = env
    g = load_npz("glue.npz")
    x, y, z = (torch.from_numpy(g[f"pqmf_base_m{M}_poly/{n}"]) for n in ("x", "y", "z"))
    q = CachedPQMF(1, 70, M).to(dev)
    assert rel_err(q.hk, torch.from_numpy(g[f"pqmf_base_m{M}_poly/hk"])) <= 1e-6
    assert rel_err(q(x.to(dev)), y) <= 1e-5
    back = q.inverse(y.to(dev)).cpu()
    assert rel_err(back[..., M:], z[..., :-M]) <= 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_pack_unpack(env, golden_blocks, dt):
    lib, _lib, layout, dev = env
    dtype, adk, _ = DT[dt]
    z = golden_blocks
    fw = torch.from_numpy(z["fourier/weight"]).reshape(-1).to(dev)
    t = torch.from_numpy(z["fourier/t"]).to(dev)
    emb = torch.from_numpy(z["fourier/y"])
    B, L, P, Cx = 4, 300, 2, 2
    x = torch.randn(B, Cx, L)
    cb = layout.chunk(dtype)
    yd = torch.full((32 // cb, B, L + 2 * P, cb), 9.0, dtype=dtype, device=dev)
    assert lib.adk_pack_input(adk, B, L, P, Cx, x.to(dev).data_ptr(), t.data_ptr(), None, 0, fw.data_ptr(), yd.data_ptr(), sp()) == 0
    torch.cuda.synchronize()
    full = layout.from_cpl(yd, 32).cpu()
    assert torch.equal(full[:, :Cx, P:P + L], quant(x, dtype))
    planes = full[:, Cx:Cx + 16, P:P + L]
    assert rel_err(planes, emb[:, :, None].expand(-1, -1, L)) <= (2e-6 if dt == "f32" else 4e-3)
    assert float(full[:, Cx + 16:].abs().max()) == 0.0
    assert float(full[:, :, :P].abs().max()) == 0.0 and float(full[:, :, P + L:].abs().max()) == 0.0
    # unpack (f32 CPL -> NCL)
    src = torch.randn(3, 20, 77)
    sdv = layout.to_cpl(src.to(dev), torch.float32, P)
    out = torch.empty(3, 20, 77, device=dev)
    assert lib.adk_unpack_output(3, 77, P, 20, sdv.data_ptr(), out.data_ptr(), sp()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), src)


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_ddim_update_bit_exact(env, eta):
    """Literal operation order of sampling.py:34-52 => bitwise equal to the CPU fp32 result."""
    lib, _lib, layout, dev = env
    from audio_diffusion_amd import step_coefficients
    steps = 7
    coef = step_coefficients(steps, eta)
    n = 2 * 2 * 1003            # not a multiple of 4: exercises the scalar tail
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, generator=g)
    xd = x.to(dev)
    pred_d = torch.empty(n, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    coef_d = coef.to(dev)
    t, alphas, sigmas = coef[:, 0], coef[:, 1], coef[:, 2]
    for i in range(steps):
        v = torch.randn(n, generator=g)
        nz = torch.randn(n, generator=g)
        pred = x * alphas[i] - v * sigmas[i]
        eps = x * sigmas[i] + v * alphas[i]
        if i < steps - 1:
            ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
            adjusted = (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()
            x = pred * alphas[i + 1] + eps * adjusted
            if eta:
                x = x + nz * ddim_sigma
        vd, nd = v.to(dev), nz.to(dev)
        flags = _lib.FLAG_NOISE if (eta and i < steps - 1) else 0
        assert lib.adk_ddim_update(flags, n, xd.data_ptr(), vd.data_ptr(), nd.data_ptr() if flags else None,
                                   pred_d.data_ptr(), coef_d.data_ptr(), step.data_ptr(), sp()) == 0, lib.adk_last_error()
        assert lib.adk_step_advance(step.data_ptr(), sp()) == 0
        torch.cuda.synchronize()
        assert torch.equal(pred_d.cpu(), pred), f"pred differs at step {i}"
        assert torch.equal(xd.cpu(), x), f"x differs at step {i}"
    assert int(step.item()) == steps
