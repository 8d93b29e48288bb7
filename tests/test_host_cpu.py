"""CPU: host logic of the product package (no kernels are launched): C-ABI surface, state_dict / init
parity with the reference, plan compilation + host-side validation, sampler coefficient tables."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, GOLDEN, load_npz, sub_sd

import audio_diffusion_amd as A
from audio_diffusion_amd import _lib
from audio_diffusion_amd.unet import _Plan, _ArenaPlanner


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "adk.h")).read()
    declared = set(re.findall(r"\b(adk_[a-z0-9_]+)\s*\(", header))
    declared.discard("adk_op")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libadk.so lacks {name}"
    assert set(_lib.EXPORTS) == declared
    assert lib.adk_version() >= 100


def test_adk_op_struct_size_matches_c():
    # 14 int32 + 2 float = 64 bytes, then 12 pointers; and what the library itself was compiled with
    assert ctypes.sizeof(_lib.AdkOp) == 64 + 12 * 8
    assert _lib.load().adk_sizeof_op() == ctypes.sizeof(_lib.AdkOp)


def test_validation_messages_without_gpu():
    lib = _lib.load()
    op = _lib.AdkOp()
    op.kind = 99
    assert lib.adk_validate_plan(ctypes.byref(op), 1) == -1
    assert b"unknown op kind" in lib.adk_last_error()
    op.kind, op.dtype, op.B, op.L, op.P, op.Cin, op.taps = _lib.OP_FIR_DOWN, 0, 1, 2, 2, 16, 8
    op.x = op.y = op.w = 1
    assert lib.adk_validate_plan(ctypes.byref(op), 1) == -1
    assert b"reflect" in lib.adk_last_error()


def test_full_model_init_matches_reference_digest():
    d = json.load(open(os.path.join(GOLDEN, "full_digest.json")))
    torch.manual_seed(0)
    m = A.DiffusionAttnUnet1D(n_attn_layers=4)
    sd = m.state_dict()
    assert list(sd.keys()) == list(d["tensors"].keys())
    h = hashlib.sha256()
    for k, v in sd.items():
        assert list(v.shape) == d["tensors"][k]["shape"], k
        h.update(k.encode())
        h.update(v.contiguous().numpy().tobytes())
    assert h.hexdigest() == d["meta"]["sd_sha256"]
    assert sum(p.numel() for p in m.parameters()) == d["meta"]["n_params"] == 221389322
    # plan for config 1 compiles and passes host validation (CPU dry run: nothing is launched)
    plan = _Plan(m, 1, 65536, torch.bfloat16, torch.device("cpu"))
    plan.validate()
    kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
    assert kinds.count(_lib.OP_ATTN) == 36 and kinds.count(_lib.OP_FIR_DOWN) == 13 and kinds.count(_lib.OP_FIR_UP) == 13
    # 203 GroupNorms: 167 of them ride in the epilogue of the conv that feeds them (ADK_FLAG_FUSE_GN: every k=5 conv of
    # a ResConvBlock but the last block's, which has no norm), the 36 attention norms stay passes of their own
    fused = sum(1 for i in range(plan.n_ops) if plan.ops[i].kind == _lib.OP_CONV and plan.ops[i].flags & _lib.FLAG_FUSE_GN)
    # ... and the 36 attention norms are folded into the qkv_proj weights (ADK_FLAG_PRE_NORM): no GroupNorm pass is left
    # (all six attention levels of this model have L <= 256: the folded projection runs inside the attention launch itself,
    # ADK_FLAG_QKV; longer sequences keep qkv_proj as a conv of its own with ADK_FLAG_PRE_NORM)
    prenorm = sum(1 for i in range(plan.n_ops) if plan.ops[i].kind == _lib.OP_ATTN and plan.ops[i].flags & _lib.FLAG_QKV)
    assert fused == 167 and prenorm == 36 and kinds.count(_lib.OP_GN_APPLY) == 203 - fused - prenorm == 0
    long_plan = _Plan(A.DiffusionAttnUnet1D(depth=4, c_mults=[32, 32, 64, 64], n_attn_layers=2), 1, 8192, torch.float16,
                      torch.device("cpu"))
    long_plan.validate()
    lk = [(long_plan.ops[i].kind, long_plan.ops[i].flags, long_plan.ops[i].L) for i in range(long_plan.n_ops)]
    assert any(k == _lib.OP_CONV and f & _lib.FLAG_PRE_NORM and L > 256 for k, f, L in lk)
    assert not any(k == _lib.OP_ATTN and f & _lib.FLAG_QKV and L > 256 for k, f, L in lk)
    # 168 k=5 + 36 out projections (the 36 qkv projections run inside the attention launches) + 5 skips: the other 14 skip convs (every one above the k-split levels but the
    # last block's) run inside the fused launch of their block's second conv (ADK_FLAG_SKIP_CONV)
    folded = sum(1 for i in range(plan.n_ops) if plan.ops[i].kind == _lib.OP_CONV and plan.ops[i].flags & _lib.FLAG_SKIP_CONV)
    # the last block's skip conv and its second conv are ONE conv over the joint buffer [h1 | x] (last_block_joint): - 1 more
    assert folded == 14 and kinds.count(_lib.OP_CONV) == 259 - folded - 36 - 1
    m.fold_skip = m.fold_last_skip = False
    nofold = _Plan(m, 1, 65536, torch.bfloat16, torch.device("cpu"))
    nofold.validate()
    assert [nofold.ops[i].kind for i in range(nofold.n_ops)].count(_lib.OP_CONV) == 259 - 36
    m.fold_skip = m.fold_last_skip = True
    assert len({plan.ops[i].i1 for i in range(plan.n_ops) if plan.ops[i].flags & _lib.FLAG_FUSE_GN
                and plan.ops[i].kind == _lib.OP_CONV}) == fused          # one ticket counter per fused conv
    m.fuse_gn = m.fold_attn_norm = False              # the unfused plan stays available (A/B, and > 512 tiles per sample)
    plain = _Plan(m, 1, 65536, torch.bfloat16, torch.device("cpu"))
    plain.validate()
    assert [plain.ops[i].kind for i in range(plain.n_ops)].count(_lib.OP_GN_APPLY) == 203


@pytest.mark.parametrize("name", ["tiny_d3.npz", "tiny_d5.npz", "tiny_d3_pqmf4.npz", "tiny_d4_noattn_nobias.npz",
                                  "tiny_learned_k7.npz", "tiny_learned_s3_k3.npz", "tiny_d3_k7_fir.npz", "tiny_d3_cond.npz"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_tiny_models_load_and_plan(name, dtype):
    z = load_npz(name)
    cfg = json.loads(bytes(z["cfg"]).decode())
    torch.manual_seed(int(z["seed"]))
    m = A.DiffusionAttnUnet1D(**cfg)
    ref_sd = sub_sd(z, "sd/")
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:                      # same seed => same init as the reference, bit for bit
        if k.startswith("pqmf.") and k != "pqmf.h" and not k.endswith("hk"):
            assert torch.allclose(sd[k], ref_sd[k], rtol=0, atol=1e-6), k
        elif k.startswith("pqmf."):
            assert torch.allclose(sd[k], ref_sd[k], rtol=0, atol=1e-6), k
        else:
            assert torch.equal(sd[k], ref_sd[k]), k
    m.load_state_dict(ref_sd, strict=True)
    B, _, T = z["x"].shape
    plan = _Plan(m, B, T, dtype, torch.device("cpu"))
    plan.validate()
    assert plan.peak > 0
    with pytest.raises(RuntimeError):
        m(torch.from_numpy(z["x"]), torch.from_numpy(z["t"]))     # CPU tensors: loud failure, no fallback


def test_ctor_errors_mirror_reference():
    with pytest.raises(ZeroDivisionError):          # attention on C < 32 (blocks/blocks.py:106)
        A.DiffusionAttnUnet1D(depth=3, c_mults=[16, 16, 16], n_attn_layers=1)
    with pytest.raises(NotImplementedError):
        A.DiffusionAttnUnet1D(depth=3, c_mults=[32, 32, 64], kernel_size=9)
    lr = A.DiffusionAttnUnet1D(depth=3, c_mults=[32, 32, 64], n_attn_layers=0, learned_resample=True, strides=[1, 4, 2])
    with pytest.raises(RuntimeError):
        lr.check_input(torch.zeros(1, 2, 1026))       # not a multiple of the level strides
    lr.check_input(torch.zeros(1, 2, 1024))
    m = A.DiffusionAttnUnet1D(depth=3, c_mults=[32, 32, 64], n_attn_layers=1)
    with pytest.raises(RuntimeError):
        m.check_input(torch.zeros(1, 2, 8))
    with pytest.raises(ValueError):
        m.check_input(torch.zeros(1, 2, 1026))


def test_dropout_is_identity_in_eval_and_refused_in_train():
    """blocks/blocks.py:96,100,111,132: the reference's blocks take dropout_rate; it is the identity in eval mode (the
    mode sample() callers use), so the plans carry no op for it - but an ACTIVE dropout must not be silently ignored."""
    from audio_diffusion_amd.blocks import SelfAttention1d, ResModConvBlock, check_dropout_inactive
    att = SelfAttention1d(64, 2, dropout_rate=0.1)          # used to raise NotImplementedError
    assert sorted(att.state_dict()) == ["norm.bias", "norm.weight", "out_proj.bias", "out_proj.weight",
                                        "qkv_proj.bias", "qkv_proj.weight"]
    blk = ResModConvBlock(16, 32, 32, 32, dropout_rate=0.2)
    for mod in (att, blk):
        mod.train()
        with pytest.raises(RuntimeError, match="dropout"):
            check_dropout_inactive(mod)
        mod.eval()
        check_dropout_inactive(mod)
    m = A.DiffusionAttnUnet1D(depth=3, c_mults=[32, 32, 64], n_attn_layers=1)
    m.train()                                             # rate 0 everywhere: train() alone is fine
    check_dropout_inactive(m)


def test_arena_planner_reuses_and_never_overlaps():
    import random
    rnd = random.Random(0)
    pl = _ArenaPlanner()
    live = {}
    for step in range(2000):
        if live and rnd.random() < 0.45:
            k = rnd.choice(list(live))
            pl.release(*live.pop(k))
        else:
            off, n = pl.alloc(rnd.randint(1, 5000))
            for o2, n2 in live.values():
                assert off + n <= o2 or o2 + n2 <= off, "overlap"
            live[step] = (off, n)
    for v in live.values():
        pl.release(*v)
    assert pl.top == 0 and not pl.holes


def test_schedule_and_coefficients_match_reference_tables(golden_blocks):
    z = golden_blocks
    for steps in (5, 10, 100):
        tab = A.step_coefficients(steps, 0.0)
        assert np.array_equal(tab[:, 0].numpy(), z[f"sched{steps}/t"])
        assert np.array_equal(tab[:, 1].numpy(), z[f"sched{steps}/alpha"])
        assert np.array_equal(tab[:, 2].numpy(), z[f"sched{steps}/sigma"])
        assert np.array_equal(tab[:-1, 3].numpy(), z[f"sched{steps}/alpha"][1:])
        assert np.array_equal(tab[:-1, 4].numpy(), z[f"sched{steps}/sigma"][1:])     # eta=0: adjusted == sigma_next
        assert tab[-1, 6] == 1 and tab[:-1, 6].abs().sum() == 0
    a, s = A.get_alphas_sigmas(torch.from_numpy(z["sched10/t"]))
    assert np.array_equal(A.alpha_sigma_to_t(a, s).numpy(), z["sched10/t_back"])
    with pytest.raises(RuntimeError):
        A.sample(lambda x, t: x, torch.zeros(1, 2, 16), 2, 0)        # CPU tensors: no fallback


def test_pqmf_design_matches_reference_tables(golden_pqmf):
    z = golden_pqmf
    for ch, att, M in ((2, 70, 4), (2, 70, 16), (2, 70, 32), (2, 100, 32)):
        q = A.CachedPQMF(ch, att, M)
        tag = f"pqmf_c{ch}_a{att}_m{M}"
        sd = q.state_dict()
        assert set(sd) == {"hk", "h", "forward_conv.weight", "inverse_conv.weight"}
        for key, ref in (("hk", "hk"), ("h", "h"), ("forward_conv.weight", "fw"), ("inverse_conv.weight", "iw")):
            assert sd[key].shape == tuple(z[f"{tag}/{ref}"].shape)
            assert float((sd[key] - torch.from_numpy(z[f"{tag}/{ref}"])).abs().max()) <= 1e-6


def test_crash_schedule_table_matches_oracle():
    """step_coefficients(warp=get_crash_schedule) == the oracle's crash-warped schedule, bit for bit."""
    import audio_diffusion_amd as A
    from oracle import sampling_oracle as S
    for steps in (5, 50):
        tab = A.step_coefficients(steps, 0.0, warp=A.get_crash_schedule)
        t, al, si = S.schedule(steps, crash=True)
        assert torch.equal(tab[:, 0], t) and torch.equal(tab[:, 1], al) and torch.equal(tab[:, 2], si)
        assert float(tab[0, 0]) == 1.0 and bool((tab[1:, 0] < tab[:-1, 0]).all())


def test_learned_resampler_folds_match_torch():
    """Host logic of SURVEY 8(f2): Downsample1d_2 / Upsample1d_2 (blocks/blocks.py:214-255) as phase fold + 3-tap conv."""
    import torch.nn.functional as F
    from audio_diffusion_amd.blocks import fold_down_weight, fold_up_weight
    torch.manual_seed(0)
    for f in (1, 2, 3, 4, 8):
        C, Co, L = 6, 5, 24 * f
        x = torch.randn(2, C, L, dtype=torch.float64)
        w, b = torch.randn(Co, C, 2 * f + 1, dtype=torch.float64), torch.randn(Co, dtype=torch.float64)
        ref = F.conv1d(x, w, b, stride=f, padding=f)
        xf = x.view(2, C, L // f, f).permute(0, 3, 1, 2).reshape(2, f * C, L // f)            # space_to_depth
        assert torch.allclose(F.conv1d(xf, fold_down_weight(w, f), b, padding=1), ref, atol=1e-12)
        if f > 1:
            wt = torch.randn(C, Co, 2 * f, dtype=torch.float64)
            ref = F.conv_transpose1d(x, wt, b, stride=f, padding=f // 2 + f % 2, output_padding=f % 2)
            w3, b3 = fold_up_weight(wt, b, f)
            z = F.conv1d(x, w3, b3, padding=1)
            got = z.view(2, f, Co, L).permute(0, 2, 3, 1).reshape(2, Co, L * f)                # depth_to_space
            assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-12)
