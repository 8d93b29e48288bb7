"""ctypes binding of libadk.so (C ABI declared in include/adk.h).

The library is built in-tree by ``make -C audio_diffusion_amd/csrc`` (or ``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a symbol is absent, importing the compute path
raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADK_LIB") or os.path.join(_HERE, "libadk.so")     # ADK_LIB: A/B runs against another build

ADK_F32, ADK_BF16, ADK_F16 = 0, 1, 2


def adk_dtype(dtype):
    """torch dtype -> ADK_* code.  float16 is the dtype of the reference's own autocast region
    (diffusion/sampling.py:30); bf16 and f16 share layout, kernels and MFMA rate."""
    import torch
    table = {torch.float32: ADK_F32, torch.bfloat16: ADK_BF16, torch.float16: ADK_F16}
    if dtype not in table:
        raise ValueError(f"compute dtype {dtype} is not supported (float32, bfloat16, float16)")
    return table[dtype]


def autocast_dtype():
    """Reduced dtype to use when the caller wrapped the call in torch.autocast: float16 if that is what the
    autocast region asks for, else bfloat16."""
    import torch
    return torch.float16 if torch.get_autocast_dtype("cuda") == torch.float16 else torch.bfloat16

OP_CONV, OP_GN_APPLY, OP_GN_STATS, OP_FIR_DOWN, OP_FIR_UP, OP_ATTN = 1, 2, 3, 4, 5, 6
OP_PACK_INPUT, OP_UNPACK_OUTPUT, OP_PQMF_ANALYSIS, OP_PQMF_SYNTHESIS = 7, 8, 9, 10
OP_DDIM_UPDATE, OP_STEP_ADVANCE, OP_AUDIO_TO_INT16, OP_SPACE_TO_DEPTH, OP_DEPTH_TO_SPACE = 11, 12, 13, 14, 15
OP_LINEAR, OP_FOURIER = 16, 17

FLAG_GELU, FLAG_OUT_F32, FLAG_NOISE, FLAG_ADA, FLAG_LOG, FLAG_NO_EMBED, FLAG_GROUP_STATS, FLAG_WT = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_CAUSAL, FLAG_ELU, FLAG_FUSE_GN, FLAG_SKIP_CONV, FLAG_PRE_NORM = 256, 512, 1024, 2048, 4096
FLAG_QKV = 8192


class AdkOp(C.Structure):
    """Mirror of ``struct adk_op`` (include/adk.h)."""
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_uint32),
        ("B", C.c_int32), ("L", C.c_int32), ("P", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32), ("taps", C.c_int32),
        ("n_partials_in", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32),
        ("f0", C.c_float), ("f1", C.c_float),
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("y", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("partials_in", C.c_void_p),
        ("partials_out", C.c_void_p),
        ("aux0", C.c_void_p), ("aux1", C.c_void_p), ("w2", C.c_void_p),
    ]


EXPORTS = [
    "adk_version", "adk_last_error", "adk_sizeof_op", "adk_run_plan", "adk_validate_plan", "adk_partials_count", "adk_conv_postile", "adk_conv_cotile_used", "adk_conv_cotile",
    "adk_conv1d", "adk_gn_apply", "adk_gn_stats", "adk_fir_down2", "adk_fir_up2", "adk_attn1d",
    "adk_pack_input", "adk_pack_input_cond", "adk_unpack_output", "adk_pqmf_analysis", "adk_pqmf_synthesis",
    "adk_ddim_update", "adk_step_advance", "adk_audio_to_int16", "adk_space_to_depth", "adk_depth_to_space",
    "adk_attn1d_hd", "adk_adagn_apply", "adk_linear", "adk_fourier_features", "adk_pack_input_plain",
    "adk_unpack_output_scaled", "adk_group_norm_noalloc", "adk_conv_gn_fusable",
]

_lib = None


class AdkError(RuntimeError):
    pass


def load():
    """Load libadk.so once; raise loudly when it is not there (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdkError(f"{LIB_PATH} not found: build it with `make -C audio_diffusion_amd/csrc` "
                       "(the HIP extension is mandatory; there is no fallback path)")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise AdkError(f"libadk.so does not export {name}")
    vp, i32, u32, f32, i64 = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_int64
    lib.adk_version.restype = C.c_int
    lib.adk_last_error.restype = C.c_char_p
    lib.adk_run_plan.argtypes = [C.POINTER(AdkOp), i32, vp]
    lib.adk_validate_plan.argtypes = [C.POINTER(AdkOp), i32]
    lib.adk_partials_count.argtypes = [C.POINTER(AdkOp)]
    lib.adk_conv_postile.argtypes = [C.POINTER(AdkOp)]
    lib.adk_conv_cotile_used.argtypes = [C.POINTER(AdkOp)]
    lib.adk_conv_cotile.argtypes = [i32, i32, i32]
    lib.adk_conv_gn_fusable.argtypes = [C.POINTER(AdkOp)]
    lib.adk_conv1d.argtypes = [i32, u32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.adk_gn_apply.argtypes = [i32, u32, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.adk_gn_stats.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
    lib.adk_fir_down2.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.adk_fir_up2.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.adk_attn1d.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
    lib.adk_pack_input.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    lib.adk_pack_input_cond.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.adk_unpack_output.argtypes = [i32, i32, i32, i32, vp, vp, vp]
    lib.adk_pqmf_analysis.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.adk_pqmf_synthesis.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.adk_ddim_update.argtypes = [u32, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.adk_step_advance.argtypes = [vp, vp]
    lib.adk_audio_to_int16.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.adk_space_to_depth.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.adk_depth_to_space.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.adk_attn1d_hd.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.adk_adagn_apply.argtypes = [i32, u32, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp, i32, vp, vp]
    lib.adk_linear.argtypes = [u32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp]
    lib.adk_fourier_features.argtypes = [u32, i32, i32, vp, vp, vp, i32, vp]
    lib.adk_pack_input_plain.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, vp]
    lib.adk_unpack_output_scaled.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.adk_group_norm_noalloc.argtypes = [i32, u32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        if name not in ("adk_last_error",):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise AdkError(f"{what} failed (code {rc}): {load().adk_last_error().decode()}")


def run_ops(ops_array, n, stream_ptr):
    check(load().adk_run_plan(ops_array, n, C.c_void_p(stream_ptr)), "adk_run_plan")
