// GroupNorm(1,C) apply + GELU + residual add, one HBM-bound pass over a chunk-plane tensor.
// GELU is the exact erf form: erff in the f32 path, a 2e-7-accurate single-exp2 form in the bf16 path (adk_common.h).
//
// Replaces nn.GroupNorm(1,C) -> nn.GELU() -> (+ skip) of ResConvBlock (reference blocks/blocks.py:71-75,
// 14-15) and the norm of SelfAttention1d (:107,115).  The per-sample statistics arrive as (sum, sumsq)
// partials written by the producing kernel's epilogue (conv / previous apply), are finalized here in
// double in a fixed order (bit-reproducible; no atomics), and the pass optionally emits partials of
// its own output for a following GroupNorm.  16-byte vector loads/stores, per-thread channel slice
// fixed so gamma/beta live in registers.
//
// The same kernel runs GroupNorm(G, C) with G > 1 groups of whole 64-byte channel planes and the ADAPTIVE
// form AdaGN (reference blocks/blocks.py:53-64: group_norm without affine, then bias + xn * (weight + 1) with
// per-SAMPLE weight/bias rows produced by a Linear on the noise-level embedding): the statistics must then
// come from ADK_OP_GN_STATS ([b][plane][row block] partials, so a group is a contiguous range of them).
#include "adk_common.h"

struct GnArgs {
    const void* x; const void* residual; void* y;
    const float* partials_in; float* partials_out;
    const float* gamma; const float* beta;       // length >= planes*CB, zero beyond C
    int B, L, P, Lp; long long M;
    int n_in; int RB, nrb, nplanes;
    double count; float eps;
    int ngroups, C;                              // groups of nplanes / ngroups planes; C = logical channels
    int mod_stride; float ada;                   // AdaGN: gamma/beta rows per sample (floats), ada = 1 (scale = gamma + 1)
};

template <typename T, bool GELU, bool APPLY, bool STATS, bool RES>
__global__ __launch_bounds__(256) void gn_kernel(const GnArgs a)
{
    constexpr int NV = ElemTraits<T>::NV, CB = ElemTraits<T>::CB;
    __shared__ double red[16];
    const int rb = blockIdx.x, plane = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, q = tid & 3, r0 = tid >> 2;

    const long long rowbase = (long long)plane * a.M + (long long)b * a.Lp;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + rowbase * 64 + q * 16;
    const unsigned char* rbp = RES ? reinterpret_cast<const unsigned char*>(a.residual) + rowbase * 64 + q * 16 : nullptr;
    unsigned char* yb = APPLY ? reinterpret_cast<unsigned char*>(a.y) + rowbase * 64 + q * 16 : nullptr;

    const int lbeg = rb * a.RB, lend = (lbeg + a.RB < a.L) ? lbeg + a.RB : a.L;
    // this thread's first piece is requested before the statistics are finalized (two barriers and a dependent load of
    // the partials): at the short levels, where a thread owns one or two rows, that takes one memory round trip out of
    // a launch that is nothing but latency
    int l = lbeg + r0;
    const bool pre = l < lend && a.RB <= 256;      // long levels keep their 4-deep batches intact
    uint4 px = make_uint4(0, 0, 0, 0), pr = make_uint4(0, 0, 0, 0);
    if (pre) {
        const long long ro = (long long)(a.P + l) * 64;
        px = *reinterpret_cast<const uint4*>(xb + ro);
        if (RES) pr = *reinterpret_cast<const uint4*>(rbp + ro);
    }

    float scale[NV], shift[NV];
    if (APPLY) {
        float mean, rstd;
        const int n_g = a.n_in / a.ngroups, g = plane / (a.nplanes / a.ngroups);
        gn_finalize(a.partials_in + ((long long)b * a.n_in + (long long)g * n_g) * 2, n_g, a.count, a.eps, mean, rstd, red);
        const int c0 = plane * CB + q * NV;
        const float* gm = a.gamma + (long long)b * a.mod_stride;
        const float* bt = a.beta + (long long)b * a.mod_stride;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const bool real = c0 + e < a.C;          // pad channels stay exactly zero
            scale[e] = real ? (gm[c0 + e] + a.ada) * rstd : 0.f;
            shift[e] = real ? __builtin_fmaf(-mean, scale[e], bt[c0 + e]) : 0.f;
        }
    }
    float s1 = 0.f, s2 = 0.f;
    auto body = [&](const uint4& xin, const uint4& rin, long long ro) {
        if constexpr (sizeof(T) == 2) {              // bf16 / f16: pairs, packed-f32 arithmetic
            const uint32_t w[4] = {xin.x, xin.y, xin.z, xin.w};
            const uint32_t rw[4] = {rin.x, rin.y, rin.z, rin.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 v = unpack2(w[i], T());
                if (APPLY) {
                    f32x2 sc, sh;
                    sc.x = scale[2 * i]; sc.y = scale[2 * i + 1]; sh.x = shift[2 * i]; sh.y = shift[2 * i + 1];
                    v = fma2(v, sc, sh);
                    if (GELU) v = gelu_fast2(v);
                    if (RES) {
                        v += unpack2(rw[i], T());
                    }
                    o[i] = pack2v(v, T());
                }
                if (STATS) { s1 += v.x; s2 += v.x * v.x; s1 += v.y; s2 += v.y * v.y; }
            }
            if (APPLY) *reinterpret_cast<uint4*>(yb + ro) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            float f[NV];
            piece_to_float(xin, f, T());
            if (APPLY) {
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    float v = __builtin_fmaf(f[e], scale[e], shift[e]);
                    f[e] = GELU ? gelu_for<T>(v) : v;
                }
                if (RES) {
                    float r[NV];
                    piece_to_float(rin, r, T());
#pragma unroll
                    for (int e = 0; e < NV; ++e) f[e] += r[e];
                }
                *reinterpret_cast<uint4*>(yb + ro) = float_to_piece(f, T());
            }
            if (STATS) {
#pragma unroll
                for (int e = 0; e < NV; ++e) { s1 += f[e]; s2 += f[e] * f[e]; }
            }
        }
    };
    constexpr int U = 4;                             // independent 16-byte loads in flight per thread
    if (pre) { body(px, pr, (long long)(a.P + l) * 64); l += 64; }
    for (; l + 64 * (U - 1) < lend; l += 64 * U) {
        uint4 xin[U], rin[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long ro = (long long)(a.P + l + 64 * u) * 64;
            xin[u] = *reinterpret_cast<const uint4*>(xb + ro);
            rin[u] = RES ? *reinterpret_cast<const uint4*>(rbp + ro) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) body(xin[u], rin[u], (long long)(a.P + l + 64 * u) * 64);
    }
    for (; l < lend; l += 64) {
        const long long ro = (long long)(a.P + l) * 64;
        const uint4 xin = *reinterpret_cast<const uint4*>(xb + ro);
        const uint4 rin = RES ? *reinterpret_cast<const uint4*>(rbp + ro) : make_uint4(0, 0, 0, 0);
        body(xin, rin, ro);
    }
    if (APPLY && (rb == 0 || rb == a.nrb - 1)) {      // halo rows of the output
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int r = r0; r < a.P; r += 64) {
            if (rb == 0) *reinterpret_cast<uint4*>(yb + (long long)r * 64) = z;
            if (rb == a.nrb - 1) *reinterpret_cast<uint4*>(yb + (long long)(a.P + a.L + r) * 64) = z;
        }
    }
    if (STATS) {
        float* redf = reinterpret_cast<float*>(red);
        block_sum2(s1, s2, redf);
        if (tid == 0) {
            float* dst = a.partials_out + (((long long)b * a.nplanes + plane) * a.nrb + rb) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
}

// rows per block: 64 KiB of a plane at the long levels, smaller below so that the mid levels still launch >= 1024
// blocks (one block per CU is latency-bound: 1.8 TB/s at L = 1024 with 1024-row blocks)
static inline int gn_rb(int L) { return L >= 8192 ? 1024 : (L >= 256 ? 256 : ((L + 63) / 64) * 64); }

int gn_partials_count(int dtype, int C, int L) {
    const int CB = dtype == ADK_F32 ? 16 : 32;
    const int RB = gn_rb(L);
    return ((C + CB - 1) / CB) * ((L + RB - 1) / RB);
}

template <typename T>
static int launch_gn_t(const adk_op& op, hipStream_t st, bool apply)
{
    constexpr int CB = ElemTraits<T>::CB;
    GnArgs a;
    a.x = op.x; a.residual = op.residual; a.y = op.y;
    a.partials_in = op.partials_in; a.partials_out = op.partials_out;
    a.gamma = op.gamma; a.beta = op.beta;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.n_in = op.n_partials_in;
    a.RB = gn_rb(op.L); a.nrb = (op.L + a.RB - 1) / a.RB; a.nplanes = (op.Cin + CB - 1) / CB;
    a.ngroups = op.i1 > 1 ? op.i1 : 1;
    a.C = op.Cin;
    a.count = (double)(op.Cin / a.ngroups) * (double)op.L; a.eps = op.f0;
    a.mod_stride = (op.flags & ADK_FLAG_ADA) ? op.i0 : 0;
    a.ada = (op.flags & ADK_FLAG_ADA) ? 1.f : 0.f;
    dim3 grid(a.nrb, a.nplanes, op.B);
    // compile-time variants: the (sum, sumsq) of the output cost 2 VALU ops per element, a runtime residual test a
    // branch per 16-byte piece
    const bool stats = a.partials_out != nullptr, res = a.residual != nullptr, gelu = (op.flags & ADK_FLAG_GELU) != 0;
#define ADK_GN_LAUNCH(G_, A_, S_, R_) hipLaunchKernelGGL((gn_kernel<T, G_, A_, S_, R_>), grid, dim3(256), 0, st, a)
    if (!apply) {
        if (!stats) return ADK_EINVAL;
        ADK_GN_LAUNCH(false, false, true, false);
    } else if (gelu) {
        if (stats) { if (res) ADK_GN_LAUNCH(true, true, true, true); else ADK_GN_LAUNCH(true, true, true, false); }
        else       { if (res) ADK_GN_LAUNCH(true, true, false, true); else ADK_GN_LAUNCH(true, true, false, false); }
    } else {
        if (stats) { if (res) ADK_GN_LAUNCH(false, true, true, true); else ADK_GN_LAUNCH(false, true, true, false); }
        else       { if (res) ADK_GN_LAUNCH(false, true, false, true); else ADK_GN_LAUNCH(false, true, false, false); }
    }
#undef ADK_GN_LAUNCH
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

int launch_gn(const adk_op& op, hipStream_t st, bool apply)
{
    if (op.dtype == ADK_F32) return launch_gn_t<float>(op, st, apply);
    if (op.dtype == ADK_F16) return launch_gn_t<f16_t>(op, st, apply);
    return launch_gn_t<bf16_t>(op, st, apply);
}
