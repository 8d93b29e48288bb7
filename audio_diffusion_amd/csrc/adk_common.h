// Shared device helpers for libadk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/adk.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Chunk-plane layout (CPL): rows of 64 bytes = CB channels; see include/adk.h.
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int CB = 16;   // channels per 64-byte row
    static constexpr int NV = 4;    // elements per 16-byte piece
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int CB = 32;
    static constexpr int NV = 8;
};
template <> struct ElemTraits<f16_t> {      // IEEE half: the dtype of the reference's own autocast region (sampling.py:30)
    static constexpr int CB = 32;
    static constexpr int NV = 8;
};

struct Geo {            // geometry of one CPL tensor
    int B, L, P, Lp;    // Lp = L + 2P
    long long M;        // rows per plane = B * Lp
};
__host__ __device__ inline Geo make_geo(int B, int L, int P) {
    Geo g; g.B = B; g.L = L; g.P = P; g.Lp = L + 2 * P; g.M = (long long)B * g.Lp; return g;
}

// ---- 16-byte piece <-> float registers -------------------------------------------------------
__device__ __forceinline__ void piece_to_float(const uint4& p, float (&f)[4], float) {
    f[0] = __uint_as_float(p.x); f[1] = __uint_as_float(p.y);
    f[2] = __uint_as_float(p.z); f[3] = __uint_as_float(p.w);
}
__device__ __forceinline__ void piece_to_float(const uint4& p, float (&f)[8], bf16_t) {
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) per element and merges the halves.
    // Do NOT "improve" this into one vector conversion per pair inside the conv epilogues: with a single
    // in-place v_cvt_pk_bf16_f32 per pair there, the (sum, sumsq) partials of the same registers came out different
    // from run to run in 1-4 of 4096 blocks (tools/find_nondeterminism.py; y itself stayed bit-identical) - an
    // issue-timing dependent hazard this instruction mix avoids.  pack2v below is the one-instruction form, used by
    // the GroupNorm pass only (its statistics are taken before the conversion in separate registers; verified
    // bit-reproducible with the same tool).
#ifdef ADK_EXP_PACK1      /* experiment / lint build only: the one-instruction form round 1 reverted (tools/isa_lint.py was run on it) */
    f32x2 v2; v2.x = lo; v2.y = hi;
    union { bf16x2 h; uint32_t u; } cv1;
    cv1.h = __builtin_convertvector(v2, bf16x2);
    return cv1.u;
#else
    union { bf16_t h[2]; uint32_t u; } cv;
    cv.h[0] = (bf16_t)lo; cv.h[1] = (bf16_t)hi;
    return cv.u;
#endif
}
// two 16-bit elements of one 32-bit word <-> two floats (element 0 in the low half)
__device__ __forceinline__ f32x2 unpack2(uint32_t w, bf16_t) {
    f32x2 v;
    v.x = __uint_as_float(w << 16); v.y = __uint_as_float(w & 0xffff0000u);
    return v;
}
__device__ __forceinline__ f32x2 unpack2(uint32_t w, f16_t) {
    union { uint32_t u; f16x2 h; } cv;
    cv.u = w;
    return __builtin_convertvector(cv.h, f32x2);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi, bf16_t) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi, f16_t) {
    union { f16_t h[2]; uint32_t u; } cv;       // plain casts: round to nearest even (not v_cvt_pkrtz)
    cv.h[0] = (f16_t)lo; cv.h[1] = (f16_t)hi;
    return cv.u;
}
__device__ __forceinline__ uint32_t pack2v(f32x2 v, bf16_t) {
    union { bf16x2 h; uint32_t u; } cv;
    cv.h = __builtin_convertvector(v, bf16x2);
    return cv.u;
}
__device__ __forceinline__ uint32_t pack2v(f32x2 v, f16_t) {
    union { f16x2 h; uint32_t u; } cv;
    cv.h = __builtin_convertvector(v, f16x2);
    return cv.u;
}
__device__ __forceinline__ void piece_to_float(const uint4& p, float (&f)[8], f16_t) {
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = unpack2(w[i], f16_t());
        f[2 * i] = v.x; f[2 * i + 1] = v.y;
    }
}
__device__ __forceinline__ uint4 float_to_piece(const float (&f)[8], f16_t) {
    return make_uint4(pack2(f[0], f[1], f16_t()), pack2(f[2], f[3], f16_t()), pack2(f[4], f[5], f16_t()),
                      pack2(f[6], f[7], f16_t()));
}
__device__ __forceinline__ uint4 float_to_piece(const float (&f)[4], float) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
__device__ __forceinline__ uint4 float_to_piece(const float (&f)[8], bf16_t) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// ---- deterministic reductions ----------------------------------------------------------------
// Cross-lane moves by DPP (data-parallel primitives: the VALU reads its operand from another lane of the same 16-lane row,
// no LDS round trip).  __shfl_xor compiles to ds_bpermute_b32 + an lgkmcnt(0) wait per step: about 700 cycles for a
// 64-lane sum, on the critical path between a tile's last MFMA and the statistics it publishes; the DPP form is ~60.
// Every caller runs with all 64 lanes active (wave-uniform control flow), which DPP needs.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

// Sum over the 64 lanes, the same value on every lane; a fixed tree (within quads, half rows, rows, then the four rows in
// the order (0 + 1) + (2 + 3)): bit-reproducible.
template <typename V> __device__ __forceinline__ V wave_sum(V v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// Block-wide sum of two values with 256 threads; every thread gets the result. `red` = 16 V's of LDS.
template <typename V> __device__ __forceinline__ void block_sum2(V& a, V& b, V* red) {
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    V sa = 0, sb = 0;
    for (int i = 0; i < nw; ++i) { sa += red[2 * i]; sb += red[2 * i + 1]; }   // fixed order
    a = sa; b = sb;
}

// Finalize GroupNorm(1,C) statistics of sample b from (sum,sumsq) partials, fixed order, in double.
__device__ __forceinline__ void gn_finalize(const float* __restrict__ partials, int n, double count, float eps,
                                            float& mean, float& rstd, double* red) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { s += (double)partials[2 * i]; q += (double)partials[2 * i + 1]; }
    block_sum2(s, q, red);
    const double mu = s / count;
    double var = q / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// The same by ONE wave on its own (no LDS, no barrier): lanes take every 64th pair, xor butterfly in double; every lane
// gets the result.  Fixed order -> bit-reproducible.
__device__ __forceinline__ void gn_finalize_wave(const float* __restrict__ partials, int n, double count, float eps,
                                                 float& mean, float& rstd) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x & 63; i < n; i += 64) {
        const float2 p = *reinterpret_cast<const float2*>(partials + 2 * i);
        s += (double)p.x; q += (double)p.y;
    }
    s = wave_sum(s); q = wave_sum(q);
    const double mu = s / count;
    double var = q / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// a * b + c with ONE rounding, written out: the GroupNorm affine must give the same bits in every instantiation that
// applies it (with / without output statistics, the branch-free epilogue and the masked one, in place or not).  Left as
// `a * b + c` the contraction is the compiler's choice per instantiation - round 3 saw the SLP vectoriser turn it into
// v_pk_mul + v_pk_add in one gn_kernel variant and v_pk_fma in its sibling (1 ulp apart, caught by the in-place test).
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// exact GELU (nn.GELU default, approximate='none'): 0.5*x*(1+erf(x/sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for the bf16 path.  gelu(v) = max(v, 0) - |v| E / 2 with E = erfc(|v| / sqrt 2) = exp2(-u Q(u)),
// u = min(|v|, 6.08) and Q a degree-5 polynomial (weighted minimax fit of -log2(erfc(z)) / z on (0, 4.3], folded
// to u = z sqrt 2): |abs error| <= 2e-7 for |v| <= 12 (6e-7 at |v| = 1000), far below bf16 rounding.
// 12 VALU instructions, ONE of them transcendental (quarter rate), against ~35 for erff and 15 (two
// transcendental) for the Abramowitz-Stegun form used before: with either of those the GroupNorm pass was
// VALU-bound (5.1 TB/s), not HBM-bound.
__device__ __forceinline__ float gelu_fast(float v) {
    const float u = fminf(fabsf(v), 6.08f);
    float q = -3.224038610e-05f;
    q = __builtin_fmaf(q, u, 7.605187619e-04f);
    q = __builtin_fmaf(q, u, -8.046323462e-03f);
    q = __builtin_fmaf(q, u, 5.334470532e-02f);
    q = __builtin_fmaf(q, u, 4.588402605e-01f);
    q = __builtin_fmaf(q, u, 1.151166223e+00f);
    const float e = __builtin_amdgcn_exp2f(-(u * q));
    return fmaxf(v, 0.f) - fabsf((0.5f * v) * e);
}
// gelu_fast on a pair: the polynomial, the products and the final subtraction are packed-f32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two elements per issue slot); min, max and exp2 are not.
// r = (u / 2) E uses the clamped u: beyond the clamp E <= 1.2e-9, so the difference is below 4e-9.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 v) {
    f32x2 u;
    u.x = fminf(fabsf(v.x), 6.08f); u.y = fminf(fabsf(v.y), 6.08f);
    f32x2 q = u * -3.224038610e-05f + 7.605187619e-04f;
    q = q * u + -8.046323462e-03f;
    q = q * u + 5.334470532e-02f;
    q = q * u + 4.588402605e-01f;
    q = q * u + 1.151166223e+00f;
    const f32x2 t = u * q;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(-t.x); e.y = __builtin_amdgcn_exp2f(-t.y);
    f32x2 m;
    m.x = fmaxf(v.x, 0.f); m.y = fmaxf(v.y, 0.f);
    return m - (u * 0.5f) * e;
}
template <typename T> __device__ __forceinline__ float gelu_for(float v);
template <> __device__ __forceinline__ float gelu_for<float>(float v) { return gelu_erf(v); }
template <> __device__ __forceinline__ float gelu_for<bf16_t>(float v) { return gelu_fast(v); }
template <> __device__ __forceinline__ float gelu_for<f16_t>(float v) { return gelu_fast(v); }

// XCD-aware bijective remap of a 1-D block id: consecutive logical ids share an XCD (8 XCDs).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, slot = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// One-time, PER-DEVICE opt-in to more than 64 KiB of dynamic LDS (hipFuncSetAttribute is kept per device, and the
// library is re-entrant): bit d of `done` = set on device d.  A race between two first callers sets it twice, harmlessly.
static inline int adk_lds_optin(std::atomic<unsigned long long>& done, const void* fn, size_t bytes)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ADK_ELAUNCH;
    if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return ADK_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return ADK_ELAUNCH;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return ADK_OK;
}
