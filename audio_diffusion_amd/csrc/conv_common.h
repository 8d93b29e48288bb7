// Shared pieces of the implicit-GEMM convolution kernels (conv_mfma.hip, conv_ksplit.hip).
#pragma once
#include "adk_common.h"

struct ConvArgs {
    const void* x; const void* w; const float* bias; const void* residual; void* y; float* partials;
    int B, L, P, Lp;
    long long M;
    int nchunks, Cout_alloc, CoutPad, tps, ncot;
    int gstats;                  // partials per 32-channel group (AdaGN / GroupNorm(C/32, C)) instead of per tile
    int shift, dil, elu;         // tap k reads row pos - shift + k*dil (centred: shift = taps/2; causal: (taps-1)*dil); ELU epilogue
    unsigned nblk;
    // GroupNorm(1,C) [+ GELU] fused into the epilogue (ADK_FLAG_FUSE_GN, conv_mfma_gn.hip): the tiles of a sample exchange
    // their (sum, sumsq) through 8-byte {epoch, value} granules in `gn_ws`
    const float* gamma; const float* beta;
    unsigned long long* gn_ws;   // [B][npb][2] granules
    const unsigned* gn_epoch;    // [0] = per-forward epoch base (advanced by an earlier kernel); the tag also carries gn_slot
    unsigned* gn_err;            // set to a non-zero code when a bounded wait runs out
    unsigned* gn_ticket;         // this op's tile counter (grows by nblk per forward)
    unsigned gn_slot, gn_cap;             // gn_cap: resident block slots of the kernel on this device
    unsigned gn_grid;                     // blocks launched when tiles go by ticket: 8 x the longest per-XCD list
    int npb, gelu;               // tiles per sample (= tps * ncot)
    float eps; double gn_count;  // Cout * L
    // ADK_FLAG_PRE_NORM: GroupNorm(1, Cin) of the INPUT folded into the conv (attention's norm -> qkv_proj): the weights carry
    // gamma, `gamma` = row sums u of the packed weights, `beta` = bias + W beta_gn, and the epilogue applies
    // y = rstd_b * acc + (beta[co] - mean_b * rstd_b * gamma[co]) with the sample's statistics from pn_partials [B][pn_n][2]
    const float* pn_partials; int pn_n;
    const void* w2; int nchunks2;   // ADK_FLAG_SKIP_CONV: packed 1x1 weights of the folded skip conv, channel chunks of its input (= residual)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    typedef f16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
#ifdef ADK_ABL_MFMA16      /* timing experiment (wrong results): the same FLOPs as two 16x16x32 MFMAs on slices of the accumulator */
        f32x4 c0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), c1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
#else
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        // lane (r, h) holds k = 4h + {0,1,2,3} of this 8-wide k group for both operands
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};

// nn.ELU (alpha = 1), applied by the conv epilogues of the SoundStream decoder (autoencoders/soundstream.py:46,94-110)
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
__device__ __forceinline__ void store4(f16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2(v[0], v[1], f16_t()), pack2(v[2], v[3], f16_t()));
}
__device__ __forceinline__ void load4(const f16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const f32x2 a = unpack2(t.x, f16_t()), b = unpack2(t.y, f16_t());
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}


// raw (unconverted) 4-element piece of an output-typed tensor: lets loads be issued without waiting for their data
template <typename TO> struct RawVec4 { typedef uint2 type; };
template <> struct RawVec4<float> { typedef float4 type; };
__device__ __forceinline__ void raw_to_float4(const float4& t, float (&v)[4], float) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void raw_to_float4(const uint2& t, float (&v)[4], bf16_t) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void raw_to_float4(const uint2& t, float (&v)[4], f16_t) {
    const f32x2 a = unpack2(t.x, f16_t()), b = unpack2(t.y, f16_t());
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// tile shape chosen on the host for one conv op (conv_mfma.hip: conv_tile)
struct ConvTile { int bn, bm; bool ksplit; };
ConvTile conv_tile(const adk_op& op);
void conv_fill_args(const adk_op& op, int BN, int BM, ConvArgs& a);
int launch_conv_ksplit(const adk_op& op, hipStream_t st);
int launch_conv_ext(const adk_op& op, hipStream_t st);      // conv_mfma_ext.hip
int launch_conv_gn(const adk_op& op, hipStream_t st);       // conv_mfma_gn.hip
int conv_gn_check(const adk_op& op, const char** why);      // host-only: can this conv op run with ADK_FLAG_FUSE_GN?
int gn_device_ok(const char** why, bool may_probe = true);                          // does the current device deal workgroups evenly to 8 XCDs (probed once)?
