// SelfAttention1d / SelfAttentionMod1d core: y = softmax((q d^-1/4)(k d^-1/4)^T) v, head dim d = 32
// (DiffusionAttnUnet1D: n_head = C/32) or 64 (AudioDenoiserModel: head_size 64), flash-style (online
// softmax, the [s,s] score matrix is never materialised).  Reference blocks/blocks.py:116-121, 137-143.
//
// Layout: qkv is CPL [B, 3C, L] with channel order (q heads | k heads | v heads); a head's d channels
// are d/32 64-byte rows (bf16) or d/16 (f32) in consecutive planes, so Q/K/V tiles of a head are row ranges.
//
// bf16 / f16: attn_mfma_kernel.  One WAVE owns TQ*32 query rows of one (sample, head) and streams that head's
// K/V in 32-key tiles: K rows go global -> registers directly in A-fragment layout, V rows global -> registers ->
// a wave-private LDS double buffer (tile kt+1 is in flight while kt computes; no block barrier anywhere).
// Everything is kept "query on the lane":
//     S^T = K Q^T          v_mfma_f32_32x32x16_{bf16,f16}, A = K rows (registers, straight from global),
//                          B = Q rows (registers, loaded once);  lane (q, h) gets 16 of the 32 keys
//     online softmax       fp32, exp2 with the d^-1/2 scale folded into the exponent; the row max / sum
//                          need ONE cross-half exchange per tile
//     O^T += V^T P^T       the fp32 P^T accumulator registers, rounded to bf16, ARE the B operand
//                          (rows 8s..8s+7 = k-step s); A = V^T gathered by ds_read_b64_tr_b16
//                          in the same (permuted) key order
//   so neither P nor the [s,s] scores ever touch LDS, and the rescale factor is a per-lane scalar.
// f32: attn_mfma_f32_kernel (s >= 128: the same structure on the exact-fp32 matrix instruction) and attn_kernel (short
// sequences: one query row per lane, K/V tiles in LDS as fp32, exact fp32 VALU math); both hold rtol 1e-4.
#include <stdlib.h>
#include "adk_common.h"

struct AttnArgs {
    const void* qkv; void* y;
    int B, L, P, Lp; long long M;
    int C, heads, ntq;
};

constexpr int ATT_TQ = 64, ATT_TK = 64;

template <typename T, int ATT_D>
__global__ __launch_bounds__(64) void attn_kernel(const AttnArgs a)
{
    constexpr int CB = ElemTraits<T>::CB, NV = ElemTraits<T>::NV;
    constexpr int PPH = ATT_D / CB;                 // planes per head (1 bf16 / f16, 2 f32)
    constexpr int PPR = ATT_D / NV;                 // 16-byte pieces per head row (4 bf16 / f16, 8 f32)
    __shared__ __attribute__((aligned(16))) float Ks[ATT_TK][ATT_D];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_TK][ATT_D];

    const int tq = blockIdx.x, hd = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const long long srow = (long long)b * a.Lp + a.P;       // first valid row of the sample
    const unsigned char* base = reinterpret_cast<const unsigned char*>(a.qkv);
    auto head_piece = [&](int which, int row, int pc) -> const uint4* {
        // piece pc (of PPR) of head hd's row: plane = (which*heads + hd)*PPH + pc / 4
        const long long plane = (long long)(which * a.heads + hd) * PPH + (pc >> 2);
        return reinterpret_cast<const uint4*>(base + ((plane * a.M + srow + row) * 64) + (pc & 3) * 16);
    };

    const int qi = tq * ATT_TQ + lane;
    const bool qvalid = qi < a.L;
    float q[ATT_D], o[ATT_D];
    const float qscale = ATT_D == 32 ? 0.17677669529663687f : 0.125f;     // (d^-1/4)^2
#pragma unroll
    for (int pc = 0; pc < PPR; ++pc) {
        float f[NV];
        piece_to_float(*head_piece(0, qvalid ? qi : 0, pc), f, T());
#pragma unroll
        for (int e = 0; e < NV; ++e) q[pc * NV + e] = f[e] * qscale;
    }
#pragma unroll
    for (int d = 0; d < ATT_D; ++d) o[d] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;

    for (int k0 = 0; k0 < a.L; k0 += ATT_TK) {
        const int kn = a.L - k0 < ATT_TK ? a.L - k0 : ATT_TK;
        __syncthreads();
        for (int p = lane; p < ATT_TK * PPR; p += 64) {
            const int row = p / PPR, pc = p % PPR;
            if (row < kn) {
                float f[NV];
                piece_to_float(*head_piece(1, k0 + row, pc), f, T());
#pragma unroll
                for (int e = 0; e < NV; ++e) Ks[row][pc * NV + e] = f[e];
                piece_to_float(*head_piece(2, k0 + row, pc), f, T());
#pragma unroll
                for (int e = 0; e < NV; ++e) Vs[row][pc * NV + e] = f[e];
            }
        }
        __syncthreads();
        for (int kk = 0; kk < kn; kk += 8) {
            float s[8];
            float cmax = mrun;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float acc = 0.f;
                if (kk + u < kn) {
                    const float4* kr = reinterpret_cast<const float4*>(&Ks[kk + u][0]);
#pragma unroll
                    for (int d4 = 0; d4 < ATT_D / 4; ++d4) {
                        const float4 kv = kr[d4];
                        acc += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
                    }
                } else {
                    acc = -3.0e38f;
                }
                s[u] = acc;
                cmax = fmaxf(cmax, acc);
            }
            const float corr = expf(mrun - cmax);
            mrun = cmax;
            lrun *= corr;
#pragma unroll
            for (int d = 0; d < ATT_D; ++d) o[d] *= corr;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kk + u < kn) {
                    const float p = expf(s[u] - cmax);
                    lrun += p;
                    const float4* vr = reinterpret_cast<const float4*>(&Vs[kk + u][0]);
#pragma unroll
                    for (int d4 = 0; d4 < ATT_D / 4; ++d4) {
                        const float4 vv = vr[d4];
                        o[4 * d4] += p * vv.x; o[4 * d4 + 1] += p * vv.y;
                        o[4 * d4 + 2] += p * vv.z; o[4 * d4 + 3] += p * vv.w;
                    }
                }
            }
        }
    }

    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y);
    if (qvalid) {
        const float inv = 1.0f / lrun;
#pragma unroll
        for (int pc = 0; pc < PPR; ++pc) {
            float f[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) f[e] = o[pc * NV + e] * inv;
            const long long plane = (long long)hd * PPH + (pc >> 2);
            *reinterpret_cast<uint4*>(yb + ((plane * a.M + srow + qi) * 64) + (pc & 3) * 16) = float_to_piece(f, T());
        }
    }
    if (tq == 0 || tq == a.ntq - 1) {               // halo rows
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int p = lane; p < PPH * a.P * 4; p += 64) {
            const long long plane = (long long)hd * PPH + p / (a.P * 4);
            const int rem = p % (a.P * 4), row = rem >> 2, pc = rem & 3;
            const long long m0 = (long long)b * a.Lp + row;
            if (tq == 0) *reinterpret_cast<uint4*>(yb + (plane * a.M + m0) * 64 + pc * 16) = z;
            if (tq == a.ntq - 1) *reinterpret_cast<uint4*>(yb + (plane * a.M + m0 + a.P + a.L) * 64 + pc * 16) = z;
        }
    }
}

// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct AttnMfmaArgs {
    const void* qkv; void* y;
    int B, L, P, Lp; long long M;
    int heads, ntq, nitems;
};

constexpr int ATT_VBUF = 2048;              // one 32-key V tile, linear [key][32] bf16

// Keys past the end of the sample get a score of -inf.  Only the LAST 32-key tile of a sample can be partial and `ragged`
// is wave-uniform; register r of a lane holds key 32 kt + (r & 3) + 8 (r >> 2) + 4 h, `lim` = last - 32 kt - 4 h.  The
// limit passes through an opaque asm INSIDE the branch: otherwise the compiler hoists the 16 compares in front of the
// branch and every tile pays for them (16 of the ~76 VALU instructions per 32 x 32 score tile; round 3, tools/loop_mix.py).
__device__ __forceinline__ void att_mask_tail(f32x16& s, const bool ragged, const int lim)
{
    if (ragged) {
        int l = lim;
#ifndef ADK_EXP_ATT_OLD           // experiment builds: the previous instruction mix, for the A/B
        asm volatile("; partial key tile" : "+v"(l));
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) > l) s[r] = -INFINITY;
    }
}

// max / sum of a value with the one 32 lanes away (the two key halves of a query column).  One v_permlane32_swap of the
// value with itself gives [lo, lo] and [hi, hi]; __shfl_xor(x, 32) was a ds_bpermute: five address instructions, an LDS
// round trip and an lgkmcnt(0) that also waited for the tile's fragment reads.  Same bits (max is exact, + commutes).
__device__ __forceinline__ float att_xhalf_max(const float x)
{
#ifdef ADK_EXP_ATT_OLD
    return fmaxf(x, __shfl_xor(x, 32, 64));
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#endif
}
__device__ __forceinline__ float att_xhalf_sum(const float x)
{
#ifdef ADK_EXP_ATT_OLD
    return x + __shfl_xor(x, 32, 64);
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

template <typename T> struct AttMma;           // 16-bit element type -> fragment type + MFMA (same shapes, same rate)
template <> struct AttMma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ f32x16 run(const Frag& a, const Frag& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct AttMma<f16_t> {
    typedef f16x8 Frag;
    static __device__ __forceinline__ f32x16 run(const Frag& a, const Frag& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T, int TQ, int DP>         // DP = 64-byte planes per head (head dim 32 * DP)
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const AttnMfmaArgs a)   // 2 blocks / CU: <= 256 registers, so the MFMAs take VGPR operands (no v_accvgpr copies around the softmax)
{
    typedef typename AttMma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x * 4 + wave;                 // (sample, head, query tile), query tile fastest
    if (item >= a.nitems) return;                           // wave-uniform; the kernel has no block barrier
    const int qt = item % a.ntq, rest = item / a.ntq, hd = rest % a.heads, b = rest / a.heads;
    const long long srow = (long long)b * a.Lp + a.P;       // first valid row of the sample
    const int last = a.L - 1;
    const long long pstride = a.M * 64;                     // bytes between the planes of one head
    const unsigned char* qg = reinterpret_cast<const unsigned char*>(a.qkv) + ((long long)hd * DP * a.M + srow) * 64;
    const unsigned char* kg = qg + (long long)a.heads * DP * pstride;
    const unsigned char* vg = kg + (long long)a.heads * DP * pstride;
    unsigned char* vlds = smem + wave * (2 * DP * ATT_VBUF);     // wave-private double buffer

    // Q fragments (B operand of S^T = K Q^T): lane (q = l31, h) holds Q[q][32 p + 16 ks + 8 h .. +7]
    Frag qf[TQ][DP][2];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        int q = (qt * TQ + u) * 32 + l31;
        q = q > last ? last : q;
#pragma unroll
        for (int p = 0; p < DP; ++p)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                qf[u][p][ks] = *reinterpret_cast<const Frag*>(qg + p * pstride + (long long)q * 64 + (2 * ks + h) * 16);
    }

    // One K/V tile = 32 keys.  K goes straight to registers in A-fragment layout (lane (key = l31, h) holds
    // K[key][32 p + 16 ks + 8 h .. +7]); V is fetched as whole rows (lane: piece (lane & 3) of row 16 j + (lane >> 2)),
    // parked in the wave's LDS buffer and gathered back transposed.  Tile kt+1 is in flight while kt computes.
    struct KvTile { uint4 k[DP][2], v[DP][2]; };
    auto fetch_tile = [&](int kt) -> KvTile {
        KvTile r;
        int key = kt * 32 + l31;
        key = key > last ? last : key;                      // keys past the sample are masked below
        int vk0 = kt * 32 + (lane >> 2), vk1 = vk0 + 16;
        vk0 = vk0 > last ? last : vk0;
        vk1 = vk1 > last ? last : vk1;
#pragma unroll
        for (int p = 0; p < DP; ++p) {
            const unsigned char* kp = kg + p * pstride + (long long)key * 64 + h * 16;
            r.k[p][0] = *reinterpret_cast<const uint4*>(kp);
            r.k[p][1] = *reinterpret_cast<const uint4*>(kp + 32);
            r.v[p][0] = *reinterpret_cast<const uint4*>(vg + p * pstride + (long long)vk0 * 64 + (lane & 3) * 16);
            r.v[p][1] = *reinterpret_cast<const uint4*>(vg + p * pstride + (long long)vk1 * 64 + (lane & 3) * 16);
        }
        return r;
    };

    f32x16 o[TQ][DP];
    float mrun[TQ], lrun[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int p = 0; p < DP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[u][p][r] = 0.f;
    }

    // V^T gather (ds_read_b64_tr_b16): lane supplies row 16 s + 8 u + 4 h + ((lane & 15) >> 2), d columns
    // 16 ((lane >> 4) & 1) + 4 (lane & 3) .. +3 of the linear [key][32] image; 32 lanes cover 256 contiguous
    // bytes, so the gather is bank-conflict free
    const int voff = (4 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const float c = DP == 1 ? 0.25503486f : 0.18033688f;    // log2(e) * d^-1/2  (q and k each scaled by d^-1/4)

    const int nt = (a.L + 31) >> 5;
    KvTile cur = fetch_tile(0);
    for (int kt = 0; kt < nt; ++kt) {
        unsigned char* vb = vlds + (kt & 1) * (DP * ATT_VBUF);
        union { uint4 q; Frag f; } kf[DP][2];
#pragma unroll
        for (int p = 0; p < DP; ++p) {
            *reinterpret_cast<uint4*>(vb + p * ATT_VBUF + lane * 16) = cur.v[p][0];
            *reinterpret_cast<uint4*>(vb + p * ATT_VBUF + 1024 + lane * 16) = cur.v[p][1];
            kf[p][0].q = cur.k[p][0]; kf[p][1].q = cur.k[p][1];
        }
        if (kt + 1 < nt) cur = fetch_tile(kt + 1);

        Frag vf[DP][2];
#pragma unroll
        for (int p = 0; p < DP; ++p)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union { s16x4 v[2]; Frag f; } cv;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    cv.v[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vb + p * ATT_VBUF + voff + (16 * s + 8 * u) * 64));
                vf[p][s] = cv.f;
            }
        const bool ragged = (kt + 1) * 32 > a.L;            // wave-uniform: only the last tile can be partial

#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int p = 0; p < DP; ++p) {
                st = AttMma<T>::run(kf[p][0].f, qf[u][p][0], st);
                st = AttMma<T>::run(kf[p][1].f, qf[u][p][1], st);
            }
            // st[r] = <k_key, q_query>, key = 32 kt + (r & 3) + 8 (r >> 2) + 4 h, query = l31
            att_mask_tail(st, ragged, last - kt * 32 - 4 * h);
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = att_xhalf_max(mx);
            const float mnew = fmaxf(mrun[u], mx * c);
            const float alpha = __builtin_amdgcn_exp2f(mrun[u] - mnew);
            const bool rescale = __ballot(mnew != mrun[u]) != 0ull;   // alpha == 1 exactly on every lane otherwise
            mrun[u] = mnew;
            // pairs: one v_pk_fma_f32 and one v_pk_add_f32 per two scores (the exp2 itself is not packable)
            f32x2 ps2 = {0.f, 0.f};
            const f32x2 c2 = {c, c}, m2 = {-mnew, -mnew};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t2;
                t2.x = st[r]; t2.y = st[r + 1];
                t2 = t2 * c2 + m2;
                t2.x = __builtin_amdgcn_exp2f(t2.x); t2.y = __builtin_amdgcn_exp2f(t2.y);
                st[r] = t2.x; st[r + 1] = t2.y;
                ps2 += t2;
            }
            lrun[u] = lrun[u] * alpha + (ps2.x + ps2.y);
            Frag pf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union { uint32_t w[4]; Frag f; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // one conversion instruction per pair (pack2v): the softmax
                    f32x2 pv;                              // tile is VALU-bound, the two-cast form costs 4 per pair
                    pv.x = st[8 * s + 2 * e]; pv.y = st[8 * s + 2 * e + 1];
                    pk.w[e] = pack2v(pv, T());
                }
                pf[s] = pk.f;
            }
#pragma unroll
            for (int p = 0; p < DP; ++p) {
                if (rescale) {                            // wave-uniform: skipped once the running maxima have settled
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[u][p][r] *= alpha;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    o[u][p] = AttMma<T>::run(vf[p][s], pf[s], o[u][p]);
            }
        }
    }

    // ---- epilogue: o[u][p][r] = O[query l31][d = 32 p + (r & 3) + 8 (r >> 2) + 4 h] (unnormalised) ----
    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + ((long long)hd * DP * a.M + srow) * 64;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const float ltot = att_xhalf_sum(lrun[u]);
        const float inv = 1.0f / ltot;
        const int q = (qt * TQ + u) * 32 + l31;
        if (q <= last) {
#pragma unroll
            for (int p = 0; p < DP; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(yb + p * pstride + (long long)q * 64 + (8 * g + 4 * h) * 2) =
                        make_uint2(pack2(o[u][p][4 * g] * inv, o[u][p][4 * g + 1] * inv, T()),
                                   pack2(o[u][p][4 * g + 2] * inv, o[u][p][4 * g + 3] * inv, T()));
        }
    }
    if (qt == 0 || qt == a.ntq - 1) {                       // halo rows of this head's planes
        const uint4 z = make_uint4(0, 0, 0, 0);
        if (lane < a.P * 4) {
            const int row = lane >> 2, pc = lane & 3;
#pragma unroll
            for (int p = 0; p < DP; ++p) {
                if (qt == 0) *reinterpret_cast<uint4*>(yb + p * pstride + (long long)(row - a.P) * 64 + pc * 16) = z;
                if (qt == a.ntq - 1) *reinterpret_cast<uint4*>(yb + p * pstride + (long long)(a.L + row) * 64 + pc * 16) = z;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// f32, long sequences: attn_mfma_f32_kernel.  The same "query on the lane" flash structure as attn_mfma_kernel on
// the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (an fp32 FMA chain: parity path, rtol 1e-4):
//     S^T = K Q^T     one MFMA per pair of head-dim elements; lane (row, h) supplies element 16 (2 t + h) + s of its row
//                     at step (t, s), i.e. half h of the wave reads plane 2 t + h of the head (a 64-byte row = 16 floats),
//                     for K and Q alike (the order of the head-dim sum is free)
//     O^T += V^T P^T  one MFMA per pair of keys: the fp32 score register st[r] (key (r & 3) + 8 (r >> 2) + 4 h) IS the B
//                     operand, A = V[that key][d = lane & 31] read from a wave-private LDS tile ([key][D + 8] floats:
//                     the two half-waves read rows 4 apart, the 8-float pad puts them on different banks)
// No block barrier; the VALU attn_kernel above stays for short sequences (s < 128), where one wave per 64 queries of a
// head leaves nothing to tile.
template <int TQ, int D>
__global__ __launch_bounds__(256, 2) void attn_mfma_f32_kernel(const AttnMfmaArgs a)
{
    constexpr int PPH = D / 16;                 // 64-byte planes per head
    constexpr int NP = PPH / 2;                 // planes per half-wave
    constexpr int DP = D / 32;                  // 32-wide output blocks of the head dim
    constexpr int VSTR = D + 8;                 // floats per key row of the LDS V tile
    constexpr int VBUF = 32 * VSTR * 4;         // bytes per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.nitems) return;                           // wave-uniform; no block barrier in this kernel
    const int qt = item % a.ntq, rest = item / a.ntq, hd = rest % a.heads, b = rest / a.heads;
    const long long srow = (long long)b * a.Lp + a.P;
    const int last = a.L - 1;
    const long long pstride = a.M * 64;
    const unsigned char* qg = reinterpret_cast<const unsigned char*>(a.qkv) + ((long long)hd * PPH * a.M + srow) * 64;
    const unsigned char* kg = qg + (long long)a.heads * PPH * pstride;
    const unsigned char* vg = kg + (long long)a.heads * PPH * pstride;
    float* vlds = reinterpret_cast<float*>(smem + wave * (2 * VBUF));

    // Q: lane (q = l31, h) holds planes 2 t + h of its query row
    float4 qf[TQ][NP][4];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        int q = (qt * TQ + u) * 32 + l31;
        q = q > last ? last : q;
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                qf[u][t][e] = *reinterpret_cast<const float4*>(qg + (2 * t + h) * pstride + (long long)q * 64 + e * 16);
    }

    struct KvTile { float4 k[NP][4]; float4 v[PPH][2]; };
    auto fetch_tile = [&](int kt) -> KvTile {
        KvTile r;
        int key = kt * 32 + l31;
        key = key > last ? last : key;                      // keys past the sample are masked below
        int vk0 = kt * 32 + (lane >> 2), vk1 = vk0 + 16;
        vk0 = vk0 > last ? last : vk0;
        vk1 = vk1 > last ? last : vk1;
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r.k[t][e] = *reinterpret_cast<const float4*>(kg + (2 * t + h) * pstride + (long long)key * 64 + e * 16);
#pragma unroll
        for (int p = 0; p < PPH; ++p) {                     // whole rows: lane = piece (lane & 3) of rows (lane >> 2), + 16
            r.v[p][0] = *reinterpret_cast<const float4*>(vg + p * pstride + (long long)vk0 * 64 + (lane & 3) * 16);
            r.v[p][1] = *reinterpret_cast<const float4*>(vg + p * pstride + (long long)vk1 * 64 + (lane & 3) * 16);
        }
        return r;
    };

    f32x16 o[TQ][DP];
    float mrun[TQ], lrun[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int p = 0; p < DP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[u][p][r] = 0.f;
    }
    const float c = D == 32 ? 0.25503486f : 0.18033688f;    // log2(e) * d^-1/2  (q and k each scaled by d^-1/4)

    const int nt = (a.L + 31) >> 5;
    KvTile cur = fetch_tile(0);
    for (int kt = 0; kt < nt; ++kt) {
        float* vb = vlds + (kt & 1) * (VBUF / 4);
        float4 kf[NP][4];
#pragma unroll
        for (int p = 0; p < PPH; ++p) {
            *reinterpret_cast<float4*>(vb + (lane >> 2) * VSTR + p * 16 + (lane & 3) * 4) = cur.v[p][0];
            *reinterpret_cast<float4*>(vb + ((lane >> 2) + 16) * VSTR + p * 16 + (lane & 3) * 4) = cur.v[p][1];
        }
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) kf[t][e] = cur.k[t][e];
        if (kt + 1 < nt) cur = fetch_tile(kt + 1);
        const bool ragged = (kt + 1) * 32 > a.L;

#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NP; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t][e].x, qf[u][t][e].x, st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t][e].y, qf[u][t][e].y, st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t][e].z, qf[u][t][e].z, st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t][e].w, qf[u][t][e].w, st, 0, 0, 0);
                }
            att_mask_tail(st, ragged, last - kt * 32 - 4 * h);
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = att_xhalf_max(mx);
            const float mnew = fmaxf(mrun[u], mx * c);
            const float alpha = __builtin_amdgcn_exp2f(mrun[u] - mnew);
            mrun[u] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] * c - mnew);
                ps += st[r];
            }
            lrun[u] = lrun[u] * alpha + ps;
#pragma unroll
            for (int p = 0; p < DP; ++p) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[u][p][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = vb[((r & 3) + 8 * (r >> 2) + 4 * h) * VSTR + 32 * p + l31];
                    o[u][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, st[r], o[u][p], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: o[u][p][r] = O[query l31][d = 32 p + (r & 3) + 8 (r >> 2) + 4 h] (unnormalised) ----
    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + ((long long)hd * PPH * a.M + srow) * 64;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const float ltot = att_xhalf_sum(lrun[u]);
        const float inv = 1.0f / ltot;
        const int q = (qt * TQ + u) * 32 + l31;
        if (q <= last) {
#pragma unroll
            for (int p = 0; p < DP; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * p + 8 * g + 4 * h;               // 4 consecutive head-dim elements
                    *reinterpret_cast<float4*>(yb + (d0 >> 4) * pstride + (long long)q * 64 + (d0 & 15) * 4) =
                        make_float4(o[u][p][4 * g] * inv, o[u][p][4 * g + 1] * inv, o[u][p][4 * g + 2] * inv, o[u][p][4 * g + 3] * inv);
                }
        }
    }
    if (qt == 0 || qt == a.ntq - 1) {                       // halo rows of this head's planes
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < a.P * 4) {
            const int row = lane >> 2, pc = lane & 3;
#pragma unroll
            for (int p = 0; p < PPH; ++p) {
                if (qt == 0) *reinterpret_cast<float4*>(yb + p * pstride + (long long)(row - a.P) * 64 + pc * 16) = z;
                if (qt == a.ntq - 1) *reinterpret_cast<float4*>(yb + p * pstride + (long long)(a.L + row) * 64 + pc * 16) = z;
            }
        }
    }
}

template <int TQ, int D>
static int launch_attn_mfma_f32(const adk_op& op, hipStream_t st)
{
    AttnMfmaArgs a;
    a.qkv = op.x; a.y = op.y;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.heads = op.Cin / D; a.ntq = (op.L + 32 * TQ - 1) / (32 * TQ);
    a.nitems = op.B * a.heads * a.ntq;
    const size_t lds = (size_t)4 * 2 * 32 * (D + 8) * 4;
    static std::atomic<unsigned long long> attr_set{0};   // d = 64: 72 KiB of dynamic LDS needs the opt-in, once per kernel and device
    if (adk_lds_optin(attr_set, reinterpret_cast<const void*>(&attn_mfma_f32_kernel<TQ, D>), lds)) return ADK_ELAUNCH;
    hipLaunchKernelGGL((attn_mfma_f32_kernel<TQ, D>), dim3((a.nitems + 3) / 4), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

// ------------------------------------------------------------------------------------------------
// Short sequences (s <= 256, head dim 32, 16-bit): norm -> qkv_proj -> attention core in ONE launch (ADK_FLAG_QKV).
// One block per (sample, head).  Phase 1: the head's 96 rows of the folded projection (ADK_FLAG_PRE_NORM form: the
// weights carry the GroupNorm's gamma, the per-sample (mean, rstd) scale the result) times the raw x[C, s], operands
// straight from global memory in fragment layout (wave w owns position tiles w and w + 4, chunks double-buffered in
// registers); q, k, v land in LDS as [position][32] rows.  Phase 2: attn_mfma_kernel's flash loop with every operand
// read from LDS (V^T by ds_read_b64_tr_b16 straight from the v rows).  Saves a launch and the qkv tensor per
// attention block at the levels where a launch costs more than its work (36 per call of the uncond model).
struct AttnQkvArgs {
    const void* x; const void* w; const float* u; const float* vb; const float* partials; void* y;
    int B, L, P, Lp; long long M;
    int C, heads, CoutPad, pn_n;
    float eps; double count;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_qkv_kernel(const AttnQkvArgs a)
{
    typedef typename AttMma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = blockIdx.x % a.heads, b = blockIdx.x / a.heads;
    const long long srow = (long long)b * a.Lp + a.P;
    const int last = a.L - 1;
    const int npt = (a.L + 31) >> 5;                        // 32-row tiles (<= 8)
    unsigned char* lq = smem;                               // [which][npt * 32][64 B]
    const int region = npt * 2048;

    float mean, rstd;
    gn_finalize_wave(a.partials + (long long)b * a.pn_n * 2, a.pn_n, a.count, a.eps, mean, rstd);
    const float mrs = mean * rstd;

    // ---- phase 1: q, k, v of this head ----
    // s > 64: wave w owns position tiles w and w + 4 and walks all channel chunks.  s <= 64 (at most two tiles): the chunk
    // chain is what costs time, so every wave takes all tiles and every fourth chunk; the four partial accumulators meet
    // in LDS and are added in wave order (fixed -> bit-reproducible).
    {
        const bool ksplit = npt <= 2;                       // block-uniform
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x) + srow * 64 + h * 16;
        const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w) + (long long)(hd * 32 + l31) * 64 + h * 16;
        const long long xplane = a.M * 64, wchunk = (long long)a.CoutPad * 64, wwhich = (long long)a.C * 64;
        const int nck = a.C >> 5;
        const int c0 = ksplit ? wave : 0, cs = ksplit ? 4 : 1;
        const int nmy = nck > c0 ? (nck - c0 + cs - 1) / cs : 0;
        int pos[2], ptile[2];
        bool act[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            ptile[jt] = ksplit ? jt : wave + 4 * jt;
            act[jt] = ptile[jt] < npt;                      // wave-uniform
            const int p = ptile[jt] * 32 + l31;
            pos[jt] = p > last ? last : p;
        }
        f32x16 acc[3][2];
#pragma unroll
        for (int wh = 0; wh < 3; ++wh)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[wh][jt][r] = 0.f;
        if (act[0] && nmy > 0) {
            Frag fa[2][3][2], fb[2][2][2];
#define ADK_AQ_LOAD(D_, I_)                                                                                  \
            {                                                                                                \
                const int kc_ = c0 + (I_) * cs;                                                              \
                const unsigned char* wp_ = wg + (long long)kc_ * wchunk;                                     \
                const unsigned char* xp_ = xg + (long long)kc_ * xplane;                                     \
                _Pragma("unroll") for (int wh = 0; wh < 3; ++wh)                                             \
                    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
                        fa[D_][wh][ks] = *reinterpret_cast<const Frag*>(wp_ + wh * wwhich + ks * 32);        \
                _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                             \
                    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
                        fb[D_][jt][ks] = *reinterpret_cast<const Frag*>(xp_ + (long long)pos[jt] * 64 + ks * 32); \
            }
            ADK_AQ_LOAD(0, 0)
            if (nmy > 1) ADK_AQ_LOAD(1, 1)
            for (int i0 = 0; i0 < nmy; i0 += 2) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    if (i0 + d < nmy) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int wh = 0; wh < 3; ++wh) {
                                acc[wh][0] = AttMma<T>::run(fa[d][wh][ks], fb[d][0][ks], acc[wh][0]);
                                if (act[1]) acc[wh][1] = AttMma<T>::run(fa[d][wh][ks], fb[d][1][ks], acc[wh][1]);
                            }
                        if (i0 + d + 2 < nmy) ADK_AQ_LOAD(d, i0 + d + 2)
                    }
                }
            }
#undef ADK_AQ_LOAD
        }
        float4* red = reinterpret_cast<float4*>(smem + 3 * region);      // [wave][which][tile][g][lane], k-split only
        if (ksplit) {
#pragma unroll
            for (int wh = 0; wh < 3; ++wh)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    if (act[jt]) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            red[(((wave * 3 + wh) * 2 + jt) * 4 + g) * 64 + lane] =
                                make_float4(acc[wh][jt][4 * g], acc[wh][jt][4 * g + 1], acc[wh][jt][4 * g + 2], acc[wh][jt][4 * g + 3]);
                    }
            __syncthreads();
        }
        // scale / shift, round, park as [position][32] rows: lane (pos = l31, h) holds channels (r & 3) + 8 (r >> 2) + 4 h.
        // k-split: wave w finishes the (which, tile) pairs w and w + 4 of the six.
#pragma unroll
        for (int wh = 0; wh < 3; ++wh)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bool mine = ksplit ? (act[jt] && ((wh * 2 + jt) & 3) == wave) : act[jt];
                if (mine) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 v = make_float4(acc[wh][jt][4 * g], acc[wh][jt][4 * g + 1], acc[wh][jt][4 * g + 2], acc[wh][jt][4 * g + 3]);
                        if (ksplit) {
                            v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                            for (int sw = 0; sw < 4; ++sw) {
                                const float4 p4 = red[(((sw * 3 + wh) * 2 + jt) * 4 + g) * 64 + lane];
                                v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
                            }
                        }
                        const int co = wh * a.C + hd * 32 + 8 * g + 4 * h;
                        const float4 u4 = *reinterpret_cast<const float4*>(a.u + co), v4 = *reinterpret_cast<const float4*>(a.vb + co);
                        const int row = ptile[jt] * 32 + l31;
                        const uint2 pk = make_uint2(
                            pack2(v.x * rstd + (v4.x - mrs * u4.x), v.y * rstd + (v4.y - mrs * u4.y), T()),
                            pack2(v.z * rstd + (v4.z - mrs * u4.z), v.w * rstd + (v4.w - mrs * u4.w), T()));
                        *reinterpret_cast<uint2*>(lq + wh * region + row * 64 + (8 * g + 4 * h) * 2) = pk;
                    }
                }
            }
    }
    __syncthreads();

    // ---- phase 2: flash loop of attn_mfma_kernel (TQ = 1), operands from LDS ----
    const unsigned char* lk = lq + region;
    const unsigned char* lv = lq + 2 * region;
    const int voff = (4 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const float c = 0.25503486f;                            // log2(e) * 32^-1/2
    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + ((long long)hd * a.M + srow) * 64;
    for (int qt = wave; qt < npt; qt += 4) {
        Frag qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const Frag*>(lq + (qt * 32 + l31) * 64 + (2 * ks + h) * 16);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float mrun = -INFINITY, lrun = 0.f;
        for (int kt = 0; kt < npt; ++kt) {
            Frag kf[2], vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[ks] = *reinterpret_cast<const Frag*>(lk + (kt * 32 + l31) * 64 + (2 * ks + h) * 16);
#pragma unroll
            for (int sgrp = 0; sgrp < 2; ++sgrp) {
                union { s16x4 v[2]; Frag f; } cv;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    cv.v[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(lv + kt * 2048 + voff + (16 * sgrp + 8 * u) * 64));
                vf[sgrp] = cv.f;
            }
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            st = AttMma<T>::run(kf[0], qf[0], st);
            st = AttMma<T>::run(kf[1], qf[1], st);
            att_mask_tail(st, (kt + 1) * 32 > a.L, last - kt * 32 - 4 * h);
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = att_xhalf_max(mx);
            const float mnew = fmaxf(mrun, mx * c);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] * c - mnew);
                ps += st[r];
            }
            lrun = lrun * alpha + ps;
            Frag pf[2];
#pragma unroll
            for (int sgrp = 0; sgrp < 2; ++sgrp) {
                union { uint32_t w[4]; Frag f; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 pv;
                    pv.x = st[8 * sgrp + 2 * e]; pv.y = st[8 * sgrp + 2 * e + 1];
                    pk.w[e] = pack2v(pv, T());
                }
                pf[sgrp] = pk.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            o = AttMma<T>::run(vf[0], pf[0], o);
            o = AttMma<T>::run(vf[1], pf[1], o);
        }
        const float ltot = att_xhalf_sum(lrun);
        const float inv = 1.0f / ltot;
        const int q = qt * 32 + l31;
        if (q <= last) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(yb + (long long)q * 64 + (8 * g + 4 * h) * 2) =
                    make_uint2(pack2(o[4 * g] * inv, o[4 * g + 1] * inv, T()), pack2(o[4 * g + 2] * inv, o[4 * g + 3] * inv, T()));
        }
    }
    if (wave == 0 && lane < a.P * 4) {                      // halo rows of this head's plane
        const uint4 z = make_uint4(0, 0, 0, 0);
        const int row = lane >> 2, pc = lane & 3;
        *reinterpret_cast<uint4*>(yb + (long long)(row - a.P) * 64 + pc * 16) = z;
        *reinterpret_cast<uint4*>(yb + (long long)(a.L + row) * 64 + pc * 16) = z;
    }
}

template <typename T>
static int launch_attn_qkv(const adk_op& op, hipStream_t st)
{
    AttnQkvArgs a;
    a.x = op.x; a.w = op.w; a.u = op.gamma; a.vb = reinterpret_cast<const float*>(op.bias); a.partials = op.partials_in; a.y = op.y;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.C = op.Cin; a.heads = op.Cin / 32; a.CoutPad = op.CoutPad; a.pn_n = op.n_partials_in;
    a.eps = op.f0; a.count = (double)op.i2 * (double)op.L;
    const int npt = (op.L + 31) / 32;
    const size_t lds = (size_t)3 * npt * 2048 + (npt <= 2 ? (size_t)4 * 3 * 2 * 4 * 64 * 16 : 0);   // + the k-split partials
    static std::atomic<unsigned long long> attr_set{0};   // > 64 KiB of dynamic LDS needs the opt-in, once per kernel and device
    if (adk_lds_optin(attr_set, reinterpret_cast<const void*>(&attn_qkv_kernel<T>), (size_t)3 * 2 * 2048 + 4 * 3 * 2 * 4 * 64 * 16))
        return ADK_ELAUNCH;
    hipLaunchKernelGGL((attn_qkv_kernel<T>), dim3(op.B * a.heads), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

// ------------------------------------------------------------------------------------------------
// Long sequences (s >= 512, head dim 32, 16-bit): attn_mfma_lds_kernel.  Same arithmetic as attn_mfma_kernel, but the four
// waves of a block own four consecutive query tiles of ONE (sample, head) and share every K/V tile: a 64-key stage
// (4 KiB of K rows + 4 KiB of V rows) reaches LDS once per block by LDS-DMA (one global_load_lds_dwordx4 per wave for K
// and one for V), three stages in a ring, one block barrier per stage.  attn_mfma_kernel has every wave stream the whole
// K/V of its head from L2 on its own: four times the L2 traffic and a dependent global round trip per 32 keys.
// K rows are XOR-swizzled through the DMA's source address (piece p of row r in slot p ^ ((r >> 2) & 3), as in the conv
// kernels) so the ds_read_b128 fragment reads are conflict-free; V rows stay linear for ds_read_b64_tr_b16.
template <int N> __device__ __forceinline__ void att_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void att_glds16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ds_read_b64_tr_b16 as inline asm: the compiler treats the intrinsic as a read that may alias the LDS-DMA writes in flight
// and puts a vmcnt(0) in front of it (draining the prefetched stages); the asm form is invisible to that bookkeeping, so
// the lgkmcnt wait is written by hand (att_lds_wait, which carries the registers as in/out operands: a data dependency
// the scheduler cannot move the consuming MFMAs across).
__device__ __forceinline__ s16x4 att_ds_read_tr(const unsigned char* p) {
    s16x4 r;
    const unsigned addr = (unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)p);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void att_lds_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

template <typename T, int TQ>
__global__ __launch_bounds__(256, 2) void attn_mfma_lds_kernel(const AttnMfmaArgs a)
{
    typedef typename AttMma<T>::Frag Frag;
    constexpr int STAGE = 8192;                             // 64 keys: K rows then V rows
    constexpr int UG = TQ < 2 ? TQ : 2;                     // query sub-tiles whose MFMAs / softmax are interleaved
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];    // [3][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (a.ntq + 3) >> 2;                       // blocks per (sample, head)
    const int qb = blockIdx.x % nqb, rest = blockIdx.x / nqb, hd = rest % a.heads, b = rest / a.heads;
    const int qt = qb * 4 + wave;                           // this wave's query tile (TQ * 32 queries)
    const bool active = qt < a.ntq;                         // wave-uniform; idle waves still stage tiles and meet the barriers
    const long long srow = (long long)b * a.Lp + a.P;
    const int last = a.L - 1;
    const long long pstride = a.M * 64;
    const unsigned char* qg = reinterpret_cast<const unsigned char*>(a.qkv) + ((long long)hd * a.M + srow) * 64;
    const unsigned char* kg = qg + (long long)a.heads * pstride;
    const unsigned char* vg = kg + (long long)a.heads * pstride;

    Frag qf[TQ][2];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        int q = (qt * TQ + u) * 32 + l31;
        q = q > last ? last : q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[u][ks] = *reinterpret_cast<const Frag*>(qg + (long long)q * 64 + (2 * ks + h) * 16);
    }

    // staging: wave w moves rows 16 w .. 16 w + 15 of a stage's K tile and of its V tile (lane: slot lane & 3 of row lane >> 2)
    const int srow16 = wave * 16 + (lane >> 2), sslot = lane & 3;
    const int kswz = (sslot ^ ((srow16 >> 2) & 3)) << 4;    // source piece of this lane's K slot
    auto stage_in = [&](int st) {
        int key = st * 64 + srow16;
        key = key > last ? last : key;                      // keys past the sample are masked below
        unsigned char* dst = smem + (st % 3) * STAGE + wave * 1024;
        att_glds16(kg + (long long)key * 64 + kswz, dst);
        att_glds16(vg + (long long)key * 64 + (sslot << 4), dst + 4096);
    };

    f32x16 o[TQ];
    float mrun[TQ], lrun[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
    }
    const int voff = (4 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int kfo0 = l31 * 64 + (((0 + h) ^ ((l31 >> 2) & 3)) << 4), kfo1 = l31 * 64 + (((2 + h) ^ ((l31 >> 2) & 3)) << 4);
    const float c = 0.25503486f;                            // log2(e) * 32^-1/2

    const int nst = (a.L + 63) >> 6;
    // The Q rows must have landed before the first LDS-DMA goes out, and the compiler has to KNOW it (intrinsic, not asm):
    // otherwise its own bookkeeping puts a vmcnt(0) - which also drains the stages in flight - in front of every MFMA
    // that reads qf.  One global round trip per block of 256 queries.
    __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xf << 8));
    asm volatile("" ::: "memory");
    stage_in(0);
    if (nst > 1) stage_in(1);
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) att_wait_vmcnt<2>(); else att_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 2 < nst) stage_in(st + 2);                 // into the buffer stage st - 1 used: every wave is past it
        if (!active) continue;
        const unsigned char* kb = smem + (st % 3) * STAGE;
        const unsigned char* vbase = kb + 4096;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kt = st * 2 + half;                   // 32-key tile
            if (kt * 32 > last) break;                      // wave-uniform
            Frag kf[2], vf[2];
            kf[0] = *reinterpret_cast<const Frag*>(kb + half * 2048 + kfo0);
            kf[1] = *reinterpret_cast<const Frag*>(kb + half * 2048 + kfo1);
            {
                union { s16x4 v[2]; Frag f; } cv0, cv1;
                cv0.v[0] = att_ds_read_tr(vbase + half * 2048 + voff);
                cv0.v[1] = att_ds_read_tr(vbase + half * 2048 + voff + 8 * 64);
                cv1.v[0] = att_ds_read_tr(vbase + half * 2048 + voff + 16 * 64);
                cv1.v[1] = att_ds_read_tr(vbase + half * 2048 + voff + 24 * 64);
                att_lds_wait(cv0.v[0], cv0.v[1], cv1.v[0], cv1.v[1]);
                vf[0] = cv0.f; vf[1] = cv1.f;
            }
            const bool ragged = (kt + 1) * 32 > a.L;
            // all score MFMAs of the tile first, then the softmax of every query sub-tile, then the P V MFMAs: the
            // matrix-pipe latency of one sub-tile hides behind the VALU work of the other
#pragma unroll
            for (int u0 = 0; u0 < TQ; u0 += UG) {
            f32x16 sc[TQ];
#pragma unroll
            for (int u = u0; u < u0 + UG; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
                sc[u] = AttMma<T>::run(kf[0], qf[u][0], sc[u]);
                sc[u] = AttMma<T>::run(kf[1], qf[u][1], sc[u]);
            }
            Frag pf[TQ][2];
            float alpha[TQ];
            bool rescale[TQ];
#pragma unroll
            for (int u = u0; u < u0 + UG; ++u) {
                att_mask_tail(sc[u], ragged, last - kt * 32 - 4 * h);
                float mx = sc[u][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[u][r]);
                mx = att_xhalf_max(mx);
                const float mnew = fmaxf(mrun[u], mx * c);
                alpha[u] = __builtin_amdgcn_exp2f(mrun[u] - mnew);
                rescale[u] = __ballot(mnew != mrun[u]) != 0ull;
                mrun[u] = mnew;
                f32x2 ps2 = {0.f, 0.f};
                const f32x2 c2 = {c, c}, m2 = {-mnew, -mnew};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 t2;
                    t2.x = sc[u][r]; t2.y = sc[u][r + 1];
                    t2 = t2 * c2 + m2;
                    t2.x = __builtin_amdgcn_exp2f(t2.x); t2.y = __builtin_amdgcn_exp2f(t2.y);
                    sc[u][r] = t2.x; sc[u][r + 1] = t2.y;
                    ps2 += t2;
                }
                lrun[u] = lrun[u] * alpha[u] + (ps2.x + ps2.y);
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    union { uint32_t w[4]; Frag f; } pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 pv;
                        pv.x = sc[u][8 * sgrp + 2 * e]; pv.y = sc[u][8 * sgrp + 2 * e + 1];
                        pk.w[e] = pack2v(pv, T());
                    }
                    pf[u][sgrp] = pk.f;
                }
            }
#pragma unroll
            for (int u = u0; u < u0 + UG; ++u) {
                if (rescale[u]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[u][r] *= alpha[u];
                }
                o[u] = AttMma<T>::run(vf[0], pf[u][0], o[u]);
                o[u] = AttMma<T>::run(vf[1], pf[u][1], o[u]);
            }
            }
        }
    }
    if (!active) return;

    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + ((long long)hd * a.M + srow) * 64;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const float ltot = att_xhalf_sum(lrun[u]);
        const float inv = 1.0f / ltot;
        const int q = (qt * TQ + u) * 32 + l31;
        if (q <= last) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(yb + (long long)q * 64 + (8 * g + 4 * h) * 2) =
                    make_uint2(pack2(o[u][4 * g] * inv, o[u][4 * g + 1] * inv, T()),
                               pack2(o[u][4 * g + 2] * inv, o[u][4 * g + 3] * inv, T()));
        }
    }
    if (qt == 0 || qt == a.ntq - 1) {                       // halo rows of this head's plane
        const uint4 z = make_uint4(0, 0, 0, 0);
        if (lane < a.P * 4) {
            const int row = lane >> 2, pc = lane & 3;
            if (qt == 0) *reinterpret_cast<uint4*>(yb + (long long)(row - a.P) * 64 + pc * 16) = z;
            if (qt == a.ntq - 1) *reinterpret_cast<uint4*>(yb + (long long)(a.L + row) * 64 + pc * 16) = z;
        }
    }
}

template <typename T, int TQ>
static int launch_attn_mfma_lds(const adk_op& op, hipStream_t st)
{
    AttnMfmaArgs a;
    a.qkv = op.x; a.y = op.y;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.heads = op.Cin / 32; a.ntq = (op.L + 32 * TQ - 1) / (32 * TQ);
    a.nitems = op.B * a.heads * ((a.ntq + 3) / 4);          // blocks
    hipLaunchKernelGGL((attn_mfma_lds_kernel<T, TQ>), dim3(a.nitems), dim3(256), 3 * 8192, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

template <typename T, int TQ, int DP>
static int launch_attn_mfma(const adk_op& op, hipStream_t st)
{
    AttnMfmaArgs a;
    a.qkv = op.x; a.y = op.y;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.heads = op.Cin / (32 * DP); a.ntq = (op.L + 32 * TQ - 1) / (32 * TQ);
    a.nitems = op.B * a.heads * a.ntq;
    const size_t lds = 4 * 2 * DP * ATT_VBUF;
    hipLaunchKernelGGL((attn_mfma_kernel<T, TQ, DP>), dim3((a.nitems + 3) / 4), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

template <typename T, int D>
static int launch_attn_valu(const adk_op& op, hipStream_t st)
{
    AttnArgs a;
    a.qkv = op.x; a.y = op.y;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.C = op.Cin; a.heads = op.Cin / D; a.ntq = (op.L + ATT_TQ - 1) / ATT_TQ;
    dim3 grid(a.ntq, a.heads, op.B);
    hipLaunchKernelGGL((attn_kernel<T, D>), grid, dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

template <typename T>
static int launch_attn_16(const adk_op& op, hipStream_t st, bool d64)
{
    if (op.P * 4 <= 64) {
        if (d64) return op.L >= 512 ? launch_attn_mfma<T, 2, 2>(op, st) : launch_attn_mfma<T, 1, 2>(op, st);
        if (op.L >= 512) {
            static const bool own = getenv("ADK_ATTN_NO_LDS") != nullptr;      // experiment: every wave streams K/V itself
            // (128 queries per wave, TQ = 4, was measured too: 280.9 vs 282.2 us at s = 4096, B = 4 and 924 vs 921 us at B = 16 -
            // the loop is bound by the softmax's VALU work, not by operand delivery; not dispatched)
            return own ? launch_attn_mfma<T, 2, 1>(op, st) : launch_attn_mfma_lds<T, 2>(op, st);
        }
        return launch_attn_mfma<T, 1, 1>(op, st);
    }
    return d64 ? launch_attn_valu<T, 64>(op, st) : launch_attn_valu<T, 32>(op, st);
}

int launch_attn(const adk_op& op, hipStream_t st)
{
    if (op.flags & ADK_FLAG_QKV) return op.dtype == ADK_F16 ? launch_attn_qkv<f16_t>(op, st) : launch_attn_qkv<bf16_t>(op, st);
    const bool d64 = op.i0 == 64;                           // head dim: op.i0 (0 = 32)
    if (op.dtype == ADK_F32) {
        if (op.L >= 128 && op.P * 4 <= 64)                  // exact-fp32 matrix instruction from s = 128 up
            return d64 ? launch_attn_mfma_f32<1, 64>(op, st)
                       : (op.L >= 512 ? launch_attn_mfma_f32<2, 32>(op, st) : launch_attn_mfma_f32<1, 32>(op, st));
        return d64 ? launch_attn_valu<float, 64>(op, st) : launch_attn_valu<float, 32>(op, st);
    }
    if (op.dtype == ADK_F16) return launch_attn_16<f16_t>(op, st, d64);
    return launch_attn_16<bf16_t>(op, st, d64);
}
