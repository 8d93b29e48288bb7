// Kernel templates of the implicit-GEMM convolution (instantiated by conv_mfma.hip: the plain centred convs of the
// hot path, EXT = false; and by conv_mfma_ext.hip: EXT = true adds the runtime tap shift (causal / dilated taps), the
// ELU epilogue and the per-32-channel-group statistics, so none of that costs the hot path a single instruction).
#pragma once
#include "conv_common.h"

#ifdef ADK_ABL_TS       // experiment builds only: per-block timestamps (make exp-ts; tools/conv_block_timeline.py)
__device__ unsigned long long g_adk_ts[16384 * 9];     // [block][8] stamps, then [block] start of the K-loop prologue (fused kernel)
#define ADK_TS(SLOT_) if (threadIdx.x == 0 && blockIdx.x < 16384) g_adk_ts[blockIdx.x * 8 + (SLOT_)] = wall_clock64();
#else
#define ADK_TS(SLOT_)
#endif

template <bool EXT, bool RES, typename TO, int BN, int BM, int WPOS, int TCO, int TPOS>
__device__ __forceinline__ void conv_epilogue_impl(const ConvArgs& a, f32x16 (&acc)[TCO][TPOS], unsigned char* smem,
                                                   int b, int t, int cot, int cobase, int l0, int tid, int l31, int h,
                                                   int wco, int wpos)
{
    constexpr int CBO = ElemTraits<TO>::CB;
    // ---- epilogue: lane (l31, h) holds D[co = 8g + 4h + e][pos = l31] in acc[..][4g + e] ----
    // 16-bit output: the two half-waves hold the two 8-byte halves of one 16-byte piece, so channel groups are
    // processed in pairs (g, g+1) and exchanged with v_permlane32_swap: lanes h=0 then store channels
    // 8g..8g+7, lanes h=1 channels 8g+8..8g+15, as ONE 16-byte store each (half the store instructions:
    // the tail of this kernel is store-issue bound, not bandwidth bound).
    TO* yg = reinterpret_cast<TO*>(a.y);
    const TO* rg = reinterpret_cast<const TO*>(a.residual);
    // (sum, sumsq) as packed pairs: one v_pk_add_f32 + one v_pk_fma_f32 per two values (the scalar form cost three
    // instructions per value in moves and half-used packed ops; the epilogue is VALU-bound: 5 of its 7 us)
    f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
    f32x2 gs1v[TCO], gs2v[TCO];                                  // per 32-channel group (ADK_FLAG_GROUP_STATS)
#pragma unroll
    for (int i = 0; i < TCO; ++i) { gs1v[i] = s1v; gs2v[i] = s1v; }
    // All loads of the epilogue are issued BEFORE its first store.  On gfx9 loads and stores share one in-order
    // counter (vmcnt), so a load issued after a store cannot be waited for without also waiting for that store's write
    // acknowledgement (~1 us under load): with bias / residual loads interleaved per channel group the epilogue ran as
    // 16-32 serial round trips and took as long as the K loop of a 128-channel level (per-block timestamps: 13 us of
    // 25).  The bias is not loaded here at all: the accumulators START at the bias (conv_acc_init, fetched while the
    // first tiles are in flight), so acc already is conv + bias.
    ADK_TS(4)
    // ADK_FLAG_PRE_NORM (EXT only): y = rstd_b * acc + (beta[co] - mean_b * rstd_b * gamma[co]); the shifts of this lane's
    // channels are built before the first store
    f32x2 pnA[TCO][4], pnB[TCO][4], pn_rs = {1.f, 1.f};
    bool prenorm = false;
    if constexpr (EXT) {
        prenorm = a.pn_partials != nullptr;
        if (prenorm) {
            float mean, rstd;
            gn_finalize_wave(a.pn_partials + (long long)b * a.pn_n * 2, a.pn_n, a.gn_count, a.eps, mean, rstd);
            pn_rs.x = pn_rs.y = rstd;
            const float mrs = mean * rstd;
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cobase + (wco * TCO + i) * 32 + 8 * g + 4 * h;
                    const float4 u = *reinterpret_cast<const float4*>(a.gamma + co), v = *reinterpret_cast<const float4*>(a.beta + co);
                    pnA[i][g].x = v.x - mrs * u.x; pnA[i][g].y = v.y - mrs * u.y;
                    pnB[i][g].x = v.z - mrs * u.z; pnB[i][g].y = v.w - mrs * u.w;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < TPOS; ++j) {
        const int l = l0 + (wpos * TPOS + j) * 32 + l31;
        const bool lvalid = l < a.L;
        const long long m = (long long)b * a.Lp + a.P + l;
        // this position's residual pieces: raw loads first (one round trip), conversion at the point of use; lanes
        // outside the tensor read a valid address of their own sample and discard it
        typedef typename RawVec4<TO>::type Raw4;
        Raw4 rraw[TCO][4];
        if constexpr (RES) {
            const long long msafe = lvalid ? m : (long long)b * a.Lp + a.P;
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int co = cobase + (wco * TCO + i) * 32 + 8 * g + 4 * h;
                    co = co < a.Cout_alloc ? co : 4 * h;
                    rraw[i][g] = *reinterpret_cast<const Raw4*>(rg + ((((long long)(co / CBO) * a.M + msafe) * CBO) + (co % CBO)));
                }
        }
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co32 = cobase + (wco * TCO + i) * 32;
            const bool ok = lvalid && co32 < a.Cout_alloc;      // Cout_alloc is a multiple of 32 when TO is 16-bit
#pragma unroll
            for (int g0 = 0; g0 < 4; g0 += 2) {
                f32x2 va[2], vb[2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int G = g0 + gg, co = co32 + 8 * G + 4 * h;
                    va[gg].x = acc[i][j][4 * G]; va[gg].y = acc[i][j][4 * G + 1];
                    vb[gg].x = acc[i][j][4 * G + 2]; vb[gg].y = acc[i][j][4 * G + 3];
                    if (lvalid && co < a.Cout_alloc) {
                        if constexpr (EXT) {
                            if (prenorm) { va[gg] = va[gg] * pn_rs + pnA[i][G]; vb[gg] = vb[gg] * pn_rs + pnB[i][G]; }
                        }
                        if constexpr (RES) {
                            float rr[4];
                            raw_to_float4(rraw[i][G], rr, TO());
                            f32x2 r0, r1;
                            r0.x = rr[0]; r0.y = rr[1]; r1.x = rr[2]; r1.y = rr[3];
                            va[gg] += r0; vb[gg] += r1;
                        }
                        if constexpr (EXT) {
                            if (a.elu) {
                                va[gg].x = elu1(va[gg].x); va[gg].y = elu1(va[gg].y);
                                vb[gg].x = elu1(vb[gg].x); vb[gg].y = elu1(vb[gg].y);
                            }
                            gs1v[i] += va[gg]; gs1v[i] += vb[gg];
                            gs2v[i] = va[gg] * va[gg] + gs2v[i]; gs2v[i] = vb[gg] * vb[gg] + gs2v[i];
                        }
                        s1v += va[gg]; s1v += vb[gg];
                        s2v = va[gg] * va[gg] + s2v; s2v = vb[gg] * vb[gg] + s2v;
                        if constexpr (sizeof(TO) == 4) {
                            const long long off = (((long long)(co / CBO) * a.M + m) * CBO) + (co % CBO);
                            const float v4[4] = {va[gg].x, va[gg].y, vb[gg].x, vb[gg].y};
                            store4(yg + off, v4);
                        }
                    }
                }
                if constexpr (sizeof(TO) == 2) {
                    uint32_t ax = pack2(va[0].x, va[0].y, TO()), ay = pack2(vb[0].x, vb[0].y, TO());
                    uint32_t bx = pack2(va[1].x, va[1].y, TO()), by = pack2(vb[1].x, vb[1].y, TO());
                    // swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second (all lanes active here)
                    const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    if (ok) {
                        // channels co32 + 8 g0 + 8 h .. +7 (h=0: [own g0 | partner's g0], h=1: [partner's g0+1 | own g0+1]):
                        // co32 is a multiple of 32 = CBO, so the plane is co32 / 32 and the piece 16 (g0 + h) bytes into the row
                        unsigned char* dst = reinterpret_cast<unsigned char*>(a.y) +
                                             ((long long)(co32 >> 5) * a.M + m) * 64 + (g0 + h) * 16;
                        *reinterpret_cast<uint4*>(dst) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
            }
        }
    }

    ADK_TS(5)
    // halo rows of the output: zeroed by the first / last tile of each sample
    if (t == 0 || t == a.tps - 1) {
        const int plane0 = cobase / CBO;
        const int cend = cobase + BN < a.Cout_alloc ? cobase + BN : a.Cout_alloc;
        const int nplanes = cend / CBO - plane0;
        const int per = a.P * 4;
        const uint4 z = make_uint4(0, 0, 0, 0);
        unsigned char* yb = reinterpret_cast<unsigned char*>(a.y);
        for (int p = tid; p < nplanes * per; p += (int)blockDim.x) {
            const int pl = plane0 + p / per, rem = p % per, row = rem >> 2, q = rem & 3;
            const long long mh = (long long)b * a.Lp + row;
            if (t == 0) *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh) * 64 + q * 16) = z;
            if (t == a.tps - 1)
                *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh + a.P + a.L) * 64 + q * 16) = z;
        }
    }

    if (EXT && a.partials && a.gstats) {
        // one (sum, sumsq) pair per (sample, 32-channel group, position tile, wave column): [b][g][t * WPOS + wpos];
        // no cross-wave step, the consumer (gn_kernel) adds a group's pairs in index order
        const int ngrp = (a.Cout_alloc + 31) >> 5;
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const float r1 = wave_sum(gs1v[i].x + gs1v[i].y), r2 = wave_sum(gs2v[i].x + gs2v[i].y);
            const int g = (cobase >> 5) + wco * TCO + i;
            if ((tid & 63) == 0 && g < ngrp) {
                float* dst = a.partials + ((((long long)b * ngrp + g) * a.tps + t) * WPOS + wpos) * 2;
                dst[0] = r1; dst[1] = r2;
            }
        }
    } else if (a.partials) {
        float* red = reinterpret_cast<float*>(smem);
        float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
        block_sum2(s1, s2, red);
        if (tid == 0) {
            float* dst = a.partials + ((long long)(b * a.tps + t) * a.ncot + cot) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
    ADK_TS(6)
}

// Accumulators start at the bias: lane (l31, h) holds D[co = 8 g + 4 h + e][pos] in acc[..][4 g + e], so the value depends
// on (i, g, e, h) only.  conv_bias_fetch issues the 8 float4 loads before the first LDS-DMA unit (older in the in-order
// vmcnt queue: the counted waits of the K loop are unaffected, and the latency hides behind the first tiles' flight);
// conv_acc_set copies them into the accumulators right after the prologue's wait, which already covers them.
template <int TCO>
__device__ __forceinline__ void conv_bias_fetch(const ConvArgs& a, float4 (&bv)[TCO][4], int cobase, int wco, int h)
{
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bv[i][g] = a.bias ? *reinterpret_cast<const float4*>(a.bias + cobase + (wco * TCO + i) * 32 + 8 * g + 4 * h)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int TCO, int TPOS>
__device__ __forceinline__ void conv_acc_set(f32x16 (&acc)[TCO][TPOS], const float4 (&bv)[TCO][4])
{
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < TPOS; ++j) {
                acc[i][j][4 * g] = bv[i][g].x; acc[i][j][4 * g + 1] = bv[i][g].y;
                acc[i][j][4 * g + 2] = bv[i][g].z; acc[i][j][4 * g + 3] = bv[i][g].w;
            }
}

// two instantiations: the residual variant keeps its raw residual pieces in registers; the plain one carries none of it
template <bool EXT, typename TO, int BN, int BM, int WPOS, int TCO, int TPOS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TCO][TPOS], unsigned char* smem,
                                              int b, int t, int cot, int cobase, int l0, int tid, int l31, int h,
                                              int wco, int wpos)
{
    if (a.residual)
        conv_epilogue_impl<EXT, true, TO, BN, BM, WPOS, TCO, TPOS>(a, acc, smem, b, t, cot, cobase, l0, tid, l31, h, wco, wpos);
    else
        conv_epilogue_impl<EXT, false, TO, BN, BM, WPOS, TCO, TPOS>(a, acc, smem, b, t, cot, cobase, l0, tid, l31, h, wco, wpos);
}

// LDS image: plain 64-byte rows filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 16 rows per wave
// instruction, destination lane-linear), XOR-swizzled through the SOURCE address: the 16-byte piece p of
// row r is stored in slot p ^ ((r >> 2) & 3), which makes every ds_read_b128 fragment read conflict-free
// (a 16-lane read group then touches 16 distinct 16-byte slots of the 256-byte bank row).
__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same load marked non-temporal (sc0 nt): for the activation tiles of a conv with ONE co tile (C = 128), which no other
// block reads again.  Round 3, same box: level-1 conv 241.6 -> 231.7 us; with two or four co tiles sharing a tile through
// the XCD's L2 the hint costs 2-6 %, so the kernel takes it only when a.ncot == 1 (profiles/r03/v4_ab_nt_x_loads.txt).
__device__ __forceinline__ void glds16_nt(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 3);
}

template <bool EXT, typename T, typename TO, int TAPS, int WCO, int WPOS, int TCO, int TPOS, int DIL = 1>
__global__ __launch_bounds__(256, 2) void conv_mfma_simple_kernel(const ConvArgs a)
{
    constexpr int BN = WCO * TCO * 32;           // output channels per block (MFMA rows)
    constexpr int BM = WPOS * TPOS * 32;         // positions per block (MFMA cols)
    constexpr int XROWS = BM + (TAPS - 1) * DIL; // DIL: tap spacing (dilated causal convs of the SoundStream decoder)
    constexpr int NXI = (XROWS + 15) / 16;       // 1-KiB LDS-DMA instructions for the X tile
    constexpr int NWI = TAPS * BN / 16;          // ... for the W tile
    static_assert(WCO * WPOS == 4, "4 waves per block");
    static_assert(BN % 32 == 0, "co tile");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;
    unsigned char* ws = smem + NXI * 1024;

    const unsigned lid = xcd_remap(blockIdx.x, a.nblk);
    const int cot = (int)(lid % (unsigned)a.ncot);
    const unsigned r1 = lid / (unsigned)a.ncot;
    const int t = (int)(r1 % (unsigned)a.tps);
    const int b = (int)(r1 / (unsigned)a.tps);
    const int cobase = cot * BN;
    const int l0 = t * BM;
    const long long mrow0 = (long long)b * a.Lp + a.P + l0 - (EXT ? a.shift : TAPS / 2);   // EXT: causal / dilated taps

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPOS, wpos = wave % WPOS;

    // ---- LDS-DMA source addressing: this lane fills slot (lane & 3) of row 16*j + (lane >> 2) ----
    const int srow = lane >> 2, sslot = lane & 3;
    const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w);
    const long long xplane = a.M * 64;
    const long long wchunk = (long long)TAPS * a.CoutPad * 64;

    const bool xnt = !EXT && __builtin_amdgcn_readfirstlane(a.ncot) == 1;    // one co tile: the X tile has no other reader
                                            // (glds16_nt; not the dilated convs, whose neighbours share wide halos through the L2)
    auto stage = [&](int kc) {
        const unsigned char* xb = xg + kc * xplane;
        const unsigned char* wb = wg + kc * wchunk;
#pragma unroll
        for (int j0 = 0; j0 < NXI; j0 += 4) {
            const int j = j0 + wave;
            if (j < NXI) {
                const int row = j * 16 + srow;
                long long m = mrow0 + row;
                m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
                if (xnt) glds16_nt(xb + m * 64 + ((sslot ^ ((row >> 2) & 3)) << 4), xs + j * 1024);
                else glds16(xb + m * 64 + ((sslot ^ ((row >> 2) & 3)) << 4), xs + j * 1024);
            }
        }
#pragma unroll
        for (int j0 = 0; j0 < NWI; j0 += 4) {
            const int j = j0 + wave;
            if (j < NWI) {
                const int row = j * 16 + srow;              // = tap * BN + co_local
                const int tap = row / BN, col = row % BN;
                glds16(wb + (long long)(tap * a.CoutPad + cobase + col) * 64 + ((sslot ^ ((row >> 2) & 3)) << 4),
                       ws + j * 1024);
            }
        }
    };

    f32x16 acc[TCO][TPOS];
    {
        float4 bv[TCO][4];
        conv_bias_fetch<TCO>(a, bv, cobase, wco, h);
        conv_acc_set<TCO, TPOS>(acc, bv);
    }

    // fragment rows: weights at (tap*BN + wco*TCO*32 + i*32 + l31), activations at (wpos*TPOS*32 + j*32 + l31 + tap)
    const unsigned char* wfrag = ws + (wco * TCO * 32 + l31) * 64;
    const unsigned char* xfrag = xs + (wpos * TPOS * 32 + l31) * 64;
    const int wsw = (l31 >> 2) & 3;              // tile offsets are multiples of 16 rows: swizzle depends on l31 only

    for (int kc = 0; kc < a.nchunks; ++kc) {
        stage(kc);
        __syncthreads();                         // waits for this wave's LDS-DMA (vmcnt) + block barrier
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int xsw = ((l31 + tap * DIL) >> 2) & 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int p = 2 * s + h;
                Frag af[TCO], bf[TPOS];
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    af[i] = *reinterpret_cast<const Frag*>(wfrag + (tap * BN + i * 32) * 64 + ((p ^ wsw) << 4));
#pragma unroll
                for (int j = 0; j < TPOS; ++j)
                    bf[j] = *reinterpret_cast<const Frag*>(xfrag + (j * 32 + tap * DIL) * 64 + ((p ^ xsw) << 4));
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int j = 0; j < TPOS; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
            }
        }
        __syncthreads();                         // every wave is done reading before the next stage lands
    }

    conv_epilogue<EXT, TO, BN, BM, WPOS, TCO, TPOS>(a, acc, smem, b, t, cot, cobase, l0, tid, l31, h, wco, wpos);
}

// ------------------------------------------------------------------------------------------------
// Pipelined variant (co tile 128): the K loop runs in PHASES of one (channel chunk, tap) each.  The
// unit of prefetch is one tap's weight tile W(kc, tap) (128 rows x 64 B = 8 KiB = 2 LDS-DMA instructions
// per wave) held in a ring of K+1 slots; the activation tile X(kc) (BM + taps - 1 rows, shared by all
// taps of the chunk) is double-buffered and issued just ahead of W(kc, 0).  Every wave issues the same
// number of LDS-DMA instructions per unit, so "unit u has landed" is one counted s_waitcnt vmcnt(N)
// (N = this wave's instructions issued AFTER unit u, a compile-time constant per tap) followed by one
// raw s_barrier per phase; K-2 units stay in flight across barriers (never vmcnt(0) in steady state).
// Fragments are register double-buffered ACROSS the barrier so no ds_read latency sits in front of an
// MFMA group:
//   phase g:  ds_read frags(g, k-step 1) | MFMAs(g, k-step 0) | wait unit g+1 -> s_barrier -> issue unit
//             g+K (into the slot phase g-1 released) | ds_read frags(g+1, k-step 0) | MFMAs(g, k-step 1)
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The same wait as an intrinsic the compiler's own waitcnt bookkeeping can see (gfx9 encoding: vmcnt in bits 3:0 and
// 15:14, expcnt / lgkmcnt left at "no wait").  Used for the prologue wait only: it tells the compiler that the bias
// loads issued before the first LDS-DMA unit have landed, so it does not put a vmcnt(0) - which would also drain the
// units in flight - in front of the first MFMA that reads the bias-initialised accumulators.
template <int N> __device__ __forceinline__ void wait_vmcnt_visible() {
    __builtin_amdgcn_s_waitcnt((N & 0xf) | ((N >> 4) << 14) | (0x7 << 4) | (0xf << 8));
    asm volatile("" ::: "memory");
}

// Everything of one output tile up to the end of its K loop (tile decode, prologue, pipelined phases, final block barrier);
// shared by conv_mfma_kernel (one tile per block) and conv_mfma_gn_kernel (persistent blocks, GroupNorm fused into the
// epilogue).  `lid` = logical tile id, co tile fastest, then position tile, then sample.
struct ConvTileId { int b, t, cot, cobase, l0; };

// KEEP = true: the accumulators are NOT reset to the bias - the K loop adds onto what they hold (the 1x1 skip conv of a
// ResidualBlock folded into the fused conv + GroupNorm kernel as extra single-tap phases, conv_mfma_gn.inc).
template <bool EXT, typename T, int TAPS, int WCO, int WPOS, int TCO, int TPOS, int DIL, bool KEEP = false>
__device__ __forceinline__ ConvTileId conv_tile_mainloop(const ConvArgs& a, const unsigned lid, unsigned char* smem,
                                                         f32x16 (&acc)[TCO][TPOS], const int tid)
{
    constexpr int BN = WCO * TCO * 32, BM = WPOS * TPOS * 32;
    constexpr int NW = WCO * WPOS;               // waves per block: 4 (two blocks per CU) or 8 (one block per CU)
    static_assert((NW == 4 || NW == 8) && BN % (16 * NW) == 0, "pipelined variant: 4 or 8 waves");
#ifndef ADK_CONV_K      // prefetch depth in units; -DADK_CONV_K=2 / 3 builds measured within 1 % of 4 on every level
#define ADK_CONV_K 4
#endif
    // unit g+K is issued during phase g.  K <= TAPS whenever X is double-buffered: X(kc+1) then goes out during chunk kc
    // (never X(kc+2) into the buffer chunk kc's last tap is still reading fragments from)
    constexpr int K = TAPS >= 3 ? (ADK_CONV_K < TAPS ? ADK_CONV_K : TAPS) : 2;
    constexpr int RW = K + 1, RX = TAPS == 1 ? K + 1 : 2;
    static_assert(RX > 2 || K <= TAPS, "prefetch depth must not reach past the X double buffer");
    constexpr int XROWS = BM + (TAPS - 1) * DIL;
    constexpr int XI = (XROWS + 16 * NW - 1) / (16 * NW);    // X LDS-DMA instructions per wave (uniform; tile padded)
    constexpr int WI = BN / (16 * NW);                        // W LDS-DMA instructions per wave per unit
    constexpr int XBYTES = XI * NW * 1024, WBYTES = BN * 64;
    typedef typename Mma<T>::Frag Frag;

#ifdef ADK_ABL_TS_GN
    ADK_TS(3 + 0 * 0)   /* overwritten below by the hardware id; the fused kernel stamps slot 0 before its ticket */
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_adk_ts[16384 * 8 + blockIdx.x] = wall_clock64();
#else
    ADK_TS(0)
#endif
#ifdef ADK_ABL_TS
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_adk_ts[blockIdx.x * 8 + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    unsigned char* xs = smem;                    // [RX][XBYTES]
    unsigned char* ws = smem + RX * XBYTES;      // [RW][WBYTES]

    const int cot = (int)(lid % (unsigned)a.ncot);
    const unsigned r1 = lid / (unsigned)a.ncot;
    const int t = (int)(r1 % (unsigned)a.tps);
    const int b = (int)(r1 / (unsigned)a.tps);
    const int cobase = cot * BN;
    const int l0 = t * BM;
    const long long mrow0 = (long long)b * a.Lp + a.P + l0 - (EXT ? a.shift : TAPS / 2);   // EXT: causal / dilated taps

    const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPOS, wpos = wave % WPOS;
    const int srow = lane >> 2, sslot = lane & 3;

    const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w);
    const long long xplane = a.M * 64;
    const long long wchunk = (long long)TAPS * a.CoutPad * 64;
    const int total = a.nchunks * TAPS;

    float4 bv[TCO][4];
    if constexpr (!KEEP) conv_bias_fetch<TCO>(a, bv, cobase, wco, h);           // bias loads go out first (see conv_bias_fetch)

    // per-lane source offsets (bytes) inside a plane / a weight chunk, fixed for the whole kernel
    unsigned xsrc[XI], wsrc[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = (j * NW + wave) * 16 + srow;
        long long m = mrow0 + row;
        m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
        xsrc[j] = (unsigned)(m * 64 + ((sslot ^ ((row >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int row = (j * NW + wave) * 16 + srow;         // co_local
        wsrc[j] = (unsigned)((cobase + row) * 64 + ((sslot ^ ((row >> 2) & 3)) << 4));
    }

    const bool xnt = !EXT && __builtin_amdgcn_readfirstlane(a.ncot) == 1;    // glds16_nt
    auto issue_unit = [&](int u) {               // u = kc * TAPS + tap, wave-uniform
        const int kc = u / TAPS, tap = u - kc * TAPS;
        if (tap == 0) {
            const unsigned char* xb = xg + kc * xplane;
            unsigned char* dst = xs + (kc % RX) * XBYTES + wave * 1024;
#pragma unroll
            for (int j = 0; j < XI; ++j) {
                if (xnt) glds16_nt(xb + xsrc[j], dst + j * (NW * 1024));
                else glds16(xb + xsrc[j], dst + j * (NW * 1024));
            }
        }
        const unsigned char* wb = wg + kc * wchunk + (long long)tap * a.CoutPad * 64;
        unsigned char* dst = ws + (u % RW) * WBYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < WI; ++j) glds16(wb + wsrc[j], dst + j * (NW * 1024));
    };

    const int wfrag = (wco * TCO * 32 + l31) * 64;
    const int xfrag = (wpos * TPOS * 32 + l31) * 64;
    const int wsw = (l31 >> 2) & 3;

#define ADK_LOAD_FRAGS(AF_, BF_, WT_, XT_, XSW_, S_)                                                        \
    {                                                                                                        \
        const int p_ = 2 * (S_) + h;                                                                         \
        _Pragma("unroll") for (int i = 0; i < TCO; ++i)                                                      \
            AF_[i] = *reinterpret_cast<const Frag*>((WT_) + i * 2048 + ((p_ ^ wsw) << 4));                   \
        _Pragma("unroll") for (int j = 0; j < TPOS; ++j)                                                     \
            BF_[j] = *reinterpret_cast<const Frag*>((XT_) + j * 2048 + ((p_ ^ (XSW_)) << 4));                \
    }
#define ADK_MMA_ALL(AF_, BF_)                                                                                \
    _Pragma("unroll") for (int i = 0; i < TCO; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < TPOS; ++j) Mma<T>::run(AF_[i], BF_[j], acc[i][j]);

    Frag a0[TCO], b0[TPOS], a1[TCO], b1[TPOS];

    // prologue: units 0..K-1 in flight, wait for unit 0, first fragments
#pragma unroll
    for (int u = 0; u < K; ++u)
        if (u < total) issue_unit(u);
    {
        constexpr int n0 = []() { int n = 0; for (int u = 1; u < K; ++u) n += WI + (u % TAPS == 0 ? XI : 0); return n; }();
        if (total >= K) wait_vmcnt_visible<n0>(); else wait_vmcnt_visible<0>();
        if constexpr (!KEEP) conv_acc_set<TCO, TPOS>(acc, bv);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ADK_LOAD_FRAGS(a0, b0, ws + wfrag, xs + xfrag, wsw, 0)
    }

// n_mid / n_last: LDS-DMA instructions this wave has issued after unit g+1 when phase g waits for it
#define ADK_PHASE(TAP_)                                                                                      \
    if constexpr (TAP_ < TAPS) {                                                                             \
        constexpr int n_mid = []() { int n = 0; for (int u = 2; u < K; ++u) n += WI + ((TAP_ + u) % TAPS == 0 ? XI : 0); return n; }(); \
        constexpr int n_last = []() { int n = 0; for (int u = 2; u < K; ++u) n += (TAP_ + u < TAPS) ? WI : 0; return n; }();          \
        const int g_ = kc * TAPS + TAP_;                                                                     \
        {                                                                                                    \
            const unsigned char* wt_ = ws + (g_ % RW) * WBYTES + wfrag;                                      \
            const unsigned char* xt_ = xs + (kc % RX) * XBYTES + xfrag + TAP_ * DIL * 64;                    \
            ADK_LOAD_FRAGS(a1, b1, wt_, xt_, ((l31 + TAP_ * DIL) >> 2) & 3, 1)                               \
        }                                                                                                    \
        ADK_MMA_ALL(a0, b0)                                                                                  \
        if (g_ + 1 < total) {                                                                                \
            if (last) wait_vmcnt<n_last>(); else wait_vmcnt<n_mid>();                                        \
            __builtin_amdgcn_s_barrier();                                                                    \
            asm volatile("" ::: "memory");                                                                   \
            if (g_ + K < total) issue_unit(g_ + K);                                                          \
            constexpr int nt_ = (TAP_ + 1) % TAPS;                                                           \
            const int nkc_ = kc + (TAP_ + 1 == TAPS ? 1 : 0);                                                \
            const unsigned char* wt_ = ws + ((g_ + 1) % RW) * WBYTES + wfrag;                                \
            const unsigned char* xt_ = xs + (nkc_ % RX) * XBYTES + xfrag + nt_ * DIL * 64;                   \
            ADK_LOAD_FRAGS(a0, b0, wt_, xt_, ((l31 + nt_ * DIL) >> 2) & 3, 0)                                \
        }                                                                                                    \
        ADK_MMA_ALL(a1, b1)                                                                                  \
    }

    for (int kc = 0; kc < a.nchunks; ++kc) {
        const bool last = kc == a.nchunks - 1;
        ADK_PHASE(0) ADK_PHASE(1) ADK_PHASE(2) ADK_PHASE(3) ADK_PHASE(4) ADK_PHASE(5) ADK_PHASE(6)
    }
#undef ADK_PHASE
#undef ADK_LOAD_FRAGS
#undef ADK_MMA_ALL
    __syncthreads();      // all fragment reads done before the epilogue reuses LDS for its reduction
    ADK_TS(1)
    ConvTileId id;
    id.b = b; id.t = t; id.cot = cot; id.cobase = cobase; id.l0 = l0;
    return id;
}

template <bool EXT, typename T, typename TO, int TAPS, int WCO, int WPOS, int TCO, int TPOS, int DIL = 1>
__global__ __launch_bounds__(WCO * WPOS * 64, (WCO * WPOS == 8 ? 1 : 2)) void conv_mfma_kernel(const ConvArgs a)
{
    constexpr int BN = WCO * TCO * 32, BM = WPOS * TPOS * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x16 acc[TCO][TPOS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const ConvTileId id = conv_tile_mainloop<EXT, T, TAPS, WCO, WPOS, TCO, TPOS, DIL>(a, xcd_remap(blockIdx.x, a.nblk), smem, acc, tid);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPOS, wpos = wave % WPOS;
    conv_epilogue<EXT, TO, BN, BM, WPOS, TCO, TPOS>(a, acc, smem, id.b, id.t, id.cot, id.cobase, id.l0, tid, l31, h, wco, wpos);
#ifdef ADK_ABL_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ADK_TS(2)
}

// ------------------------------------------------------------------------------------------------
template <bool EXT, typename T, typename TO, int TAPS, int WCO, int WPOS, int TCO, int TPOS, int DIL = 1>
static int launch_variant(const adk_op& op, hipStream_t st)
{
    constexpr int BN = WCO * TCO * 32, BM = WPOS * TPOS * 32;
    ConvArgs a;
    conv_fill_args(op, BN, BM, a);
    if constexpr (BN >= 128) {
        constexpr int K = TAPS >= 3 ? (ADK_CONV_K < TAPS ? ADK_CONV_K : TAPS) : 2;
        constexpr int NW = WCO * WPOS;
        constexpr int XI = (BM + (TAPS - 1) * DIL + 16 * NW - 1) / (16 * NW);
        const size_t lds = (size_t)(TAPS == 1 ? K + 1 : 2) * XI * NW * 1024 + (size_t)(K + 1) * BN * 64;
        static std::atomic<unsigned long long> attr_set{0};   // > 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel and device
        if (adk_lds_optin(attr_set, reinterpret_cast<const void*>(&conv_mfma_kernel<EXT, T, TO, TAPS, WCO, WPOS, TCO, TPOS, DIL>), lds))
            return ADK_ELAUNCH;
        hipLaunchKernelGGL((conv_mfma_kernel<EXT, T, TO, TAPS, WCO, WPOS, TCO, TPOS, DIL>), dim3(a.nblk), dim3(NW * 64), lds, st, a);
    } else {
        const size_t lds = (size_t)((BM + (TAPS - 1) * DIL + 15) / 16) * 1024 + (size_t)TAPS * BN * 64;
        hipLaunchKernelGGL((conv_mfma_simple_kernel<EXT, T, TO, TAPS, WCO, WPOS, TCO, TPOS, DIL>), dim3(a.nblk), dim3(256), lds, st, a);
    }
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

template <bool EXT, typename T, typename TO, int TAPS, int DIL = 1>
static int launch_tiles(const adk_op& op, hipStream_t st)
{
    const ConvTile t = conv_tile(op);
    if (t.bn == 32) return launch_variant<EXT, T, TO, TAPS, 1, 4, 1, 1, DIL>(op, st);
    if (t.bm == 256) return launch_variant<EXT, T, TO, TAPS, 2, 2, 2, 4, DIL>(op, st);
    if (t.bm == 64) return launch_variant<EXT, T, TO, TAPS, 2, 2, 2, 1, DIL>(op, st);
    return launch_variant<EXT, T, TO, TAPS, 2, 2, 2, 2, DIL>(op, st);
}

