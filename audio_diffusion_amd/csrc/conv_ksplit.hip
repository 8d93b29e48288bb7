// Implicit-GEMM Conv1d for SHORT levels (L <= 256): small (co, pos) tiles, K range split over the 4 waves.
//
// At L <= 256 the pipelined kernel of conv_mfma.hip has too few 128-row tiles to fill 256 CUs and runs a
// serial K loop of up to 400 barrier-separated phases per block (21-28 us for < 1 % of the FLOPs).  Here a
// block owns a (TCO*32 co) x (TPOS*32 pos) tile of ONE sample and its four waves each take every fourth
// channel chunk: no barrier and no LDS in the K loop.  Both MFMA operands go straight from global memory
// to VGPRs in fragment layout (one 16-byte piece per lane: 32 rows x 64 B = 2 KiB contiguous per
// instruction; the weights are read once per block, the activation rows are L1/L2 hits re-read per tap),
// register double-buffered one chunk (or four 1x1 units) ahead.  The four partial accumulators meet in LDS
// and are summed in FIXED wave order (bit-reproducible); wave w finishes channel group 8w..8w+3 (+4h) of
// every 32-channel sub-tile: + bias, + residual, GroupNorm partials, halo rows, store.
// Same packed weights, same CPL activations, same epilogue semantics as conv_mfma.hip.
//
// GN = true (ADK_FLAG_FUSE_GN): GroupNorm(1,C) + GELU (+ residual) ride in the epilogue, as in conv_mfma_gn.inc: the
// blocks of a sample exchange their (sum, sumsq) through epoch-tagged granules inside the launch.  A grid that fits the
// GPU (B = 16: 256 blocks) keeps its co-tile-major block order - every block is resident, so any order is safe; a
// larger grid hands its tiles out by ticket in sample-major start order (conv_mfma_gn.inc has the argument).
#include <stdlib.h>
#include "conv_common.h"
#include "conv_gn_exchange.h"

template <typename T, typename TO, int TAPS, int TCO, int TPOS, int DEPTH, bool GN>
__global__ __launch_bounds__(256, (TCO * TPOS >= 4 ? 1 : 2)) void conv_ksplit_kernel(const ConvArgs a)
{
    constexpr int BN = TCO * 32, BM = TPOS * 32, S = TCO * TPOS;
    constexpr int CBO = ElemTraits<TO>::CB;
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // co-tile-major logical order: after the XCD remap an XCD owns a contiguous range of co tiles, so each of
    // the 8 L2s fetches only its share of the weights (which come from HBM on every denoiser call)
    unsigned lid = xcd_remap(blockIdx.x, a.nblk);
    const unsigned nbt = (unsigned)(a.B * a.tps);
    int cot = (int)(lid / nbt);
    unsigned r1 = lid % nbt;
    int t = (int)(r1 % (unsigned)a.tps);
    int b = (int)(r1 / (unsigned)a.tps);
    unsigned epoch = 0;
    if constexpr (GN) {
        const unsigned ebase = a.gn_epoch[0];
        epoch = ebase * 4096u + a.gn_slot + 1u;
        if (a.nblk > a.gn_cap) {
            // Not all blocks resident at once: tiles go by ticket, sample-major, ONE LIST PER XCD exactly as in the pipelined
            // kernel (conv_mfma_gn.inc has the argument): tile q of the sample-major order belongs to list q % 8, a block takes
            // the next tile of the list of the XCD it runs on, the grid is 8 x the longest list.  XCD k then only ever holds
            // list-k tiles in sample-major start order, at most ceil(npb / 8) - 1 of them wait for a sample's missing tile,
            // and that tile is the next workgroup in XCD k's own queue: no spare blocks, nothing statistical.
            unsigned* tick = reinterpret_cast<unsigned*>(smem);
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned xk = xcc & 7u;
            if (threadIdx.x == 0) {
                const unsigned old = __hip_atomic_fetch_add((gu32*)(a.gn_ticket) + xk * GN_LIST_STRIDE, 1u, ADK_RLX_AGENT);
                *tick = old - (ebase - 1u) * (a.gn_grid >> 3);     // this list's counter grows by grid / 8 per forward
            }
            __syncthreads();
            const unsigned n = __builtin_amdgcn_readfirstlane(*tick);
            __syncthreads();
            lid = xk + 8u * n;
            if (lid >= a.nblk) {                             // beyond the end of its list; beyond the grid's share: counter out of step
                if (n >= (a.gn_grid >> 3) && threadIdx.x == 0) __hip_atomic_store((gu32*)(a.gn_err), 0x71c1u, ADK_RLX_AGENT);
                return;
            }
            b = (int)(lid / (unsigned)a.npb);
            r1 = lid % (unsigned)a.npb;
            cot = (int)(r1 / (unsigned)a.tps);
            t = (int)(r1 % (unsigned)a.tps);
        }
    }
    const int cobase = cot * BN;
    const int l0 = t * BM;
    const int mrow0 = b * a.Lp + a.P + l0 - a.shift;           // >= 0: P >= shift (validated on the host)
    const int dil = a.dil;
    const int mlast = (int)a.M - 1;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nck = a.nchunks > wave ? (a.nchunks - wave + 3) >> 2 : 0;      // this wave's chunks: wave, wave+4, ...

    const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x) + h * 16;
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w) + (long long)(cobase + l31) * 64 + h * 16;
    const long long xplane = a.M * 64;
    const long long wtap = (long long)a.CoutPad * 64;

    // The epilogue's operands are fetched FIRST, before the K loop: wave w finishes D[co = cobase + 32 i + 8 w + 4 h + e]
    // [pos = l0 + 32 j + l31], so its bias / residual pieces are known now, and a load issued after the first store
    // would wait for that store's acknowledgement (loads and stores share vmcnt on gfx9; these kernels are
    // latency-bound).  Lanes outside the tensor read a valid address of their own sample and discard it.
    typedef typename RawVec4<TO>::type Raw4;
    float4 bpre[TCO];
    Raw4 rpre[TCO][TPOS];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
        bpre[i] = a.bias ? *reinterpret_cast<const float4*>(a.bias + cobase + i * 32 + 8 * wave + 4 * (lane >> 5))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.residual) {
        constexpr int CBR = ElemTraits<TO>::CB;
        const TO* rgp = reinterpret_cast<const TO*>(a.residual);
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPOS; ++j) {
                int co = cobase + i * 32 + 8 * wave + 4 * (lane >> 5);
                co = co < a.Cout_alloc ? co : 4 * (lane >> 5);
                const int l = l0 + j * 32 + (lane & 31);
                const long long m = (long long)b * a.Lp + a.P + (l < a.L ? l : 0);
                rpre[i][j] = *reinterpret_cast<const Raw4*>(rgp + ((((long long)(co / CBR) * a.M + m) * CBR) + (co % CBR)));
            }
    }

    // ADK_FLAG_PRE_NORM (plain epilogue only): y = rstd_b * acc + (beta[co] - mean_b * rstd_b * gamma[co]).  pn_rs = 1 and
    // bpre = bias otherwise: v * 1 + bias is the plain result, bit for bit.
    float pn_rs = 1.f;
    if constexpr (!GN) {
        if (a.pn_partials) {
            float mean, rstd;
            gn_finalize_wave(a.pn_partials + (long long)b * a.pn_n * 2, a.pn_n, a.gn_count, a.eps, mean, rstd);
            pn_rs = rstd;
            const float mrs = mean * rstd;
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int c = cobase + i * 32 + 8 * wave + 4 * (lane >> 5);
                const float4 u = *reinterpret_cast<const float4*>(a.gamma + c), v = *reinterpret_cast<const float4*>(a.beta + c);
                bpre[i] = make_float4(v.x - mrs * u.x, v.y - mrs * u.y, v.z - mrs * u.z, v.w - mrs * u.w);
            }
        }
    }

    Frag fa[DEPTH][TAPS][TCO][2], fb[DEPTH][TAPS][TPOS][2];
    f32x16 acc[TCO][TPOS];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPOS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define ADK_KS_LOAD(D_, TAP_, C_)                                                                           \
    {                                                                                                        \
        const int kc_ = wave + 4 * (C_);                                                                     \
        const unsigned char* wp_ = wg + ((long long)kc_ * TAPS + (TAP_)) * wtap;                             \
        const unsigned char* xp_ = xg + kc_ * xplane;                                                        \
        _Pragma("unroll") for (int i = 0; i < TCO; ++i)                                                      \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                    \
                fa[D_][TAP_][i][s] = *reinterpret_cast<const Frag*>(wp_ + i * 2048 + s * 32);                \
        _Pragma("unroll") for (int j = 0; j < TPOS; ++j) {                                                   \
            int row_ = mrow0 + j * 32 + l31 + (TAP_) * dil;                                                  \
            row_ = row_ > mlast ? mlast : row_;          /* rows past the tensor only feed masked columns */ \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                    \
                fb[D_][TAP_][j][s] = *reinterpret_cast<const Frag*>(xp_ + (long long)row_ * 64 + s * 32);    \
        }                                                                                                    \
    }

#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nck) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) ADK_KS_LOAD(d, tap, d)
        }

    for (int c0 = 0; c0 < nck; c0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int c = c0 + d;
            if (c < nck) {
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int i = 0; i < TCO; ++i)
#pragma unroll
                            for (int j = 0; j < TPOS; ++j) Mma<T>::run(fa[d][tap][i][s], fb[d][tap][j][s], acc[i][j]);
                    if (c + DEPTH < nck) ADK_KS_LOAD(d, tap, c + DEPTH)     // refill the registers just consumed
                }
            }
        }
    }
#undef ADK_KS_LOAD

    // ---- cross-wave reduction: red[src wave][sub-tile][g][lane] (float4), summed in wave order ----
    float4* red = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPOS; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[((wave * S + i * TPOS + j) * 4 + g) * 64 + lane] =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    __syncthreads();

    // wave w now owns D[co = cobase + 32 i + 8 w + 4 h + e][pos = l0 + 32 j + l31]
    TO* yg = reinterpret_cast<TO*>(a.y);
    const TO* rg = reinterpret_cast<const TO*>(a.residual);
    if constexpr (GN) {
        // raw = conv + bias stays in registers (S x 4 values per lane) across the exchange of the sample statistics
        float vv[TCO][TPOS][4];
        float4 gm[TCO], bt[TCO];
        float r1s = 0.f, r2s = 0.f;
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int c = cobase + i * 32 + 8 * wave + 4 * h;
            gm[i] = *reinterpret_cast<const float4*>(a.gamma + c);
            bt[i] = *reinterpret_cast<const float4*>(a.beta + c);
        }
#pragma unroll
        for (int j = 0; j < TPOS; ++j) {
            const bool lvalid = l0 + j * 32 + l31 < a.L;
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int co = cobase + i * 32 + 8 * wave + 4 * h;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sw = 0; sw < 4; ++sw) {
                    const float4 p = red[((sw * S + i * TPOS + j) * 4 + wave) * 64 + lane];
                    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
                }
                v[0] += bpre[i].x; v[1] += bpre[i].y; v[2] += bpre[i].z; v[3] += bpre[i].w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vv[i][j][e] = v[e];
                    if (lvalid && co < a.Cout_alloc) { r1s += v[e]; r2s += v[e] * v[e]; }
                }
            }
        }
        float* redf = reinterpret_cast<float*>(smem);
        block_sum2(r1s, r2s, redf);                         // starts with a barrier: the reads of `red` are done
        float mean, rstd;
        gn_exchange(a, b, t * a.ncot + cot, epoch, r1s, r2s, redf, tid, mean, rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < TPOS; ++j) {
            const int l = l0 + j * 32 + l31;
            const bool lvalid = l < a.L;
            const long long m = (long long)b * a.Lp + a.P + l;
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int co = cobase + i * 32 + 8 * wave + 4 * h;
                if (lvalid && co < a.Cout_alloc) {
                    const float g4[4] = {gm[i].x, gm[i].y, gm[i].z, gm[i].w}, b4[4] = {bt[i].x, bt[i].y, bt[i].z, bt[i].w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sc = g4[e] * rstd, sh = __builtin_fmaf(-mean, sc, b4[e]);
                        v[e] = __builtin_fmaf(vv[i][j][e], sc, sh);
                        if (a.gelu) v[e] = gelu_for<T>(v[e]);
                    }
                    if (rg) {
                        float rr[4];
                        raw_to_float4(rpre[i][j], rr, TO());
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rr[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
                    const long long off = (((long long)(co / CBO) * a.M + m) * CBO) + (co % CBO);
                    store4(yg + off, v);
                }
            }
        }
        if (t == 0 || t == a.tps - 1) {
            const int plane0 = cobase / CBO;
            const int cend = cobase + BN < a.Cout_alloc ? cobase + BN : a.Cout_alloc;
            const int nplanes = (cend + CBO - 1) / CBO - plane0;
            const int per = a.P * 4;
            const uint4 z = make_uint4(0, 0, 0, 0);
            unsigned char* yb = reinterpret_cast<unsigned char*>(a.y);
            for (int p = tid; p < nplanes * per; p += 256) {
                const int pl = plane0 + p / per, rem = p % per, row = rem >> 2, q = rem & 3;
                const long long mh = (long long)b * a.Lp + row;
                if (t == 0) *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh) * 64 + q * 16) = z;
                if (t == a.tps - 1)
                    *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh + a.P + a.L) * 64 + q * 16) = z;
            }
        }
        if (a.partials) {                                    // statistics of the output, for a following attention norm
            block_sum2(s1, s2, redf);
            if (tid == 0) {
                float* dst = a.partials + ((long long)(b * a.tps + t) * a.ncot + cot) * 2;
                dst[0] = s1; dst[1] = s2;
            }
        }
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    float gs1[TCO], gs2[TCO];
#pragma unroll
    for (int i = 0; i < TCO; ++i) { gs1[i] = 0.f; gs2[i] = 0.f; }
#pragma unroll
    for (int j = 0; j < TPOS; ++j) {
        const int l = l0 + j * 32 + l31;
        const bool lvalid = l < a.L;
        const long long m = (long long)b * a.Lp + a.P + l;
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = cobase + i * 32 + 8 * wave + 4 * h;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) {
                const float4 p = red[((sw * S + i * TPOS + j) * 4 + wave) * 64 + lane];
                v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
            }
            v[0] = v[0] * pn_rs + bpre[i].x; v[1] = v[1] * pn_rs + bpre[i].y;
            v[2] = v[2] * pn_rs + bpre[i].z; v[3] = v[3] * pn_rs + bpre[i].w;
            if (lvalid && co < a.Cout_alloc) {
                const long long off = (((long long)(co / CBO) * a.M + m) * CBO) + (co % CBO);
                if (rg) {
                    float rr[4];
                    raw_to_float4(rpre[i][j], rr, TO());
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[e];
                }
                if (a.elu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = elu1(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; gs1[i] += v[e]; gs2[i] += v[e] * v[e]; }
                store4(yg + off, v);
            }
        }
    }

    // halo rows of the output: zeroed by the first / last tile of each sample
    if (t == 0 || t == a.tps - 1) {
        const int plane0 = cobase / CBO;
        const int cend = cobase + BN < a.Cout_alloc ? cobase + BN : a.Cout_alloc;
        const int nplanes = (cend + CBO - 1) / CBO - plane0;
        const int per = a.P * 4;
        const uint4 z = make_uint4(0, 0, 0, 0);
        unsigned char* yb = reinterpret_cast<unsigned char*>(a.y);
        for (int p = tid; p < nplanes * per; p += 256) {
            const int pl = plane0 + p / per, rem = p % per, row = rem >> 2, q = rem & 3;
            const long long mh = (long long)b * a.Lp + row;
            if (t == 0) *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh) * 64 + q * 16) = z;
            if (t == a.tps - 1)
                *reinterpret_cast<uint4*>(yb + ((long long)pl * a.M + mh + a.P + a.L) * 64 + q * 16) = z;
        }
    }

    if (a.partials && a.gstats) {                       // [b][g][t * 4 + wave]: this wave's 8 channels of each group
        const int ngrp = (a.Cout_alloc + 31) >> 5;
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const float r1 = wave_sum(gs1[i]), r2 = wave_sum(gs2[i]);
            const int g = (cobase >> 5) + i;
            if (lane == 0 && g < ngrp) {
                float* dst = a.partials + ((((long long)b * ngrp + g) * a.tps + t) * 4 + wave) * 2;
                dst[0] = r1; dst[1] = r2;
            }
        }
    } else if (a.partials) {
        block_sum2(s1, s2, reinterpret_cast<float*>(smem));     // starts with a barrier: the reads of `red` are done
        if (tid == 0) {
            float* dst = a.partials + ((long long)(b * a.tps + t) * a.ncot + cot) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
}

template <typename T, typename TO, int TAPS, int TCO, int TPOS>
static int launch_ks(const adk_op& op, hipStream_t st)
{
    constexpr int DEPTH = TAPS == 1 ? 4 : ((TCO * TPOS == 1 && TAPS <= 5) ? 2 : 1);
    ConvArgs a;
    conv_fill_args(op, TCO * 32, TPOS * 32, a);
    const size_t lds = (size_t)TCO * TPOS * 16384;
    if constexpr (TAPS != 1 && sizeof(T) == sizeof(TO)) {
        if (op.flags & ADK_FLAG_FUSE_GN) {
            const void* fn = reinterpret_cast<const void*>(&conv_ksplit_kernel<T, TO, TAPS, TCO, TPOS, DEPTH, true>);
            static std::atomic<int> cap[64];                 // per device; atomic: the library is re-entrant
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ADK_ELAUNCH;
            int capd = cap[dev].load(std::memory_order_acquire);
            if (!capd) {
                int per_cu = 0, cus = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess) return ADK_ELAUNCH;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return ADK_ELAUNCH;
                const int lb = TCO * TPOS >= 4 ? 1 : 2;      // __launch_bounds__
                if (per_cu > lb) per_cu = lb;
                if (per_cu < 1 || cus < 1) return ADK_ELAUNCH;
                capd = per_cu * cus;
                cap[dev].store(capd, std::memory_order_release);
            }
            if (a.npb > capd) return ADK_EINVAL;
            a.gn_cap = (unsigned)capd;
            // ticketed (more blocks than resident slots): one list per XCD, grid = 8 x the longest list, a sample's share of a
            // list must fit the XCD's slots, and the device must deal workgroups evenly to 8 XCDs; else every block is resident
            unsigned grid = a.nblk;
            if (a.nblk > a.gn_cap) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
                if (!gn_device_ok(nullptr, !capturing)) return ADK_EINVAL;
                if (((unsigned)a.npb + 7u) / 8u > a.gn_cap / 8u) return ADK_EINVAL;
                a.gn_grid = grid = 8u * ((a.nblk + 7u) / 8u);
            }
            hipLaunchKernelGGL((conv_ksplit_kernel<T, TO, TAPS, TCO, TPOS, DEPTH, true>), dim3(grid), dim3(256), lds, st, a);
            return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
        }
    }
    hipLaunchKernelGGL((conv_ksplit_kernel<T, TO, TAPS, TCO, TPOS, DEPTH, false>), dim3(a.nblk), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

template <typename T, typename TO, int TAPS>
static int launch_ks_tiles(const adk_op& op, hipStream_t st)
{
    const ConvTile t = conv_tile(op);
    if (t.bn == 32 && t.bm == 32) return launch_ks<T, TO, TAPS, 1, 1>(op, st);
    if (t.bn == 32 && t.bm == 64) return launch_ks<T, TO, TAPS, 1, 2>(op, st);
    if (t.bn == 64 && t.bm == 64) return launch_ks<T, TO, TAPS, 2, 2>(op, st);
    return ADK_EINVAL;
}

template <typename T, typename TO>
static int launch_ks_taps(const adk_op& op, hipStream_t st)
{
    if (op.taps == 1) return launch_ks_tiles<T, TO, 1>(op, st);
    if (op.taps == 5) return launch_ks_tiles<T, TO, 5>(op, st);
    if (op.taps == 3) return launch_ks_tiles<T, TO, 3>(op, st);
    if (op.taps == 7) return launch_ks_tiles<T, TO, 7>(op, st);
    return ADK_EINVAL;
}

int launch_conv_ksplit(const adk_op& op, hipStream_t st)
{
    if (op.dtype == ADK_F32) return launch_ks_taps<float, float>(op, st);
    if (op.dtype == ADK_F16)
        return (op.flags & ADK_FLAG_OUT_F32) ? launch_ks_taps<f16_t, float>(op, st) : launch_ks_taps<f16_t, f16_t>(op, st);
    if (op.flags & ADK_FLAG_OUT_F32) return launch_ks_taps<bf16_t, float>(op, st);
    return launch_ks_taps<bf16_t, bf16_t>(op, st);
}
