// K loop of the 128co x 256pos pipelined tile on v_mfma_f32_16x16x32_{f16,bf16} (fused conv + GroupNorm kernel, 16-bit types).
//
// Same LDS-DMA pipeline, units, ring and counted vmcnt waits as conv_tile_mainloop (conv_mfma_kernels.h); what differs is
// the matrix instruction and therefore the fragment schedule and the accumulator layout.  Why: two co-resident blocks keep
// the matrix pipe busy, and the pipe then runs at the clock the chip holds under that load - and it holds a higher one on
// the 16x16x32 shape than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7; measured here
// with a FLOP-equal stand-in before this was written: -4 % on the whole call).
//
//   * One MFMA covers the unit's whole K (32 channels = one 64-byte row): per unit (chunk, tap) a wave reads NA = 4 weight
//     fragments (16 rows each) and NB = 8 activation fragments (16 positions each) - the same 12 ds_read_b128 as before -
//     and issues 32 MFMAs.  Lane (l15, q = lane >> 4) reads 16-byte slot q of row l15.
//   * Fragment schedule: the B fragments roll through a ring of 4 (B[y+3] is requested when step y starts, into the
//     register B[y-1] just vacated), the A fragments are double-buffered across units (unit g+1's are requested during the
//     first four steps of unit g): 48 fragment registers, as before.  So unit g+1 must be readable when phase g STARTS:
//     one wait + barrier per phase, at its top (the old loop had it between its two MFMA groups).
//   * The A double buffer's parity is compile-time: the chunk loop is unrolled by two (all tap counts are odd).
//   * Swizzle: slot p of row r is stored in slot p ^ (((r >> 2) & 1) << 1) - conflict-free for these reads at every tap
//     shift (brute-forced over the four 16-lane groups of ds_read_b128; the identity XOR of the 32x32 image is 2-way here).
//   * Accumulators: acc[i][j] (32co x 32pos, 16 registers) = four 16x16 sub-tiles G = 2 ci + pj, registers 4G..4G+3:
//     channel 32 i + 16 ci + 4 q + e, position 32 j + 16 pj + l15 (conv_epilogue_gn's M16 branch).
#pragma once
#include <type_traits>
#include "conv_mfma_kernels.h"

template <typename T> struct Mma16;
template <> struct Mma16<f16_t> {
    typedef f16x8 Frag;
    static __device__ __forceinline__ f32x4 run(const Frag& a, const Frag& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ f32x4 run(const Frag& a, const Frag& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <int G> __device__ __forceinline__ void mma16_sub(const typename Mma16<f16_t>::Frag& a, const typename Mma16<f16_t>::Frag& b, f32x16& c)
{
    f32x4 s = __builtin_shufflevector(c, c, 4 * G, 4 * G + 1, 4 * G + 2, 4 * G + 3);
    s = Mma16<f16_t>::run(a, b, s);
    c[4 * G] = s[0]; c[4 * G + 1] = s[1]; c[4 * G + 2] = s[2]; c[4 * G + 3] = s[3];
}
template <int G> __device__ __forceinline__ void mma16_sub(const typename Mma16<bf16_t>::Frag& a, const typename Mma16<bf16_t>::Frag& b, f32x16& c)
{
    f32x4 s = __builtin_shufflevector(c, c, 4 * G, 4 * G + 1, 4 * G + 2, 4 * G + 3);
    s = Mma16<bf16_t>::run(a, b, s);
    c[4 * G] = s[0]; c[4 * G + 1] = s[1]; c[4 * G + 2] = s[2]; c[4 * G + 3] = s[3];
}

// Fragment reads and their waits are inline asm: the compiler otherwise gathers the reads into groups right in front of an
// lgkmcnt(0) (three exposed LDS round trips per phase in its schedule).  asm volatile statements keep their order, LDS
// returns in order, so each wait is a COUNTED lgkmcnt(N) - N = reads issued after the fragment it is for - and names the
// registers it releases (a data dependency the consuming MFMAs cannot be moved across).
template <int OFF, typename Frag> __device__ __forceinline__ void lds_read128(Frag& r, unsigned addr) {
    static_assert(sizeof(Frag) == 16 && OFF >= 0 && OFF < 65536, "one ds_read_b128, immediate offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
}
// tools/isa_lint.py checks this pattern in the machine code that ships (no instruction may name a fragment register
// between its ds_read_b128 and the counted wait that covers it; MFMA wait states; no stray vector-memory instruction in
// the counted-vmcnt loop).  Its positive controls are two deliberately broken builds of this file, never shipped:
//   -DADK_LINT_CONTROL_WEAK_WAIT  every counted fragment wait one too weak (rule a must fire);
//   -DADK_LINT_CONTROL_ACC_TIE    the wrong turn described below, re-created (rule b must fire).
#ifdef ADK_LINT_CONTROL_WEAK_WAIT
#define ADK_LDS_WAIT_SLACK 1
#else
#define ADK_LDS_WAIT_SLACK 0
#endif
template <int N, typename Frag> __device__ __forceinline__ void lds_wait(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N + ADK_LDS_WAIT_SLACK) : "memory");
}
#ifdef ADK_LINT_CONTROL_ACC_TIE
template <int N, typename Frag> __device__ __forceinline__ void lds_wait_tie(Frag& f, f32x16& c0, f32x16& c1) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f), "+v"(c0), "+v"(c1) : "n"(N) : "memory");
}
#endif
// (NEVER pass an accumulator through such an asm to order it after the MFMAs that write it: a version of this loop tied
// each wait to the previous step's accumulator tuple as an in/out operand.  The compiler then takes the asm - not the
// MFMA still in flight - for the producer of those registers, its hazard recogniser stops inserting the wait states an
// MFMA needs before it reads, as SrcC, registers that overlap but do not coincide with an in-flight MFMA's destination,
// and whole samples came out different from run to run wherever the register assignment had such a pair: on the odd
// chunk's tail of the unrolled loop, under load only.  Found by tools/soak_fused.py, isolated with tools/bisect_plan.py
// and tools/rerun_fused.py.)
template <int N, typename Frag> __device__ __forceinline__ void lds_wait(Frag& f, Frag (&g)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f), "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_addr32(const unsigned char* p) {
    return (unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)p);
}

template <typename T, int TAPS, int WCO, int WPOS, int TCO, int TPOS, bool KEEP = false>
__device__ __forceinline__ ConvTileId conv_tile_mainloop16(const ConvArgs& a, const unsigned lid, unsigned char* smem,
                                                           f32x16 (&acc)[TCO][TPOS], const int tid)
{
    static_assert(sizeof(T) == 2 && TCO == 2 && TPOS == 4 && WCO * WPOS == 4, "the 128co x 256pos tile, 16-bit operands");
    static_assert(TAPS % 2 == 1, "odd tap counts (the A double buffer's parity is unrolled over two chunks)");
    constexpr int BN = WCO * TCO * 32, BM = WPOS * TPOS * 32;
    constexpr int NW = 4;
    constexpr int K = TAPS >= 3 ? (ADK_CONV_K < TAPS ? ADK_CONV_K : TAPS) : 2;
    constexpr int RW = K + 1, RX = TAPS == 1 ? K + 1 : 2;
    constexpr int XROWS = BM + TAPS - 1;
    constexpr int XI = (XROWS + 16 * NW - 1) / (16 * NW);
    constexpr int WI = BN / (16 * NW);
    constexpr int XBYTES = XI * NW * 1024, WBYTES = BN * 64;
    constexpr int NA = 2 * TCO, NB = 2 * TPOS;
    typedef typename Mma16<T>::Frag Frag;

#ifdef ADK_ABL_TS_GN
    ADK_TS(3 + 0 * 0)
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
    const long long mrow0 = (long long)b * a.Lp + a.P + l0 - TAPS / 2;

    const int lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPOS, wpos = wave % WPOS;
    const int srow = lane >> 2, sslot = lane & 3;

    const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w);
    const long long xplane = a.M * 64;
    const long long wchunk = (long long)TAPS * a.CoutPad * 64;
    const int total = a.nchunks * TAPS;

    // bias of this lane's channels: 32 i + 16 ci + 4 q .. + 3; requested before the first LDS-DMA unit (conv_bias_fetch)
    // The loads are UNCONDITIONAL (a conv without bias reads gamma - always there in this kernel, same extent - and
    // discards it): loads inside a branch leave the compiler unable to count them, and it then guards the accumulator
    // initialisation with a vmcnt(0) of its own, which also drains the prefetched units 1..K-1.
    float4 bv[TCO][2];
    if constexpr (!KEEP) {
        const float* bsrc = a.bias ? a.bias : a.gamma;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
                bv[i][ci] = *reinterpret_cast<const float4*>(bsrc + cobase + (wco * TCO + i) * 32 + 16 * ci + 4 * q);
    }

    unsigned xsrc[XI], wsrc[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = (j * NW + wave) * 16 + srow;
        long long m = mrow0 + row;
        m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
        xsrc[j] = (unsigned)(m * 64 + ((sslot ^ (((row >> 2) & 1) << 1)) << 4));
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int row = (j * NW + wave) * 16 + srow;
        wsrc[j] = (unsigned)((cobase + row) * 64 + ((sslot ^ (((row >> 2) & 1) << 1)) << 4));
    }

    // (the 32-bit per-lane offsets pass through an opaque asm at every use: the 64-bit source addresses are then formed
    // where they are used instead of living - and spilling - as seven loop-invariant register pairs)
    const bool xnt = __builtin_amdgcn_readfirstlane(a.ncot) == 1;
    auto issue_unit = [&](int u, auto opaque) {  // u = kc * TAPS + tap, wave-uniform
        const int kc = u / TAPS, tap = u - kc * TAPS;
        if (tap == 0) {
            const unsigned char* xb = xg + kc * xplane;
            unsigned char* dst = xs + (kc % RX) * XBYTES + wave * 1024;
#ifndef ADK_EXP_NO_NT_X
            if (xnt) {                            // one co tile: nobody else reads this tile, keep it out of the L2's way
#pragma unroll
                for (int j = 0; j < XI; ++j) {
                    if constexpr (decltype(opaque)::value) asm volatile("" : "+v"(xsrc[j]));
                    glds16_nt(xb + xsrc[j], dst + j * (NW * 1024));
                }
            } else
#endif
            {
#pragma unroll
                for (int j = 0; j < XI; ++j) {
                    if constexpr (decltype(opaque)::value) asm volatile("" : "+v"(xsrc[j]));
                    glds16(xb + xsrc[j], dst + j * (NW * 1024));
                }
            }
        }
        const unsigned char* wb = wg + kc * wchunk + (long long)tap * a.CoutPad * 64;
        unsigned char* dst = ws + (u % RW) * WBYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            if constexpr (decltype(opaque)::value) asm volatile("" : "+v"(wsrc[j]));
            glds16(wb + wsrc[j], dst + j * (NW * 1024));
        }
    };

    // this lane's fragment addresses: A fragment x at + x KiB, B fragment y at + y KiB
    const int wlane = (wco * TCO * 32 + l15) * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) << 4);
    auto xlane = [&](int tap) { return (wpos * TPOS * 32 + l15 + tap) * 64 + ((q ^ ((((l15 + tap) >> 2) & 1) << 1)) << 4); };

    Frag aq[2][NA], bq[4];

    // prologue: units 0..K-1 in flight, wait for unit 0, unit 0's A fragments and first three B fragments
#pragma unroll
    for (int u = 0; u < K; ++u)
        if (KEEP ? u < total : true) issue_unit(u, std::false_type());   // a conv's K loop has >= TAPS >= K units; unconditional,
                                                                         // so that the compiler can COUNT them (bias wait below)
    {
        constexpr int n0 = []() { int n = 0; for (int u = 1; u < K; ++u) n += WI + (u % TAPS == 0 ? XI : 0); return n; }();
        if (!KEEP || total >= K) wait_vmcnt_visible<n0>(); else wait_vmcnt_visible<0>();
        if constexpr (!KEEP) {
            const float keep = a.bias ? 1.f : 0.f;           // applied only now: the values must not be touched before the wait above
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TPOS; ++j)
#pragma unroll
                    for (int G = 0; G < 4; ++G) {
                        acc[i][j][4 * G] = bv[i][G >> 1].x * keep; acc[i][j][4 * G + 1] = bv[i][G >> 1].y * keep;
                        acc[i][j][4 * G + 2] = bv[i][G >> 1].z * keep; acc[i][j][4 * G + 3] = bv[i][G >> 1].w * keep;
                    }
        }
#ifndef ADK_EXP_NO_ACC_PIN
        // pin the bias -> accumulator copies HERE: sunk below the asm reads, the compiler guards them with a vmcnt(0) of its
        // own (it cannot count across inline asm), which also drains units 1..K-1.  No MFMA is in flight yet, so naming
        // the accumulators in an asm is harmless at this one place (see the note at lds_wait).
        if constexpr (!KEEP) {
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TPOS; ++j) asm volatile("" : "+v"(acc[i][j]));
        }
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned wt = lds_addr32(ws + wlane), xt = lds_addr32(xs + xlane(0));
        lds_read128<0>(aq[0][0], wt); lds_read128<1024>(aq[0][1], wt); lds_read128<2048>(aq[0][2], wt); lds_read128<3072>(aq[0][3], wt);
        lds_read128<0>(bq[0], xt); lds_read128<1024>(bq[1], xt); lds_read128<2048>(bq[2], xt);
    }

    auto phase = [&](auto tapc, auto parc, const int kc, const bool last) {
        constexpr int TAP_ = decltype(tapc)::value, PAR_ = decltype(parc)::value;
        constexpr int n_mid = []() { int n = 0; for (int u = 2; u < K; ++u) n += WI + ((TAP_ + u) % TAPS == 0 ? XI : 0); return n; }();
        constexpr int n_last = []() { int n = 0; for (int u = 2; u < K; ++u) n += (TAP_ + u < TAPS) ? WI : 0; return n; }();
        const int g = kc * TAPS + TAP_;
        // unit g+1 has landed (this wave's pieces: counted wait; the others': barrier), and every wave is done with unit g-1
        if (last) wait_vmcnt<n_last>(); else wait_vmcnt<n_mid>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + K < total) issue_unit(g + K, std::true_type());
        constexpr int nt = (TAP_ + 1) % TAPS;
        const int nkc = kc + (TAP_ + 1 == TAPS ? 1 : 0);
        const unsigned xt_c = lds_addr32(xs + (kc % RX) * XBYTES + xlane(TAP_));
        const unsigned xt_n = lds_addr32(xs + (nkc % RX) * XBYTES + xlane(nt));     // after the tile's last unit: a stale slot, discarded
        const unsigned wt_n = lds_addr32(ws + ((g + 1) % RW) * WBYTES + wlane);
        // step y: wait for B[y] (requested three steps ago; at y = 0 also this unit's A fragments), request B[y+3] (of the
        // next unit from y = 5) and, in the first four steps, the next unit's A[y]; then the four MFMAs A[0..3] x B[y].
        // Reads per step: 2, 2, 2, 2, 1, 1, 1, 1 -> reads younger than B[y] when step y starts: 2, 3, 4, 4, 4, 3, 2, 2.
#ifdef ADK_LINT_CONTROL_ACC_TIE
#define ADK_STEP_WAIT(Y_, NYOUNGER_) lds_wait_tie<NYOUNGER_>(bq[(Y_) & 3], acc[0][((Y_) - 1) >> 1], acc[1][((Y_) - 1) >> 1]);
#else
#define ADK_STEP_WAIT(Y_, NYOUNGER_) lds_wait<NYOUNGER_>(bq[(Y_) & 3]);
#endif
#define ADK_STEP(Y_, NYOUNGER_)                                                                              \
        if constexpr ((Y_) == 0) lds_wait<NYOUNGER_>(bq[0], aq[PAR_]); else ADK_STEP_WAIT(Y_, NYOUNGER_)        \
        if constexpr ((Y_) + 3 < NB) lds_read128<((Y_) + 3) * 1024>(bq[((Y_) + 3) & 3], xt_c);               \
        else lds_read128<((Y_) + 3 - NB) * 1024>(bq[((Y_) + 3) & 3], xt_n);                                  \
        if constexpr ((Y_) < NA) lds_read128<((Y_) & 3) * 1024>(aq[PAR_ ^ 1][(Y_) & 3], wt_n);               \
        mma16_sub<((Y_) & 1)>(aq[PAR_][0], bq[(Y_) & 3], acc[0][(Y_) >> 1]);                                 \
        mma16_sub<2 + ((Y_) & 1)>(aq[PAR_][1], bq[(Y_) & 3], acc[0][(Y_) >> 1]);                             \
        mma16_sub<((Y_) & 1)>(aq[PAR_][2], bq[(Y_) & 3], acc[1][(Y_) >> 1]);                                 \
        mma16_sub<2 + ((Y_) & 1)>(aq[PAR_][3], bq[(Y_) & 3], acc[1][(Y_) >> 1]);
        ADK_STEP(0, 2) ADK_STEP(1, 3) ADK_STEP(2, 4) ADK_STEP(3, 4) ADK_STEP(4, 4) ADK_STEP(5, 3) ADK_STEP(6, 2) ADK_STEP(7, 2)
#undef ADK_STEP
#undef ADK_STEP_WAIT
    };
    auto chunk = [&](auto parc, const int kc) {
        constexpr int P0 = decltype(parc)::value;
        const bool last = kc == a.nchunks - 1;
        phase(std::integral_constant<int, 0>(), std::integral_constant<int, P0>(), kc, last);
        if constexpr (TAPS > 1) phase(std::integral_constant<int, 1>(), std::integral_constant<int, P0 ^ 1>(), kc, last);
        if constexpr (TAPS > 2) phase(std::integral_constant<int, 2>(), std::integral_constant<int, P0>(), kc, last);
        if constexpr (TAPS > 3) phase(std::integral_constant<int, 3>(), std::integral_constant<int, P0 ^ 1>(), kc, last);
        if constexpr (TAPS > 4) phase(std::integral_constant<int, 4>(), std::integral_constant<int, P0>(), kc, last);
        if constexpr (TAPS > 5) phase(std::integral_constant<int, 5>(), std::integral_constant<int, P0 ^ 1>(), kc, last);
        if constexpr (TAPS > 6) phase(std::integral_constant<int, 6>(), std::integral_constant<int, P0>(), kc, last);
    };
    int kc = 0;
    for (; kc + 1 < a.nchunks; kc += 2) {
        chunk(std::integral_constant<int, 0>(), kc);
        chunk(std::integral_constant<int, 1>(), kc + 1);
    }
    if (kc < a.nchunks) chunk(std::integral_constant<int, 0>(), kc);
    // The last phase has requested fragments of a unit that does not exist (7 reads of stale LDS).  Nobody consumes them, but
    // they still WRITE their registers when they return, and the compiler - which sees asm outputs that are never used -
    // is free to give those registers to other values at once.  Keep every fragment register occupied until the reads
    // have returned (seen as run-to-run differences of a 100-step sample before this was here).
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(aq[0][0]), "+v"(aq[0][1]), "+v"(aq[0][2]), "+v"(aq[0][3]), "+v"(aq[1][0]), "+v"(aq[1][1]), "+v"(aq[1][2]),
                   "+v"(aq[1][3]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])
                 :
                 : "memory");
    __syncthreads();      // all fragment reads done before the epilogue reuses LDS
    ADK_TS(1)
    ConvTileId id;
    id.b = b; id.t = t; id.cot = cot; id.cobase = cobase; id.l0 = l0;
    return id;
}
