// Exchange of GroupNorm statistics between the tiles of one sample INSIDE a launch (ADK_FLAG_FUSE_GN; the design and
// its deadlock argument: conv_mfma_gn.inc).  cdna_hip_programming.md, Guideline 16, form R2: naturally aligned 8-byte
// {tag = epoch, value} granules, one agent-scope store each, re-read with agent-scope loads by ONE wave until every
// tag matches; no fence, no flag, nothing to zero (the epoch is unique per launch).
#pragma once
#include "conv_common.h"

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define ADK_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

#ifndef ADK_GN_POLL_SLEEP      /* s_sleep argument between two polls (x 64 cycles). Same-box, whole sampler:
                                * 2 / 8 / 32 within 0.2 % of each other, 100 is 2 % slower (profiles/r03/v3_ab_poll_sleep.txt) */
#define ADK_GN_POLL_SLEEP 8
#endif
constexpr unsigned GN_SPIN_LIMIT = 1u << 20;     // ~1 us per poll: gives up after about a second
constexpr unsigned GN_LIST_STRIDE = 512;         // words between the ticket counters of two XCDs' lists (separate L2 lines):
                                                 // counter of (slot, XCD k) = counters[k * 512 + slot], slot < 511
constexpr int GN_MAX_NPB = 512;                  // tiles per sample: 2 * 512 granules = 16 per lane of the sweeping wave

// One lane publishes this tile's (sum, sumsq).
__device__ __forceinline__ void gn_publish(const ConvArgs& a, int b, int tile, unsigned epoch, float s1, float s2)
{
    gu64* g = (gu64*)(a.gn_ws) + ((size_t)b * a.npb + (size_t)tile) * 2;
    __hip_atomic_store(g, ((unsigned long long)epoch << 32) | __float_as_uint(s1), ADK_RLX_AGENT);
    __hip_atomic_store(g + 1, ((unsigned long long)epoch << 32) | __float_as_uint(s2), ADK_RLX_AGENT);
}

// ONE wave (all 64 lanes) gathers the sample's granules and returns mean / rstd on every lane.  Lane parity: even lanes
// read sums, odd lanes sums of squares; both are added in granule order in double, whatever the arrival order.
__device__ __forceinline__ void gn_gather(const ConvArgs& a, int b, unsigned epoch, int lane, float& mean, float& rstd)
{
    gu64* gs = (gu64*)(a.gn_ws) + (size_t)b * a.npb * 2;
    const int ng = 2 * a.npb;
    // Only the granules that have not arrived yet are read again: a sample's tiles finish over ~10 us, every one of up to
    // 511 waiting blocks polls, and re-reading the whole list (up to 8 KiB, agent scope: it bypasses the L2s) on every
    // poll would put a second output tensor's worth of traffic on the fabric while the late tiles still need it.
    unsigned long long x[16];
    unsigned have = 0;                                // bit k: x[k] holds this launch's value
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (lane + 64 * k >= ng) { x[k] = (unsigned long long)epoch << 32; have |= 1u << k; }      // value +0.0
    // A launch whose error word is already set (an earlier wait of this forward ran out, or the ticket counters are out of
    // step) does not wait again: one poll, then on with what is there.  The forward's output is discarded by the host
    // anyway (the word is sticky until the host clears it), so a mis-stepped run costs ONE bounded wait in all, not one per
    // resident block per launch.  The word is read with the first poll's loads and again every 64 polls.
    unsigned limit = __hip_atomic_load((gu32*)(a.gn_err), ADK_RLX_AGENT) ? 1u : GN_SPIN_LIMIT;
    for (unsigned spins = 0;;) {
        unsigned long long v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)                   // all the loads of this poll go out before the first tag is looked at
            v[k] = (have & (1u << k)) ? x[k] : __hip_atomic_load(gs + lane + 64 * k, ADK_RLX_AGENT);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if ((unsigned)(v[k] >> 32) == epoch) { x[k] = v[k]; have |= 1u << k; }
        if (__all(have == 0xffffu)) break;
        ++spins;
        if ((spins & 63u) == 0u && spins < limit && __hip_atomic_load((gu32*)(a.gn_err), ADK_RLX_AGENT)) limit = spins;
        if (spins >= limit) {                         // bounded: a block that is not resident must not hang the GPU
            // 0x6sssbbb: which op (slot) and which sample ran out of patience (the FIRST such code of a forward stays)
            if (lane == 0 && limit == GN_SPIN_LIMIT)
                __hip_atomic_store((gu32*)(a.gn_err), 0x60000000u | (a.gn_slot << 12) | ((unsigned)b & 0xfffu), ADK_RLX_AGENT);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (!(have & (1u << k))) x[k] = 0;
            break;
        }
        __builtin_amdgcn_s_sleep(ADK_GN_POLL_SLEEP);
    }
    double acc_d = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc_d += (double)__uint_as_float((unsigned)x[k]);
    // within each parity class (even lanes: sums, odd lanes: sums of squares), by DPP moves that keep a lane's parity:
    // xor 2 inside a quad, then rotations by 4 and 8 inside the 16-lane row; lanes 0 / 1 of the four rows then hold the
    // row totals of the two classes, added in the order (0 + 1) + (2 + 3).  A fixed tree, as the shuffle butterfly was.
    acc_d += dpp_mov<DPP_QUAD_XOR2>(acc_d);
    acc_d += dpp_mov<DPP_ROW_ROR4>(acc_d);
    acc_d += dpp_mov<DPP_ROW_ROR8>(acc_d);
    const double tot = (lane_value(acc_d, 0) + lane_value(acc_d, 16)) + (lane_value(acc_d, 32) + lane_value(acc_d, 48));
    const double q = (lane_value(acc_d, 1) + lane_value(acc_d, 17)) + (lane_value(acc_d, 33) + lane_value(acc_d, 49));
    const double mu = tot / a.gn_count;
    double var = q / a.gn_count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
}

// publish + gather + broadcast to the block through two LDS words (red[8], red[9]); every thread calls it.
__device__ __forceinline__ void gn_exchange(const ConvArgs& a, int b, int tile, unsigned epoch, float s1, float s2,
                                            float* red, int tid, float& mean, float& rstd)
{
    if (tid == 0) gn_publish(a, b, tile, epoch, s1, s2);
    if ((tid >> 6) == 0) {
        float mu, rs;
        gn_gather(a, b, epoch, tid & 63, mu, rs);
        if (tid == 0) { red[8] = mu; red[9] = rs; }
    }
    __syncthreads();
    mean = red[8]; rstd = red[9];
}
