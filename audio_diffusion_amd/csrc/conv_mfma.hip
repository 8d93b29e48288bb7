// Implicit-GEMM Conv1d (k taps, stride 1, zero "same" padding) on MFMA, chunk-plane layout.
//
// Replaces nn.Conv1d in ResConvBlock / skip / qkv_proj / out_proj (reference blocks/blocks.py:68-73,
// 109-110).  D[co][pos] = sum_tap sum_ci W[tap][co][ci] * X[pos + tap - taps/2][ci]:
//   A operand (MFMA rows)  = packed weights  [Cin/CB][taps][CoutPad][CB]
//   B operand (MFMA cols)  = activation rows (CPL: one 64-byte row per position per channel chunk)
// One K-stage = one channel chunk (64 B per row): the X tile (BM + taps - 1 rows) is staged ONCE in
// LDS and re-read at `taps` row shifts; the zero padding of the convolution is read from the halo
// rows of the layout.  Tiles reach LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no ds_write),
// XOR-swizzled on the source address so every ds_read_b128 fragment read is conflict-free.  Two or three
// blocks are co-resident per CU: one block's DMA overlaps another block's MFMAs.
//   bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate);  f32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
// Epilogue: + bias, (+ residual), per-tile (sum, sumsq) partials for the following GroupNorm(1,C)
// (deterministic: fixed-order tree, no atomics), halo rows of the output zeroed by the edge tiles.
#include <atomic>
#include <mutex>
#include "conv_mfma_kernels.h"

void conv_fill_args(const adk_op& op, int BN, int BM, ConvArgs& a)
{
    const int CB = op.dtype == ADK_F32 ? 16 : 32;
    const int CBO = (op.dtype == ADK_F32 || (op.flags & ADK_FLAG_OUT_F32)) ? 16 : 32;
    a.x = op.x; a.w = op.w; a.bias = reinterpret_cast<const float*>(op.bias); a.residual = op.residual; a.y = op.y; a.partials = op.partials_out;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.M = (long long)op.B * a.Lp;
    a.nchunks = op.Cin / CB;
    a.Cout_alloc = (op.Cout + CBO - 1) / CBO * CBO;
    a.CoutPad = op.CoutPad;
    a.tps = (op.L + BM - 1) / BM;
    a.ncot = op.CoutPad / BN;
    a.nblk = (unsigned)(op.B * a.tps * a.ncot);
    a.gstats = (op.flags & ADK_FLAG_GROUP_STATS) ? 1 : 0;
    a.dil = op.i0 > 1 ? op.i0 : 1;
    a.shift = (op.flags & ADK_FLAG_CAUSAL) ? (op.taps - 1) * a.dil : (op.taps / 2) * a.dil;
    a.elu = (op.flags & ADK_FLAG_ELU) ? 1 : 0;
    a.gamma = a.beta = nullptr; a.gn_ws = nullptr; a.gn_epoch = nullptr; a.gn_err = nullptr;
    a.gn_slot = a.gn_cap = 0; a.gn_grid = a.nblk; a.gn_ticket = nullptr; a.npb = a.tps * a.ncot; a.gelu = 0; a.eps = 0.f; a.gn_count = 1.0;
    if (op.flags & ADK_FLAG_FUSE_GN) {
        a.gamma = op.gamma; a.beta = op.beta;
        a.gn_ws = reinterpret_cast<unsigned long long*>(op.aux1);
        a.gn_epoch = reinterpret_cast<const unsigned*>(op.aux0);
        a.gn_err = const_cast<unsigned*>(reinterpret_cast<const unsigned*>(op.aux0)) + 1;
        a.gn_slot = (unsigned)op.i1;
        a.gn_ticket = a.gn_err + 1 + op.i1;                  // aux0 = {epoch base, error, ticket counters[8 lists][512]}
        a.gelu = (op.flags & ADK_FLAG_GELU) ? 1 : 0;
        a.eps = op.f0; a.gn_count = (double)op.Cout * (double)op.L;
    }
    a.pn_partials = nullptr; a.pn_n = 0;
    if (op.flags & ADK_FLAG_PRE_NORM) {                      // bias travels as `beta`: it is applied AFTER the per-sample scale
        a.pn_partials = op.partials_in; a.pn_n = op.n_partials_in;
        a.gamma = op.gamma; a.beta = reinterpret_cast<const float*>(op.bias); a.bias = nullptr;
        a.eps = op.f0; a.gn_count = (double)op.i2 * (double)op.L;
    }
    a.w2 = nullptr; a.nchunks2 = 0;
    if (op.flags & ADK_FLAG_SKIP_CONV) { a.w2 = op.w2; a.nchunks2 = op.i2 / CB; }
}

// ADK_FLAG_FUSE_GN: which conv ops the fused conv + GroupNorm kernel takes (host-only; the launch re-checks the
// resident-block capacity of the actual device).  512 = block slots of an MI355X (256 CUs x 2 blocks of this kernel).
int launch_conv_gn_f32(const adk_op& op, hipStream_t st);
int launch_conv_gn_bf16(const adk_op& op, hipStream_t st);
int launch_conv_gn_f16(const adk_op& op, hipStream_t st);
int conv_gn_check(const adk_op& op, const char** why)
{
    static const char* none = "";
    const char* dummy;
    if (!why) why = &dummy;
    *why = none;
    const ConvTile t = conv_tile(op);
    if (!t.ksplit && t.bn != 128) { *why = "not for the narrow-output kernel (Cout <= 32 above the short levels)"; return 0; }
    if (op.taps != 3 && op.taps != 5 && op.taps != 7) { *why = "taps must be 3, 5 or 7"; return 0; }
    if ((op.flags & (ADK_FLAG_CAUSAL | ADK_FLAG_ELU | ADK_FLAG_GROUP_STATS | ADK_FLAG_OUT_F32)) || op.i0 > 1) {
        *why = "not with causal / dilated / ELU / group-statistics / f32-output convs"; return 0;
    }
    const int npb = ((op.L + t.bm - 1) / t.bm) * (op.CoutPad / t.bn);
    if (npb > 512) { *why = "more than 512 tiles per sample (they must all be in flight together)"; return 0; }
    if (!t.ksplit && ((op.L + t.bm - 1) / t.bm + 7) / 8 * (op.CoutPad / t.bn) > 64) {
        *why = "a sample's share of one XCD's ticket list exceeds the XCD's 64 block slots"; return 0;
    }
    return 1;
}

// ---- does THIS device deal workgroups the way the fused launch's ticket lists assume? -----------------------------
// The per-XCD ticket lists (conv_mfma_gn.inc) rest on dispatch behaviour HIP does not promise: 8 XCDs (HW_REG_XCC_ID in
// 0..7), and a 1-D grid that is a multiple of 8 dealt round-robin so that exactly grid / 8 workgroups start on each XCD.
// A partitioned GPU (CPX / DPX / QPX) or a part with another XCD count breaks both, and the counters would then run out
// of step: every resident block would spin to its bound and the forward would be thrown away.  So before the first
// fused launch on a device (adk_conv_gn_fusable at plan time, else the first launch) a probe kernel counts the
// workgroups per XCC_ID for two grids; the fusion is refused on that device unless both came out exactly even.
// Nothing is allocated: the counters are a file-scope __device__ array (address by hipGetSymbolAddress, per device).
static __device__ __align__(64) unsigned g_xcd_probe[16];
__global__ void xcd_probe_kernel(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(out + (xcc < 8u ? xcc : 8u), 1u);
    }
}

int gn_device_ok(const char** why, bool may_probe)
{
    static std::atomic<int> state[64];            // 0 = not probed, 1 = ok, 2 = refused
    static std::mutex mu;
    static const char* reason[64];
    const char* dummy;
    if (!why) why = &dummy;
    *why = "";
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *why = "no current HIP device"; return 0; }
    int s = state[dev].load(std::memory_order_acquire);
    if (!s && !may_probe) { *why = "device not probed yet (call adk_conv_gn_fusable outside stream capture first)"; return 0; }
    if (!s) {
        std::lock_guard<std::mutex> lock(mu);
        s = state[dev].load(std::memory_order_acquire);
        if (!s) {
            unsigned* ptr = nullptr;
            unsigned host[16];
            int cus = 0;
            const char* r = nullptr;
            bool transient = false;                // a failure that says nothing about the device (e.g. stream capture in progress)
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8 || cus % 8)
                r = "the device's CU count is not a multiple of 8";
            else if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(g_xcd_probe)) != hipSuccess || !ptr)
                r = "hipGetSymbolAddress(g_xcd_probe) failed";
            for (unsigned grid : {512u, 1032u}) {
                if (r) break;
                if (hipMemset(ptr, 0, sizeof host) != hipSuccess) { r = "probe: hipMemset failed (stream capture in progress?)"; transient = true; break; }
                hipLaunchKernelGGL(xcd_probe_kernel, dim3(grid), dim3(64), 0, 0, ptr);
                if (hipGetLastError() != hipSuccess || hipMemcpy(host, ptr, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) {
                    r = "probe: launch / copy failed (stream capture in progress?)"; transient = true; break;
                }
                for (int k = 0; k < 8; ++k)
                    if (host[k] != grid / 8) r = "workgroups are not dealt evenly to 8 XCDs on this device (partition mode?)";
                if (host[8]) r = "XCC_ID outside 0..7 on this device";
            }
            (void)hipGetLastError();
            if (r && transient) { *why = r; return 0; }          // not cached: ask again outside the capture
            reason[dev] = r;
            s = r ? 2 : 1;
            state[dev].store(s, std::memory_order_release);
        }
    }
    if (s == 2) *why = reason[dev];
    return s == 1;
}

// adk_conv_gn_fusable: the shape check and - for the pipelined kernel, whose ticket lists need it - the device probe
int conv_gn_fusable_here(const adk_op& op)
{
    if (!conv_gn_check(op, nullptr)) return 0;
    if (conv_tile(op).ksplit) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {      // host-only dry run of a plan (no GPU in this process):
        (void)hipGetLastError();                                   // the shape answer; nothing can be launched here anyway,
        return 1;                                                  // and every launch path probes its device itself
    }
    return gn_device_ok(nullptr, true);
}

int launch_conv_gn(const adk_op& op, hipStream_t st)
{
    // the probe synchronises: never start it from inside a stream capture (it would invalidate the capture)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (!gn_device_ok(nullptr, !capturing)) return ADK_EINVAL;
    if (op.dtype == ADK_F32) return launch_conv_gn_f32(op, st);
    if (op.dtype == ADK_F16) return launch_conv_gn_f16(op, st);
    return launch_conv_gn_bf16(op, st);
}

// tile-shape selection (host); adk_conv_cotile / adk_conv_postile / adk_partials_count report it.
static inline int conv_bn(int Cout) { return Cout <= 32 ? 32 : 128; }
ConvTile conv_tile(const adk_op& op)
{
    if (conv_bn(op.Cout) == 32) return {32, 128, false};
    // Short levels: too few 128-row tiles to fill 256 CUs, serial K loop -> small tiles whose K range is split
    // over the block's 4 waves (conv_ksplit.hip).  Measured on MI355X at B=16, C=512 (us, k=5): L=64 ks32x64 10.7
    // vs pipelined 128x64 16.3; L=128 ks64x64 14.6 vs 17.0; L=256 pipelined 128x64 19.3 vs ks64x64 26.1 (the
    // k-split kernel's direct operand loads saturate the L1 at ~470 TF/s); 1x1: L=128 pipelined 11.3 vs 13.8.
    if (op.L <= 32) return {32, 32, true};
    if (op.L <= 64) return {32, 64, true};
    if (op.L <= 128 && op.taps != 1) return {64, 64, true};
    if (op.L <= 256) return {128, 64, false};
    // wide tile (W amortised over 256 positions) from L = 2048 up.  The choice depends on L only, never on B:
    // the tile shape fixes the order in which GroupNorm partials are summed, and a sample's result must not
    // depend on how many other samples share its launch (batch sharding across GPUs stays bit-exact).
    // (256co x 256pos and 128co x 512pos one-wave-per-SIMD tiles were measured 7-20 % slower.)
    // (an 8-wave 128co x 512pos tile, which halves the weight bytes staged per FLOP, measured within +-1 % of this one
    // on every level and on the whole call: the kernel is not bound by the L2 -> LDS fill)
#if defined(ADK_EXP_BM512) || defined(ADK_EXP_BM1024)      // experiment builds only: position tile of the L = 512 / 1024 levels
#ifdef ADK_EXP_BM512
    if (op.L > 256 && op.L <= 512) return {128, ADK_EXP_BM512, false};
#endif
#ifdef ADK_EXP_BM1024
    if (op.L > 512 && op.L <= 1024) return {128, ADK_EXP_BM1024, false};
#endif
#endif
    return {128, op.L >= 2048 ? 256 : 128, false};
}

int conv_cotile(int Cout) { return conv_bn(Cout); }
int conv_postile(const adk_op& op) { return conv_tile(op).bm; }
int conv_cotile_used(const adk_op& op) { return conv_tile(op).bn; }
int conv_partials_count(const adk_op& op)
{
    const ConvTile t = conv_tile(op);
    if (op.flags & ADK_FLAG_GROUP_STATS)        // [group][position tile][wave column]: 4 columns in the k-split and
        return ((op.Cout + 31) / 32) * ((op.L + t.bm - 1) / t.bm) * ((t.ksplit || t.bn == 32) ? 4 : 2);   // Cout<=32 kernels, else 2
    return ((op.L + t.bm - 1) / t.bm) * (op.CoutPad / t.bn);
}

template <typename T, typename TO>
static int launch_taps(const adk_op& op, hipStream_t st)
{
    if (op.taps == 1) return launch_tiles<false, T, TO, 1>(op, st);
    if (op.taps == 5) return launch_tiles<false, T, TO, 5>(op, st);
    if (op.taps == 3) return launch_tiles<false, T, TO, 3>(op, st);      // learned resamplers (blocks/blocks.py:214-255)
    if (op.taps == 7) return launch_tiles<false, T, TO, 7>(op, st);      // kernel_size=7 models (train_uncond.py:97-106)
    return ADK_EINVAL;
}

int launch_conv(const adk_op& op, hipStream_t st)
{
    const ConvTile ct = conv_tile(op);
    if (ct.ksplit) return launch_conv_ksplit(op, st);
    if (op.flags & ADK_FLAG_FUSE_GN) return launch_conv_gn(op, st);
    if ((op.flags & (ADK_FLAG_CAUSAL | ADK_FLAG_ELU | ADK_FLAG_GROUP_STATS | ADK_FLAG_PRE_NORM)) || op.i0 > 1) return launch_conv_ext(op, st);
    if (op.dtype == ADK_F32) return launch_taps<float, float>(op, st);
    if (op.dtype == ADK_F16)
        return (op.flags & ADK_FLAG_OUT_F32) ? launch_taps<f16_t, float>(op, st) : launch_taps<f16_t, f16_t>(op, st);
    if (op.flags & ADK_FLAG_OUT_F32) return launch_taps<bf16_t, float>(op, st);
    return launch_taps<bf16_t, bf16_t>(op, st);
}

#ifdef ADK_ABL_TS
extern "C" int adk_exp_ts(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_adk_ts), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
