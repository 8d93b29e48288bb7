// Depthwise FIR x2 down/up-sampling with reflect padding, chunk-plane layout (HBM-bound stencils).
//
// Replaces Downsample1d.forward / Upsample1d.forward (reference blocks/blocks.py:191-196, 207-212).  The
// reference builds a dense diagonal [C,C,K] weight per call and runs a full conv; the result only
// depends on the K taps, so this is the depthwise stencil:
//   down: y[lo]   = sum_j k[j] * xr[2*lo + j - (K/2-1)]
//   up:   y[2m]   = sum_{j odd}  k2[j] * xr[m + (K-1-j)/2 - K/4]
//         y[2m+1] = sum_{j even} k2[j] * xr[m + (K-j)/2   - K/4]          (k2 = module buffer = 2*table)
// with xr the reflect extension (index -i -> i, L-1+i -> L-1-i).  Each thread owns a 16-byte channel
// slice and walks rows, so every global access is a coalesced 64-byte row segment.
#include "adk_common.h"

struct FirArgs {
    const void* x; void* y; const float* taps;
    int B, L, P;          // input geometry
    int Lo;               // output length
    int RB, nrb;          // output rows per block
};

__device__ __forceinline__ int reflect(int i, int L) {
    i = i < 0 ? -i : i;
    return i >= L ? 2 * (L - 1) - i : i;
}

template <typename T, int K, bool UP, int RUN>
__global__ __launch_bounds__(256) void fir_kernel(const FirArgs a)
{
    constexpr int NV = ElemTraits<T>::NV;
    const int rb = blockIdx.x, plane = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, q = tid & 3, r0 = tid >> 2;
    float k[K];
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = a.taps[j];

    const int Lpi = a.L + 2 * a.P, Lpo = a.Lo + 2 * a.P;
    const long long Mi = (long long)a.B * Lpi, Mo = (long long)a.B * Lpo;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) +
                              (((long long)plane * Mi + (long long)b * Lpi + a.P) * 64) + q * 16;
    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + (((long long)plane * Mo + (long long)b * Lpo) * 64) + q * 16;

    // Each thread produces RUN consecutive output rows (down) / output row pairs (up) from a sliding window of
    // input rows, so an input row is fetched 1 + (K-2)/(2 RUN) times instead of K/2 (down) or K/2+1 (up) times, and a
    // thread's accesses are one sequential stream.  RUN = 16 on the long levels (measured on [16,128,65536]: down
    // 110 -> 88 us, up 105 -> 90 us against RUN = 4; RUN = 8 in between), 4 where there are too few rows to spread.
    if (!UP) {
        const int lbeg = rb * a.RB, lend = lbeg + a.RB < a.Lo ? lbeg + a.RB : a.Lo;
        for (int lo = lbeg + r0 * RUN; lo < lend; lo += 64 * RUN) {
            float acc[RUN][NV];
#pragma unroll
            for (int u = 0; u < RUN; ++u)
#pragma unroll
                for (int e = 0; e < NV; ++e) acc[u][e] = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * (RUN - 1) + K; ++i) {          // input row 2*lo - (K/2-1) + i feeds output u with tap i - 2u
                const int li = reflect(2 * lo + i - (K / 2 - 1), a.L);
                float f[NV];
                piece_to_float(*reinterpret_cast<const uint4*>(xb + (long long)li * 64), f, T());
#pragma unroll
                for (int u = 0; u < RUN; ++u) {
                    const int j = i - 2 * u;
                    if (j >= 0 && j < K) {
#pragma unroll
                        for (int e = 0; e < NV; ++e) acc[u][e] += k[j] * f[e];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RUN; ++u)
                if (lo + u < lend) *reinterpret_cast<uint4*>(yb + (long long)(a.P + lo + u) * 64) = float_to_piece(acc[u], T());
        }
    } else {
        // a.RB counts INPUT rows m here; each produces output rows 2m, 2m+1
        const int mbeg = rb * a.RB, mend = mbeg + a.RB < a.L ? mbeg + a.RB : a.L;
        for (int m = mbeg + r0 * RUN; m < mend; m += 64 * RUN) {
            float ev[RUN][NV], od[RUN][NV];
#pragma unroll
            for (int u = 0; u < RUN; ++u)
#pragma unroll
                for (int e = 0; e < NV; ++e) { ev[u][e] = 0.f; od[u][e] = 0.f; }
#pragma unroll
            for (int i = 0; i < RUN + K / 2; ++i) {                 // input row m - K/4 + i feeds pair u with offset d = i - K/4 - u
                float f[NV];
                piece_to_float(*reinterpret_cast<const uint4*>(xb + (long long)reflect(m - K / 4 + i, a.L) * 64), f, T());
#pragma unroll
                for (int u = 0; u < RUN; ++u) {
                    const int d = i - K / 4 - u;
                    if (d >= -(K / 4) && d <= K / 2 - K / 4) {
                        const int je = K - 1 - 2 * (d + K / 4);     // odd tap feeding y[2m]
                        const int jo = K - 2 * (d + K / 4);         // even tap feeding y[2m+1]
                        if (je >= 0 && je < K) {
#pragma unroll
                            for (int e = 0; e < NV; ++e) ev[u][e] += k[je] * f[e];
                        }
                        if (jo >= 0 && jo < K) {
#pragma unroll
                            for (int e = 0; e < NV; ++e) od[u][e] += k[jo] * f[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RUN; ++u)
                if (m + u < mend) {
                    *reinterpret_cast<uint4*>(yb + (long long)(a.P + 2 * (m + u)) * 64) = float_to_piece(ev[u], T());
                    *reinterpret_cast<uint4*>(yb + (long long)(a.P + 2 * (m + u) + 1) * 64) = float_to_piece(od[u], T());
                }
        }
    }
    if (rb == 0 || rb == a.nrb - 1) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int r = r0; r < a.P; r += 64) {
            if (rb == 0) *reinterpret_cast<uint4*>(yb + (long long)r * 64) = z;
            if (rb == a.nrb - 1) *reinterpret_cast<uint4*>(yb + (long long)(a.P + a.Lo + r) * 64) = z;
        }
    }
}

template <typename T, bool UP>
static int launch_fir_t(const adk_op& op, hipStream_t st)
{
    constexpr int CB = ElemTraits<T>::CB;
    FirArgs a;
    a.x = op.x; a.y = op.y; a.taps = reinterpret_cast<const float*>(op.w);
    a.B = op.B; a.L = op.L; a.P = op.P;
    a.Lo = UP ? op.L * 2 : op.L / 2;
    const int rows = UP ? op.L : a.Lo;           // loop domain
    const bool wide = rows >= 2048;              // RUN = 16: 1024 rows per block
    a.RB = wide ? 1024 : (rows >= 1024 ? 512 : ((rows + 63) / 64) * 64);
    a.nrb = (rows + a.RB - 1) / a.RB;
    dim3 grid(a.nrb, (op.Cin + CB - 1) / CB, op.B);
#define ADK_FIR_LAUNCH(K_)                                                                              \
    if (wide) hipLaunchKernelGGL((fir_kernel<T, K_, UP, 16>), grid, dim3(256), 0, st, a);               \
    else hipLaunchKernelGGL((fir_kernel<T, K_, UP, 4>), grid, dim3(256), 0, st, a);
    switch (op.taps) {
        case 4: ADK_FIR_LAUNCH(4) break;
        case 8: ADK_FIR_LAUNCH(8) break;
        case 12: ADK_FIR_LAUNCH(12) break;
        default: return ADK_EINVAL;
    }
#undef ADK_FIR_LAUNCH
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

int launch_fir(const adk_op& op, hipStream_t st, bool up)
{
    if (op.dtype == ADK_F32) return up ? launch_fir_t<float, true>(op, st) : launch_fir_t<float, false>(op, st);
    if (op.dtype == ADK_F16) return up ? launch_fir_t<f16_t, true>(op, st) : launch_fir_t<f16_t, false>(op, st);
    return up ? launch_fir_t<bf16_t, true>(op, st) : launch_fir_t<bf16_t, false>(op, st);
}

// ------------------------------------------------------------------------------------------------
// Learned resamplers (reference blocks/blocks.py:214-255: Downsample1d_2 = strided Conv1d(k = 2f+1, stride f,
// padding f), Upsample1d_2 = ConvTranspose1d(k = 2f, stride f)) are run as a PHASE SHUFFLE + a 3-tap stride-1
// convolution on the MFMA kernels (the weight repack is host-side, audio_diffusion_amd/unet.py):
//   space_to_depth: y[p*C + c][m] = x[c][m*f + p]        [B, C, L]   -> [B, f*C, L/f]   (before the down conv)
//   depth_to_space: y[c][j*f + s] = x[s*C + c][j]        [B, f*C, L] -> [B, C, L*f]     (after the up conv)
// With C a multiple of the 64-byte channel chunk a plane maps to a plane, so both are row copies (HBM-bound).
struct ShuffleArgs { const void* x; void* y; int B, Li, Lo, P, f, planes_c; };   // planes_c = planes of C channels

template <bool TO_DEPTH>
__global__ __launch_bounds__(256) void phase_shuffle_kernel(const ShuffleArgs a)
{
    const int q = threadIdx.x & 3, r0 = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int plane_o = blockIdx.y, b = blockIdx.z;
    const int Lpi = a.Li + 2 * a.P, Lpo = a.Lo + 2 * a.P;
    const long long Mi = (long long)a.B * Lpi, Mo = (long long)a.B * Lpo;
    unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + ((long long)plane_o * Mo + (long long)b * Lpo) * 64 + q * 16;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + q * 16;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (r0 < a.Lo) {
        long long src;
        if (TO_DEPTH) {                       // output plane (p, pc): phase p of input plane pc
            const int p = plane_o / a.planes_c, pc = plane_o % a.planes_c;
            src = (long long)pc * Mi + (long long)b * Lpi + a.P + (long long)r0 * a.f + p;
        } else {                              // output row j*f + s of plane pc: row j of input plane (s, pc)
            const int j = r0 / a.f, s = r0 % a.f;
            src = (long long)(s * a.planes_c + plane_o) * Mi + (long long)b * Lpi + a.P + j;
        }
        *reinterpret_cast<uint4*>(yb + (long long)(a.P + r0) * 64) = *reinterpret_cast<const uint4*>(xb + src * 64);
    }
    if (blockIdx.x == 0 && (threadIdx.x >> 2) < a.P) {          // halo rows of the output
        const int r = threadIdx.x >> 2;
        *reinterpret_cast<uint4*>(yb + (long long)r * 64) = z;
        *reinterpret_cast<uint4*>(yb + (long long)(a.P + a.Lo + r) * 64) = z;
    }
}

int launch_phase_shuffle(const adk_op& op, hipStream_t st, bool to_depth)
{
    const int CB = op.dtype == ADK_F32 ? 16 : 32;
    ShuffleArgs a;
    a.x = op.x; a.y = op.y; a.B = op.B; a.P = op.P; a.f = op.i0;
    a.planes_c = op.Cin / CB;                 // op.Cin = C (channels of the NON-folded tensor)
    a.Li = op.L;
    a.Lo = to_depth ? op.L / a.f : op.L * a.f;
    const int planes_o = to_depth ? a.planes_c * a.f : a.planes_c;
    dim3 grid((a.Lo + 63) / 64, planes_o, op.B);
    if (to_depth) hipLaunchKernelGGL(phase_shuffle_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(phase_shuffle_kernel<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}
