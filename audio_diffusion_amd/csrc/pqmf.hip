// PQMF analysis / synthesis filterbank (f32, NCL boundary tensors).
//
// Replaces CachedPQMF.forward / .inverse (reference diffusion/pqmf.py:271-302), including the
// channel->batch folding for stereo (:274-275, 280-281, 288-289, 299-300), reverse_half (:16-20)
// and the phase flip + interleave of the inverse (:293-297).  The two convolutions the reference
// delegates to cached_conv.Conv1d are restated as: zero padding of K/2 on both sides, then a plain
// strided (analysis, stride M) / unit-stride (synthesis) correlation.
//   analysis : y[b, ch*M + k, n] = sgn(k,n) * sum_j fw[k][j] * x[b, ch, n*M + j - K/2]
//   synthesis: z[m][n] = M * sum_c sum_t iw[m][c][t] * sgn(c,n+t-Kt/2) * x[b, ch*M + c, n + t - Kt/2]
//              y[b, ch, n*M + (M-1-m)] = z[m][n]
// sgn(k,n) = -1 for odd band k at even frame n.
//
// M = 16 / 32 (the bank sizes the models use): frame GEMMs on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (D[band][frame] += W[band][j] X[j][frame]; bit-for-bit an fp32 FMA chain).  In direct
// form the bank is 513 MACs per output sample = 128 FLOP per HBM byte, far above the fp32 vector ridge, so
// the stencil only approaches its HBM bound when the filter matrix is applied by the matrix cores.  A block
// stages the input segment of 128 frames in LDS once (row pitch M+1: the stride-M frame reads of the 32
// lanes hit 32 different banks); the filter is read coalesced from the transposed filter (L1/L2 resident).
// Other M: the generic LDS-tiled VALU stencil below.
#include "adk_common.h"

constexpr int PQ_FT = 64;     // frames per block

struct PqmfArgs {
    const float* x; const float* w; float* y;
    int B, Ch, T, M, K, N;
};

__global__ __launch_bounds__(256) void pqmf_analysis_kernel(const PqmfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M = a.M, K = a.K, half = K / 2;
    const int seg = (PQ_FT - 1) * M + K;
    float* xs = sm;                         // [seg]
    float* os = sm + ((seg + 3) & ~3);      // [M][PQ_FT + 1]
    const int n0 = blockIdx.x * PQ_FT, bc = blockIdx.y;
    const float* xin = a.x + (long long)bc * a.T;
    const int t0 = n0 * M - half;
    for (int i = threadIdx.x; i < seg; i += 256) {
        const int t = t0 + i;
        xs[i] = (t >= 0 && t < a.T) ? xin[t] : 0.f;
    }
    __syncthreads();
    const int k = threadIdx.x % M, nfb = 256 / M;
    for (int nf = threadIdx.x / M; nf < PQ_FT; nf += nfb) {
        const float* xp = xs + nf * M;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) acc += a.w[j * M + k] * xp[j];
        const int n = n0 + nf;
        os[k * (PQ_FT + 1) + nf] = ((k & 1) && !(n & 1)) ? -acc : acc;
    }
    __syncthreads();
    float* yout = a.y + (long long)bc * M * a.N;
    for (int i = threadIdx.x; i < M * PQ_FT; i += 256) {
        const int kk = i / PQ_FT, nf = i % PQ_FT;
        if (n0 + nf < a.N) yout[(long long)kk * a.N + n0 + nf] = os[kk * (PQ_FT + 1) + nf];
    }
}

__global__ __launch_bounds__(256) void pqmf_synthesis_kernel(const PqmfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M = a.M, Kt = a.K, half = Kt / 2;
    const int W = PQ_FT + Kt - 1;
    float* xs = sm;                         // [M][W], sign-flipped, zero padded
    const int n0 = blockIdx.x * PQ_FT, bc = blockIdx.y;
    const float* xin = a.x + (long long)bc * M * a.N;
    for (int i = threadIdx.x; i < M * W; i += 256) {
        const int c = i / W, n = n0 + (i % W) - half;
        float v = (n >= 0 && n < a.N) ? xin[(long long)c * a.N + n] : 0.f;
        xs[i] = ((c & 1) && !(n & 1)) ? -v : v;
    }
    __syncthreads();
    const int m = threadIdx.x % M, nfb = 256 / M;
    float* yout = a.y + (long long)bc * a.N * M;
    for (int nf = threadIdx.x / M; nf < PQ_FT; nf += nfb) {
        const int n = n0 + nf;
        float acc = 0.f;
        for (int c = 0; c < M; ++c) {
            const float* xp = xs + c * W + nf;
            const float* wp = a.w + (long long)c * Kt * M + m;
            for (int t = 0; t < Kt; ++t) acc += wp[t * M] * xp[t];
        }
        if (n < a.N) yout[(long long)n * M + (M - 1 - m)] = acc * (float)M;
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int PQ_FTM = 128;   // frames per block of the MFMA kernels (4 waves x 32 frames)

__global__ __launch_bounds__(256) void pqmf_analysis_mfma_kernel(const PqmfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M = a.M, K = a.K, half = K / 2;
    const int lm = 31 - __builtin_clz(M);                   // M is a power of two <= 32
    const int seg = (PQ_FTM - 1) * M + K;
    const int n0 = blockIdx.x * PQ_FTM, bc = blockIdx.y;
    const float* xin = a.x + (long long)bc * a.T;
    const int t0 = n0 * M - half;
    for (int i = threadIdx.x; i < seg; i += 256) {
        const int t = t0 + i;
        sm[(i >> lm) * (M + 1) + (i & (M - 1))] = (t >= 0 && t < a.T) ? xin[t] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A operand: lane (band = l31, h) supplies W[band][2q + h]; B operand: lane (frame = l31, h) supplies
    // X[2q + h][frame] = segment[frame * M + 2q + h]
    const bool bvalid = l31 < M;
    const float* wp = a.w + l31;
    const int fbase = (wave * 32 + l31) * M;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto xat = [&](int i) { return sm[(i >> lm) * (M + 1) + (i & (M - 1))]; };
    const float* wl = wp + h * M;
    const int ib = fbase + h;
    const int nfull = K >> 1;               // k-pairs with both taps inside the filter
    int q = 0;
    for (; q + 8 <= nfull; q += 8) {
        float wv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            wv[u] = bvalid ? wl[2 * (q + u) * M] : 0.f;
            xv[u] = xat(ib + 2 * (q + u));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u], xv[u], acc, 0, 0, 0);
    }
    for (; 2 * q < K; ++q) {                // remaining pairs; the last one is half empty when K is odd
        const int j = 2 * q + h;
        const bool jv = j < K;
        const float wv = (bvalid && jv) ? wp[j * M] : 0.f;
        const float xv = jv ? xat(fbase + j) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv, acc, 0, 0, 0);
    }
    // acc[r] = D[band = (r & 3) + 8 (r >> 2) + 4 h][frame = l31]: a register is 2 bands x 32 consecutive frames
    const int n = n0 + wave * 32 + l31;
    float* yout = a.y + (long long)bc * M * a.N;
    if (n < a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int band = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (band < M) yout[(long long)band * a.N + n] = ((band & 1) && !(n & 1)) ? -acc[r] : acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void pqmf_synthesis_mfma_kernel(const PqmfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M = a.M, Kt = a.K, half = Kt / 2;
    const int W = PQ_FTM + Kt - 1;
    float* xs = sm;                         // [M][W], sign-flipped, zero padded
    const int n0 = blockIdx.x * PQ_FTM, bc = blockIdx.y;
    const float* xin = a.x + (long long)bc * M * a.N;
    for (int i = threadIdx.x; i < M * W; i += 256) {
        const int c = i / W, n = n0 + (i % W) - half;
        const float v = (n >= 0 && n < a.N) ? xin[(long long)c * a.N + n] : 0.f;
        xs[i] = ((c & 1) && !(n & 1)) ? -v : v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // K dimension = (band c, tap t) flattened: kk = c * Kt + t.  A: lane (phase m = l31, h) supplies
    // iw[m][kk = 2q + h]; B: lane (frame = l31, h) supplies xs[c][frame + t]
    const bool mvalid = l31 < M;
    const float* wp = a.w + l31;
    const int KK = M * Kt, nq = KK >> 1;    // M is even: every k-pair is full
    int c = 0, t = h;                       // (c, t) of kk = 2q + h, advanced by 2 per step (Kt >= 3)
    const int fr = wave * 32 + l31;
    const float* wl = wp + h * M;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int q = 0;
    for (; q + 8 <= nq; q += 8) {
        float wv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            wv[u] = mvalid ? wl[2 * (q + u) * M] : 0.f;
            xv[u] = xs[c * W + fr + t];
            t += 2;
            const bool wrap = t >= Kt;
            t = wrap ? t - Kt : t;
            c += wrap ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u], xv[u], acc, 0, 0, 0);
    }
    for (; q < nq; ++q) {
        const float wv = mvalid ? wl[2 * q * M] : 0.f;
        const float xv = xs[c * W + fr + t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv, acc, 0, 0, 0);
        t += 2;
        const bool wrap = t >= Kt;
        t = wrap ? t - Kt : t;
        c += wrap ? 1 : 0;
    }
    __syncthreads();                        // everyone is done with xs: reuse LDS to transpose the tile
    float* zs = sm;                         // [PQ_FTM][M + 1]
    const float scale = (float)M;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) zs[fr * (M + 1) + (M - 1 - m)] = acc[r] * scale;
    }
    __syncthreads();
    float* yout = a.y + ((long long)bc * a.N + n0) * M;
    const int lm = 31 - __builtin_clz(M);
    const int nvalid = (a.N - n0 < PQ_FTM ? a.N - n0 : PQ_FTM) * M;
    for (int i = threadIdx.x; i < nvalid; i += 256) yout[i] = zs[(i >> lm) * (M + 1) + (i & (M - 1))];
}

int launch_pqmf(const adk_op& op, hipStream_t st, bool synth)
{
    PqmfArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.w = reinterpret_cast<const float*>(op.w);
    a.y = reinterpret_cast<float*>(op.y);
    a.B = op.B; a.Ch = op.Cin; a.M = op.i0; a.K = op.taps;
    if (!synth) { a.T = op.L; a.N = op.L / a.M; } else { a.N = op.L; a.T = op.L * a.M; }
    if ((a.M == 32 || a.M == 16) && a.K >= 3) {
        dim3 gridm((a.N + PQ_FTM - 1) / PQ_FTM, op.B * op.Cin);
        if (!synth) {
            const int seg = (PQ_FTM - 1) * a.M + a.K;
            const size_t lds = (size_t)(seg / a.M + 2) * (a.M + 1) * sizeof(float);
            hipLaunchKernelGGL(pqmf_analysis_mfma_kernel, gridm, dim3(256), lds, st, a);
        } else {
            const size_t xsz = (size_t)a.M * (PQ_FTM + a.K - 1), zsz = (size_t)PQ_FTM * (a.M + 1);
            hipLaunchKernelGGL(pqmf_synthesis_mfma_kernel, gridm, dim3(256), (xsz > zsz ? xsz : zsz) * sizeof(float), st, a);
        }
        return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
    }
    dim3 grid((a.N + PQ_FT - 1) / PQ_FT, op.B * op.Cin);
    if (!synth) {
        const int seg = (PQ_FT - 1) * a.M + a.K;
        const size_t lds = (size_t)(((seg + 3) & ~3) + a.M * (PQ_FT + 1)) * sizeof(float);
        hipLaunchKernelGGL(pqmf_analysis_kernel, grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = (size_t)a.M * (PQ_FT + a.K - 1) * sizeof(float);
        hipLaunchKernelGGL(pqmf_synthesis_kernel, grid, dim3(256), lds, st, a);
    }
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}
