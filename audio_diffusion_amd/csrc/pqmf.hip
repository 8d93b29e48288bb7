// PQMF analysis / synthesis filterbank as LDS-tiled stencils (f32, NCL boundary tensors).
//
// Replaces CachedPQMF.forward / .inverse (reference diffusion/pqmf.py:271-302), including the
// channel->batch folding for stereo (:274-275, 280-281, 288-289, 299-300), reverse_half (:16-20)
// and the phase flip + interleave of the inverse (:293-297).  The two convolutions the reference
// delegates to cached_conv.Conv1d are restated as: zero padding of K/2 on both sides, then a plain
// strided (analysis, stride M) / unit-stride (synthesis) correlation.
//   analysis : y[b, ch*M + k, n] = sgn(k,n) * sum_j fw[k][j] * x[b, ch, n*M + j - K/2]
//   synthesis: z[m][n] = M * sum_c sum_t iw[m][c][t] * sgn(c,n+t-Kt/2) * x[b, ch*M + c, n + t - Kt/2]
//              y[b, ch, n*M + (M-1-m)] = z[m][n]
// sgn(k,n) = -1 for odd band k at even frame n.  A block stages the input segment of FT frames in LDS
// once; lanes run over bands (filter reads coalesced from the transposed filter, input reads are LDS
// broadcasts); analysis output is transposed through LDS so global stores run along frames.
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

int launch_pqmf(const adk_op& op, hipStream_t st, bool synth)
{
    PqmfArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.w = reinterpret_cast<const float*>(op.w);
    a.y = reinterpret_cast<float*>(op.y);
    a.B = op.B; a.Ch = op.Cin; a.M = op.i0; a.K = op.taps;
    if (!synth) { a.T = op.L; a.N = op.L / a.M; } else { a.N = op.L; a.T = op.L * a.M; }
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
