// Noise-level conditioning of the k-diffusion denoiser (AudioDenoiserModel, reference
// decoders/diffusion_decoder.py:296-305): tiny per-sample dense layers, fp32, one wave per output.
//   fourier : c_noise = log(sigma) / 4 (or sigma), FourierFeatures(1, F)      (blocks/blocks.py:161-163)
//   linear  : y[b][o] = act(sum_i W[o][i] x[b][i] + bias[o] + residual[b][o])  -> MappingNet (:292-301),
//             mapping_cond (:277,301) and ALL AdaGN mappers of the network at once (blocks/blocks.py:59,62:
//             their weights are stacked on the host, so one launch yields every layer's (weight, bias) rows)
// The products are summed lane-strided then by a fixed shuffle tree: bit-reproducible, batch-invariant.
#include "adk_common.h"

struct LinArgs {
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int B, I, O, xs, rs, ys, gelu;
};

__global__ __launch_bounds__(256) void linear_kernel(const LinArgs a)
{
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (o >= a.O) return;                                   // wave-uniform; no barrier in this kernel
    const float* wr = a.w + (long long)o * a.I;
    const float* xr = a.x + (long long)b * a.xs;
    float acc = 0.f;
    for (int i = lane; i < a.I; i += 64) acc = __builtin_fmaf(wr[i], xr[i], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        if (a.bias) acc += a.bias[o];
        if (a.res) acc += a.res[(long long)b * a.rs + o];
        a.y[(long long)b * a.ys + o] = a.gelu ? gelu_erf(acc) : acc;
    }
}

// Wide layers (the stacked AdaGN mappers: ~1e5 outputs): w is given TRANSPOSED, [Cin][Cout], so a wave owns 64
// consecutive outputs, every weight load is one coalesced 256-byte row segment, the matrix is read once for all
// samples and a dot product is a sequential fp32 chain over i (no cross-lane step).
__global__ __launch_bounds__(256) void linear_wt_kernel(const LinArgs a)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.O) return;
    constexpr int NB = 8;
    for (int b0 = 0; b0 < a.B; b0 += NB) {
        float acc[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) acc[u] = 0.f;
#pragma unroll 16
        for (int i = 0; i < a.I; ++i) {                   // 16 independent weight loads in flight per thread
            const float w = a.w[(long long)i * a.O + o];
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (b0 + u < a.B) acc[u] = __builtin_fmaf(w, a.x[(long long)(b0 + u) * a.xs + i], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (b0 + u < a.B) {
                float v = acc[u];
                if (a.bias) v += a.bias[o];
                if (a.res) v += a.res[(long long)(b0 + u) * a.rs + o];
                a.y[(long long)(b0 + u) * a.ys + o] = a.gelu ? gelu_erf(v) : v;
            }
    }
}

int launch_linear(const adk_op& op, hipStream_t st)
{
    LinArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.w = reinterpret_cast<const float*>(op.w);
    a.bias = reinterpret_cast<const float*>(op.bias); a.res = reinterpret_cast<const float*>(op.residual);
    a.y = reinterpret_cast<float*>(op.y);
    a.B = op.B; a.I = op.Cin; a.O = op.Cout; a.xs = op.i0; a.rs = op.i1; a.ys = op.i2;
    a.gelu = (op.flags & ADK_FLAG_GELU) ? 1 : 0;
    if (op.flags & ADK_FLAG_WT) hipLaunchKernelGGL(linear_wt_kernel, dim3((op.Cout + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(linear_kernel, dim3((op.Cout + 3) / 4, op.B), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

#pragma clang fp contract(off)
struct FourierArgs { const float* sigma; const float* w; float* y; int B, half, ys, log_sigma; };

__global__ __launch_bounds__(256) void fourier_kernel(const FourierArgs a)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.half) return;
    float s = a.sigma[b];
    if (a.log_sigma) s = logf(s) / 4.f;                     // diffusion_decoder.py:297-298
    const float tp = 6.283185307179586f * s;                // 2 * math.pi * input @ weight.T, each product rounded
    const float f = tp * a.w[j];
    a.y[(long long)b * a.ys + j] = cosf(f);
    a.y[(long long)b * a.ys + a.half + j] = sinf(f);
}

int launch_fourier(const adk_op& op, hipStream_t st)
{
    FourierArgs a;
    a.sigma = reinterpret_cast<const float*>(op.x); a.w = reinterpret_cast<const float*>(op.w);
    a.y = reinterpret_cast<float*>(op.y);
    a.B = op.B; a.half = op.Cout / 2; a.ys = op.i2; a.log_sigma = (op.flags & ADK_FLAG_LOG) ? 1 : 0;
    hipLaunchKernelGGL(fourier_kernel, dim3((a.half + 255) / 256, op.B), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}
