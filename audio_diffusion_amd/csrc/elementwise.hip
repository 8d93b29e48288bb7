// Boundary + sampler elementwise kernels (HBM-bound, vectorized).
//   pack_input   : FourierFeatures(t) -> expand_to_planes -> cat([x, emb]) -> CPL   (reference
//                  blocks/blocks.py:161-167, decoders/diffusion_decoder.py:144-171)
//   unpack_output: CPL f32 -> [B,C,L] f32
//   ddim_update  : literal-order v-DDIM step (reference diffusion/sampling.py:34-52)
//   step_advance : ++*step, so one captured hipGraph replays every step of the loop
#include "adk_common.h"

// The DDIM update must round every product like the reference's separate ATen mul/add kernels do:
// no fused multiply-add anywhere in this file (hipcc contracts by default).
#pragma clang fp contract(off)

struct PackArgs {
    const float* x; const float* t; const int32_t* step; const float* fw; void* y;
    const float* aug; const float* cond;          // optional: noise-aug levels [B]; conditioning [B, Cc, Lc]
    const float* noise;                            // optional: N(0,1) draw [B, Cc, L] added to cond scaled by aug[b]
    int B, L, P, Lp, Cx, nplanes, t_stride, Cc, Lc; long long M;
    float cscale;                                  // Lc / L, rounded to f32 like ATen's area_pixel_compute_scale
    int nemb;                                      // embedding channels: 16, 32 (noise-aug), or 0 (AudioDenoiserModel)
    const float* xscale;                           // optional f32 [B]: x * c_in[b] (Karras preconditioner, Denoiser.forward)
};

// Channel order of the network input (decoders/diffusion_decoder.py:146-171):
//   [ x (Cx) | timestep embedding (16) | noise-aug level embedding (16, only with cond_noise_aug) | cond (Cc) ]
// cond is resampled to L with F.interpolate(mode='linear', align_corners=False) semantics (:150-151):
//   src = scale * (l + 0.5) - 0.5, clamped at 0; out = (1 - w) * in[i0] + w * in[min(i0 + 1, Lc - 1)].
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const PackArgs a)
{
    constexpr int NV = ElemTraits<T>::NV, CB = ElemTraits<T>::CB;
    __shared__ float emb[32];
    const int plane = blockIdx.y, b = blockIdx.z;
    if (threadIdx.x < 32 && a.nemb) {
        const bool second = threadIdx.x >= 16;
        float tv = 0.f;
        if (!second) tv = a.step ? a.t[(long long)(*a.step) * a.t_stride] : a.t[b];
        else if (a.aug) tv = a.aug[b];
        const int j = threadIdx.x & 7;
        // 2 * math.pi * input @ weight.T : ((2*pi) * t) * w, each product rounded to f32
        const float tp = 6.283185307179586f * tv;
        const float f = tp * a.fw[j];
        emb[threadIdx.x] = (threadIdx.x & 15) < 8 ? cosf(f) : sinf(f);
    }
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;          // row within the sample, halo included
    if (r >= a.Lp) return;
    const int l = r - a.P;
    const bool valid = l >= 0 && l < a.L;
    const int c_cond = a.Cx + a.nemb;
    const float xs = a.xscale ? a.xscale[b] : 1.f;
    int i0 = 0, i1 = 0;
    float w0 = 1.f, w1 = 0.f;
    if (a.cond && valid) {
        if (a.Lc == a.L) { i0 = i1 = l; }
        else {
            float src = a.cscale * ((float)l + 0.5f) - 0.5f;
            src = src < 0.f ? 0.f : src;
            i0 = (int)src;
            i0 = i0 > a.Lc - 1 ? a.Lc - 1 : i0;
            i1 = i0 < a.Lc - 1 ? i0 + 1 : i0;
            w1 = src - (float)i0;
            w0 = 1.f - w1;
        }
    }
    unsigned char* yrow = reinterpret_cast<unsigned char*>(a.y) + ((long long)plane * a.M + (long long)b * a.Lp + r) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float f[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int c = plane * CB + q * NV + e;
            float v = 0.f;
            if (valid) {
                if (c < a.Cx) { v = a.x[((long long)b * a.Cx + c) * a.L + l]; if (a.xscale) v = v * xs; }
                else if (c < c_cond) v = emb[c - a.Cx];
                else if (c < c_cond + a.Cc) {
                    const float* cp = a.cond + ((long long)b * a.Cc + (c - c_cond)) * a.Lc;
                    v = a.Lc == a.L ? cp[i0] : w0 * cp[i0] + w1 * cp[i1];
                    if (a.noise) v = v + a.noise[((long long)b * a.Cc + (c - c_cond)) * a.L + l] * a.aug[b];   // :160
                }
            }
            f[e] = v;
        }
        *reinterpret_cast<uint4*>(yrow + q * 16) = float_to_piece(f, T());
    }
}

int launch_pack_input(const adk_op& op, hipStream_t st)
{
    PackArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.t = reinterpret_cast<const float*>(op.aux0);
    a.step = reinterpret_cast<const int32_t*>(op.bias); a.fw = reinterpret_cast<const float*>(op.w); a.y = op.y;
    a.aug = op.gamma; a.cond = reinterpret_cast<const float*>(op.residual);
    a.noise = op.gamma ? reinterpret_cast<const float*>(op.aux1) : nullptr;
    a.Cc = a.cond ? op.Cout : 0; a.Lc = a.cond ? op.i1 : 0;
    a.cscale = a.cond ? (float)a.Lc / (float)op.L : 0.f;
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.Cx = op.Cin; a.t_stride = op.i0;
    a.M = (long long)op.B * a.Lp;
    a.nemb = (op.flags & ADK_FLAG_NO_EMBED) ? 0 : (a.aug ? 32 : 16);
    a.xscale = op.beta;
    const int Cp = (op.Cin + a.nemb + a.Cc + 31) / 32 * 32;
    const int CB = op.dtype == ADK_F32 ? 16 : 32;
    a.nplanes = Cp / CB;
    dim3 grid((a.Lp + 255) / 256, a.nplanes, op.B);
    if (op.dtype == ADK_F32) hipLaunchKernelGGL(pack_input_kernel<float>, grid, dim3(256), 0, st, a);
    else if (op.dtype == ADK_F16) hipLaunchKernelGGL(pack_input_kernel<f16_t>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pack_input_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

struct UnpackArgs { const float* x; float* y; int B, L, P, Lp, C; long long M;
                    const float* cout; const float* cskip; const float* xin; };   // optional: y = v * c_out[b] + xin * c_skip[b]

__global__ __launch_bounds__(256) void unpack_output_kernel(const UnpackArgs a)
{
    const int l = blockIdx.x * 256 + threadIdx.x, plane = blockIdx.y, b = blockIdx.z;
    if (l >= a.L) return;
    const float* row = a.x + ((long long)plane * a.M + (long long)b * a.Lp + a.P + l) * 16;
    float f[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(row)[q];
        f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = plane * 16 + e;
        if (c < a.C) {
            const long long o = ((long long)b * a.C + c) * a.L + l;
            float v = f[e];
            if (a.cout) {                            // Denoiser.forward (decoders/diffusion_decoder.py:335-337), literal order
                const float vo = v * a.cout[b], xs = a.xin[o] * a.cskip[b];
                v = vo + xs;
            }
            a.y[o] = v;
        }
    }
}

int launch_unpack_output(const adk_op& op, hipStream_t st)
{
    UnpackArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.y = reinterpret_cast<float*>(op.y);
    a.B = op.B; a.L = op.L; a.P = op.P; a.Lp = op.L + 2 * op.P; a.C = op.Cin; a.M = (long long)op.B * a.Lp;
    a.cout = op.gamma; a.cskip = op.beta; a.xin = reinterpret_cast<const float*>(op.residual);
    dim3 grid((op.L + 255) / 256, (op.Cin + 15) / 16, op.B);
    hipLaunchKernelGGL(unpack_output_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

// ---- v-DDIM step -------------------------------------------------------------------------------
struct DdimArgs { float* x; const float* v; const float* noise; float* pred; const float* coef; const int32_t* step; long long n4, n; };

template <bool NOISE>
__global__ __launch_bounds__(256) void ddim_update_kernel(const DdimArgs a)
{
    const float* c = a.coef + (long long)(a.step ? *a.step : 0) * 8;
    const float alpha = c[1], sigma = c[2], alpha_n = c[3], adj = c[4], dsig = c[5];
    const bool last = c[6] != 0.f;
    auto one = [&](float x, float v, float nz, float& pred) -> float {
        // literal order of sampling.py:34-35,48,52; no FMA contraction so CPU fp32 matches bitwise
        // (plain operators: contraction is off in this file; the __f*_rn intrinsics would inline with
        // the header's contract flag and fuse)
        const float xa = x * alpha, vs = v * sigma;
        pred = xa - vs;
        const float xsg = x * sigma, va = v * alpha;
        const float eps = xsg + va;
        if (last) return x;
        const float pa = pred * alpha_n, ea = eps * adj;
        float xn = pa + ea;
        if (NOISE) { const float nd = nz * dsig; xn = xn + nd; }
        return xn;
    };
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(a.x)[i];
        const float4 v = reinterpret_cast<const float4*>(a.v)[i];
        float4 nz = make_float4(0, 0, 0, 0);
        if (NOISE) nz = reinterpret_cast<const float4*>(a.noise)[i];
        float4 p, xn;
        xn.x = one(x.x, v.x, nz.x, p.x); xn.y = one(x.y, v.y, nz.y, p.y);
        xn.z = one(x.z, v.z, nz.z, p.z); xn.w = one(x.w, v.w, nz.w, p.w);
        reinterpret_cast<float4*>(a.pred)[i] = p;
        reinterpret_cast<float4*>(a.x)[i] = xn;
    }
    // tail (n not a multiple of 4)
    for (long long i = a.n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        float p;
        const float xn = one(a.x[i], a.v[i], NOISE ? a.noise[i] : 0.f, p);
        a.pred[i] = p; a.x[i] = xn;
    }
}

int launch_ddim_update(const adk_op& op, hipStream_t st)
{
    DdimArgs a;
    a.x = reinterpret_cast<float*>(op.y); a.v = reinterpret_cast<const float*>(op.x);
    a.noise = reinterpret_cast<const float*>(op.residual); a.pred = reinterpret_cast<float*>(op.aux1);
    a.coef = reinterpret_cast<const float*>(op.aux0); a.step = reinterpret_cast<const int32_t*>(op.bias);
    a.n = ((long long)op.i1 << 31) | (long long)(uint32_t)op.i0;
    a.n4 = a.n / 4;
    long long blocks = (a.n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (op.flags & ADK_FLAG_NOISE) hipLaunchKernelGGL(ddim_update_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(ddim_update_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

__global__ void step_advance_kernel(int32_t* step) { if (threadIdx.x == 0) step[0] = step[0] + 1; }

int launch_step_advance(const adk_op& op, hipStream_t st)
{
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<int32_t*>(op.y));
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}

// ---- demo output stage: rearrange(b d n -> d (b n)).clamp(-1, 1).mul(32767).to(int16) ---------------------
// (reference train_uncond.py:182-187).  x f32 [B, C, T] -> y int16 [C, B*T]; truncation toward zero like torch.
struct PcmArgs { const float* x; short* y; int B, C, T; };

__global__ __launch_bounds__(256) void audio_to_int16_kernel(const PcmArgs a)
{
    const int bc = blockIdx.y, b = bc / a.C, c = bc % a.C;
    const float* xin = a.x + (long long)bc * a.T;
    short* yout = a.y + ((long long)c * a.B + b) * a.T;
    const int n4 = ((reinterpret_cast<uintptr_t>(xin) & 15) == 0 && (reinterpret_cast<uintptr_t>(yout) & 7) == 0) ? a.T / 4 : 0;
    auto cv = [](float v) -> int { v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v); return (int)(v * 32767.f); };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(xin)[i];
        const unsigned lo = (unsigned)(cv(v.x) & 0xffff) | ((unsigned)cv(v.y) << 16);
        const unsigned hi = (unsigned)(cv(v.z) & 0xffff) | ((unsigned)cv(v.w) << 16);
        reinterpret_cast<uint2*>(yout)[i] = make_uint2(lo, hi);
    }
    for (int i = n4 * 4 + blockIdx.x * 256 + threadIdx.x; i < a.T; i += gridDim.x * 256) yout[i] = (short)cv(xin[i]);
}

int launch_audio_to_int16(const adk_op& op, hipStream_t st)
{
    PcmArgs a;
    a.x = reinterpret_cast<const float*>(op.x); a.y = reinterpret_cast<short*>(op.y);
    a.B = op.B; a.C = op.Cin; a.T = op.L;
    int bx = (op.L / 4 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(audio_to_int16_kernel, dim3(bx, op.B * op.Cin), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ADK_OK : ADK_ELAUNCH;
}
