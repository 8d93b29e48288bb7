// EXT = true instantiations of the convolution kernels (conv_mfma_kernels.h): causal / dilated taps, ELU epilogue,
// per-group GroupNorm statistics.  Users: the k-diffusion denoiser family (AdaGN statistics; taps 1, 3) and the
// SoundStream-XL decoder (CausalConv1d k = 7, dilation 1 / 3 / 9, k = 3 folded transposed convs, ELU; taps 1, 3, 7).
#include "conv_mfma_kernels.h"

template <typename T, typename TO>
static int launch_taps_ext(const adk_op& op, hipStream_t st)
{
    if (op.taps == 1) return launch_tiles<true, T, TO, 1>(op, st);
    if (op.taps == 3) return launch_tiles<true, T, TO, 3>(op, st);
    if (op.taps == 7) {
        if (op.i0 == 3) return launch_tiles<true, T, TO, 7, 3>(op, st);
        if (op.i0 == 9) return launch_tiles<true, T, TO, 7, 9>(op, st);
        return launch_tiles<true, T, TO, 7>(op, st);
    }
    return ADK_EINVAL;
}

int launch_conv_ext(const adk_op& op, hipStream_t st)
{
    if (op.dtype == ADK_F32) return launch_taps_ext<float, float>(op, st);
    if (op.dtype == ADK_F16)
        return (op.flags & ADK_FLAG_OUT_F32) ? launch_taps_ext<f16_t, float>(op, st) : launch_taps_ext<f16_t, f16_t>(op, st);
    if (op.flags & ADK_FLAG_OUT_F32) return launch_taps_ext<bf16_t, float>(op, st);
    return launch_taps_ext<bf16_t, bf16_t>(op, st);
}
