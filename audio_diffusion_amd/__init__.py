"""audio_diffusion_amd — MI355X-native sampling hot path of zqevans/audio-diffusion.

Public surface (same names as the reference):
    DiffusionAttnUnet1D                      decoders/diffusion_decoder.py:56-176
    sample, get_alphas_sigmas, alpha_sigma_to_t    diffusion/sampling.py:6-55
    sample_crash, get_crash_schedule         train_uncond.py:35-38,45-86 (the training demos' timestep warp)
    sample_diffae                            diffusion/inference.py:14-57 (encode / decode DDIM variant)
    audio_to_int16                           train_uncond.py:182-187 (rearrange + clamp + int16 output stage)
    CachedPQMF                               diffusion/pqmf.py:236-302 (design + forward/inverse on libadk)
    AudioDenoiserModel, Denoiser             decoders/diffusion_decoder.py:261-337 (k-diffusion denoiser family)
    get_sigmas_karras, get_sigmas_exponential, sample_heun, sample_lms     diffusion/inference.py:60-73, 94-154
Compute runs exclusively in libadk.so (hand-written gfx950 HIP kernels, C ABI in include/adk.h).
"""
from .unet import DiffusionAttnUnet1D
from .sampling import (sample, sample_crash, sample_diffae, audio_to_int16, get_alphas_sigmas, alpha_sigma_to_t,
                       get_crash_schedule, step_coefficients, get_sigmas_karras, get_sigmas_exponential, sample_heun,
                       sample_lms, to_d)
from .pqmf import CachedPQMF
from .kdiffusion import AudioDenoiserModel, Denoiser
from . import _lib, blocks, distributed

__all__ = ["DiffusionAttnUnet1D", "sample", "sample_crash", "sample_diffae", "audio_to_int16", "get_alphas_sigmas",
           "alpha_sigma_to_t", "get_crash_schedule", "step_coefficients", "CachedPQMF", "AudioDenoiserModel", "Denoiser", "get_sigmas_karras", "get_sigmas_exponential", "sample_heun", "sample_lms", "to_d"]
