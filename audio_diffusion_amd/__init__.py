"""audio_diffusion_amd — MI355X-native sampling hot path of zqevans/audio-diffusion.

Public surface (same names as the reference):
    DiffusionAttnUnet1D                      decoders/diffusion_decoder.py:56-176
    sample, get_alphas_sigmas, alpha_sigma_to_t    diffusion/sampling.py:6-55
    CachedPQMF                               diffusion/pqmf.py:236-302 (design + forward/inverse on libadk)
Compute runs exclusively in libadk.so (hand-written gfx950 HIP kernels, C ABI in include/adk.h).
"""
from .unet import DiffusionAttnUnet1D
from .sampling import sample, get_alphas_sigmas, alpha_sigma_to_t, step_coefficients
from .pqmf import CachedPQMF
from . import _lib

__all__ = ["DiffusionAttnUnet1D", "sample", "get_alphas_sigmas", "alpha_sigma_to_t", "step_coefficients",
           "CachedPQMF"]
