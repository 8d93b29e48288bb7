"""Chunk-plane layout (CPL) <-> [B, C, L] conversion with torch ops (plumbing for tests, debugging and
callers that want to feed libadk kernels directly; the denoiser itself packs/unpacks in HIP kernels).

CPL: element (b, c, l) at (((c // CB) * B + b) * (L + 2P) + P + l) * CB + c % CB, CB = 16 (f32) / 32 (bf16, f16);
halo rows and pad channels are zero.  See include/adk.h.
"""
import torch


def chunk(dtype):
    return 16 if dtype == torch.float32 else 32


def to_cpl(x, dtype, P):
    B, C, L = x.shape
    cb = chunk(dtype)
    planes = (C + cb - 1) // cb
    buf = torch.zeros(planes, B, L + 2 * P, cb, dtype=dtype, device=x.device)
    xp = torch.zeros(B, planes * cb, L, dtype=x.dtype, device=x.device)
    xp[:, :C] = x
    buf[:, :, P:P + L, :] = xp.view(B, planes, cb, L).permute(1, 0, 3, 2).to(dtype)
    return buf


def from_cpl(buf, C):
    """buf [planes, B, Lp, CB] -> ([B, C, L+2P] fp32 INCLUDING halo rows)."""
    planes, B, Lp, cb = buf.shape
    return buf.permute(1, 0, 3, 2).reshape(B, planes * cb, Lp)[:, :C].float()
