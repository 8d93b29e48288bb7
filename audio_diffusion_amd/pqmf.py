"""Host side of the PQMF filterbank: prototype design (once per model) and the parameter container.

Mirrors the reference interface ``CachedPQMF(a_n_channels, attenuation, n_band)`` with buffers ``hk``,
``h`` and parameters ``forward_conv.weight`` / ``inverse_conv.weight`` (reference diffusion/pqmf.py:182-265)
so reference checkpoints load unchanged.  The transforms themselves run in libadk
(``adk_pqmf_analysis`` / ``adk_pqmf_synthesis``); nothing here computes audio on the CPU.
"""
import math

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F
from scipy import optimize, signal


def design_prototype(attenuation, n_band):
    """Kaiser-window low-pass whose cutoff minimises the reconstruction error of the M-band
    cosine-modulated bank (reference diffusion/pqmf.py:58-92)."""

    def lowpass(wc):
        n, beta = signal.kaiserord(attenuation, wc / np.pi)
        n = 2 * (n // 2) + 1
        # firwin(..., nyq=pi) in old SciPy == fs=2*pi
        return signal.firwin(n, wc, window=("kaiser", beta), scale=False, fs=2 * np.pi)

    def objective(wc):
        h = lowpass(wc)
        g = np.convolve(h, h[::-1], "full")
        return np.max(np.abs(g[g.shape[-1] // 2::2 * n_band][1:]))

    wc = optimize.fmin(objective, 1 / n_band, disp=0)[0]
    return lowpass(wc)


def cosine_modulate(h, n_band):
    """reference diffusion/pqmf.py:46-55, same fp32 operation order."""
    k = torch.arange(n_band).reshape(-1, 1)
    n = h.shape[-1]
    t = torch.arange(-(n // 2), n // 2 + 1)
    phase = (-1) ** k * math.pi / 4
    return 2 * h * torch.cos((2 * k + 1) * math.pi / (2 * n_band) * t + phase)


def _pad_to_pow2(x):
    target = 2 ** math.ceil(math.log2(x.shape[-1]))
    extra = target - x.shape[-1]
    return F.pad(x, (extra // 2, extra // 2 + extra % 2))


def _odd(x):
    return x if x.shape[-1] % 2 else F.pad(x, (0, 1))


class _FilterConv(nn.Module):
    """Holds one filter tensor under the key ``<name>.weight`` (the reference stores its filters in
    cached_conv.Conv1d modules).  ``nn.Conv1d`` is instantiated only to consume the same RNG draws as
    the reference constructor, then overwritten."""

    def __init__(self, weight, stride):
        super().__init__()
        probe = nn.Conv1d(weight.shape[1], weight.shape[0], weight.shape[2], stride=stride, bias=False)
        with torch.no_grad():
            probe.weight.copy_(weight)
        self.weight = probe.weight
        self.stride = stride


class CachedPQMF(nn.Module):
    def __init__(self, a_n_channels, attenuation, n_band):
        super().__init__()
        if n_band < 2 or n_band & (n_band - 1):
            raise AssertionError("when using the polyphase algorithm, n_band must be a power of 2")
        h = torch.from_numpy(design_prototype(attenuation, n_band)).float()
        hk = _pad_to_pow2(cosine_modulate(h, n_band))
        self.register_buffer("hk", hk)
        self.register_buffer("h", h)
        self.n_band = n_band
        self.a_n_channels = a_n_channels
        fwd = _odd(hk).unsqueeze(1)                                              # [M, 1, K]
        inv = hk.flip(-1).reshape(n_band, hk.shape[-1] // n_band, n_band).permute(2, 0, 1)
        inv = _odd(inv.contiguous())                                             # [M, M, K/M + 1]
        self.forward_conv = _FilterConv(fwd, n_band)
        self.inverse_conv = _FilterConv(inv, 1)

    def packed_filters(self, device):
        """fwT [K][M] and iwT [M(c)][Kt][M(m)] fp32, the layouts adk_pqmf_* read coalesced."""
        fw = self.forward_conv.weight.detach().to(device=device, dtype=torch.float32)
        iw = self.inverse_conv.weight.detach().to(device=device, dtype=torch.float32)
        return fw[:, 0, :].t().contiguous(), iw.permute(1, 2, 0).contiguous()

    # ---- the reference's call interface (diffusion/pqmf.py:271-302), executed by libadk ----------------
    def _filters(self, device):
        key = str(device)
        cache = self.__dict__.setdefault("_packed", {})
        ver = (self.forward_conv.weight._version, self.inverse_conv.weight._version)
        if key not in cache or cache[key][0] != ver:
            cache[key] = (ver, self.packed_filters(device))
        return cache[key][1]

    @torch.no_grad()
    def forward(self, x):
        """[B, C, T] -> [B, C*M, T/M]: analysis conv (stride M, zero padding K//2), reverse_half sign
        pattern, stereo folded channel -> batch -> channel (reference pqmf.py:271-283)."""
        from . import _lib
        if not x.is_cuda:
            raise RuntimeError("CachedPQMF (MI355X build) runs on ROCm devices only; there is no CPU path")
        B, C, T = x.shape
        M = self.n_band
        if T % M:
            raise ValueError(f"T={T} is not a multiple of n_band={M}")
        fwT, _ = self._filters(x.device)
        xin = x.detach().to(torch.float32).contiguous()
        y = torch.empty(B, C * M, T // M, device=x.device, dtype=torch.float32)
        lib = _lib.load()
        _lib.check(lib.adk_pqmf_analysis(B, C, T, M, fwT.shape[0], xin.data_ptr(), fwT.data_ptr(), y.data_ptr(),
                                         torch.cuda.current_stream(x.device).cuda_stream), "adk_pqmf_analysis")
        return y

    @torch.no_grad()
    def inverse(self, x):
        """[B, C*M, N] -> [B, C, N*M] (reference pqmf.py:285-302); reconstruction is delayed by M samples."""
        from . import _lib
        if not x.is_cuda:
            raise RuntimeError("CachedPQMF (MI355X build) runs on ROCm devices only; there is no CPU path")
        B, CM, N = x.shape
        M = self.n_band
        if CM % M:
            raise ValueError(f"channel count {CM} is not a multiple of n_band={M}")
        _, iwT = self._filters(x.device)
        xin = x.detach().to(torch.float32).contiguous()
        y = torch.empty(B, CM // M, N * M, device=x.device, dtype=torch.float32)
        lib = _lib.load()
        _lib.check(lib.adk_pqmf_synthesis(B, CM // M, N, M, iwT.shape[1], xin.data_ptr(), iwT.data_ptr(), y.data_ptr(),
                                          torch.cuda.current_stream(x.device).cuda_stream), "adk_pqmf_synthesis")
        return y
