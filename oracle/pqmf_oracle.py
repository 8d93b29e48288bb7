"""CPU restatement of the PQMF filterbank used by the denoiser (TEST ORACLE).

Reference followed: diffusion/pqmf.py
  :16-20   reverse_half           :23-32  center_pad_next_pow_2 / make_odd
  :35-55   get_qmf_bank           :58-92  kaiser_filter / loss_wc / get_prototype
  :95-110  polyphase_forward      :236-302 CachedPQMF (the class the denoiser imports)

Third-party arithmetic NOT in the reference tree: ``cached_conv`` (caillonantoine/cached_conv,
unpinned HEAD, reference setup.py:37) supplies ``Conv1d`` and ``get_padding`` used at
pqmf.py:248-265.  Its published behaviour is restated here:
    get_padding(k, mode="centered") -> (p//2, p//2) with p = (k-1)*dilation + 1, (0,0) for k=1
    Conv1d.forward(x) = F.conv1d(F.pad(x, padding /*zeros*/), weight, stride=stride)
PARITY UNPINNED for sample values at that boundary: the reference's only test
(test/pqmf_test.py:9-26) asserts output shape.  Extra anchors used by tests/: equality with the
reference's own cached_conv-free ``polyphase_forward`` + ``reverse_half`` (importable), and the
round-trip reconstruction (delay M samples).

scipy API drift restated: ``firwin(..., nyq=np.pi)`` (removed in SciPy 1.12) == ``fs=2*np.pi``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import fmin
from scipy.signal import firwin, kaiserord


def _kaiser_lowpass(wc, atten, n_taps=None):
    """pqmf.py:58-73."""
    n_est, beta = kaiserord(atten, wc / np.pi)
    n_est = 2 * (n_est // 2) + 1
    n = n_taps if n_taps is not None else n_est
    return firwin(n, wc, window=("kaiser", beta), scale=False, fs=2 * np.pi)


def _pr_error(wc, atten, bands, n_taps):
    """pqmf.py:76-83."""
    h = _kaiser_lowpass(wc, atten, n_taps)
    g = np.convolve(h, h[::-1], "full")
    g = abs(g[g.shape[-1] // 2::2 * bands][1:])
    return np.max(g)


def prototype(atten, bands, n_taps=None):
    """pqmf.py:86-92: Nelder-Mead over the cutoff, start 1/M."""
    wc = fmin(lambda w: _pr_error(w, atten, bands, n_taps), 1 / bands, disp=0)[0]
    return _kaiser_lowpass(wc, atten, n_taps)


def modulate(h, bands):
    """pqmf.py:35-55 (same op order, fp32)."""
    k = torch.arange(bands).reshape(-1, 1)
    n = h.shape[-1]
    t = torch.arange(-(n // 2), n // 2 + 1)
    phase = (-1) ** k * math.pi / 4
    mod = torch.cos((2 * k + 1) * math.pi / (2 * bands) * t + phase)
    return 2 * h * mod


def pad_pow2(x):
    """pqmf.py:23-26."""
    nxt = 2 ** math.ceil(math.log2(x.shape[-1]))
    pad = nxt - x.shape[-1]
    return F.pad(x, (pad // 2, pad // 2 + int(pad % 2)))


def design(atten, bands):
    """PQMF.__init__ + CachedPQMF.__init__ (pqmf.py:193-210, 237-265).

    Returns dict(h, hk, forward_w [M,1,K], inverse_w [M,M,K/M+1]) as fp32 tensors."""
    h = torch.from_numpy(prototype(atten, bands)).float()
    hk = pad_pow2(modulate(h, bands))
    fw = hk if hk.shape[-1] % 2 else F.pad(hk, (0, 1))             # make_odd :29-32
    fw = fw.unsqueeze(1)
    taps = hk.shape[-1] // bands
    # rearrange(hk.flip(-1), "c (t m) -> m c t", m=bands)
    iw = hk.flip(-1).reshape(bands, taps, bands).permute(2, 0, 1).contiguous()
    iw = iw if iw.shape[-1] % 2 else F.pad(iw, (0, 1))
    return dict(h=h, hk=hk, forward_w=fw.contiguous(), inverse_w=iw.contiguous())


def flip_alternate(x):
    """reverse_half, pqmf.py:16-20: negate odd bands at even frames."""
    sign = torch.ones_like(x)
    sign[..., 1::2, ::2] = -1
    return x * sign


def analysis(x, forward_w, bands):
    """CachedPQMF.forward, pqmf.py:271-283.  x [B,C,T] -> [B, C*M, T/M]."""
    b, c, _ = x.shape
    k = forward_w.shape[-1]
    half = ((k - 1) + 1) // 2 if k > 1 else 0                      # cached_conv.get_padding
    y = x.permute(1, 0, 2).reshape(c * b, 1, -1)                   # cat(split(x,1,1),0)
    y = F.conv1d(F.pad(y, (half, half)), forward_w, stride=bands)
    y = flip_alternate(y)
    n = y.shape[-1]
    return y.reshape(c, b, bands, n).permute(1, 0, 2, 3).reshape(b, c * bands, n)


def synthesis(x, inverse_w, bands):
    """CachedPQMF.inverse, pqmf.py:285-302.  x [B, C*M, N] -> [B, C, N*M]."""
    b, cm, n = x.shape
    c = cm // bands
    k = inverse_w.shape[-1]
    half = ((k - 1) + 1) // 2 if k > 1 else 0
    y = x.reshape(b, c, bands, n).permute(1, 0, 2, 3).reshape(c * b, bands, n)
    y = flip_alternate(y)
    y = F.conv1d(F.pad(y, (half, half)), inverse_w) * bands
    y = y.flip(1).permute(0, 2, 1).reshape(c * b, 1, n * bands)    # interleave phases
    return y.reshape(c, b, n * bands).permute(1, 0, 2).contiguous()


# ---- base (non-cached) PQMF paths, pqmf.py:95-233: CPU cross-check of the cached bank ----------------
def base_forward(x, hk, polyphase=True):
    """PQMF.forward, pqmf.py:212-222 (x [B,1,T], hk [M,K] with K a power of two): polyphase_forward :95-110 or
    classic_forward :140-157, then reverse_half."""
    m, k = hk.shape
    if polyphase:
        b, c, t = x.shape
        xp = x.reshape(b, c, t // m, m).permute(0, 1, 3, 2).reshape(b, c * m, t // m)       # "b c (t m) -> b (c m) t"
        w = hk.reshape(m, k // m, m).permute(0, 2, 1)                                          # "c (t m) -> c m t"
        y = F.conv1d(xp, w, padding=w.shape[-1] // 2)[..., :-1]
    else:
        y = F.conv1d(x, hk.unsqueeze(1), stride=m, padding=k // 2)[..., :-1]
    return flip_alternate(y)


def base_inverse(y, hk, polyphase=True):
    """PQMF.inverse, pqmf.py:224-233: reverse_half, then polyphase_inverse :113-137 or classic_inverse :160-179."""
    m, k = hk.shape
    y = flip_alternate(y)
    if polyphase:
        w = hk.flip(-1).reshape(m, k // m, m).permute(2, 0, 1)                                # "c (t m) -> m c t"
        pad = w.shape[-1] // 2 + 1
        z = F.conv1d(y, w, padding=pad)[..., :-1] * m
        z = z.flip(1)
        b, cm, t = z.shape
        z = z.reshape(b, cm // m, m, t).permute(0, 1, 3, 2).reshape(b, cm // m, t * m)         # "b (c m) t -> b c (t m)"
        return z[..., 2 * w.shape[1]:]
    up = torch.zeros(*y.shape[:2], m * y.shape[-1]).to(y)
    up[..., ::m] = y * m
    return F.conv1d(up, hk.flip(-1).unsqueeze(0), padding=k // 2)[..., 1:]
