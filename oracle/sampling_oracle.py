"""CPU restatement of the v-objective DDIM sampler (TEST ORACLE).

Reference followed: diffusion/sampling.py
  :6-9    get_alphas_sigmas      :11-14  alpha_sigma_to_t      :16-55  sample
The literal operation order of :34-48 is kept so fp32 results are bit-comparable with the
reference on CPU.  (torch.cuda.amp.autocast at :30 is a no-op for CPU tensors.)
"""
import math

import torch


def crash_schedule(t):
    """train_uncond.py:35-38 (get_crash_schedule): sigma = sin(t pi/2)^2, alpha = sqrt(1 - sigma^2), back to t.
    Pinned by tests/golden/glue.npz (crash/*): tools/make_goldens.py compiles just that script's function definitions
    (ast; the script itself needs prefigure / wandb / aeiou and is never imported) and records their outputs."""
    sigma = torch.sin(t * math.pi / 2) ** 2
    alpha = (1 - sigma ** 2) ** 0.5
    return alpha_sigma_to_t(alpha, sigma)


def schedule(steps, crash=False):
    """sampling.py:22-24: t = linspace(1,0,steps+1)[:-1]; alpha=cos(t*pi/2); sigma=sin(t*pi/2).
    crash=True applies train_uncond.py:53 (t = get_crash_schedule(t)) first."""
    t = torch.linspace(1, 0, steps + 1)[:-1]
    if crash:
        t = crash_schedule(t)
    return t, torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def alpha_sigma_to_t(alpha, sigma):
    """sampling.py:11-14."""
    return torch.atan2(sigma, alpha) / math.pi * 2


@torch.no_grad()
def ddim_sample(model, x, steps, eta, trace=None, crash=False, **extra):
    """sampling.py:16-55.  ``trace`` (optional list) receives each step's v.  crash=True: train_uncond.py:45-86."""
    ones = x.new_ones([x.shape[0]])
    t, alphas, sigmas = schedule(steps, crash)
    pred = None
    for i in range(steps):
        v = model(x, ones * t[i], **extra).float()
        if trace is not None:
            trace.append(v.clone())
        pred = x * alphas[i] - v * sigmas[i]
        eps = x * sigmas[i] + v * alphas[i]
        if i < steps - 1:
            ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * \
                (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
            adjusted = (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()
            x = pred * alphas[i + 1] + eps * adjusted
            if eta:
                x = x + torch.randn_like(x) * ddim_sigma
    return pred


@torch.no_grad()
def diffae_sample(model, signal, steps, eta, noise):
    """diffusion/inference.py:14-57: z_sem = model.encode(signal) once, start from ``noise`` (the reference draws
    randn_like(signal) at :26; passed in here so a fixture can pin it), then the loop of :29-54 on model.decode."""
    z_sem = model.encode(signal)
    return ddim_sample(lambda x, t: model.decode(x, t, z_sem), noise, steps, eta)
