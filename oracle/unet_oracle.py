"""Functional CPU restatement of the DiffusionAttnUnet1D denoiser (TEST ORACLE).

Operates on a plain ``state_dict`` (reference key names) plus a config dict; no
nn.Module tree.  Plain PyTorch CPU fp32 ops only.  See ``oracle/__init__.py`` for who
may import this.

Reference followed (paths relative to the reference checkout):
  decoders/diffusion_decoder.py:56-176   DiffusionAttnUnet1D ctor/forward
  blocks/blocks.py:8-15, 66-76           ResidualBlock / ResConvBlock
  blocks/blocks.py:103-122               SelfAttention1d
  blocks/blocks.py:146-152               SkipBlock (cat([main(x), x]))
  blocks/blocks.py:154-167               FourierFeatures / expand_to_planes
  blocks/blocks.py:169-212               _kernels, Downsample1d, Upsample1d
"""
import math

import torch
import torch.nn.functional as F

from . import pqmf_oracle

DEFAULT_CONFIG = dict(
    io_channels=2, depth=14, n_attn_layers=6,
    c_mults=[128, 128, 256, 256] + [512] * 10,
    cond_dim=0, cond_noise_aug=False, pqmf_bands=1, kernel_size=5,
    learned_resample=False, strides=[2] * 14, conv_bias=True,
)

# blocks/blocks.py:169-180
RESAMPLE_KERNELS = {
    "linear": [1 / 8, 3 / 8, 3 / 8, 1 / 8],
    "cubic": [-0.01171875, -0.03515625, 0.11328125, 0.43359375,
              0.43359375, 0.11328125, -0.03515625, -0.01171875],
    "lanczos3": [0.003689131001010537, 0.015056144446134567, -0.03399861603975296,
                 -0.066637322306633, 0.13550527393817902, 0.44638532400131226,
                 0.44638532400131226, 0.13550527393817902, -0.066637322306633,
                 -0.03399861603975296, 0.015056144446134567, 0.003689131001010537],
}


def make_config(**kw):
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(kw)
    return cfg


def fourier_features(weight, t):
    """blocks/blocks.py:161-163.  weight [F/2, 1], t [B] -> [B, F]."""
    f = 2 * math.pi * t[:, None] @ weight.T
    return torch.cat([f.cos(), f.sin()], dim=-1)


def downsample_fir(x, taps):
    """blocks/blocks.py:191-196 restated as a depthwise conv (the reference builds a dense
    diagonal [C,C,K] weight; off-diagonal zeros contribute exact zeros)."""
    taps = torch.as_tensor(taps, dtype=x.dtype)
    k = taps.numel()
    pad = k // 2 - 1
    c = x.shape[1]
    xp = F.pad(x, (pad, pad), mode="reflect")
    return F.conv1d(xp, taps.view(1, 1, k).expand(c, 1, k).contiguous(), stride=2, groups=c)


def upsample_fir(x, taps2):
    """blocks/blocks.py:207-212; ``taps2`` is the stored buffer (= 2 * table)."""
    taps2 = torch.as_tensor(taps2, dtype=x.dtype)
    k = taps2.numel()
    pad = k // 2 - 1
    c = x.shape[1]
    xp = F.pad(x, ((pad + 1) // 2,) * 2, mode="reflect")
    return F.conv_transpose1d(xp, taps2.view(1, 1, k).expand(c, 1, k).contiguous(),
                              stride=2, padding=pad * 2 + 1, groups=c)


def self_attention(sd, p, x):
    """blocks/blocks.py:113-122 (dropout p=0)."""
    n, c, s = x.shape
    heads = c // 32                                   # diffusion_decoder.py:100-101
    h = F.group_norm(x, 1, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    qkv = F.conv1d(h, sd[p + "qkv_proj.weight"], sd[p + "qkv_proj.bias"])
    qkv = qkv.view(n, heads * 3, c // heads, s).transpose(2, 3)
    q, k, v = qkv.chunk(3, dim=1)
    scale = k.shape[3] ** -0.25
    att = ((q * scale) @ (k.transpose(2, 3) * scale)).softmax(3)
    y = (att @ v).transpose(2, 3).contiguous().view(n, c, s)
    return x + F.conv1d(y, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def res_conv_block(sd, p, x, is_last=False):
    """blocks/blocks.py:66-76 + 14-15."""
    w0 = sd[p + "main.0.weight"]
    pad = w0.shape[2] // 2
    h = F.conv1d(x, w0, sd.get(p + "main.0.bias"), padding=pad)
    h = F.gelu(F.group_norm(h, 1, sd[p + "main.1.weight"], sd[p + "main.1.bias"], 1e-5))
    h = F.conv1d(h, sd[p + "main.3.weight"], sd.get(p + "main.3.bias"), padding=pad)
    if not is_last:
        h = F.gelu(F.group_norm(h, 1, sd[p + "main.4.weight"], sd[p + "main.4.bias"], 1e-5))
    skip_w = sd.get(p + "skip.weight")
    return h + (x if skip_w is None else F.conv1d(x, skip_w))


def attn_levels(cfg):
    """Levels that carry SelfAttention1d (diffusion_decoder.py:85,96): i >= depth-n-1,
    level 1 never (its branch :122-132 has none)."""
    d, n = cfg["depth"], cfg["n_attn_layers"]
    return [i for i in range(2, d + 1) if n > 0 and i >= d - n - 1]


def _level(sd, cfg, p, i, x, taps):
    """One SkipBlock at level i>1 (diffusion_decoder.py:97-121); p = prefix of its .main"""
    depth = cfg["depth"]
    attn = i in attn_levels(cfg)
    learned = cfg["learned_resample"]
    f = cfg["strides"][i - 1]
    if learned:     # Downsample1d_2 (blocks/blocks.py:214-226): Conv1d(k = 2f+1, stride f, padding f)
        h = F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=f, padding=f)
    else:
        h = downsample_fir(x, sd[p + "0.kernel"])
    for j in (1, 3, 5):
        h = res_conv_block(sd, f"{p}{j}.", h)
        if attn:
            h = self_attention(sd, f"{p}{j + 1}.", h)
    if i < depth:
        h = _level(sd, cfg, f"{p}7.main.", i + 1, h, taps)
    for j in (8, 10, 12):
        h = res_conv_block(sd, f"{p}{j}.", h)
        if attn:
            h = self_attention(sd, f"{p}{j + 1}.", h)
    if not learned:
        h = upsample_fir(h, sd[p + "14.kernel"])
    elif f == 1:    # Upsample1d_2 (blocks/blocks.py:229-255): factor 1 is a plain k=3 conv
        h = F.conv1d(h, sd[p + "14.weight"], sd[p + "14.bias"], padding=1)
    else:           # ConvTranspose1d(k = 2f, stride f, padding f//2 + f%2, output_padding f%2)
        h = F.conv_transpose1d(h, sd[p + "14.weight"], sd[p + "14.bias"], stride=f,
                               padding=f // 2 + f % 2, output_padding=f % 2)
    return torch.cat([h, x], dim=1)                    # blocks.py:151-152: main first


def net_forward(sd, cfg, h):
    """self.net(...) (diffusion_decoder.py:124-132)."""
    for j in (0, 1, 2):
        h = res_conv_block(sd, f"net.{j}.", h)
    if cfg["depth"] > 1:
        h = _level(sd, cfg, "net.3.main.", 2, h, None)
    for j in (4, 5):
        h = res_conv_block(sd, f"net.{j}.", h)
    return res_conv_block(sd, "net.6.", h, is_last=True)


def unet_forward(sd, cfg, x, t, cond=None, cond_aug_scale=None):
    """DiffusionAttnUnet1D.forward (diffusion_decoder.py:139-176).  With cond_noise_aug the level must be given
    (``cond_aug_scale``; the reference's Sobol draw at :155-156 is random) and the noise is torch's CPU randn, so
    only cond_aug_scale = 0 is bit-reproducible."""
    bands = cfg["pqmf_bands"]
    if bands > 1:
        x = pqmf_oracle.analysis(x, sd["pqmf.forward_conv.weight"], bands)
    emb = fourier_features(sd["timestep_embed.weight"], t)
    planes = emb[..., None].repeat([1, 1, x.shape[2]])   # blocks.py:166-167
    inputs = [x, planes]
    if cond is not None:
        if cond.shape[2] != x.shape[2]:
            cond = F.interpolate(cond, (x.shape[2],), mode="linear", align_corners=False)
        if cfg["cond_noise_aug"]:                              # :153-167
            if cond_aug_scale is None:
                raise NotImplementedError("oracle needs an explicit cond_aug_scale (the reference draws it from Sobol)")
            aug = torch.tensor([cond_aug_scale]).repeat([cond.shape[0]]).to(cond)
            cond = cond + torch.randn_like(cond) * aug[:, None, None]
            aug_emb = fourier_features(sd["timestep_embed.weight"], aug)
            inputs.append(aug_emb[..., None].repeat([1, 1, x.shape[2]]))
        inputs.append(cond)
    out = net_forward(sd, cfg, torch.cat(inputs, dim=1))
    if bands > 1:
        out = pqmf_oracle.synthesis(out, sd["pqmf.inverse_conv.weight"], bands)
    return out


def make_model_fn(sd, cfg):
    """Return ``model(x, t, **extra)`` closed over a state_dict (all tensors CPU fp32)."""
    sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}

    @torch.no_grad()
    def model(x, t, **extra):
        return unet_forward(sd, cfg, x, t, **extra)

    return model
