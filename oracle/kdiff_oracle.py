"""Functional CPU restatement of the k-diffusion denoiser family (TEST ORACLE): AudioDenoiserModel + Denoiser.

Operates on a plain ``state_dict`` (reference key names) plus a config dict; plain PyTorch CPU fp32 ops only.
See ``oracle/__init__.py`` for who may import this.  Pinned by tests/test_oracle_golden.py against fixtures
generated from the reference itself (tools/make_goldens.py tier K).

Reference followed (paths relative to the reference checkout):
  decoders/diffusion_decoder.py:261-312   AudioDenoiserModel ctor/forward
  decoders/diffusion_decoder.py:314-337   Denoiser.get_scalings / forward
  blocks/blocks.py:53-64                  AdaGN
  blocks/blocks.py:89-101, 42-50          ResModConvBlock / ConditionedResidualBlock
  blocks/blocks.py:124-144                SelfAttentionMod1d
  blocks/blocks.py:214-255                Downsample1d_2 / Upsample1d_2
  blocks/blocks.py:260-316                DBlock / UBlock / MappingNet / UNet
"""
import math

import torch
import torch.nn.functional as F


def make_config(c_in, feats_in, depths, channels, self_attn_depths, strides, mapping_cond_dim=0, unet_cond_dim=0):
    return dict(c_in=c_in, feats_in=feats_in, depths=list(depths), channels=list(channels),
                self_attn_depths=list(self_attn_depths), strides=list(strides), mapping_cond_dim=mapping_cond_dim,
                unet_cond_dim=unet_cond_dim)


def ada_gn(sd, p, x, cond, groups):
    """blocks/blocks.py:61-64."""
    wb = F.linear(cond, sd[p + "mapper.weight"], sd[p + "mapper.bias"])
    weight, bias = wb.chunk(2, dim=-1)
    h = F.group_norm(x, groups, eps=1e-5)
    return torch.addcmul(bias[..., None], h, weight[..., None] + 1)


def res_mod_conv_block(sd, p, x, cond, group_size=32):
    """blocks/blocks.py:89-101 (dropout 0) + :48-50."""
    c_in = x.shape[1]
    h = F.gelu(ada_gn(sd, p + "main.0.", x, cond, max(1, c_in // group_size)))
    h = F.conv1d(h, sd[p + "main.2.weight"], sd[p + "main.2.bias"], padding=1)
    h = F.gelu(ada_gn(sd, p + "main.4.", h, cond, max(1, h.shape[1] // group_size)))
    h = F.conv1d(h, sd[p + "main.6.weight"], sd[p + "main.6.bias"], padding=1)
    skip_w = sd.get(p + "skip.weight")
    return h + (x if skip_w is None else F.conv1d(x, skip_w))


def self_attention_mod(sd, p, x, cond, group_size=32, head_size=64):
    """blocks/blocks.py:134-144; n_head = max(1, c // head_size) (:269)."""
    n, c, s = x.shape
    heads = max(1, c // head_size)
    qkv = F.conv1d(ada_gn(sd, p + "norm_in.", x, cond, max(1, c // group_size)),
                   sd[p + "qkv_proj.weight"], sd[p + "qkv_proj.bias"])
    qkv = qkv.view(n, heads * 3, c // heads, s).transpose(2, 3)
    q, k, v = qkv.chunk(3, dim=1)
    scale = k.shape[3] ** -0.25
    att = ((q * scale) @ (k.transpose(2, 3) * scale)).softmax(3)
    y = (att @ v).transpose(2, 3).contiguous().view(n, c, s)
    return x + F.conv1d(y, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _layers(sd, p, first, n_layers, attn, x, cond):
    k = first
    for _ in range(n_layers):
        x = res_mod_conv_block(sd, f"{p}{k}.", x, cond)
        k += 1
        if attn:
            x = self_attention_mod(sd, f"{p}{k}.", x, cond)
            k += 1
    return x, k


def d_block(sd, p, x, cond, n_layers, ratio, attn):
    """blocks/blocks.py:260-270."""
    k = 0
    if ratio > 1:                       # Downsample1d_2: Conv1d(k = 2f+1, stride f, padding f)
        x = F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=ratio, padding=ratio)
        k = 1
    return _layers(sd, p, k, n_layers, attn, x, cond)[0]


def u_block(sd, p, x, cond, skip, n_layers, ratio, attn):
    """blocks/blocks.py:273-290."""
    if skip is not None:
        x = torch.cat([x, skip], dim=1)
    x, k = _layers(sd, p, 0, n_layers, attn, x, cond)
    if ratio > 1:                       # Upsample1d_2: ConvTranspose1d(k = 2f, stride f, padding f//2 + f%2, output_padding f%2)
        x = F.conv_transpose1d(x, sd[f"{p}{k}.weight"], sd[f"{p}{k}.bias"], stride=ratio,
                               padding=ratio // 2 + ratio % 2, output_padding=ratio % 2)
    return x


def denoiser_model_forward(sd, cfg, x, sigma, mapping_cond=None, unet_cond=None, log_sigma=True):
    """AudioDenoiserModel.forward (decoders/diffusion_decoder.py:296-312)."""
    if log_sigma:
        sigma = sigma.log() / 4
    f = 2 * math.pi * sigma[:, None] @ sd["timestep_embed.weight"].T          # blocks/blocks.py:161-163
    emb = torch.cat([f.cos(), f.sin()], dim=-1)
    if mapping_cond is not None:
        emb = emb + F.linear(mapping_cond, sd["mapping_cond.weight"])
    cond = F.gelu(F.linear(emb, sd["mapping.0.weight"], sd["mapping.0.bias"]))
    cond = F.gelu(F.linear(cond, sd["mapping.2.weight"], sd["mapping.2.bias"]))
    if unet_cond is not None:
        unet_cond = F.interpolate(unet_cond, (x.shape[2],), mode="linear", align_corners=False)
        x = torch.cat([x, unet_cond], dim=1)
    x = F.conv1d(x, sd["proj_in.weight"], sd["proj_in.bias"])
    n = len(cfg["depths"])
    skips = []
    for i in range(n):                                                          # blocks/blocks.py:309-316
        ratio = 1 if i == 0 else cfg["strides"][i - 1]
        x = d_block(sd, f"u_net.d_blocks.{i}.", x, cond, cfg["depths"][i], ratio, cfg["self_attn_depths"][i])
        skips.append(x)
    for i in range(n):
        j = n - 1 - i
        ratio = 1 if j == 0 else cfg["strides"][j - 1]
        x = u_block(sd, f"u_net.u_blocks.{i}.", x, cond, skips[j] if i > 0 else None, cfg["depths"][j], ratio,
                    cfg["self_attn_depths"][j])
    return F.conv1d(x, sd["proj_out.weight"], sd["proj_out.bias"])


def karras_scalings(sigma, sigma_data):
    """Denoiser.get_scalings (decoders/diffusion_decoder.py:322-326)."""
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2) ** 0.5
    c_in = 1 / (sigma ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out, c_in


def make_model_fn(sd, cfg, sigma_data=None):
    """``model(x, sigma, **extra)`` closed over a state_dict; with ``sigma_data`` the Denoiser wrapper (:335-337)."""
    sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}

    @torch.no_grad()
    def model(x, sigma, **extra):
        if sigma_data is None:
            return denoiser_model_forward(sd, cfg, x, sigma, **extra)
        c_skip, c_out, c_in = [c[:, None, None] for c in karras_scalings(sigma, sigma_data)]
        return denoiser_model_forward(sd, cfg, x * c_in, sigma, **extra) * c_out + x * c_skip

    return model
