"""Functional CPU restatement of the SoundStream-XL decoder (TEST ORACLE).

Operates on a plain ``state_dict`` (reference key names); plain PyTorch CPU fp32 ops only.  See
``oracle/__init__.py`` for who may import this.  Pinned by tests/test_oracle_golden-style checks in
tests/test_soundstream_cpu.py against fixtures generated from the reference (tools/make_goldens.py tier S).

Reference followed (autoencoders/soundstream.py):
  :11-17   CausalConv1d (left padding dilation * (k - 1))
  :20-34   CausalConvTranspose1d (conv_transpose1d, tail of causal_padding = stride samples dropped)
  :37-52   ResidualUnit            :86-115  DecoderBlock            :143-165  SoundStreamXLDecoder
"""
import torch
import torch.nn.functional as F


def causal_conv(sd, p, x, dilation=1):
    w = sd[p + "weight"]
    return F.conv1d(F.pad(x, [dilation * (w.shape[2] - 1), 0]), w, sd[p + "bias"], dilation=dilation)


def causal_conv_transpose(sd, p, x, stride):
    w = sd[p + "weight"]                                   # [Cin, Cout, 2 * stride]
    cut = (w.shape[2] - 1) + 1 - stride                    # :23 with dilation 1, output_padding 0
    return F.conv_transpose1d(x, w, sd[p + "bias"], stride=stride)[..., :-cut]


def residual_unit(sd, p, x, dilation):
    h = F.elu(causal_conv(sd, p + "layers.0.", x, dilation))
    return x + F.conv1d(h, sd[p + "layers.2.weight"], sd[p + "layers.2.bias"])


def decoder_block(sd, p, x, stride):
    x = causal_conv_transpose(sd, p + "layers.0.", x, stride)
    for k, d in zip(range(2, 13, 2), (1, 3, 9, 1, 3, 9)):
        x = residual_unit(sd, f"{p}layers.{k}.", F.elu(x), d)
    return x


@torch.no_grad()
def decoder_forward(sd, strides, x):
    """SoundStreamXLDecoder.forward; ``strides`` as given to the constructor (the blocks use them last to first)."""
    sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
    n = len(strides)
    x = F.elu(causal_conv(sd, "layers.0.", x))
    for i, s in enumerate(reversed(strides)):
        x = F.elu(decoder_block(sd, f"layers.{2 + 2 * i}.", x, s))
    return causal_conv(sd, f"layers.{2 + 2 * n}.", x)


def encoder_block(sd, p, x, stride):
    """autoencoders/soundstream.py:55-83: six residual units (each followed by ELU), then the strided causal conv."""
    for k, d in zip(range(0, 11, 2), (1, 3, 9, 1, 3, 9)):
        x = F.elu(residual_unit(sd, f"{p}layers.{k}.", x, d))
    w = sd[p + "layers.12.weight"]
    return F.conv1d(F.pad(x, [w.shape[2] - 1, 0]), w, sd[p + "layers.12.bias"], stride=stride)


@torch.no_grad()
def encoder_forward(sd, strides, x):
    """SoundStreamXLEncoder.forward (autoencoders/soundstream.py:118-140)."""
    sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
    x = F.elu(causal_conv(sd, "layers.0.", x))
    for i, s in enumerate(strides):
        x = F.elu(encoder_block(sd, f"layers.{2 + 2 * i}.", x, s))
    return causal_conv(sd, f"layers.{2 + 2 * len(strides)}.", x)
