"""SoundStreamXLDecoder (and its mirror, SoundStreamXLEncoder): drop-ins for the reference latent <-> audio
autoencoder halves (autoencoders/soundstream.py:11-165); the decoder is the stage after latent sampling.
Executed as flat plans of libadk launches.

* Same constructor signature/defaults, ``state_dict`` keys and seeded initialisation as the reference.
* Every op is a convolution on the MFMA kernels:
    CausalConv1d(k=7, dilation d)   one conv launch, tap k reads x[t - (6-k) d] (left halo of 6 d zero rows = the causal
                                    padding), ELU fused into the epilogue;
    ResidualUnit                    = that conv + a 1x1 conv whose epilogue adds the skip AND applies the ELU that follows
                                    every unit (so each tensor is stored once, already activated);
    CausalConvTranspose1d(k=2s, stride s)[..., :-s]
                                    y[j s + r] = W[:, :, r]^T x[j] + W[:, :, r + s]^T x[j-1]: one 3-tap conv producing
                                    s phases (+ ELU) and a depth_to_space row shuffle (as the learned resamplers of unet.py).
    CausalConv1d(k=2s, stride s)    (encoder) space_to_depth + one causal 3-tap conv over s*C channels (fold_down_causal_weight).
* No CPU or eager fallback.
"""
import torch
from torch import nn

from . import _lib
from .blocks import fold_up_weight, fold_down_causal_weight
from .unet import _Plan, _ceil_to


class CausalConv1d(nn.Conv1d):
    """Parameter container (autoencoders/soundstream.py:11-17)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.causal_padding = self.dilation[0] * (self.kernel_size[0] - 1)

    def forward(self, x):
        raise RuntimeError("CausalConv1d is a parameter container; run it through SoundStreamXLDecoder")


class CausalConvTranspose1d(nn.ConvTranspose1d):
    """Parameter container (autoencoders/soundstream.py:20-34)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.causal_padding = (self.dilation[0] * (self.kernel_size[0] - 1) + self.output_padding[0] + 1 - self.stride[0])

    def forward(self, x, output_size=None):
        raise RuntimeError("CausalConvTranspose1d is a parameter container; run it through SoundStreamXLDecoder")


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run it through SoundStreamXLDecoder")


class ResidualUnit(_Container):
    """keys: layers.0 (causal k=7 conv), layers.2 (1x1 conv) (autoencoders/soundstream.py:37-52)."""

    def __init__(self, in_channels, out_channels, dilation):
        super().__init__()
        self.dilation = dilation
        self.layers = nn.Sequential(
            CausalConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=7, dilation=dilation),
            nn.ELU(),
            nn.Conv1d(in_channels=out_channels, out_channels=out_channels, kernel_size=1))


class DecoderBlock(_Container):
    """keys: layers.0 (transposed conv), layers.{2,4,...,12} (residual units, dilations 1,3,9,1,3,9) (soundstream.py:86-115)."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        mods = [CausalConvTranspose1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=stride)]
        for d in (1, 3, 9, 1, 3, 9):
            mods += [nn.ELU(), ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=d)]
        self.layers = nn.Sequential(*mods)
        self.stride = stride


class EncoderBlock(_Container):
    """keys: layers.{0,2,...,10} (residual units, dilations 1,3,9,1,3,9), layers.12 (strided causal conv) (soundstream.py:55-83)."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        mods = []
        for d in (1, 3, 9, 1, 3, 9):
            mods += [ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=d), nn.ELU()]
        mods.append(CausalConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=stride))
        self.layers = nn.Sequential(*mods)
        self.stride = stride


class _SSPlan(_Plan):
    """Compiled SoundStreamXLDecoder.forward for one (B, latent length, dtype)."""

    def causal(self, x, key, mod, elu, **kw):
        return self.conv(x, key, mod.weight, mod.bias, mod.out_channels, dilation=mod.dilation[0], causal=True, elu=elu, **kw)[0]

    def residual_unit(self, ru, key, x, elu_after):
        """x + conv1x1(ELU(causal_conv7(x))) [-> ELU]  (soundstream.py:51-52).  Frees x."""
        h = self.causal(x, key + "layers.0", ru.layers[0], elu=True)
        c1 = ru.layers[2]
        y, _ = self.conv(h, key + "layers.2", c1.weight, c1.bias, c1.out_channels, residual=x, elu=elu_after)
        self.free(h, x)
        return y

    def _build(self):
        if isinstance(self.model, SoundStreamXLEncoder):
            return self._build_encoder()
        m = self.model
        L = self.T
        x0 = self.tensor(_ceil_to(m.latent_dim, 32), L)
        x0.C = m.latent_dim
        self._emit(_lib.OP_PACK_INPUT, {"x": ("input", 0), "y": ("arena", x0.off)}, flags=_lib.FLAG_NO_EMBED, L=L, Cin=m.latent_dim)
        layers = m.layers
        h = self.causal(x0, "layers.0", layers[0], elu=True)               # CausalConv1d(latent -> 16*capacity) + ELU
        self.free(x0)
        for li in range(2, len(layers) - 1, 2):                             # DecoderBlock, ELU pairs
            blk = layers[li]
            key = f"layers.{li}.layers."
            up = blk.layers[0]
            s = blk.stride
            w3, b3 = fold_up_weight(up.weight.detach().float().cpu(), None if up.bias is None else up.bias.detach().float().cpu(),
                                    s, pad=0)
            z, _ = self.conv(h, key + f"0.fold{s}", w3, b3, s * up.out_channels, elu=True)      # ELU commutes with the shuffle
            self.free(h)
            h = self.shuffle(z, s, False)
            self.free(z)
            for k in range(2, 13, 2):                                       # every unit is followed by an ELU (:97-110, :158)
                h = self.residual_unit(blk.layers[k], f"{key}{k}.", h, elu_after=True)
        last = layers[len(layers) - 1]
        o = self.causal(h, f"layers.{len(layers) - 1}", last, elu=False, out_f32=True)
        self.free(h)
        self._emit(_lib.OP_UNPACK_OUTPUT, {"x": ("arena", o.off), "y": ("output", 0)}, L=o.L, Cin=m.out_channels)


    def _build_encoder(self):
        m = self.model
        L = self.T
        x0 = self.tensor(32, L)
        x0.C = m.in_channels
        self._emit(_lib.OP_PACK_INPUT, {"x": ("input", 0), "y": ("arena", x0.off)}, flags=_lib.FLAG_NO_EMBED, L=L, Cin=m.in_channels)
        layers = m.layers
        h = self.causal(x0, "layers.0", layers[0], elu=True)
        self.free(x0)
        for li in range(2, len(layers) - 1, 2):                             # EncoderBlock, ELU pairs (:131-133)
            blk = layers[li]
            key = f"layers.{li}.layers."
            for k in range(0, 11, 2):
                h = self.residual_unit(blk.layers[k], f"{key}{k}.", h, elu_after=True)
            down, s = blk.layers[12], blk.stride
            folded = self.shuffle(h, s, True)
            self.free(h)
            h, _ = self.conv(folded, f"{key}12.fold{s}", fold_down_causal_weight(down.weight.detach().float().cpu(), s), down.bias,
                             down.out_channels, causal=True, elu=True)
            self.free(folded)
        last = layers[len(layers) - 1]
        o = self.causal(h, f"layers.{len(layers) - 1}", last, elu=False, out_f32=True)
        self.free(h)
        self._emit(_lib.OP_UNPACK_OUTPUT, {"x": ("arena", o.off), "y": ("output", 0)}, L=o.L, Cin=m.latent_dim)


class _SSBase(nn.Module):
    """Cache management shared by the two halves (same contract as DiffusionAttnUnet1D)."""

    kernel_size = 7
    halo_rows = 54                        # 6 taps x dilation 9: the deepest causal padding, kept as zero rows in HBM

    def _init_caches(self):
        self.compute_dtype = None          # None: fp32 (bf16 under torch.autocast)
        self._wcache = {}
        self._plans = {}

    def invalidate_cache(self):
        self._wcache = {}
        self._plans = {}

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_cache()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_cache()
        return out

    def total_stride(self):
        f = 1
        for s in self.strides:
            f *= s
        return f

    def plan_for(self, B, L, dtype, device):
        key = (B, L, dtype, str(device))
        plan = self._plans.get(key)
        if plan is None:
            plan = _SSPlan(self, B, L, dtype, device)
            self._plans[key] = plan
        return plan

    def _run(self, x, channels_in, channels_out, out_len, what):
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} (MI355X build) runs on ROCm devices only; there is no CPU path")
        if x.dim() != 3 or x.shape[1] != channels_in:
            raise RuntimeError(f"expected {what} of shape [B, {channels_in}, L], got {tuple(x.shape)} "
                               "(the reference fails in its first conv with a channel mismatch)")
        dtype = self.compute_dtype or (_lib.autocast_dtype() if torch.is_autocast_enabled() else torch.float32)
        xin = x.detach().to(torch.float32).contiguous()
        plan = self.plan_for(x.shape[0], x.shape[2], dtype, x.device)
        out = torch.empty(x.shape[0], channels_out, out_len, dtype=torch.float32, device=x.device)
        plan.bind_named(input=xin.data_ptr(), output=out.data_ptr())
        plan.run(torch.cuda.current_stream(x.device).cuda_stream)
        return out


class SoundStreamXLEncoder(_SSBase):
    """Drop-in for the reference class of the same name (autoencoders/soundstream.py:118-140): audio -> latents."""

    def __init__(self, in_channels=2, capacity=32, latent_dim=128, c_mults=[2, 4, 4, 4, 8, 16], strides=[2, 2, 2, 4, 5, 8]):
        super().__init__()
        c_mults = [1] + list(c_mults)
        self.depth = len(c_mults)
        self.in_channels, self.capacity, self.latent_dim, self.strides = in_channels, capacity, latent_dim, list(strides)
        if capacity % 32:
            raise NotImplementedError("libadk's space_to_depth needs channel counts that are multiples of 32 (capacity % 32 == 0)")
        if in_channels > 32:
            raise NotImplementedError("in_channels > 32 is not supported by the input boundary")
        layers = [CausalConv1d(in_channels=in_channels, out_channels=c_mults[0] * capacity, kernel_size=7), nn.ELU()]
        for i in range(self.depth - 1):
            layers.append(EncoderBlock(in_channels=c_mults[i] * capacity, out_channels=c_mults[i + 1] * capacity, stride=strides[i]))
            layers.append(nn.ELU())
        layers.append(CausalConv1d(in_channels=c_mults[-1] * capacity, out_channels=latent_dim, kernel_size=3))
        self.layers = nn.Sequential(*layers)
        self._init_caches()

    @torch.no_grad()
    def forward(self, x):
        f = self.total_stride()
        if x.dim() == 3 and x.shape[2] % f:
            raise NotImplementedError(f"audio length {x.shape[2]} must be a multiple of the total stride {f} "
                                      "(the reference rounds up at every strided conv; pad the clip)")
        return self._run(x, self.in_channels, self.latent_dim, x.shape[2] // f if x.dim() == 3 else 0, "audio")


class SoundStreamXLDecoder(_SSBase):
    """Drop-in for the reference class of the same name (autoencoders/soundstream.py:143-165)."""

    def __init__(self, out_channels=2, capacity=32, latent_dim=128, c_mults=[2, 4, 4, 4, 8, 16], strides=[2, 2, 2, 4, 5, 8]):
        super().__init__()
        c_mults = [1] + list(c_mults)
        self.depth = len(c_mults)
        self.out_channels, self.capacity, self.latent_dim, self.strides = out_channels, capacity, latent_dim, list(strides)
        if capacity % 32:
            raise NotImplementedError("libadk's depth_to_space needs channel counts that are multiples of 32 (capacity % 32 == 0)")
        layers = [CausalConv1d(in_channels=latent_dim, out_channels=c_mults[-1] * capacity, kernel_size=7), nn.ELU()]
        for i in range(self.depth - 1, 0, -1):
            layers.append(DecoderBlock(in_channels=c_mults[i] * capacity, out_channels=c_mults[i - 1] * capacity,
                                       stride=strides[i - 1]))
            layers.append(nn.ELU())
        layers.append(CausalConv1d(in_channels=c_mults[0] * capacity, out_channels=out_channels, kernel_size=7))
        self.layers = nn.Sequential(*layers)
        self._init_caches()

    def upsampling(self):
        return self.total_stride()

    @torch.no_grad()
    def forward(self, x):
        return self._run(x, self.latent_dim, self.out_channels, (x.shape[2] if x.dim() == 3 else 0) * self.total_stride(), "latents")
