"""Parameter containers mirroring the reference building blocks (blocks/blocks.py) key-for-key.

These modules own parameters/buffers only, laid out so ``state_dict()`` has the reference's names and
shapes and ``torch.manual_seed(s)`` + construction reproduces the reference initialisation (same
module creation order => same RNG consumption).  Compute never happens here: the denoiser compiles
the tree into a flat plan of libadk kernel launches (see ``unet.py``).
"""
import torch
from torch import nn

# blocks/blocks.py:169-180
FIR_TABLES = {
    "linear": (1 / 8, 3 / 8, 3 / 8, 1 / 8),
    "cubic": (-0.01171875, -0.03515625, 0.11328125, 0.43359375,
              0.43359375, 0.11328125, -0.03515625, -0.01171875),
    "lanczos3": (0.003689131001010537, 0.015056144446134567, -0.03399861603975296,
                 -0.066637322306633, 0.13550527393817902, 0.44638532400131226,
                 0.44638532400131226, 0.13550527393817902, -0.066637322306633,
                 -0.03399861603975296, 0.015056144446134567, 0.003689131001010537),
}


class _NoCompute(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run it through "
                           "DiffusionAttnUnet1D (libadk plan), there is no eager path")


def check_dropout_inactive(model):
    """Dropout (blocks/blocks.py:96,100,111,132) is the identity in eval mode, which is the mode the reference samples in;
    the plans therefore carry no dropout op.  A model left in train() mode with a non-zero rate would silently differ
    from the reference (which WOULD drop): refuse that instead."""
    if model.training:
        for name, m in model.named_modules():
            if isinstance(m, nn.Dropout) and m.p > 0 and m.training:
                raise RuntimeError(f"{name}: dropout p={m.p} is active (model.train()); this build covers the sampling path - "
                                   "call model.eval(), as the reference's sample() callers do")


class ResConvBlock(_NoCompute):
    """keys: main.{0,3}.{weight,bias}, main.{1,4}.{weight,bias}, skip.weight (blocks/blocks.py:66-76)."""

    def __init__(self, c_in, c_mid, c_out, is_last=False, kernel_size=5, conv_bias=True):
        super().__init__()
        # the reference builds the skip conv before the main convs: keep that order for the RNG
        skip = None if c_in == c_out else nn.Conv1d(c_in, c_out, 1, bias=False)
        pad = kernel_size // 2
        self.main = nn.Sequential(
            nn.Conv1d(c_in, c_mid, kernel_size, padding=pad, bias=conv_bias),
            nn.GroupNorm(1, c_mid),
            nn.GELU(),
            nn.Conv1d(c_mid, c_out, kernel_size, padding=pad, bias=conv_bias),
            nn.Identity() if is_last else nn.GroupNorm(1, c_out),
            nn.Identity() if is_last else nn.GELU(),
        )
        self.skip = skip if skip is not None else nn.Identity()
        self.is_last = is_last
        self.c_in, self.c_mid, self.c_out, self.kernel_size = c_in, c_mid, c_out, kernel_size


class SelfAttention1d(_NoCompute):
    """keys: norm.*, qkv_proj.*, out_proj.* (blocks/blocks.py:103-111)."""

    def __init__(self, c_in, n_head=1, dropout_rate=0.):
        super().__init__()
        if n_head == 0:
            raise ZeroDivisionError("integer modulo by zero")        # reference: c_in % n_head at :106
        assert c_in % n_head == 0
        if c_in // n_head != 32:
            raise NotImplementedError("libadk attention supports head dim 32 (n_head = c // 32)")
        self.norm = nn.GroupNorm(1, c_in)
        self.n_head = n_head
        self.qkv_proj = nn.Conv1d(c_in, c_in * 3, 1)
        self.out_proj = nn.Conv1d(c_in, c_in, 1)
        self.dropout = nn.Dropout(dropout_rate, inplace=True)       # identity in eval mode (check_dropout_inactive)
        self.c_in = c_in


class SkipBlock(_NoCompute):
    """keys under main.N (blocks/blocks.py:146-152); output = cat([main(x), x], dim=1)."""

    def __init__(self, *main):
        super().__init__()
        self.main = nn.Sequential(*main)


class FourierFeatures(_NoCompute):
    """key: weight [out/2, in] (blocks/blocks.py:154-159)."""

    def __init__(self, in_features, out_features, std=1.):
        super().__init__()
        assert out_features % 2 == 0
        self.weight = nn.Parameter(torch.randn([out_features // 2, in_features]) * std)


class Downsample1d(_NoCompute):
    """buffer: kernel [K] (blocks/blocks.py:183-189)."""

    def __init__(self, kernel="linear", pad_mode="reflect"):
        super().__init__()
        if pad_mode != "reflect":
            raise NotImplementedError("only reflect padding is implemented")
        taps = torch.tensor(FIR_TABLES[kernel])
        self.pad = taps.shape[0] // 2 - 1
        self.register_buffer("kernel", taps)


class Upsample1d(_NoCompute):
    """buffer: kernel [K] = 2 * table (blocks/blocks.py:199-205)."""

    def __init__(self, kernel="linear", pad_mode="reflect"):
        super().__init__()
        if pad_mode != "reflect":
            raise NotImplementedError("only reflect padding is implemented")
        taps = torch.tensor(FIR_TABLES[kernel]) * 2
        self.pad = taps.shape[0] // 2 - 1
        self.register_buffer("kernel", taps)


def Downsample1d_2(in_channels, out_channels, factor, kernel_multiplier=2):
    """Learned downsampler, key-for-key the reference factory (blocks/blocks.py:214-226): a strided nn.Conv1d
    (parameter container here; libadk runs it as space_to_depth + a 3-tap convolution, see fold_down_weight)."""
    assert kernel_multiplier % 2 == 0, "Kernel multiplier must be even"
    if kernel_multiplier != 2:
        raise NotImplementedError("libadk folds Downsample1d_2 for kernel_multiplier=2 (the reference default)")
    return nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=factor * kernel_multiplier + 1,
                     stride=factor, padding=factor * (kernel_multiplier // 2))


def Upsample1d_2(in_channels, out_channels, factor, use_nearest=False):
    """Learned upsampler, key-for-key the reference factory (blocks/blocks.py:229-255)."""
    if factor == 1:
        return nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1)
    if use_nearest:
        raise NotImplementedError("use_nearest=True is never used by DiffusionAttnUnet1D")
    return nn.ConvTranspose1d(in_channels=in_channels, out_channels=out_channels, kernel_size=factor * 2,
                              stride=factor, padding=factor // 2 + factor % 2, output_padding=factor % 2)


def fold_down_weight(w, factor):
    """Conv1d(k = 2f+1, stride f, padding f) == 3-tap stride-1 conv over the phase-folded input
    x'[p*C + c][m] = x[c][m*f + p]:   y[m] = sum_{q,p} w[:, :, q*f + p] x'[p][m - 1 + q]   (q*f + p <= 2f).
    w [Co, C, 2f+1] -> [Co, f*C, 3]."""
    co, c, k = w.shape
    f = factor
    assert k == 2 * f + 1
    out = w.new_zeros(co, f, c, 3)
    for q in range(3):
        for p in range(f):
            if q * f + p <= 2 * f:
                out[:, p, :, q] = w[:, :, q * f + p]
    return out.reshape(co, f * c, 3)


def fold_down_causal_weight(w, factor):
    """CausalConv1d(k = 2f, stride f) (left padding 2f-1; autoencoders/soundstream.py:78-79) == CAUSAL 3-tap stride-1 conv over
    the phase-folded input x'[p*C + c][m] = x[c][m*f + p]:  y[j] = sum_k w[k] x[(j-2) f + k + 1], so with q = k + 1
    rows j-2 / j-1 / j carry phases q (1..f-1) / q - f (0..f-1) / 0.   w [Co, C, 2f] -> [Co, f*C, 3]."""
    co, c, k = w.shape
    f = factor
    assert k == 2 * f
    out = w.new_zeros(co, f, c, 3)
    for kk in range(k):
        q = kk + 1
        out[:, q % f, :, q // f] = w[:, :, kk]
    return out.reshape(co, f * c, 3)


def fold_up_weight(wt, bias, factor, pad=None):
    """ConvTranspose1d(k = 2f, stride f, padding pad = f//2 + f%2, output_padding f%2) == 3-tap stride-1 conv
    producing f phases per input position, y[co][j*f + s] = z[s*Co + co][j]:  with r = (s + pad) mod f,
        r >= pad:  z[s][j] = wt[:, :, r + f]^T x[j - 1] + wt[:, :, r]^T x[j]
        r <  pad:  z[s][j] = wt[:, :, r + f]^T x[j]     + wt[:, :, r]^T x[j + 1]
    wt [C, Co, 2f] (ConvTranspose layout) -> weight [f*Co, C, 3], bias [f*Co]."""
    c, co, k = wt.shape
    f = factor
    assert k == 2 * f
    if pad is None:                # Upsample1d_2's padding; pad = 0: CausalConvTranspose1d (soundstream.py), whose
        pad = f // 2 + f % 2       # trimmed tail [..., :-f] leaves exactly y[j*f + s] = W[s]^T x[j] + W[s+f]^T x[j-1]
    out = wt.new_zeros(f, co, c, 3)
    for s in range(f):
        r = (s + pad) % f
        hi, lo = wt[:, :, r + f].t(), wt[:, :, r].t()          # [Co, C]
        if r >= pad:
            out[s, :, :, 0], out[s, :, :, 1] = hi, lo
        else:
            out[s, :, :, 1], out[s, :, :, 2] = hi, lo
    b = None if bias is None else bias.repeat(f)
    return out.reshape(f * co, c, 3), b


# ---- k-diffusion ("audio_v1") family: parameter containers of AudioDenoiserModel ---------------------------------
class AdaGN(_NoCompute):
    """key: mapper.{weight,bias} = Linear(feats_in, 2*c_out) (blocks/blocks.py:53-59)."""

    def __init__(self, feats_in, c_out, num_groups, eps=1e-5, cond_key="cond"):
        super().__init__()
        self.num_groups, self.eps, self.cond_key, self.c_out = num_groups, eps, cond_key, c_out
        self.mapper = nn.Linear(feats_in, c_out * 2)


class ResModConvBlock(_NoCompute):
    """keys: main.{0,4}.mapper.*, main.{2,6}.{weight,bias}, skip.weight (blocks/blocks.py:89-101):
    AdaGN -> GELU -> Conv1d(3) -> AdaGN -> GELU -> Conv1d(3), plus the (1x1 or identity) skip."""

    def __init__(self, feats_in, c_in, c_mid, c_out, group_size=32, dropout_rate=0.):
        super().__init__()
        skip = None if c_in == c_out else nn.Conv1d(c_in, c_out, 1, bias=False)      # built first, like the reference
        self.main = nn.Sequential(
            AdaGN(feats_in, c_in, max(1, c_in // group_size)),
            nn.GELU(),
            nn.Conv1d(c_in, c_mid, 3, padding=1),
            nn.Dropout(dropout_rate, inplace=True),
            AdaGN(feats_in, c_mid, max(1, c_mid // group_size)),
            nn.GELU(),
            nn.Conv1d(c_mid, c_out, 3, padding=1),
            nn.Dropout(dropout_rate, inplace=True),
        )
        self.skip = skip if skip is not None else nn.Identity()
        self.c_in, self.c_mid, self.c_out = c_in, c_mid, c_out


class SelfAttentionMod1d(_NoCompute):
    """keys: norm_in.mapper.*, qkv_proj.*, out_proj.* (blocks/blocks.py:124-132)."""

    def __init__(self, c_in, n_head, norm, dropout_rate=0.):
        super().__init__()
        assert c_in % n_head == 0
        if c_in // n_head not in (32, 64):
            raise NotImplementedError(f"libadk attention supports head dims 32 and 64, got {c_in // n_head}")
        self.norm_in = norm(c_in)
        self.n_head = n_head
        self.qkv_proj = nn.Conv1d(c_in, c_in * 3, 1)
        self.out_proj = nn.Conv1d(c_in, c_in, 1)
        self.dropout = nn.Dropout(dropout_rate)                     # identity in eval mode (check_dropout_inactive)
        self.c_in = c_in


def _mod_layers(n_layers, feats_in, c_in, c_mid, c_out, group_size, head_size, dropout_rate, self_attn):
    mods = []
    for i in range(n_layers):
        ci = c_in if i == 0 else c_mid
        co = c_mid if i < n_layers - 1 else c_out
        mods.append(ResModConvBlock(feats_in, ci, c_mid, co, group_size, dropout_rate))
        if self_attn:
            mods.append(SelfAttentionMod1d(co, max(1, co // head_size),
                                           lambda c, co=co: AdaGN(feats_in, c, max(1, co // group_size)), dropout_rate))
    return mods


class DBlock(nn.Sequential):
    """[Downsample1d_2] + n_layers x (ResModConvBlock [+ SelfAttentionMod1d]) (blocks/blocks.py:260-270)."""

    def __init__(self, n_layers, feats_in, c_in, c_mid, c_out, group_size=32, head_size=64, dropout_rate=0.,
                 downsample_ratio=1, self_attn=False):
        mods = [Downsample1d_2(c_in, c_in, downsample_ratio)] if downsample_ratio > 1 else []
        mods += _mod_layers(n_layers, feats_in, c_in, c_mid, c_out, group_size, head_size, dropout_rate, self_attn)
        super().__init__(*mods)
        self.ratio = downsample_ratio

    def forward(self, *a, **k):
        raise RuntimeError("DBlock is a parameter container; run it through AudioDenoiserModel")


class UBlock(nn.Sequential):
    """n_layers x (ResModConvBlock [+ SelfAttentionMod1d]) + [Upsample1d_2]; input = cat([x, skip]) (blocks.py:273-290)."""

    def __init__(self, n_layers, feats_in, c_in, c_mid, c_out, group_size=32, head_size=64, dropout_rate=0.,
                 upsample_ratio=1, self_attn=False):
        mods = _mod_layers(n_layers, feats_in, c_in, c_mid, c_out, group_size, head_size, dropout_rate, self_attn)
        if upsample_ratio > 1:
            mods.append(Upsample1d_2(c_out, c_out, upsample_ratio))
        super().__init__(*mods)
        self.ratio = upsample_ratio

    def forward(self, *a, **k):
        raise RuntimeError("UBlock is a parameter container; run it through AudioDenoiserModel")


class MappingNet(nn.Sequential):
    """keys: {0,2}.{weight,bias}; orthogonal init of the Linear weights (blocks/blocks.py:292-301)."""

    def __init__(self, feats_in, feats_out, n_layers=2):
        layers = []
        for i in range(n_layers):
            layers.append(nn.Linear(feats_in if i == 0 else feats_out, feats_out))
            layers.append(nn.GELU())
        super().__init__(*layers)
        for layer in self:
            if isinstance(layer, nn.Linear):
                nn.init.orthogonal_(layer.weight)

    def forward(self, *a, **k):
        raise RuntimeError("MappingNet is a parameter container; run it through AudioDenoiserModel")


class UNet(_NoCompute):
    """keys: d_blocks.N.*, u_blocks.N.* (deepest first) (blocks/blocks.py:303-316)."""

    def __init__(self, d_blocks, u_blocks):
        super().__init__()
        self.d_blocks = nn.ModuleList(d_blocks)
        self.u_blocks = nn.ModuleList(u_blocks)
