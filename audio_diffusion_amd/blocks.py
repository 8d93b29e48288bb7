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
        if dropout_rate:
            raise NotImplementedError("inference path: dropout must be 0")
        self.norm = nn.GroupNorm(1, c_in)
        self.n_head = n_head
        self.qkv_proj = nn.Conv1d(c_in, c_in * 3, 1)
        self.out_proj = nn.Conv1d(c_in, c_in, 1)
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
