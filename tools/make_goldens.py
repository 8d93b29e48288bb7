#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (this container only).

Run:  python tools/make_goldens.py [--tier B T C R K S G F] [--ref /root/reference]
Writes small fixtures (inputs + expected outputs, never source text) to tests/golden/.

The reference's hot path is Python; three of its imports are unavailable or have drifted in this
image, so the script installs in-process stand-ins BEFORE importing (none of them is executed on
the DiffusionAttnUnet1D path except ``cached_conv``, whose published semantics are restated):
  * scipy.signal.kaiser (removed in SciPy 1.13; imported but unused at diffusion/pqmf.py:4)
  * cached_conv.{Conv1d,get_padding,USE_BUFFER_CONV} (third party, not vendored; pqmf.py:10-11)
  * local_attention / x_transformers (imported by autoencoders/transformer_ae.py:3-4, unused here)
  * vector_quantize_pytorch.ResidualVQ (imported at autoencoders/soundstream.py:5, referenced only by a commented-out class)
  * scipy.signal.firwin(nyq=...) keyword removed in SciPy 1.12 -> fs=2*nyq (pqmf.py:72)
This file never travels as a dependency of the product; it is the committed recipe for the
fixtures.
"""
import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def install_shims():
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser

    cc = types.ModuleType("cached_conv")
    cc.USE_BUFFER_CONV = False

    def get_padding(kernel_size, stride=1, dilation=1, mode="centered"):
        if kernel_size == 1:
            return (0, 0)
        p = (kernel_size - 1) * dilation + 1
        return (p // 2, p // 2)

    class Conv1d(nn.Conv1d):
        def __init__(self, *args, **kwargs):
            self._pad = kwargs.get("padding", (0, 0))
            kwargs["padding"] = 0
            super().__init__(*args, **kwargs)

        def forward(self, x):
            return F.conv1d(F.pad(x, self._pad), self.weight, self.bias, self.stride,
                            0, self.dilation, self.groups)

    cc.get_padding = get_padding
    cc.Conv1d = Conv1d
    sys.modules["cached_conv"] = cc

    la = types.ModuleType("local_attention")
    lat = types.ModuleType("local_attention.transformer")
    lat.LocalMHA = lat.FeedForward = object
    la.transformer = lat
    sys.modules["local_attention"] = la
    sys.modules["local_attention.transformer"] = lat
    vq = types.ModuleType("vector_quantize_pytorch")       # imported at autoencoders/soundstream.py:5, used only by a
    vq.ResidualVQ = object                                  # commented-out class; never executed on the decoder path
    sys.modules["vector_quantize_pytorch"] = vq
    xt = types.ModuleType("x_transformers")
    xt.ContinuousTransformerWrapper = xt.Encoder = object
    sys.modules["x_transformers"] = xt


def import_reference(ref):
    sys.path.insert(0, ref)
    install_shims()
    import scipy.signal
    import diffusion.pqmf as P
    P.firwin = lambda N, wc, window=None, scale=True, nyq=None: scipy.signal.firwin(
        N, wc, window=window, scale=scale, fs=2 * nyq)
    from decoders.diffusion_decoder import DiffusionAttnUnet1D
    from diffusion import sampling
    from blocks import blocks
    return P, DiffusionAttnUnet1D, sampling, blocks


def npz(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def sd_arrays(sd, prefix="sd/"):
    return {prefix + k: v for k, v in sd.items()}


def tier_blocks(P, blocks, sampling):
    print("tier B: per-block goldens")
    torch.manual_seed(100)
    out = {}
    # ResConvBlock variants
    for tag, (ci, cm, co, last) in dict(rcb_diff=(48, 64, 32, False), rcb_same=(32, 32, 32, False),
                                        rcb_last=(32, 32, 2, True)).items():
        m = blocks.ResConvBlock(ci, cm, co, is_last=last).eval()
        with torch.no_grad():
            for p in m.parameters():           # make the affine terms non-trivial
                p.mul_(0.5).add_(0.05 * torch.randn_like(p))
        x = torch.randn(2, ci, 96)
        with torch.no_grad():
            y = m(x)
        out.update({f"{tag}/x": x, f"{tag}/y": y, **sd_arrays(m.state_dict(), f"{tag}/sd/")})
    # SelfAttention1d
    for s in (8, 64, 256):
        m = blocks.SelfAttention1d(64, 2).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(2, 64, s)
        with torch.no_grad():
            y = m(x)
        out.update({f"attn{s}/x": x, f"attn{s}/y": y, **sd_arrays(m.state_dict(), f"attn{s}/sd/")})
    # resamplers
    for kern in ("linear", "cubic", "lanczos3"):
        for L in (8, 16, 4096):
            x = torch.randn(2, 3, L)
            d, u = blocks.Downsample1d(kern), blocks.Upsample1d(kern)
            with torch.no_grad():
                try:
                    out[f"down_{kern}_{L}/y"] = d(x)
                    out[f"down_{kern}_{L}/x"] = x
                except RuntimeError as e:       # reflect pad larger than input
                    print("   down", kern, L, "raises:", str(e)[:60])
                try:
                    out[f"up_{kern}_{L}/y"] = u(x)
                    out[f"up_{kern}_{L}/x"] = x
                except RuntimeError as e:
                    print("   up", kern, L, "raises:", str(e)[:60])
            out[f"down_{kern}/kernel"] = d.kernel
            out[f"up_{kern}/kernel"] = u.kernel
    # FourierFeatures + expand_to_planes
    ff = blocks.FourierFeatures(1, 16)
    t = torch.tensor([0.0, 0.25, 0.5, 1.0])
    with torch.no_grad():
        e = ff(t[:, None])
    out.update({"fourier/weight": ff.weight, "fourier/t": t, "fourier/y": e,
                "fourier/planes": blocks.expand_to_planes(e, (4, 16, 5))})
    # SkipBlock ordering
    sb = blocks.SkipBlock(nn.Identity())
    x = torch.randn(1, 2, 4)
    out.update({"skip/x": x, "skip/y": sb(x)})
    # DDIM schedule tables (fp32 bit patterns)
    for steps in (5, 10, 100):
        t = torch.linspace(1, 0, steps + 1)[:-1]
        a, s = sampling.get_alphas_sigmas(t)
        out.update({f"sched{steps}/t": t, f"sched{steps}/alpha": a, f"sched{steps}/sigma": s})
    tt = sampling.alpha_sigma_to_t(out["sched10/alpha"], out["sched10/sigma"])
    out["sched10/t_back"] = tt
    npz("blocks.npz", **out)

    print("tier B: PQMF goldens (cached_conv restated -> values 'parity unpinned', see oracle)")
    out = {}
    for ch, att, M in ((1, 70, 32), (2, 70, 32), (2, 70, 4), (2, 70, 16), (2, 100, 32)):
        tag = f"pqmf_c{ch}_a{att}_m{M}"
        torch.manual_seed(7)
        q = P.CachedPQMF(ch, att, M).eval()
        T = M * 96
        x = torch.randn(2, ch, T)
        with torch.no_grad():
            y = q(x)
            z = q.inverse(y)
            # cached_conv-free anchor from the reference itself (base-class polyphase path)
            yb = P.reverse_half(P.polyphase_forward(x[:, :1], q.hk))
        out.update({f"{tag}/h": q.h, f"{tag}/hk": q.hk, f"{tag}/fw": q.forward_conv.weight,
                    f"{tag}/iw": q.inverse_conv.weight, f"{tag}/x": x, f"{tag}/y": y, f"{tag}/z": z,
                    f"{tag}/y_polyphase_ch0": yb})
    npz("pqmf.npz", **out)


def run_model_golden(name, Model, sampling, kwargs, x_shape, seed, steps=5):
    torch.manual_seed(seed)
    m = Model(**kwargs).eval()
    torch.manual_seed(seed + 1)
    x = torch.randn(*x_shape)
    t = torch.rand(x_shape[0])
    with torch.no_grad():
        y = m(x, t)
    vs = []
    orig = m.forward

    def rec(xx, tt, **kw):
        v = orig(xx, tt, **kw)
        vs.append(v.clone())
        return v
    m.forward = rec
    pred = sampling.sample(m, x, steps, 0)
    m.forward = orig
    cfg = {k: v for k, v in kwargs.items()}
    npz(name, x=x, t=t, y=y, pred=pred, v_steps=torch.stack(vs), seed=np.int64(seed),
        steps=np.int64(steps), cfg=np.bytes_(json.dumps(cfg).encode()), **sd_arrays(m.state_dict()))


def tier_tiny(Model, sampling):
    print("tier T: tiny full models")
    run_model_golden("tiny_d3.npz", Model, sampling,
                     dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1), (2, 2, 1024), seed=0)
    run_model_golden("tiny_d5.npz", Model, sampling,
                     dict(depth=5, c_mults=[32, 32, 64, 64, 96], n_attn_layers=1), (2, 2, 2048), seed=10)
    run_model_golden("tiny_d3_pqmf4.npz", Model, sampling,
                     dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1, pqmf_bands=4), (2, 2, 4096), seed=20)
    run_model_golden("tiny_d4_noattn_nobias.npz", Model, sampling,
                     dict(depth=4, c_mults=[32, 64, 64, 64], n_attn_layers=0, conv_bias=False, io_channels=1),
                     (3, 1, 512), seed=30)


def run_cond_golden(name, Model, sampling, kwargs, x_shape, cond_len, seed, aug_scale=None, steps=4):
    """Conditioning path (decoders/diffusion_decoder.py:148-169): cond is linearly interpolated to the signal
    length; with cond_noise_aug the level is fixed to ``aug_scale`` (0.0 keeps the call deterministic: the
    reference multiplies its randn by it) and a second Fourier embedding is inserted before cond."""
    import contextlib
    import io
    torch.manual_seed(seed)
    m = Model(**kwargs).eval()
    torch.manual_seed(seed + 1)
    x = torch.randn(*x_shape)
    t = torch.rand(x_shape[0])
    cond = torch.randn(x_shape[0], kwargs["cond_dim"], cond_len)
    extra = dict(cond=cond)
    if aug_scale is not None:
        extra["cond_aug_scale"] = aug_scale
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        y = m(x, t, **extra)
        pred = sampling.sample(m, x, steps, 0, **extra)
    npz(name, x=x, t=t, cond=cond, y=y, pred=pred, seed=np.int64(seed), steps=np.int64(steps),
        aug_scale=np.float32(-1.0 if aug_scale is None else aug_scale),
        cfg=np.bytes_(json.dumps(dict(kwargs)).encode()), **sd_arrays(m.state_dict()))


def tier_learned(Model, sampling):
    print("tier R: learned resamplers + kernel_size 7 / 3 (SURVEY 8 f2)")
    run_model_golden("tiny_learned_k7.npz", Model, sampling,
                     dict(depth=4, c_mults=[32, 32, 64, 64], n_attn_layers=1, learned_resample=True,
                          strides=[9, 2, 4, 1], kernel_size=7), (2, 2, 1024), seed=50, steps=4)
    run_model_golden("tiny_learned_s3_k3.npz", Model, sampling,
                     dict(depth=3, c_mults=[32, 64, 64], n_attn_layers=0, learned_resample=True,
                          strides=[1, 3, 8], kernel_size=3, io_channels=1), (3, 1, 1536), seed=51, steps=4)
    run_model_golden("tiny_d3_k7_fir.npz", Model, sampling,
                     dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1, kernel_size=7), (2, 2, 512), seed=52, steps=4)


def tier_cond(Model, sampling):
    print("tier C: conditioning path (SURVEY 8 f1)")
    run_cond_golden("tiny_d3_cond.npz", Model, sampling,
                    dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1, cond_dim=4), (2, 2, 1024), 37, seed=40)
    run_cond_golden("tiny_d3_cond_samelen.npz", Model, sampling,
                    dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=0, cond_dim=32), (2, 2, 512), 512, seed=41)
    run_cond_golden("tiny_d3_cond_aug0.npz", Model, sampling,
                    dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1, cond_dim=3, cond_noise_aug=True),
                    (2, 2, 1024), 100, seed=42, aug_scale=0.0)


def tier_kdiff():
    """AudioDenoiserModel + Denoiser (decoders/diffusion_decoder.py:261-337), SURVEY 8 f4.  proj_out is zero-initialised
    by the reference (:281-282), so its parameters are re-drawn to make the fixtures informative."""
    print("tier K: k-diffusion denoiser family (SURVEY 8 f4)")
    from decoders.diffusion_decoder import AudioDenoiserModel, Denoiser
    cases = {
        # c_in, feats_in, depths, channels, self_attn_depths, strides, extra ctor kwargs, L, cond_len, sigma_data
        "kdiff_tiny": (dict(c_in=2, feats_in=32, depths=[1, 2, 2], channels=[32, 64, 64],
                            self_attn_depths=[False, False, True], strides=[4, 2]), 256, 0, 0.5),
        "kdiff_attn128": (dict(c_in=4, feats_in=64, depths=[2, 1, 1], channels=[64, 128, 128],
                               self_attn_depths=[False, True, True], strides=[2, 3]), 384, 0, 1.0),
        "kdiff_cond": (dict(c_in=2, feats_in=32, depths=[1, 1], channels=[32, 32], self_attn_depths=[False, True],
                            strides=[2], mapping_cond_dim=5, unet_cond_dim=3), 128, 19, 0.25),
    }
    for i, (name, (kw, L, cond_len, sigma_data)) in enumerate(cases.items()):
        torch.manual_seed(60 + i)
        m = AudioDenoiserModel(**kw).eval()
        torch.manual_seed(160 + i)
        with torch.no_grad():
            m.proj_out.weight.normal_(0, 0.2)
            m.proj_out.bias.normal_(0, 0.1)
        x = torch.randn(2, kw["c_in"], L)
        sigma = torch.tensor([0.05, 2.0])
        extra = {}
        if cond_len:
            extra = dict(mapping_cond=torch.randn(2, kw["mapping_cond_dim"]), unet_cond=torch.randn(2, kw["unet_cond_dim"], cond_len))
        with torch.no_grad():
            y = m(x, sigma, **extra)
            y_nolog = m(x, sigma, log_sigma=False, **extra)
            # Denoiser.forward (:336) builds its scalings with `append_dims(x, x.ndim) for x in ...`: the comprehension
            # variable shadows the input, nothing is appended, and the broadcast only works for batch size 1 -- so the
            # reference is called one sample at a time (per-sample scaling is what it computes there)
            dn = Denoiser(m, sigma_data=sigma_data)
            d = torch.cat([dn(x[b:b + 1], sigma[b:b + 1], **{k: v[b:b + 1] for k, v in extra.items()}) for b in range(2)])
        samp = {}
        if name == "kdiff_tiny":            # the reference's own samplers around its own Denoiser (batch 1: see above)
            from diffusion import inference as INF
            sig_k = INF.get_sigmas_karras(6, 1e-2, 3.0, rho=9.)
            sig_e = INF.get_sigmas_exponential(5, 1e-2, 3.0)
            x1 = x[:1] * 3.0
            with torch.no_grad():
                samp = {"samp/sigmas_karras": sig_k, "samp/sigmas_exp": sig_e, "samp/x": x1,
                        "samp/heun": INF.sample_heun(dn, x1, sig_k, disable=True),
                        "samp/euler": INF.sample_heun(dn, x1, sig_e, disable=True, second_order=False),
                        "samp/lms": INF.sample_lms(dn, x1, sig_k, disable=True, order=3)}
        npz(name + ".npz", seed=np.int64(60 + i), x=x, sigma=sigma, y=y, **samp, y_nolog=y_nolog, denoised=d, sigma_data=np.float32(sigma_data),
            cfg=np.bytes_(json.dumps(kw).encode()), **{"extra/" + k: v for k, v in extra.items()}, **sd_arrays(m.state_dict()))


def tier_soundstream():
    """SoundStreamXLDecoder (autoencoders/soundstream.py:143-165), SURVEY 8 f4: small widths, every stride parity."""
    print("tier S: SoundStream-XL decoder (SURVEY 8 f4)")
    from autoencoders.soundstream import SoundStreamXLDecoder
    cases = {"sstream_tiny": (dict(out_channels=2, capacity=32, latent_dim=24, c_mults=[2, 2, 4], strides=[2, 5, 4]), 2, 13),
             "sstream_s8": (dict(out_channels=1, capacity=32, latent_dim=8, c_mults=[1, 2], strides=[3, 8]), 3, 40)}
    for i, (name, (kw, B, L)) in enumerate(cases.items()):
        torch.manual_seed(70 + i)
        m = SoundStreamXLDecoder(**kw).eval()
        torch.manual_seed(170 + i)
        z = torch.randn(B, kw["latent_dim"], L)
        with torch.no_grad():
            y = m(z)
        npz(name + ".npz", seed=np.int64(70 + i), z=z, y=y, cfg=np.bytes_(json.dumps(kw).encode()), **sd_arrays(m.state_dict()))
    from autoencoders.soundstream import SoundStreamXLEncoder
    ecases = {"sstream_enc_tiny": (dict(in_channels=2, capacity=32, latent_dim=24, c_mults=[2, 2, 4], strides=[2, 5, 4]), 2, 40 * 13),
              "sstream_enc_s8": (dict(in_channels=1, capacity=32, latent_dim=8, c_mults=[1, 2], strides=[3, 8]), 3, 24 * 20)}
    for i, (name, (kw, B, T)) in enumerate(ecases.items()):
        torch.manual_seed(80 + i)
        m = SoundStreamXLEncoder(**kw).eval()
        torch.manual_seed(180 + i)
        x = torch.randn(B, kw["in_channels"], T)
        with torch.no_grad():
            y = m(x)
        npz(name + ".npz", seed=np.int64(80 + i), x=x, y=y, cfg=np.bytes_(json.dumps(kw).encode()), **sd_arrays(m.state_dict()))


def tier_full(Model, sampling):
    print("tier F: full 221M-param uncond model (seeds 0/1), digests + 10-step output")
    torch.manual_seed(0)
    m = Model(n_attn_layers=4).eval()
    sd = m.state_dict()
    h = hashlib.sha256()
    per = {}
    for k, v in sd.items():
        b = v.detach().contiguous().numpy().tobytes()
        h.update(k.encode())
        h.update(b)
        flat = v.detach().reshape(-1)
        per[k] = dict(shape=list(v.shape), sum=float(flat.double().sum()),
                      head=[float(z) for z in flat[:4]])
    n_params = sum(p.numel() for p in m.parameters())
    torch.manual_seed(1)
    x = torch.randn(1, 2, 65536)
    vs = []
    orig = m.forward

    def rec(xx, tt, **kw):
        v = orig(xx, tt, **kw)
        if len(vs) < 1:
            vs.append(v.clone())
        return v
    m.forward = rec
    import time
    t0 = time.time()
    pred = sampling.sample(m, x, 10, 0)
    dt = time.time() - t0
    meta = dict(n_params=n_params, n_tensors=len(sd), sd_sha256=h.hexdigest(),
                pred_sha256=hashlib.sha256(pred.numpy().tobytes()).hexdigest(),
                pred_mean=float(pred.mean()), pred_std=float(pred.std()), pred_absmax=float(pred.abs().max()),
                wall_s=dt, threads=torch.get_num_threads(), torch=torch.__version__)
    print("  ", meta)
    with open(os.path.join(OUT, "full_digest.json"), "w") as f:
        json.dump(dict(meta=meta, tensors=per), f)
    npz("full_seed0_steps10.npz", pred=pred, v0=vs[0])


def extract_functions(path, names):
    """Compile just the named top-level function definitions of a reference script (ast; the script itself is never
    imported or run: train_uncond.py needs prefigure / wandb / aeiou, absent here) into a fresh namespace."""
    import ast
    import math
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), [n.name for n in keep]
    ns = {"torch": torch, "math": math, "trange": range}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def tier_glue(P, Model, ref):
    """SURVEY 8 f3 + a16: the training demos' crash-warped sampler (train_uncond.py:35-86), the diffusion-autoencoder
    DDIM (diffusion/inference.py:14-57) and the base (non-cached) PQMF paths (diffusion/pqmf.py:95-233)."""
    import contextlib
    import io
    import warnings
    print("tier G: crash schedule, diff-AE DDIM, base PQMF")
    out = {}
    ns = extract_functions(os.path.join(ref, "train_uncond.py"),
                           ["get_alphas_sigmas", "get_crash_schedule", "alpha_sigma_to_t", "sample"])
    for steps in (5, 6, 10, 100, 250):
        out[f"crash/t{steps}"] = ns["get_crash_schedule"](torch.linspace(1, 0, steps + 1)[:-1])
    # the same tiny model as tiny_d3.npz (seed 0), so no second copy of its weights is stored
    kwargs = dict(depth=3, c_mults=[32, 32, 64], n_attn_layers=1)
    torch.manual_seed(0)
    m = Model(**kwargs).eval()
    torch.manual_seed(1)
    x = torch.randn(2, 2, 1024)
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        out["crash/x"] = x
        out["crash/pred6"] = ns["sample"](m, x, 6, 0)

        from diffusion import inference as INF

        class ToyDiffAE:                      # any object with encode / decode (inference.py:23,33)
            def encode(self, signal):
                return signal.mean(dim=2, keepdim=True)

            def decode(self, xx, tt, z_sem):
                return m(xx + 0.1 * z_sem, tt)

        torch.manual_seed(11)
        noise = torch.randn_like(x)            # the draw inference.py:26 makes under this seed
        torch.manual_seed(11)
        out["diffae/signal"], out["diffae/noise"] = x, noise
        out["diffae/pred4"] = INF.sample(ToyDiffAE(), x, 4, 0)
    out["model_sd_check"] = m.state_dict()["net.0.main.0.weight"][:2, :2]

    for M in (4, 16, 32):                  # 32 = config 4's bank
        torch.manual_seed(7)
        xb = torch.randn(2, 1, M * 96)
        for poly in (True, False):
            q = P.PQMF(70, M, polyphase=poly).eval()
            tag = f"pqmf_base_m{M}_{'poly' if poly else 'classic'}"
            with torch.no_grad():
                y = q(xb)
                z = q.inverse(y)
            out.update({f"{tag}/x": xb, f"{tag}/y": y, f"{tag}/z": z, f"{tag}/hk": q.hk})
    npz("glue.npz", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--tier", nargs="+", default=["B", "T", "C", "R", "K", "S", "G", "F"])
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    P, Model, sampling, blocks = import_reference(a.ref)
    if "B" in a.tier:
        tier_blocks(P, blocks, sampling)
    if "T" in a.tier:
        tier_tiny(Model, sampling)
    if "C" in a.tier:
        tier_cond(Model, sampling)
    if "R" in a.tier:
        tier_learned(Model, sampling)
    if "K" in a.tier:
        tier_kdiff()
    if "S" in a.tier:
        tier_soundstream()
    if "G" in a.tier:
        tier_glue(P, Model, a.ref)
    if "F" in a.tier:
        tier_full(Model, sampling)


if __name__ == "__main__":
    main()
