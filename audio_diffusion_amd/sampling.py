"""v-objective DDIM sampler: drop-in for the reference ``diffusion/sampling.py`` (same names, same
signatures: ``get_alphas_sigmas``, ``alpha_sigma_to_t``, ``sample(model, x, steps, eta, **extra_args)``).

MI355X design: the per-step scalars (t, alpha, sigma, ...) are computed on the host exactly like the
reference (CPU fp32 torch ops, sampling.py:22-24,42-44) and uploaded ONCE as a coefficient table; a
device-side step counter indexes it, so the whole fixed-shape step (denoiser plan + fused literal-order
update kernel + counter++) is captured in one hipGraph and replayed ``steps`` times with no host work in
between.  eta > 0 draws its noise with torch's device RNG per step (eager loop).
"""
import math

import torch

from . import _lib
from .unet import DiffusionAttnUnet1D, FusedLaunchError


def get_alphas_sigmas(t):
    """Scaling factors for the clean signal (alpha) and the noise (sigma) at timestep t
    (reference sampling.py:6-9)."""
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def alpha_sigma_to_t(alpha, sigma):
    """Inverse of get_alphas_sigmas (reference sampling.py:11-14)."""
    return torch.atan2(sigma, alpha) / math.pi * 2


def get_crash_schedule(t):
    """Timestep warp used by the training demos (reference train_uncond.py:35-38)."""
    sigma = torch.sin(t * math.pi / 2) ** 2
    alpha = (1 - sigma ** 2) ** 0.5
    return alpha_sigma_to_t(alpha, sigma)


def step_coefficients(steps, eta, warp=None):
    """Host table [steps, 8] = (t, alpha, sigma, alpha_next, adjusted_sigma, ddim_sigma, is_last, 0),
    computed with the reference's CPU fp32 expressions (sampling.py:22-24, 42-44).  ``warp`` maps the linear
    timesteps first (train_uncond.py:51-55 applies get_crash_schedule there)."""
    t = torch.linspace(1, 0, steps + 1)[:-1]
    if warp is not None:
        t = warp(t)
    alphas, sigmas = get_alphas_sigmas(t)
    tab = torch.zeros(steps, 8, dtype=torch.float32)
    tab[:, 0], tab[:, 1], tab[:, 2] = t, alphas, sigmas
    for i in range(steps):
        if i < steps - 1:
            ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * \
                (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
            adjusted = (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()
            tab[i, 3], tab[i, 4], tab[i, 5] = alphas[i + 1], adjusted, ddim_sigma
        else:
            tab[i, 6] = 1.0
    return tab


class _FusedSampler:
    """Captured step loop for one (model, B, T, dtype, steps)."""
    replay_window = 16          # steps the host may run ahead of the GPU (0 = unbounded)

    def __init__(self, model, x, steps, dtype, use_graph=True, cond_shape=None):
        dev = x.device
        self.steps = steps
        # cond (constant over the loop) lives in a buffer of its own: the captured graph holds its address
        self.cond = torch.empty(cond_shape, dtype=torch.float32, device=dev) if cond_shape else None
        self.x = torch.empty_like(x, dtype=torch.float32)
        self.v = torch.empty_like(self.x)
        self.pred = torch.empty_like(self.x)
        self.coef = torch.zeros(steps, 8, dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.plan = model.plan_for(x.shape[0], x.shape[2], dtype, dev,
                                   t_table=(self.coef.data_ptr(), self.step.data_ptr()),
                                   cond_len=cond_shape[2] if cond_shape else 0)
        self.plan.bind(self.x.data_ptr(), 0, self.v.data_ptr(), self.cond.data_ptr() if cond_shape else 0)
        tail = (_lib.AdkOp * 2)()
        n = self.x.numel()
        tail[0].kind, tail[0].i0, tail[0].i1 = _lib.OP_DDIM_UPDATE, n & 0x7fffffff, n >> 31
        tail[0].y, tail[0].x, tail[0].aux1 = self.x.data_ptr(), self.v.data_ptr(), self.pred.data_ptr()
        tail[0].aux0, tail[0].bias = self.coef.data_ptr(), self.step.data_ptr()
        tail[1].kind, tail[1].y = _lib.OP_STEP_ADVANCE, self.step.data_ptr()
        self.tail = tail
        self.graph = None
        self.use_graph = use_graph

    def _one_step(self, stream_ptr):
        self.plan.run(stream_ptr)
        _lib.run_ops(self.tail, 2, stream_ptr)

    def run(self, x, coef_host, cond=None):
        if self.cond is not None:
            self.cond.copy_(cond)
        self.x.copy_(x)
        self.coef.copy_(coef_host, non_blocking=False)
        self.step.zero_()
        if self.use_graph and self.graph is None:
            # warm-up step outside capture (module loading), then capture ONE step
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                self._one_step(side.cuda_stream)
            torch.cuda.current_stream(x.device).wait_stream(side)
            torch.cuda.synchronize(x.device)
            self.x.copy_(x)
            self.step.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._one_step(torch.cuda.current_stream(x.device).cuda_stream)
            self.graph = g
            # capture does not execute: state is still at step 0
        stream = torch.cuda.current_stream(x.device)
        sp = stream.cuda_stream
        # The host enqueues a replay in microseconds and a step takes 15-90 ms, so without a bound all `steps` launches
        # (100 x ~300 kernel nodes) sit in the queue at once.  That is legal, and it is also what made the process die
        # with SIGSEGV inside rocprofv3's interception frames (fp32, 100 steps; DESIGN.md section 8: not with 20 steps,
        # not in an eager loop, not when the run-ahead is bounded).  Keep at most `replay_window` steps in flight: the
        # host waits for the event of step i - window before it enqueues step i, so the queue never runs dry either.
        window = int(self.replay_window or 0)
        events = []
        for i in range(self.steps):
            if window and i >= window:
                events[i - window].synchronize()
            if self.graph is not None:
                self.graph.replay()
            else:
                self._one_step(sp)
            if window and i + window < self.steps:
                ev = torch.cuda.Event()
                ev.record(stream)
                events.append(ev)
            if i == 0:
                # The fused conv + GroupNorm launches wait, bounded, for the other tiles of a sample; a wait that ran out
                # (or tickets out of step) leaves an error word instead of a hang.  A GPU that cannot run them (shared
                # with another process, partitioned) shows it on the FIRST step: look now - one 4-byte read per
                # sample() call - instead of paying the bounded waits of every remaining step first.
                self._check()
        out = self.pred.clone()
        self._check()                 # ... and never hand back audio computed past a flagged launch
        return out

    def close(self):
        """Destroy the captured graph while everything its nodes point at (arena, coefficient table, step counter, x / v /
        pred) is still allocated."""
        if self.graph is not None:
            self.graph.reset()
            self.graph = None

    def _check(self):
        err = self.plan.gn_error(clear=True)          # the host clears the sticky word where it raises
        if err:
            raise FusedLaunchError(f"libadk: a fused conv + GroupNorm launch flagged error {err:#x} during sample() "
                                   "(a tile waited in vain for its sample's statistics); the output was discarded")


@torch.no_grad()
def sample(model, x, steps, eta, **extra_args):
    """Draws samples from a model given starting noise (reference sampling.py:16-55).

    Returns the LAST step's ``pred``; ``x`` is not modified; ``randn_like`` is drawn only when ``eta`` is
    truthy; ``extra_args`` are forwarded to ``model`` (generic path)."""
    return _sample(model, x, steps, eta, None, extra_args)


@torch.no_grad()
def sample_crash(model, x, steps, eta):
    """The training demos' sampler: same loop on the crash-warped timesteps (reference train_uncond.py:45-86)."""
    return _sample(model, x, steps, eta, get_crash_schedule, {})


@torch.no_grad()
def sample_diffae(model, signal, steps, eta):
    """Diffusion-autoencoder DDIM (reference diffusion/inference.py:14-57): ``z_sem = model.encode(signal)``,
    start from fresh noise, denoise with ``model.decode(x, t, z_sem)``.  ``model`` is any object with those two
    methods; the per-step update runs in the fused kernel."""
    z_sem = model.encode(signal)
    x = torch.randn_like(signal)
    return _sample(lambda xx, tt: model.decode(xx, tt, z_sem), x, steps, eta, None, {})


@torch.no_grad()
def audio_to_int16(fakes):
    """``rearrange(fakes, 'b d n -> d (b n)').clamp(-1, 1).mul(32767).to(torch.int16)`` in one kernel — the output
    stage every caller of ``sample`` applies before saving (reference train_uncond.py:182-187)."""
    if not fakes.is_cuda or fakes.dim() != 3:
        raise RuntimeError("audio_to_int16 expects a [B, C, T] tensor on a ROCm device")
    lib = _lib.load()
    x = fakes.detach().to(torch.float32).contiguous()
    B, C, T = x.shape
    y = torch.empty(C, B * T, dtype=torch.int16, device=x.device)
    _lib.check(lib.adk_audio_to_int16(B, C, T, x.data_ptr(), y.data_ptr(),
                                      torch.cuda.current_stream(x.device).cuda_stream), "adk_audio_to_int16")
    return y


def _sample(model, x, steps, eta, warp, extra_args):
    if not x.is_cuda:
        raise RuntimeError("sample(): MI355X build runs on ROCm devices only; there is no CPU path")
    if steps < 1:
        raise ValueError("steps must be >= 1")
    coef = step_coefficients(steps, eta, warp)
    x = x.detach().to(torch.float32).contiguous()
    # fused (captured) loop: our denoiser, eta = 0, and at most a constant `cond` (no per-step RNG: cond_noise_aug
    # redraws its noise on every call in the reference, so it takes the generic path)
    fused = isinstance(model, DiffusionAttnUnet1D) and set(extra_args) <= {"cond"} and \
        not (extra_args and model.cond_noise_aug)
    if fused and not eta:
        model.check_input(x)
        cond = model.prepare_cond(x, extra_args.get("cond"))[0]
        dtype = model.resolve_dtype(in_sampler=True)
        cshape = tuple(cond.shape) if cond is not None else None
        key = ("sampler", x.shape[0], x.shape[2], dtype, str(x.device), steps, cshape)
        for attempt in (0, 1):
            smp = model._plans.get(key)
            if smp is None:
                smp = _FusedSampler(model, x, steps, dtype, use_graph=getattr(model, "use_graph", True), cond_shape=cshape)
                model._plans[key] = smp
            try:
                return smp.run(x, coef, cond)
            except FusedLaunchError as e:
                # The in-launch exchange of GroupNorm statistics assumes this process has the GPU's block slots to itself
                # (DESIGN.md section 4).  When it does not (another process on the same GPU), fall back ONCE to the plan with
                # GroupNorm as passes of its own - slower, no in-kernel waiting - and say so.
                if attempt or not getattr(model, "fuse_gn", True):
                    raise
                import warnings
                warnings.warn(f"{e}; re-running with model.fuse_gn = False (GroupNorm as separate passes)")
                model.fuse_gn = False
                model.release_plans() if hasattr(model, "release_plans") else model._plans.clear()

    # generic path: any callable model (or eta > 0); the update still runs as one fused kernel
    lib = _lib.load()
    dev = x.device
    coef_d = coef.to(dev)
    step_d = torch.zeros(1, dtype=torch.int32, device=dev)
    xs = x.clone()
    pred = torch.empty_like(xs)
    ts = xs.new_ones([xs.shape[0]])
    prev = getattr(model, "compute_dtype", None)
    auto = isinstance(model, DiffusionAttnUnet1D) and prev is None
    if auto:
        model.compute_dtype = model.resolve_dtype(in_sampler=True)      # the reference wraps the call in autocast (:30)
    try:
        for i in range(steps):
            v = model(xs, ts * coef[i, 0].item(), **extra_args).float().contiguous()
            noise = torch.randn_like(xs) if (eta and i < steps - 1) else None
            sp = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.adk_ddim_update(_lib.FLAG_NOISE if noise is not None else 0, xs.numel(), xs.data_ptr(),
                                           v.data_ptr(), noise.data_ptr() if noise is not None else None,
                                           pred.data_ptr(), coef_d.data_ptr(), step_d.data_ptr(), sp), "adk_ddim_update")
            _lib.check(lib.adk_step_advance(step_d.data_ptr(), sp), "adk_step_advance")
    finally:
        if auto:
            model.compute_dtype = prev
    # (a flagged fused launch inside model(...) has already raised or fallen back there: DiffusionAttnUnet1D.forward)
    return pred


# ---- Karras-ODE samplers for the k-diffusion denoiser family ------------------------------------------
# Host-side loops around ``denoiser(x, sigma)`` (kdiffusion.Denoiser / AudioDenoiserModel): the schedules and the
# per-step updates are a handful of elementwise torch ops on the device; all the work is in the denoiser call.
def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device="cpu"):
    """Karras et al. (2022) noise schedule with a trailing 0 (reference diffusion/inference.py:60-67)."""
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sig = (hi + torch.linspace(0, 1, n) * (lo - hi)) ** rho
    return torch.cat([sig, sig.new_zeros([1])]).to(device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device="cpu"):
    """Log-linear noise schedule with a trailing 0 (reference diffusion/inference.py:70-73)."""
    sig = torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp()
    return torch.cat([sig, sig.new_zeros([1])])


def to_d(x, sigma, denoised):
    """ODE derivative dx/dsigma from a denoiser output (reference diffusion/inference.py:94-96)."""
    return (x - denoised) / sigma.reshape(sigma.shape + (1,) * (x.ndim - sigma.ndim))


@torch.no_grad()
def sample_heun(model, x, sigmas, extra_args=None, callback=None, disable=None, second_order=True, s_churn=0.):
    """Heun (2nd order) / Euler sampler of the Karras ODE, drop-in for reference diffusion/inference.py:99-122
    (``disable`` is accepted for signature parity; there is no progress bar)."""
    extra_args = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    n = len(sigmas) - 1
    gamma = min(s_churn / n, 2 ** 0.5 - 1) if n else 0.
    for i in range(n):
        noise = torch.randn_like(x)                       # drawn every step like the reference (keeps its RNG stream)
        s_hat = sigmas[i] * (gamma + 1)
        if gamma > 0:
            x = x + noise * (s_hat ** 2 - sigmas[i] ** 2) ** 0.5
        den = model(x, s_hat * ones, **extra_args)
        d = to_d(x, s_hat, den)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": s_hat, "denoised": den})
        dt = sigmas[i + 1] - s_hat
        if sigmas[i + 1] == 0 or not second_order:
            x = x + d * dt                                # Euler step (always the last one)
            continue
        x_mid = x + d * dt
        d_mid = to_d(x_mid, sigmas[i + 1], model(x_mid, sigmas[i + 1] * ones, **extra_args))
        x = x + (d + d_mid) / 2 * dt
    return x


def _lms_coeff(order, t, i, j):
    """Integral of the j-th Lagrange basis polynomial over [t_i, t_{i+1}] (reference diffusion/inference.py:125-135)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        p = 1.
        for k in range(order):
            if k != j:
                p *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return p
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, x, sigmas, extra_args=None, callback=None, disable=None, order=4):
    """Linear multistep sampler, drop-in for reference diffusion/inference.py:138-154."""
    extra_args = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    t = sigmas.cpu()
    hist = []
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra_args)
        hist = (hist + [to_d(x, sigmas[i], den)])[-order:]
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": den})
        k = min(i + 1, order)
        x = x + sum(_lms_coeff(k, t, i, j) * d for j, d in zip(range(k), reversed(hist)))
    return x
