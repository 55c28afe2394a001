"""GaussianDiffusion / DiffusionDDIMSR — sigma-parametrised VP diffusion with the DPM-Solver++(2M)
SDE sampler and DDIM inversion, MI355X-native (configs 3/5 of BASELINE.json).

Interface parity (reference: tools/modules/diffusions/diffusion_gauss.py:145-464 GaussianDiffusion,
:85-142 sample_dpmpp_2m_sde, :22-76 Brownian noise; diffusion_ddim.py:18-25 DiffusionDDIMSR): same
constructor arguments, same `sample(...)`, `denoise(...)`, `diffuse(...)`,
`ddim_reverse_sample(_loop)(...)`, `_sigma_to_t/_t_to_sigma` signatures and return values; the model
is still called as `model(xt, t=t, **kwargs)` (any callable).

Underneath: classifier-free guidance, guide_rescale (two per-sample std reductions), the
x0 / eps algebra and the exponential-integrator update run as fused HIP kernels (vgen_cfg_stats,
vgen_gauss_x0, vgen_dpmpp2m_sde_step — one launch per solver update —, vgen_lincomb4); the per-step scalars (sigma ratios, expm1 terms) are computed on the
host in fp32 exactly as the reference's 0-dim tensor arithmetic; cond/uncond evaluate as one batch
when the model exposes `forward_units`.

Transcription note: the HOST-side scalar algebra of this file — the step-list preamble of `sample()` (discretisation,
sigma interpolation `_sigma_to_t / _t_to_sigma`, the penultimate-sigma drop) and the per-step solver coefficients — follows
the reference's expressions statement by statement (diffusion_gauss.py:249-373, :436-464, :113-139): these are 0-dim fp32
tensor computations whose rounding order decides the timestep the UNet is asked for, so they are restated, not
redesigned.  Everything that touches a latent-sized tensor is this repo's own kernels.

Brownian noise: the reference draws the SDE noise from `torchsde.BrownianTree` (torchsde==0.2.6,
third-party, not vendored -> parity of the stochastic term is unpinned).  torchsde is
used when importable; otherwise `IntervalNoise` draws one N(0, I) sample per (consecutive,
non-overlapping) step interval from a seeded generator, which has the same distribution.  A sampler
can also be injected (`noise_sampler_cls`) — the tests inject the same deterministic stub on both
sides to compare the full update.
"""
from __future__ import annotations

import random

import torch

from . import ops
from .schedules import sigma_schedule
from .session import SessionCache, eval_units

_PRED = {"eps": 0, "v": 1, "x0": 2}


def _i(tensor, t, x):
    shape = (x.size(0),) + (1,) * (x.ndim - 1)
    return tensor[t.to(tensor.device)].view(shape).to(x.device)


class IntervalNoise:
    """N(0, I) per requested interval, seeded; stands in for BrownianTreeNoiseSampler when torchsde
    is unavailable (identical in distribution for the solver's consecutive intervals)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda v: v):
        self.shape, self.device, self.dtype = x.shape, x.device, x.dtype
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(int(seed) if seed is not None else random.randint(0, 2 ** 31))

    def __call__(self, sigma, sigma_next):
        return torch.randn(self.shape, generator=self.gen, dtype=torch.float32).to(self.device)


def _default_noise_sampler_cls():
    try:
        import torchsde  # noqa: F401
    except Exception:
        return IntervalNoise

    class _Tree:
        """BrownianTreeNoiseSampler of the reference (diffusion_gauss.py:52-76) on torchsde."""

        def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda v: v):
            import torchsde
            self.transform = transform
            t0, t1 = transform(torch.as_tensor(sigma_min)), transform(torch.as_tensor(sigma_max))
            t0, t1, self.sign = (t0, t1, 1) if t0 < t1 else (t1, t0, -1)
            if seed is None:
                seed = torch.randint(0, 2 ** 63 - 1, []).item()
            self.tree = torchsde.BrownianTree(t0, torch.zeros_like(x), t1, entropy=seed)

        def __call__(self, sigma, sigma_next):
            t0, t1 = self.transform(torch.as_tensor(sigma)), self.transform(torch.as_tensor(sigma_next))
            a, b, sign = (t0, t1, 1) if t0 < t1 else (t1, t0, -1)
            return self.tree(a, b) * (self.sign * sign) / (t1 - t0).abs().sqrt()

    return _Tree


class GaussianDiffusion(object):
    def __init__(self, sigmas, prediction_type="eps"):
        assert prediction_type in {"x0", "eps", "v"}
        self.sigmas = sigmas.float()
        self.alphas = torch.sqrt(1 - sigmas ** 2).float()
        self.num_timesteps = len(sigmas)
        self.prediction_type = prediction_type
        self.partition = None
        self.sessions = SessionCache()         # cached sampling sessions (vgen_amd/session.py): one replay per step
        self.noise_sampler_cls = None          # default: torchsde tree if importable, else IntervalNoise

    # -- q(x_t | x_0) ---------------------------------------------------------------------------------
    def diffuse(self, x0, t, noise=None):
        noise = torch.randn_like(x0) if noise is None else noise
        return _i(self.alphas, t, x0) * x0 + _i(self.sigmas, t, x0) * noise

    # -- model evaluation -------------------------------------------------------------------------------
    def _eval(self, xt, t, model, model_kwargs, guide_scale):
        if guide_scale is None:
            assert isinstance(model_kwargs, dict)
            return model(xt, t=t, **model_kwargs), None
        assert isinstance(model_kwargs, list) and len(model_kwargs) == 2
        if guide_scale == 1.:
            return model(xt, t=t, **model_kwargs[0]), None
        outs = eval_units(self.sessions, self.partition, model, xt, t, model_kwargs, self.num_timesteps)
        if outs is not None:
            return outs
        return model(xt, t=t, **model_kwargs[0]), model(xt, t=t, **model_kwargs[1])

    def _x0_eps(self, xt, t, model, model_kwargs, guide_scale, guide_rescale, clamp, percentile, want_eps):
        if clamp is not None or percentile is not None:
            raise NotImplementedError("clamp / percentile are unused by the inference configs")
        y_out, u_out = self._eval(xt, t, model, model_kwargs, guide_scale)
        if u_out is None:
            guide_rescale = None               # reference: rescale only inside the CFG branch (:208-218)
        else:
            assert guide_rescale is None or 0 <= guide_rescale <= 1
        coef = torch.stack([self.alphas.to(xt.device)[t], self.sigmas.to(xt.device)[t]], dim=1).contiguous()
        f32 = lambda v: None if v is None else v.float().contiguous()
        return ops.backend().gauss_denoise(f32(xt), f32(y_out), f32(u_out),
                                           0.0 if guide_scale is None else float(guide_scale),
                                           guide_rescale, coef.float(), _PRED[self.prediction_type], want_eps)

    def denoise(self, xt, t, s, model, model_kwargs={}, guide_scale=None, guide_rescale=None, clamp=None,
                percentile=None):
        """Returns (mu, var, log_var, x0, eps) like the reference (diffusion_gauss.py:163-247)."""
        s = t - 1 if s is None else s
        sigmas, alphas = _i(self.sigmas, t, xt), _i(self.alphas, t, xt)
        alphas_s = _i(self.alphas, s.clamp(0), xt)
        alphas_s[s < 0] = 1.
        sigmas_s = torch.sqrt(1 - alphas_s ** 2)
        betas = 1 - (alphas / alphas_s) ** 2
        coef1 = betas * alphas_s / sigmas ** 2
        coef2 = (alphas * sigmas_s ** 2) / (alphas_s * sigmas ** 2)
        var = betas * (sigmas_s / sigmas) ** 2
        log_var = torch.log(var).clamp_(-20, 20)
        x0, eps = self._x0_eps(xt, t, model, model_kwargs, guide_scale, guide_rescale, clamp, percentile, True)
        mu = coef1 * x0 + coef2 * xt
        return mu, var, log_var, x0, eps

    # -- sigma <-> t ----------------------------------------------------------------------------------------
    def _log_sigmas(self):
        return torch.sqrt(self.sigmas ** 2 / (1 - self.sigmas ** 2)).log()

    def _sigma_to_t(self, sigma):
        if sigma == float("inf"):
            t = torch.full_like(sigma, len(self.sigmas) - 1)
        else:
            log_sigmas = self._log_sigmas().to(sigma)
            log_sigma = sigma.log()
            dists = log_sigma - log_sigmas[:, None]
            low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=log_sigmas.shape[0] - 2)
            high_idx = low_idx + 1
            low, high = log_sigmas[low_idx], log_sigmas[high_idx]
            w = ((low - log_sigma) / (low - high)).clamp(0, 1)
            t = ((1 - w) * low_idx + w * high_idx).view(sigma.shape)
        if t.ndim == 0:
            t = t.unsqueeze(0)
        return t

    def _t_to_sigma(self, t):
        t = t.float()
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        log_sigmas = self._log_sigmas().to(t)
        log_sigma = (1 - w) * log_sigmas[low_idx] + w * log_sigmas[high_idx]
        log_sigma[torch.isnan(log_sigma) | torch.isinf(log_sigma)] = float("inf")
        return log_sigma.exp()

    # -- DPM-Solver++(2M) SDE -------------------------------------------------------------------------------------
    @torch.no_grad()
    def _dpmpp_2m_sde(self, noise, model_fn, sigmas, eta=1., s_noise=1., solver_type="midpoint", seed=None):
        """reference: sample_dpmpp_2m_sde (diffusion_gauss.py:85-142).  `sigmas` is a CPU fp32 vector; the
        per-step scalars follow the reference's expressions, the tensor updates are vgen_lincomb4."""
        assert solver_type in {"heun", "midpoint"}
        be = ops.backend()
        sig = sigmas.detach().float().cpu()
        noise = noise.float().contiguous()
        x = be.lincomb4(noise, None, None, None, float(sig[0]), 0, 0, 0)
        sigma_min, sigma_max = sig[sig > 0].min(), sig[sig < float("inf")].max()
        cls = self.noise_sampler_cls or _default_noise_sampler_cls()
        # the reference never forwards `seed` to its BrownianTree sampler (entropy comes from torch's global
        # RNG, diffusion_gauss.py:31-32); only the torchsde-free fallback is seeded explicitly
        sampler = cls(x, sigma_min, sigma_max, seed=seed) if cls is IntervalNoise else cls(x, sigma_min, sigma_max)
        old_denoised, h_last = None, None
        for i in range(len(sig) - 1):
            if sig[i] == float("inf"):
                denoised = model_fn(noise, sig[i])
                x = be.lincomb4(denoised, noise, None, None, 1.0, float(sig[i + 1]), 0, 0)
                h = None
            else:
                c_in = 1 / (sig[i] ** 2 + 1. ** 2) ** 0.5
                denoised = model_fn(be.lincomb4(x, None, None, None, float(c_in), 0, 0, 0), sig[i])
                if sig[i + 1] == 0:
                    x = denoised
                    h = None
                else:
                    t, s = -sig[i].log(), -sig[i + 1].log()
                    h = s - t
                    eta_h = eta * h
                    ca = float(sig[i + 1] / sig[i] * (-eta_h).exp())
                    cb = float((-h - eta_h).expm1().neg())
                    cc, use_old = 0.0, old_denoised is not None
                    if use_old:
                        r = h_last / h
                        if solver_type == "heun":
                            cc = float(((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r))
                        else:
                            cc = float(0.5 * (-h - eta_h).expm1().neg() * (1 / r))
                    cn = (float(sig[i + 1]), float((-2 * eta_h).expm1().neg().sqrt()), float(s_noise))
                    nz = None
                    if cn[0] * cn[1] * cn[2] != 0.0:
                        nz = sampler(sig[i], sig[i + 1]).to(device=x.device, dtype=torch.float32).contiguous()
                    # exponential-integrator step + 2M correction + noise injection: ONE launch (vgen_dpmpp2m_sde_step)
                    x = be.dpmpp2m_sde_step(x, denoised, old_denoised if use_old else None, nz, ca, cb, cc, cn)
            old_denoised = denoised
            h_last = h
        return x

    @torch.no_grad()
    def sample(self, noise, model, model_kwargs={}, condition_fn=None, guide_scale=None, guide_rescale=None,
               clamp=None, percentile=None, solver="euler_a", steps=20, t_max=None, t_min=None,
               discretization=None, discard_penultimate_step=None, return_intermediate=None,
               show_progress=False, seed=-1, **kwargs):
        assert isinstance(steps, (int, torch.LongTensor))
        assert t_max is None or (0 < t_max <= self.num_timesteps - 1)
        assert t_min is None or (0 <= t_min < self.num_timesteps - 1)
        assert discretization in (None, "leading", "linspace", "trailing")
        assert discard_penultimate_step in (None, True, False)
        assert return_intermediate in (None, "x0", "xt")
        if solver != "dpmpp_2m_sde":
            raise KeyError(solver)            # the reference's solver table holds only this entry (:281-284)
        discretization = discretization or "linspace"
        seed = seed if seed >= 0 else random.randint(0, 2 ** 31)
        if isinstance(steps, torch.LongTensor):
            discard_penultimate_step = False
        if discard_penultimate_step is None:
            discard_penultimate_step = True

        intermediates = []

        def model_fn(xt, sigma):
            t = self._sigma_to_t(sigma).repeat(len(xt)).round().long().to(xt.device)
            x0, _ = self._x0_eps(xt, t, model, model_kwargs, guide_scale, guide_rescale, clamp, percentile, False)
            if return_intermediate == "xt":
                intermediates.append(xt)
            elif return_intermediate == "x0":
                intermediates.append(x0)
            return x0

        if isinstance(steps, int):
            steps += 1 if discard_penultimate_step else 0
            t_max = self.num_timesteps - 1 if t_max is None else t_max
            t_min = 0 if t_min is None else t_min
            if discretization == "leading":
                steps = torch.arange(t_min, t_max + 1, (t_max - t_min + 1) / steps).flip(0)
            elif discretization == "linspace":
                steps = torch.linspace(t_max, t_min, steps)
            else:
                steps = torch.arange(t_max, t_min - 1, -((t_max - t_min + 1) / steps))
            steps = steps.clamp_(t_min, t_max)
        steps = torch.as_tensor(steps, dtype=torch.float32)
        sigmas = self._t_to_sigma(steps)
        sigmas = torch.cat([sigmas, sigmas.new_zeros([1])])
        if discard_penultimate_step:
            sigmas = torch.cat([sigmas[:-2], sigmas[-1:]])
        x0 = self._dpmpp_2m_sde(noise, model_fn, sigmas, seed=seed, **kwargs)
        return (x0, intermediates) if return_intermediate is not None else x0

    # -- DDIM inversion -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def ddim_reverse_sample(self, xt, t, model, model_kwargs={}, clamp=None, percentile=None, guide_scale=None,
                            guide_rescale=None, ddim_timesteps=20, reverse_steps=600):
        stride = reverse_steps // ddim_timesteps
        x0, eps = self._x0_eps(xt, t, model, model_kwargs, guide_scale, guide_rescale, clamp, percentile, True)
        s = (t + stride).clamp(0, reverse_steps - 1)
        alphas_s = _i(self.alphas, s.clamp(0), xt)
        alphas_s[s < 0] = 1.
        sigmas_s = torch.sqrt(1 - alphas_s ** 2)
        a, b = alphas_s.flatten(), sigmas_s.flatten()
        if bool((a == a[0]).all()) and bool((b == b[0]).all()):
            mu = ops.backend().lincomb4(x0, eps, None, None, float(a[0]), float(b[0]), 0, 0)
        else:                                   # per-sample timesteps: rare, not on the engine path
            mu = alphas_s * x0 + sigmas_s * eps
        return mu, x0

    @torch.no_grad()
    def ddim_reverse_sample_loop(self, x0, model, model_kwargs={}, clamp=None, percentile=None, guide_scale=None,
                                 guide_rescale=None, ddim_timesteps=20, reverse_steps=600):
        b = x0.size(0)
        xt = x0
        for step in torch.arange(0, reverse_steps, reverse_steps // ddim_timesteps):
            t = torch.full((b,), int(step), dtype=torch.long, device=xt.device)
            xt, _ = self.ddim_reverse_sample(xt, t, model, model_kwargs, clamp, percentile, guide_scale,
                                             guide_rescale, ddim_timesteps, reverse_steps)
        return xt


class DiffusionDDIMSR(object):
    """reference: DiffusionDDIMSR (diffusion_ddim.py:18-25): two GaussianDiffusions read from attribute-style
    sub-configs (`.schedule`, `.schedule_param`, `.mean_type`)."""

    def __init__(self, reverse_diffusion, forward_diffusion, **kwargs):
        g = lambda c, k: c[k] if isinstance(c, dict) else getattr(c, k)
        self.reverse_diffusion = GaussianDiffusion(
            sigmas=sigma_schedule(g(reverse_diffusion, "schedule"), **g(reverse_diffusion, "schedule_param")),
            prediction_type=g(reverse_diffusion, "mean_type"))
        self.forward_diffusion = GaussianDiffusion(
            sigmas=sigma_schedule(g(forward_diffusion, "schedule"), **g(forward_diffusion, "schedule_param")),
            prediction_type=g(forward_diffusion, "mean_type"))
