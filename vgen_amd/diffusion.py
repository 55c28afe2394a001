"""DiffusionDDIM — sampling side of the reference's DDIM diffusion, MI355X-native.

Interface parity (reference: tools/modules/diffusions/diffusion_ddim.py:28-78 ctor,
:147-206 p_mean_variance, :208-241 ddim_sample, :243-254 ddim_sample_loop,
:256-288 ddim_reverse_sample(_loop), :91-97 q_sample, :507-511 _scale_timesteps):
same registry name, same constructor keywords, same method signatures / return values, the
sampler still accepts ANY `model(xt, t, **kwargs)` callable.

What is different underneath:
  * the float64 schedule tables are cast to fp32 once and cached per device (the reference
    re-uploads a float64 table on every `_i()` call, diffusion_ddim.py:13-16),
  * classifier-free guidance + the v/eps/x0 algebra + the DDIM update are ONE fused HIP kernel
    (vgen_cfg_ddim_step) that reproduces the reference's fp32 operation order bit for bit,
  * when the model exposes `forward_units` (vgen_amd.unet) the cond/uncond pair is evaluated as
    one batch (weights stream from HBM once per step), and when a UnitPartition is attached the
    units are spread over the ranks with a single all-gather per step (vgen_amd/parallel.py).
The ancestral (DDPM) sampler p_sample / p_sample_loop (:116-144) and the closed-form q(.) helpers
(:99-114) ride on the same fused x0 evaluation.  Training-only members of the reference class (loss, VLB)
are out of scope; its plms_sample cannot run as shipped (`eps_cache` is read at
:335 but is not a parameter of :290) and has no caller, so it is not restated.
"""
from __future__ import annotations


import torch

from . import ops
from .schedules import beta_schedule
from .session import SessionCache, eval_units

_MEAN = {"eps": 0, "v": 1, "x0": 2}
# rows of the per-device fp32 table
_AC, _SQRT_AC, _SQRT_1M, _SQRT_RECIP, _SQRT_RECIPM1, _AC_NEXT = range(6)


class DiffusionDDIM(object):
    def __init__(self, schedule="linear_sd", schedule_param={}, mean_type="eps",
                 var_type="learned_range", loss_type="mse", epsilon=1e-12,
                 rescale_timesteps=False, noise_strength=0.0, **kwargs):
        assert mean_type in ["x0", "x_{t-1}", "eps", "v"]
        assert var_type in ["learned", "learned_range", "fixed_large", "fixed_small"]
        betas = beta_schedule(schedule, **schedule_param)
        assert min(betas) > 0 and max(betas) <= 1
        if not isinstance(betas, torch.DoubleTensor):
            betas = torch.tensor(betas, dtype=torch.float64)
        self.betas = betas
        self.num_timesteps = len(betas)
        self.mean_type, self.var_type, self.loss_type = mean_type, var_type, loss_type
        self.epsilon, self.rescale_timesteps, self.noise_strength = epsilon, rescale_timesteps, noise_strength

        alphas = 1 - self.betas
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.alphas_cumprod_prev = torch.cat([alphas.new_ones([1]), self.alphas_cumprod[:-1]])
        self.alphas_cumprod_next = torch.cat([self.alphas_cumprod[1:], alphas.new_zeros([1])])
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = torch.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(self.posterior_variance.clamp(1e-20))
        self.posterior_mean_coef1 = betas * torch.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - self.alphas_cumprod)

        # construction is as permissive as the reference's (its default var_type is 'learned_range', :34): DDIM sampling
        # never reads var_type; the ancestral p_sample / p_mean_variance do and say so for the learned variants there
        self._tab = {}
        self._coef_tabs = {}
        self.sessions = SessionCache()  # per-(model, kwarg sets, shape) sampling sessions (vgen_amd/session.py)
        self.partition = None          # optional vgen_amd.parallel.UnitPartition
        self.rng_parity = True         # draw the (unused when eta == 0) per-step noise like the reference

    # -- tables ---------------------------------------------------------------------------------
    def _table(self, device):
        tab = self._tab.get(device)
        if tab is None:
            ac_ext = torch.cat([self.alphas_cumprod, self.alphas_cumprod.new_zeros([1])])
            pad = lambda v: torch.cat([v, v.new_zeros([1])])
            rows = [ac_ext, pad(self.sqrt_alphas_cumprod), pad(self.sqrt_one_minus_alphas_cumprod),
                    pad(self.sqrt_recip_alphas_cumprod), pad(self.sqrt_recipm1_alphas_cumprod), ac_ext]
            tab = torch.stack(rows).to(torch.float32).to(device)     # fp64 -> fp32 like `_i(...).to(x)`
            self._tab[device] = tab
        return tab

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * 1000.0 / self.num_timesteps
        return t

    # -- q(x_t | x_0) ---------------------------------------------------------------------------
    def q_sample(self, x0, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x0)
            if self.noise_strength > 0:
                b, c, f, _, _ = x0.shape
                noise = noise + self.noise_strength * torch.randn(b, c, f, 1, 1, device=x0.device)
        tab = self._table(x0.device)
        shape = (x0.size(0),) + (1,) * (x0.ndim - 1)
        return tab[_SQRT_AC][t].view(shape) * x0 + tab[_SQRT_1M][t].view(shape) * noise

    def _g(self, table, t, x):
        """table[t] as fp32 on x's device, broadcastable over x (the reference's `_i`, diffusion_ddim.py:10-16)."""
        shape = (x.size(0),) + (1,) * (x.ndim - 1)
        return table.to(x.device)[t].view(shape).to(x)

    def q_mean_variance(self, x0, t):
        """Distribution of q(x_t | x_0) (diffusion_ddim.py:99-105)."""
        return (self._g(self.sqrt_alphas_cumprod, t, x0) * x0, self._g(1.0 - self.alphas_cumprod, t, x0),
                self._g(self.log_one_minus_alphas_cumprod, t, x0))

    def q_posterior_mean_variance(self, x0, xt, t):
        """Distribution of q(x_{t-1} | x_t, x_0) (diffusion_ddim.py:107-114)."""
        mu = self._g(self.posterior_mean_coef1, t, xt) * x0 + self._g(self.posterior_mean_coef2, t, xt) * xt
        return mu, self._g(self.posterior_variance, t, xt), self._g(self.posterior_log_variance_clipped, t, xt)

    # -- model evaluation -------------------------------------------------------------------------
    def _eval_model(self, xt, t, model, model_kwargs, guide_scale):
        ts = self._scale_timesteps(t)
        if guide_scale is None:
            return model(xt, ts, **model_kwargs), None
        assert isinstance(model_kwargs, list) and len(model_kwargs) == 2
        outs = eval_units(self.sessions, self.partition, model, xt, ts, model_kwargs,
                          None if self.rescale_timesteps else self.num_timesteps)
        if outs is not None:
            return outs
        return model(xt, ts, **model_kwargs[0]), model(xt, ts, **model_kwargs[1])

    def _x0_coefs(self, tab, t):
        if self.mean_type == "v":
            return tab[_SQRT_AC][t], tab[_SQRT_1M][t]
        if self.mean_type == "eps":
            return tab[_SQRT_RECIP][t], tab[_SQRT_RECIPM1][t]
        if self.mean_type == "x0":
            z = torch.zeros_like(tab[_AC][t])
            return z, z
        raise NotImplementedError("mean_type 'x_{t-1}' takes the general path (_p_mean_variance_general)")

    def _coef_rows(self, tab, t, kind, stride, eta):
        """The 7 coefficients of vgen_cfg_ddim_step for timesteps `t` (any shape of long indices), fp32, in the
        reference's expression order (diffusion_ddim.py:232-234 for 'ddim', :262-270 for 'reverse'; 'x0' is
        p_mean_variance alone: alpha_prev = 1, sigma = 0)."""
        a0, a1 = self._x0_coefs(tab, t)
        alphas = tab[_AC][t]
        if kind == "ddim":
            alphas_prev = tab[_AC][(t - stride).clamp(0)]
            sigmas = eta * torch.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
            mask = t.ne(0).float()
        elif kind == "reverse":
            alphas_prev = tab[_AC_NEXT][(t + stride).clamp(0, self.num_timesteps)]
            sigmas = torch.zeros_like(alphas_prev)
            mask = torch.zeros_like(alphas_prev)
        else:
            alphas_prev = torch.ones_like(alphas)
            sigmas = torch.zeros_like(alphas)
            mask = torch.zeros_like(alphas)
        return torch.stack([a0, a1, tab[_SQRT_RECIP][t], tab[_SQRT_RECIPM1][t], alphas_prev, sigmas, mask],
                           dim=-1).contiguous()

    def _coef_table(self, device, kind, stride, eta):
        """Coefficient rows of ALL integer timesteps ([T, 7]): a sampling session's step graph gathers row t on
        the device, so a step needs no host-side scalar work."""
        key = (str(device), kind, int(stride), float(eta))
        tabc = self._coef_tabs.get(key)
        if tabc is None:
            t = torch.arange(self.num_timesteps, dtype=torch.long, device=device)
            tabc = self._coef_rows(self._table(device), t, kind, stride, eta)
            self._coef_tabs[key] = tabc
        return tabc

    def _session(self, xt, t, model, model_kwargs, guide_scale):
        """The UnitSession for this (model, kwarg sets, latent shape), or None: arbitrary callables, float /
        rescaled timesteps and partitioned steps take the eager path."""
        if self.sessions is None or self.partition is not None or self.rescale_timesteps \
                or t.dtype != torch.long or xt.dim() != 5:
            return None
        kws = model_kwargs if guide_scale is not None else [model_kwargs]
        if not isinstance(kws, (list, tuple)) or len(kws) != (2 if guide_scale is not None else 1):
            return None
        if not all(isinstance(kw, dict) for kw in kws):
            return None
        return self.sessions.get(model, tuple(xt.shape), xt.device, list(kws), torch.long, self.num_timesteps)

    def _fused(self, xt, t, model, model_kwargs, guide_scale, kind, stride, eta, noise, clamp, percentile,
               alias_ok=False):
        """(x_next, x0) = fused CFG + x0 + update of `kind` ('ddim' | 'reverse' | 'x0') after evaluating the model."""
        if clamp is not None or percentile is not None:
            raise NotImplementedError("clamp / percentile take the general path (_p_mean_variance_general)")
        sess = self._session(xt, t, model, model_kwargs, guide_scale)
        if sess is not None:
            return sess.ddim_step(xt, t, self._coef_table(xt.device, kind, stride, eta),
                                  0.0 if guide_scale is None else float(guide_scale), _MEAN[self.mean_type],
                                  noise, clone=not alias_ok)
        part = self.partition
        if part is not None and getattr(part, "graph_collective", False) and guide_scale is not None and noise is None \
                and not self.rescale_timesteps and isinstance(model_kwargs, (list, tuple)) and len(model_kwargs) == 2:
            # r04: the whole partitioned step (local forward -> all-gather -> update) as one launch sequence / one hipGraph
            r = part.ddim_step(model, xt, t, list(model_kwargs), self._coef_table(xt.device, kind, stride, eta),
                               float(guide_scale), _MEAN[self.mean_type], self.num_timesteps)
            if r is not None:
                return r if alias_ok else (r[0].clone(), r[1].clone())
        y_out, u_out = self._eval_model(xt, t, model, model_kwargs, guide_scale)
        coef = self._coef_rows(self._table(xt.device), t, kind, stride, eta)
        xt32 = xt.float().contiguous()
        y32 = y_out.float().contiguous()
        u32 = None if u_out is None else u_out.float().contiguous()
        return ops.backend().cfg_ddim_step(xt32, y32, u32, noise, coef,
                                           0.0 if guide_scale is None else float(guide_scale),
                                           guide_scale is not None, _MEAN[self.mean_type], True)

    # -- the options no inference config uses ------------------------------------------------------
    def _uncommon(self, clamp, percentile):
        return clamp is not None or percentile is not None or self.mean_type == "x_{t-1}" \
            or self.var_type in ("learned", "learned_range")

    def _p_mean_variance_general(self, xt, t, model, model_kwargs, clamp, percentile, guide_scale):
        """p_mean_variance with the reference's rarely used options — learned / learned_range variances (a 2 C-channel
        model output, diffusion_ddim.py:164-177), mean_type 'x_{t-1}' (:184-187), clamp / percentile on x0 (:199-204).
        The model is still evaluated through `_eval_model` (the HIP units); the algebra on the latent-sized tensors is
        written with a handful of torch elementwise ops in the reference's expression order (off the benchmarked path:
        the fused kernel covers what the inference yamls use)."""
        y_out, u_out = self._eval_model(xt, t, model, model_kwargs, guide_scale)
        if u_out is None:
            out = y_out.float()
        else:
            y_out, u_out = y_out.float(), u_out.float()
            dim = y_out.size(1) if self.var_type.startswith("fixed") else y_out.size(1) // 2
            out = torch.cat([u_out[:, :dim] + guide_scale * (y_out[:, :dim] - u_out[:, :dim]), y_out[:, dim:]], dim=1)
        xt = xt.float()
        g = lambda tab: self._g(tab, t, xt)
        if self.var_type == "learned":
            out, log_var = out.chunk(2, dim=1)
            var = torch.exp(log_var)
        elif self.var_type == "learned_range":
            out, fraction = out.chunk(2, dim=1)
            min_log_var = g(self.posterior_log_variance_clipped)
            max_log_var = g(torch.log(self.betas))
            fraction = (fraction + 1) / 2.0
            log_var = fraction * max_log_var + (1 - fraction) * min_log_var
            var = torch.exp(log_var)
        elif self.var_type == "fixed_large":
            var = g(torch.cat([self.posterior_variance[1:2], self.betas[1:]]))
            log_var = torch.log(var)
        else:
            var = g(self.posterior_variance)
            log_var = g(self.posterior_log_variance_clipped)
        if self.mean_type == "x_{t-1}":
            mu = out
            x0 = g(1.0 / self.posterior_mean_coef1) * mu - g(self.posterior_mean_coef2 / self.posterior_mean_coef1) * xt
        else:
            if self.mean_type == "x0":
                x0 = out
            elif self.mean_type == "eps":
                x0 = g(self.sqrt_recip_alphas_cumprod) * xt - g(self.sqrt_recipm1_alphas_cumprod) * out
            else:
                x0 = g(self.sqrt_alphas_cumprod) * xt - g(self.sqrt_one_minus_alphas_cumprod) * out
            mu, _, _ = self.q_posterior_mean_variance(x0, xt, t)
        if percentile is not None:
            assert percentile > 0 and percentile <= 1
            sh = (-1,) + (1,) * (x0.ndim - 1)           # the reference's view(-1, 1, 1, 1) serves 4-D images only
            sq = torch.quantile(x0.flatten(1).abs(), percentile, dim=1).clamp_(1.0).view(*sh)
            x0 = torch.min(sq, torch.max(-sq, x0)) / sq
        elif clamp is not None:
            x0 = x0.clamp(-clamp, clamp)
        return mu, var, log_var, x0

    # -- p(x_{t-1} | x_t) pieces used by the samplers ---------------------------------------------
    @torch.no_grad()
    def p_mean_variance(self, xt, t, model, model_kwargs={}, clamp=None, percentile=None, guide_scale=None):
        """Returns (mu, var, log_var, x0) like the reference; x0 comes from the fused kernel."""
        if self._uncommon(clamp, percentile):
            return self._p_mean_variance_general(xt, t, model, model_kwargs, clamp, percentile, guide_scale)
        _, x0 = self._fused(xt, t, model, model_kwargs, guide_scale, "x0", 0, 0.0, None, clamp, percentile)
        mu, var, log_var = self.q_posterior_mean_variance(x0, xt.float(), t)
        if self.var_type == "fixed_large":
            var = self._g(torch.cat([self.posterior_variance[1:2], self.betas[1:]]), t, mu)
            log_var = torch.log(var)
        return mu, var, log_var, x0

    @torch.no_grad()
    def p_sample(self, xt, t, model, model_kwargs={}, clamp=None, percentile=None, condition_fn=None, guide_scale=None):
        """One ancestral step x_t -> x_{t-1} ~ N(mu, var) (diffusion_ddim.py:116-132); returns (x_{t-1}, x0)."""
        mu, var, log_var, x0 = self.p_mean_variance(xt, t, model, model_kwargs, clamp, percentile, guide_scale)
        noise = torch.randn_like(xt)
        mask = t.ne(0).float().view(-1, *((1,) * (xt.ndim - 1)))      # no noise when t == 0
        if condition_fn is not None:                                  # classifier guidance (:127-130)
            grad = condition_fn(xt, self._scale_timesteps(t), **model_kwargs)
            mu = mu.float() + var * grad.float()
        return mu + mask * torch.exp(0.5 * log_var) * noise, x0

    @torch.no_grad()
    def p_sample_loop(self, noise, model, model_kwargs={}, clamp=None, percentile=None, condition_fn=None,
                      guide_scale=None):
        """All num_timesteps ancestral steps (diffusion_ddim.py:134-145)."""
        b = noise.size(0)
        xt = noise
        for step in torch.arange(self.num_timesteps).flip(0):
            t = torch.full((b,), int(step), dtype=torch.long, device=xt.device)
            xt, _ = self.p_sample(xt, t, model, model_kwargs, clamp, percentile, condition_fn, guide_scale)
        return xt

    @torch.no_grad()
    def ddim_sample(self, xt, t, model, model_kwargs={}, clamp=None, percentile=None, condition_fn=None,
                    guide_scale=None, ddim_timesteps=20, eta=0.0, _alias_ok=False):
        """One DDIM step (diffusion_ddim.py:208-241).  For vgen_amd models the whole step — both CFG branches as
        one UNet batch plus the fused update — is a hipGraph replay of a cached sampling session; any other
        `model` callable runs eagerly through the same update kernel."""
        stride = self.num_timesteps // ddim_timesteps
        if condition_fn is not None or self._uncommon(clamp, percentile):
            return self._ddim_sample_general(xt, t, model, model_kwargs, clamp, percentile, condition_fn, guide_scale,
                                             stride, eta)
        noise = None
        if self.rng_parity or eta != 0.0:
            noise = torch.randn_like(xt)             # drawn every step by the reference (:237)
        if eta == 0.0:
            noise = None                             # sigma == 0: the term is exactly +0
        return self._fused(xt, t, model, model_kwargs, guide_scale, "ddim", stride, eta,
                           noise if noise is None else noise.float().contiguous(), clamp, percentile,
                           alias_ok=_alias_ok)

    def _ddim_sample_general(self, xt, t, model, model_kwargs, clamp, percentile, condition_fn, guide_scale, stride, eta):
        """ddim_sample with the options of `_p_mean_variance_general` and / or classifier guidance (diffusion_ddim.py:
        217-241 in the reference's expression order)."""
        _, _, _, x0 = self._p_mean_variance_general(xt, t, model, model_kwargs, clamp, percentile, guide_scale)
        xt = xt.float()
        g = lambda tab: self._g(tab, t, xt)
        if condition_fn is not None:
            alpha = g(self.alphas_cumprod)
            eps = (g(self.sqrt_recip_alphas_cumprod) * xt - x0) / g(self.sqrt_recipm1_alphas_cumprod)
            eps = eps - (1 - alpha).sqrt() * condition_fn(xt, self._scale_timesteps(t), **model_kwargs)
            x0 = g(self.sqrt_recip_alphas_cumprod) * xt - g(self.sqrt_recipm1_alphas_cumprod) * eps
        eps = (g(self.sqrt_recip_alphas_cumprod) * xt - x0) / g(self.sqrt_recipm1_alphas_cumprod)
        alphas = g(self.alphas_cumprod)
        alphas_prev = self._g(self.alphas_cumprod, (t - stride).clamp(0), xt)
        sigmas = eta * torch.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
        noise = torch.randn_like(xt)
        direction = torch.sqrt(1 - alphas_prev - sigmas ** 2) * eps
        mask = t.ne(0).float().view(-1, *((1,) * (xt.ndim - 1)))
        return torch.sqrt(alphas_prev) * x0 + direction + mask * sigmas * noise, x0

    @torch.no_grad()
    def ddim_sample_loop(self, noise, model, model_kwargs={}, clamp=None, percentile=None,
                         condition_fn=None, guide_scale=None, ddim_timesteps=20, eta=0.0):
        b = noise.size(0)
        xt = noise
        steps = (1 + torch.arange(0, self.num_timesteps, self.num_timesteps // ddim_timesteps)) \
            .clamp(0, self.num_timesteps - 1).flip(0)
        for step in steps:
            t = torch.full((b,), int(step), dtype=torch.long, device=xt.device)
            # _alias_ok: inside the loop the session hands back its own x_{t-1} buffer (no copies between steps)
            xt, _ = self.ddim_sample(xt, t, model, model_kwargs, clamp, percentile, condition_fn,
                                     guide_scale, ddim_timesteps, eta, _alias_ok=True)
        return xt.clone()

    @torch.no_grad()
    def ddim_reverse_sample(self, xt, t, model, model_kwargs={}, clamp=None, percentile=None,
                            guide_scale=None, ddim_timesteps=20, _alias_ok=False):
        """x_t -> x_{t+stride} along the deterministic DDIM ODE (diffusion_ddim.py:256-274):
        mu = sqrt(a_next) * x0 + sqrt(1 - a_next) * eps — the same kernel with sigma = 0."""
        stride = self.num_timesteps // ddim_timesteps
        if self._uncommon(clamp, percentile):
            _, _, _, x0 = self._p_mean_variance_general(xt, t, model, model_kwargs, clamp, percentile, guide_scale)
            xt = xt.float()
            eps = (self._g(self.sqrt_recip_alphas_cumprod, t, xt) * xt - x0) / self._g(self.sqrt_recipm1_alphas_cumprod, t, xt)
            alphas_next = self._g(torch.cat([self.alphas_cumprod, self.alphas_cumprod.new_zeros([1])]),
                                  (t + stride).clamp(0, self.num_timesteps), xt)
            return torch.sqrt(alphas_next) * x0 + torch.sqrt(1 - alphas_next) * eps, x0
        return self._fused(xt, t, model, model_kwargs, guide_scale, "reverse", stride, 0.0, None, clamp, percentile,
                           alias_ok=_alias_ok)

    @torch.no_grad()
    def ddim_reverse_sample_loop(self, x0, model, model_kwargs={}, clamp=None, percentile=None,
                                 guide_scale=None, ddim_timesteps=20):
        b = x0.size(0)
        xt = x0
        steps = torch.arange(0, self.num_timesteps, self.num_timesteps // ddim_timesteps)
        for step in steps:
            t = torch.full((b,), int(step), dtype=torch.long, device=xt.device)
            xt, _ = self.ddim_reverse_sample(xt, t, model, model_kwargs, clamp, percentile, guide_scale,
                                             ddim_timesteps, _alias_ok=True)
        return xt.clone()
