"""Noise schedules (host side, float64) for the sampling path.

Semantics follow tools/modules/diffusions/schedules.py of the reference — `beta_schedule`
(:5-21), `sigma_schedule` (:24-43), `cosine_schedule` (:72-79), `linear_sd_schedule` (:62-63),
`quadratic_schedule` (:66-69), `logsnr_cosine_interp_schedule` (:52-60,127-140),
`rescale_zero_terminal_snr` (:143-165) — and are checked BIT-EXACT in float64 against it
(tests/test_schedules.py + tests/golden/schedules.pt).  The tables are tiny and computed once;
they stay on the host, the per-step scalars are handed to the fused HIP update kernel.
"""
from __future__ import annotations

import math

import torch

F64 = torch.float64


def _cosine_betas(n: int, cosine_s: float = 0.008, **_):
    # alpha_bar(u) = cos^2((u + s)/(1 + s) * pi/2); beta_k = min(1 - ab(t_{k+1})/ab(t_k), 0.999)
    def ab(u):
        return math.cos((u + cosine_s) / (1 + cosine_s) * math.pi / 2) ** 2

    vals = [min(1.0 - ab((k + 1) / n) / ab(k / n), 0.999) for k in range(n)]
    return torch.tensor(vals, dtype=F64)


def _linear_betas(n: int, init_beta, last_beta, **_):
    # the reference's `linear` ignores its own defaults for last_beta (typo at schedules.py:49);
    # behave identically: both ends must be given.
    scale = 1000.0 / n
    init_beta = init_beta or scale * 0.0001
    return torch.linspace(init_beta, last_beta, n, dtype=F64)


def _linear_sd_betas(n: int, init_beta, last_beta, **_):
    return torch.linspace(init_beta ** 0.5, last_beta ** 0.5, n, dtype=F64) ** 2


def _quadratic_betas(n: int, init_beta, last_beta, **_):
    init_beta = init_beta or 0.0015
    last_beta = last_beta or 0.0195
    return torch.linspace(init_beta ** 0.5, last_beta ** 0.5, n, dtype=F64) ** 2


def _logsnr_cosine(n, logsnr_min, logsnr_max):
    t_min = math.atan(math.exp(-0.5 * logsnr_min))
    t_max = math.atan(math.exp(-0.5 * logsnr_max))
    t = torch.linspace(1, 0, n)
    return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)))


def _logsnr_cosine_interp_sigmas(n, scale_min=2, scale_max=4, logsnr_min=-15, logsnr_max=15, **_):
    t = torch.linspace(1, 0, n)
    lo = _logsnr_cosine(n, logsnr_min, logsnr_max)
    lo = lo + 2 * math.log(1 / scale_min)
    hi = _logsnr_cosine(n, logsnr_min, logsnr_max)
    hi = hi + 2 * math.log(1 / scale_max)
    logsnrs = t * lo + (1 - t) * hi
    return torch.sqrt(torch.sigmoid(-logsnrs))


def rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """Shift/scale sqrt(alpha_bar) so that alpha_bar[-1] == 0 while alpha_bar[0] is preserved."""
    ab_sqrt = (1 - betas).cumprod(0).sqrt()
    first = ab_sqrt[0].clone()
    last = ab_sqrt[-1].clone()
    ab_sqrt -= last
    ab_sqrt *= first / (first - last)
    ab = ab_sqrt ** 2
    alphas = torch.cat([ab[0:1], ab[1:] / ab[:-1]])
    return 1 - alphas


_BETAS = {"linear": _linear_betas, "linear_sd": _linear_sd_betas, "quadratic": _quadratic_betas,
          "cosine": _cosine_betas}


def beta_schedule(schedule="cosine", num_timesteps=1000, zero_terminal_snr=False, **kwargs):
    betas = _BETAS[schedule](num_timesteps, **kwargs)
    if zero_terminal_snr and abs(betas.max() - 1.0) > 0.0001:
        betas = rescale_zero_terminal_snr(betas)
    return betas


def betas_to_sigmas(betas):
    return torch.sqrt(1 - torch.cumprod(1 - betas, dim=0))


def sigma_schedule(schedule="cosine", num_timesteps=1000, zero_terminal_snr=False, **kwargs):
    if schedule == "logsnr_cosine_interp":
        sigma = _logsnr_cosine_interp_sigmas(num_timesteps, **kwargs)
    else:
        sigma = betas_to_sigmas(_BETAS[schedule](num_timesteps, **kwargs))
    if zero_terminal_snr and abs(sigma.max() - 1.0) > 0.0001:
        sigma = rescale_zero_terminal_snr(sigma)
    return sigma
