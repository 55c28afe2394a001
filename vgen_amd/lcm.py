"""LCMScheduler for the VideoLCM engine.

PARITY UNPINNED.  The reference imports `LCMScheduler` from the third-party `diffusers` package
(tools/inferences/inference_videolcm_entrance.py:49), which is neither vendored under the reference tree nor
version-pinned in its requirements, and no reference test touches it.  This is a restatement of the
published latent-consistency multistep sampler as that class implements it, with the constructor
arguments the engine passes (:171: v_prediction, scaled_linear betas 0.00085..0.012, zero-terminal-SNR
rescale, clip_sample False) and its call pattern (:179 set_timesteps, :236 scale_model_input, :255 step):

  timesteps   : origin grid t_k = (k + 1) * (T / original_steps) - 1, k = 0..original_steps-1, reversed;
                `num_inference_steps` of them at indices floor(linspace(0, original_steps, n, endpoint=False))
                (50 origin steps, n = 4 -> 999, 759, 499, 259)
  boundary    : s = t * timestep_scaling (10), sigma_data = 0.5;  c_skip = sd^2 / (s^2 + sd^2),
                c_out = s / sqrt(s^2 + sd^2)
  x0          : v-pred  sqrt(a_t) x - sqrt(1 - a_t) v   |  eps-pred (x - sqrt(1 - a_t) e) / sqrt(a_t)  |  sample
  denoised    : c_out * x0 + c_skip * x
  x_prev      : sqrt(a_prev) * denoised + sqrt(1 - a_prev) * noise   (last step: denoised)

The update runs on the device through `vgen_lincomb4` (fp32, no contraction, the operation order written
above), classifier-free guidance through `vgen_gauss_x0` (u + g * (y - u)); tables are built in fp32 torch
like the original.  tests/test_lcm.py checks it against the CPU restatement in oracle/torch_ref.py and (r06) against
known-answer vectors derived by hand from the published formulas in pure-Python float64 (tests/golden/make_lcm_kat.py ->
lcm_kat.json: timestep lists, alphas_cumprod, boundary scalings, a whole 4-step loop) — pinned to the ALGORITHM; parity
against the `diffusers` package itself stays unpinned (absent here).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops
from .session import SessionCache, eval_units


def _rescale_zero_terminal_snr_f32(betas: torch.Tensor) -> torch.Tensor:
    a = (1.0 - betas).cumprod(0).sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = (a - aT) * (a0 / (a0 - aT))
    ab = a ** 2
    al = torch.cat([ab[:1], ab[1:] / ab[:-1]])
    return 1.0 - al


class LCMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 trained_betas=None, original_inference_steps=50, clip_sample=False, clip_sample_range=1.0,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", thresholding=False,
                 timestep_spacing="leading", timestep_scaling=10.0, rescale_betas_zero_snr=False, **kwargs):
        if thresholding:
            raise NotImplementedError("LCMScheduler: dynamic thresholding")
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"LCMScheduler: beta_schedule {beta_schedule}")
        if rescale_betas_zero_snr:
            betas = _rescale_zero_terminal_snr_f32(betas)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, 0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.original_inference_steps = original_inference_steps
        self.prediction_type = prediction_type
        self.clip_sample, self.clip_sample_range = clip_sample, clip_sample_range
        self.timestep_scaling = timestep_scaling
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._step_index = None
        # sampling sessions live on the scheduler (the engine builds ONE scheduler and loops over its prompts,
        # inference_videolcm_entrance.py:171-257): the second video re-binds the first one's session (one K/V GEMM)
        # instead of re-capturing the model graph for a 4-step loop
        self.sessions = SessionCache(capacity=1)

    def reset(self):
        """Drop the cached sampling session (the model reference, its captured graphs and their memory pool) — e.g. between
        two models driven by the same scheduler object (the cache otherwise lives as long as the scheduler)."""
        self.sessions.clear()
        self._step_index = None

    # -- schedule ------------------------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps, device=None, original_inference_steps=None, strength=1.0):
        orig = original_inference_steps or self.original_inference_steps
        if orig > self.num_train_timesteps or num_inference_steps > orig:
            raise ValueError("LCMScheduler.set_timesteps: num_inference_steps <= original_inference_steps <= T")
        k = self.num_train_timesteps // orig
        origin = np.asarray(list(range(1, int(orig * strength) + 1))) * k - 1
        origin = origin[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=num_inference_steps, endpoint=False)).astype(np.int64)
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(origin[idx]).to(device=device, dtype=torch.long)
        self._step_index = None
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def boundary_scalings(self, timestep):
        sd = 0.5
        s = float(timestep) * self.timestep_scaling
        return sd ** 2 / (s ** 2 + sd ** 2), s / math.sqrt(s ** 2 + sd ** 2)

    def _index_of(self, timestep):
        ts = self.timesteps.tolist()
        t = int(timestep)
        hits = [i for i, v in enumerate(ts) if v == t]
        if not hits:
            raise ValueError(f"LCMScheduler.step: timestep {t} is not in the schedule {ts}")
        return hits[1] if len(hits) > 1 else hits[0]

    # -- one step ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, model_output, timestep, sample, generator=None, return_dict=True, noise=None):
        if self.num_inference_steps is None:
            raise ValueError("LCMScheduler.step: call set_timesteps first")
        if self._step_index is None:
            self._step_index = self._index_of(timestep)
        i = self._step_index
        prev_t = int(self.timesteps[i + 1]) if i + 1 < len(self.timesteps) else int(timestep)
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        c_skip, c_out = self.boundary_scalings(timestep)
        be = ops.backend()
        x = sample.float().contiguous()
        v = model_output.float().contiguous()
        sa, sb = float(a_t.sqrt()), float((1.0 - a_t).sqrt())
        if self.prediction_type == "v_prediction":
            x0 = be.lincomb4(x, v, None, None, sa, -sb, 0.0, 0.0)
        elif self.prediction_type == "epsilon":
            x0 = be.lincomb4(be.lincomb4(x, v, None, None, 1.0, -sb, 0.0, 0.0), None, None, None, 1.0 / sa, 0.0, 0.0, 0.0)
        elif self.prediction_type == "sample":
            x0 = v
        else:
            raise ValueError(self.prediction_type)
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        denoised = be.lincomb4(x0, x, None, None, float(c_out), float(c_skip), 0.0, 0.0)
        if i != self.num_inference_steps - 1:
            if noise is None:
                noise = torch.randn(v.shape, generator=generator, device=v.device, dtype=torch.float32)
            prev = be.lincomb4(denoised, noise.float().contiguous(), None, None, float(a_prev.sqrt()),
                               float((1.0 - a_prev).sqrt()), 0.0, 0.0)
        else:
            prev = denoised
        self._step_index += 1
        if not return_dict:
            return (prev, denoised)
        return {"prev_sample": prev, "denoised": denoised}

    # -- the engine's loop (inference_videolcm_entrance.py:228-257) -----------------------------------------
    @torch.no_grad()
    def sample_loop(self, noise, model, model_kwargs, guidance_scale=None, generator=None, step_noise=None):
        """latents <- noise; for t in timesteps: v = model(latents, t) [+ CFG]; latents = step(v, t, latents)."""
        be = ops.backend()
        latents = noise.float().contiguous()
        B = latents.shape[0]
        cfg = guidance_scale is not None and len(model_kwargs) > 1 and guidance_scale != 1.0
        coef = torch.ones((B, 2), dtype=torch.float32, device=latents.device)
        self._step_index = None
        sessions = self.sessions
        for k, t in enumerate(self.timesteps):
            tt = t.repeat(B).to(device=latents.device, dtype=latents.dtype)
            x_in = self.scale_model_input(latents, t)
            # the engine's kwarg sets are the same objects on every step: one sampling session per loop, a step
            # is one model-graph replay (K/V, condition stems computed once; vgen_amd/session.py)
            outs = eval_units(sessions, None, model, x_in, tt,
                              [model_kwargs[0], model_kwargs[1]] if cfg else [model_kwargs[0]])
            if outs is not None:
                y_out, u_out = (outs[0], outs[1]) if cfg else (outs[0], None)
            else:
                y_out = model(x_in, tt, t_w=None, **model_kwargs[0])
                u_out = model(x_in, tt, t_w=None, **model_kwargs[1]) if cfg else None
            if cfg:
                v, _ = be.gauss_denoise(latents, y_out.float().contiguous(), u_out.float().contiguous(),
                                        float(guidance_scale), None, coef, 2, False)
            else:
                v = y_out
            nz = None if step_noise is None else step_noise[k]
            latents = self.step(v, t, latents, generator=generator, return_dict=False, noise=nz)[0]
        return latents
