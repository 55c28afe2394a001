"""Seeded synthetic parameters.

Pretrained VGen checkpoints cannot be fetched offline, and the reference zero-initialises five layer families
(ResBlock out-conv, temporal conv4, proj_out, fps embedding, head conv: tools/modules/unet/util.py:873-875,1683-1684,
351,1229; unet_t2v.py:208) — a default-initialised model is a per-channel constant.  Benchmarks, fixtures and tests
therefore all run on the same recipe: EVERY parameter re-randomised from one CPU generator, keys visited in sorted
order, so any process that knows the parameter shapes reproduces the same tensors bit for bit (the golden fixtures
under tests/golden/ store shapes + seed, not weights).
"""
from __future__ import annotations

import math

import torch


def seeded_state_dict(shapes: dict, seed: int = 0, gain: float = 0.8, recipe: str = "gauss") -> dict:
    """>= 2-D weights ~ N(0, gain / sqrt(fan_in)), norm scales ~ 1 + 0.1 N, biases ~ 0.1 N (fp32, CPU).
    recipe="student4": the >= 2-D weights are heavy-tailed instead — Student-t with 4 degrees of freedom scaled to the
    same variance (z / sqrt(chi2_4 / 4) / sqrt(2)): outliers of 10+ sigma in every large matrix, which is what a 16-bit
    hi/lo weight split and the fp16 range have to survive on trained checkpoints (r04 fixtures)."""
    assert recipe in ("gauss", "student4"), recipe
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            w = torch.randn(shp, generator=g)
            if recipe == "student4":
                chi = torch.randn((4,) + shp, generator=g).square_().sum(0)
                w = w / (chi / 4.0).sqrt_() * (0.5 ** 0.5)
            sd[k] = w * (gain / math.sqrt(fan_in))
        elif k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(shp, generator=g)
    return sd


def shapes_of(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}
