"""Unit partition of a denoise step across the GPUs of one node.

A denoise step evaluates U = P x G independent UNet forwards ("units": P prompts/seeds in flight,
G = 2 classifier-free-guidance branches).  Units share nothing but weights, so they shard with no
data-path collective inside the UNet; the only exchange is ONE all-gather of the unit outputs per
step (RCCL over xGMI when the process group is `nccl`; 229 KB bf16 / 459 KB fp32 per unit at the
t2v shape — latency-bound), after which every rank applies the cheap fused CFG + DDIM update
redundantly for all prompts.  The reference has no such path (every rank re-runs the whole prompt
list: tools/inferences/inference_text2video_entrance.py:93,165-171); this is the design the
north-star asks for.

unit u = p * G + g  ->  rank u % W.  One process per GPU; torch.distributed supplies the
process group (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class UnitPartition:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # -- static helpers --------------------------------------------------------------------------
    def owner(self, u: int) -> int:
        return u % self.world

    def my_units(self, U: int) -> List[int]:
        return [u for u in range(U) if self.owner(u) == self.rank]

    def slots(self, U: int) -> int:
        return (U + self.world - 1) // self.world

    # -- the per-step exchange -----------------------------------------------------------------------
    def gather_units(self, mine: Sequence[torch.Tensor], U: int, like: torch.Tensor) -> List[torch.Tensor]:
        """`mine`: outputs of my_units(U) in order, each shaped like `like` (one unit).
        Returns the U unit outputs on every rank after ONE all-gather."""
        S = self.slots(U)
        buf = like.new_zeros((S,) + tuple(like.shape))
        for i, o in enumerate(mine):
            buf[i].copy_(o)
        if self.world == 1:
            return [buf[i] for i in range(U)]
        allb = like.new_empty((self.world * S,) + tuple(like.shape))
        dist.all_gather_into_tensor(allb, buf, group=self.group)
        # unit u sits in rank (u % W)'s slot (u // W)
        return [allb[(u % self.world) * S + (u // self.world)] for u in range(U)]

    # -- classifier-free guidance for P prompts stacked in the batch dim ---------------------------------
    def run_units(self, model, xt, t, model_kwargs):
        """xt [P, C, F, H, W], t [P], model_kwargs = [cond_kwargs, uncond_kwargs] with per-prompt `y`
        stacked along dim 0.  Returns (y_out, u_out), each [P, ...], identical on every rank."""
        G = len(model_kwargs)
        P = xt.shape[0]
        U = P * G
        mine = self.my_units(U)
        outs = []
        if mine:
            xs = torch.stack([xt[u // G] for u in mine])
            ts = torch.stack([t[u // G] for u in mine])
            ys = torch.stack([model_kwargs[u % G]["y"][u // G] for u in mine])
            extra = {k: v for k, v in model_kwargs[0].items() if k not in ("y", "fps")}
            o = model(xs, ts, y=ys, **extra)
            outs = [o[i] for i in range(len(mine))]
            like = o[0]
        else:
            like = xt.new_zeros((getattr(model, "out_dim", xt.shape[1]),) + tuple(xt.shape[2:]),
                                dtype=torch.float32)
        allu = self.gather_units(outs, U, like)
        return tuple(torch.stack([allu[p * G + g] for p in range(P)]) for g in range(G))
