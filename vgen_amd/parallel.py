"""Unit partition of a denoise step across the GPUs of one node.

A denoise step evaluates U = P x G independent UNet forwards ("units": P prompts/seeds in flight,
G = 2 classifier-free-guidance branches).  Units share nothing but weights, so they shard with no
data-path collective inside the UNet; the only exchange is ONE all-gather of the unit outputs per
step (RCCL over xGMI when the process group is `nccl`; 459 KB fp32 per unit at the t2v shape —
latency-bound, so the payload stays fp32 and exact), after which every rank applies the cheap fused
CFG + DDIM update redundantly for all prompts.  The reference has no such path (every rank re-runs
the whole prompt list: tools/inferences/inference_text2video_entrance.py:93,165-171); this is the
design the north-star asks for.

unit u = p * G + g  ->  rank u % W, or — when every rank can own whole prompts (P % W == 0) — rank (u // G) % W.
One process per GPU; torch.distributed supplies the process group (backend "nccl" == RCCL on ROCm, "gloo" in the
CPU tests).

The local units run through a sampling session (vgen_amd/session.py): their condition stems and K/V are
prompt constants computed once, and the local UNet batch is one hipGraph replay that writes straight into
this rank's all-gather slot; the gather and the update kernel stay eager between two replays (a collective
inside a captured graph could not be validated on the 1-GPU development boxes).
"""
from __future__ import annotations

from typing import List, Sequence

import os

import torch
import torch.distributed as dist

# test switch: run the all-gather even with a 1-rank group (exercises RCCL on a 1-GPU box)
_FORCE_COLLECTIVE = os.environ.get("VGEN_FORCE_COLLECTIVE") == "1"


# kwargs of the reference's UNets that are batched over the prompts (dim 0 = batch): y / fps / image / local_image of
# UNetSD_T2VBase / I2VGen (unet_t2v.py:210-223, unet_i2vgen.py:243-262) and the composer conditions of VideoLCM / TFT2V
# (unet_videolcm.py:541-560).  Everything else (config objects, flags, shared tensors) is passed through.
# ONE shared set object, mutated in place: `from vgen_amd.parallel import PER_PROMPT_KEYS` stays current
PER_PROMPT_KEYS = {
    "y", "fps", "image", "local_image", "depth", "sketch", "canny", "masked", "motion", "single_sketch", "histogram",
    "video_mask", "focus_present_mask", "x_lr", "zero_y", "y_words", "t_w"}
SHARED_KEYS = set()        # tensor kwargs declared SHARED by all prompts even when their leading dim happens to equal P


def register_per_prompt_keys(*names):
    """Custom models: declare further kwargs that are batched over the prompts (sliced per rank like `y`)."""
    PER_PROMPT_KEYS.update(names)
    SHARED_KEYS.difference_update(names)


def register_shared_keys(*names):
    """Custom models: declare tensor kwargs every prompt shares (passed through unsliced) — the escape hatch for a table
    whose leading dimension happens to equal the number of prompts, which _slice_kwargs otherwise refuses to guess about."""
    SHARED_KEYS.update(names)
    PER_PROMPT_KEYS.difference_update(names)


def _slice_kwargs(kw, ps, P):
    """kwargs of the prompts `ps`.  Only the known per-prompt keys are indexed, and only when their leading dim IS
    the prompt count; a per-prompt tensor given as a broadcast [1, ...] is expanded first.  (r02 indexed every tensor
    whose leading dim happened to equal P and left broadcast rows unsliced: ADVICE r02.)  A tensor under an UNKNOWN
    name whose leading dim equals the prompt count while only a subset of the prompts is evaluated is ambiguous (shared
    table or per-prompt batch?): it raises instead of being paired silently with a smaller batch —
    register_per_prompt_keys() declares it per-prompt; reshape it or pass it under a registered name otherwise."""
    out = {}
    for k, v in kw.items():
        if k not in PER_PROMPT_KEYS and k not in SHARED_KEYS and torch.is_tensor(v) and v.dim() >= 1 and P > 1 and v.shape[0] == P and len(ps) != P:
            raise ValueError(f"kwarg {k!r}: tensor with leading dim == prompt count {P} under a name that is not in "
                             f"PER_PROMPT_KEYS — cannot tell a per-prompt batch from a shared tensor; call "
                             f"vgen_amd.parallel.register_per_prompt_keys({k!r}) if it is batched over the prompts, "
                             f"register_shared_keys({k!r}) if every prompt shares it")
        if k in PER_PROMPT_KEYS and torch.is_tensor(v) and v.dim() >= 1:
            if v.shape[0] == P:
                out[k] = v[ps]
            elif v.shape[0] == 1:
                out[k] = v.expand(len(ps), *v.shape[1:])
            else:
                raise ValueError(f"kwarg {k!r}: leading dim {v.shape[0]} is neither the prompt count {P} nor 1")
        else:
            out[k] = v
    return out


# r04 switch: model -> all-gather -> fused update of a partitioned DDIM step as ONE launch sequence, captured as one
# hipGraph per step when the group's backend is nccl (= RCCL); default off until it has run on a multi-GPU node
_GRAPH_COLLECTIVE = os.environ.get("VGEN_PARTITION_GRAPH") == "1"


class UnitPartition:
    def __init__(self, group=None, graph_collective=None):
        self.group = group
        self.graph_collective = _GRAPH_COLLECTIVE if graph_collective is None else bool(graph_collective)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        from .session import SessionCache
        self.sessions = SessionCache()
        self._sub = {}              # (kwargs identity, local prompts) -> per-rank slices of the kwarg sets
        self._gidx = {}             # (P, G, device) -> gather order of gather_grouped

    # -- unit -> (rank, slot) ----------------------------------------------------------------------
    # unit u = p * G + g.  Two layouts:
    #   'unit'   (default)            owner u % W, slot u // W — the CFG pair of ONE prompt lands on two ranks (P = 1,
    #                                 W = 2 is the north-star's "cond on GPU 0, uncond on GPU 1");
    #   'prompt' (when P % W == 0)    owner (u // G) % W — a rank owns whole prompts, so its local batch is an ordinary
    #                                 CFG batch and shares the context-independent prefix between the branches
    #                                 (UNetSD_T2VBase._body shared_groups); slot = g * (P / W) + p // W, i.e. the
    #                                 session's own group-major order: its output buffer IS the all-gather payload.
    def layout(self, P: int, G: int) -> str:
        return "prompt" if (G > 1 and P % self.world == 0) else "unit"

    def owner(self, u: int, P: int = 0, G: int = 1) -> int:
        return (u // G) % self.world if P and self.layout(P, G) == "prompt" else u % self.world

    def slot(self, u: int, P: int = 0, G: int = 1) -> int:
        if P and self.layout(P, G) == "prompt":
            return (u % G) * (P // self.world) + (u // G) // self.world
        return u // self.world

    def my_units(self, U: int, P: int = 0, G: int = 1) -> List[int]:
        mine = [u for u in range(U) if self.owner(u, P, G) == self.rank]
        return sorted(mine, key=lambda u: self.slot(u, P, G))

    def slots(self, U: int) -> int:
        return (U + self.world - 1) // self.world

    # -- the per-step exchange -----------------------------------------------------------------------
    def gather_stacked(self, mine: torch.Tensor, U: int, P: int = 0, G: int = 1) -> List[torch.Tensor]:
        """`mine`: [len(my_units), ...] outputs of this rank's units in slot order.  Returns the U unit outputs on
        every rank after ONE all-gather."""
        S = self.slots(U)
        if self.world == 1:
            return [mine[self.slot(u, P, G)] for u in range(U)]
        if mine.shape[0] == S and mine.is_contiguous():
            buf = mine
        else:                                                   # ragged tail: this rank owns fewer than S units
            buf = mine.new_zeros((S,) + tuple(mine.shape[1:]))
            buf[: mine.shape[0]].copy_(mine)
        allb = mine.new_empty((self.world * S,) + tuple(mine.shape[1:]))
        dist.all_gather_into_tensor(allb, buf, group=self.group)
        return [allb[self.owner(u, P, G) * S + self.slot(u, P, G)] for u in range(U)]

    def gather_grouped(self, mine: torch.Tensor, P: int, G: int):
        """ONE all-gather of this rank's unit outputs (slot order), then the G per-branch batches [P, ...] every
        sampler consumes — one index_select over the gathered buffer with a cached index (r02 built them with P-element
        python lists and a torch.stack per branch on every step)."""
        U = P * G
        S = self.slots(U)
        if self.world == 1 and not _FORCE_COLLECTIVE:
            # a COPY, not the session's static output buffer: the returned branches must survive the next eval / graph
            # replay (multistep samplers keep a model output across steps: old_denoised of DPM-Solver++) — ADVICE r03
            allb = mine.clone()
        else:
            if mine.shape[0] == S and mine.is_contiguous():
                buf = mine
            else:                                               # ragged tail: this rank owns fewer than S units
                buf = mine.new_zeros((S,) + tuple(mine.shape[1:]))
                buf[: mine.shape[0]].copy_(mine)
            allb = mine.new_empty((self.world * S,) + tuple(mine.shape[1:]))
            dist.all_gather_into_tensor(allb, buf, group=self.group)
        key = (P, G, str(mine.device))
        hit = self._gidx.get(key)
        if hit is None:
            # (g, p) -> position of unit u = p * G + g in the gathered buffer
            pos = [self.owner(p * G + g, P, G) * S + self.slot(p * G + g, P, G) for g in range(G) for p in range(P)]
            identity = pos == list(range(U))
            hit = (None if identity else torch.tensor(pos, dtype=torch.long, device=mine.device), identity)
            self._gidx[key] = hit
        idx, identity = hit
        ordered = allb[:U] if identity else allb.index_select(0, idx)
        return tuple(ordered.view(G, P, *allb.shape[1:])[g] for g in range(G))

    def gather_units(self, mine: Sequence[torch.Tensor], U: int, like: torch.Tensor) -> List[torch.Tensor]:
        """List form of gather_stacked, 'unit' layout (each element shaped like `like`)."""
        if len(mine):
            st = torch.stack(list(mine))
        else:
            st = like.new_zeros((0,) + tuple(like.shape))
        return self.gather_stacked(st, U)

    # -- classifier-free guidance for P prompts stacked in the batch dim ---------------------------------
    def _local_kwargs(self, model_kwargs, ps, P, device):
        """The kwarg sets restricted to the local prompts `ps` — built once per (kwargs identity, ps) so that the
        sampling session keyed on tensor identity is found again on every step."""
        from .session import _kw_key
        key = (_kw_key(model_kwargs), tuple(ps))
        hit = self._sub.get(key)
        if hit is None:
            idx = torch.tensor(ps, dtype=torch.long, device=device)
            hit = (idx, [_slice_kwargs(kw, idx, P) for kw in model_kwargs], [dict(kw) for kw in model_kwargs])
            if len(self._sub) >= 4:
                self._sub.clear()
            self._sub[key] = hit
        return hit[0], hit[1]

    def run_units(self, model, xt, t, model_kwargs, num_timesteps=None):
        """xt [P, C, F, H, W], t [P], model_kwargs = list of G kwarg sets whose tensors are stacked over the P
        prompts along dim 0.  Returns G tensors [P, out_dim, F, H, W], identical on every rank."""
        G = len(model_kwargs)
        P = xt.shape[0]
        U = P * G
        mine = self.my_units(U, P, G)
        out_dim = getattr(model, "out_dim", xt.shape[1])
        unit_shape = (out_dim,) + tuple(xt.shape[2:])
        is_unit_model = xt.dim() == 5 and hasattr(model, "_prepare_units")
        nt = num_timesteps if t.dtype == torch.long else None
        local = None
        if is_unit_model and self.layout(P, G) == "prompt":
            # whole prompts per rank: an ordinary CFG session over the local prompts
            ps = sorted({u // G for u in mine})
            idx, sub = self._local_kwargs(model_kwargs, ps, P, xt.device)
            sess = self.sessions.get(model, (len(ps),) + tuple(xt.shape[1:]), xt.device, sub, t.dtype, nt)
            if sess is not None:
                sess.eval(xt.index_select(0, idx), t.index_select(0, idx))
                local = sess.out                                               # [G * len(ps), ...] = slot order
        if local is None and is_unit_model:
            # session over the local units (session index g * P + p)
            sess = self.sessions.get(model, tuple(xt.shape), xt.device, list(model_kwargs), t.dtype, nt,
                                     units=[(u % G) * P + (u // G) for u in mine])
            if sess is not None:
                local = sess.eval(xt, t)                                       # [len(mine), out_dim, F, H, W]
        if local is None:
            outs = {}
            for g in range(G):
                ps = [u // G for u in mine if u % G == g]
                if not ps:
                    continue
                idx = torch.tensor(ps, device=xt.device)
                o = model(xt[idx], t[idx], **_slice_kwargs(model_kwargs[g], idx, P))
                for i, p in enumerate(ps):
                    outs[p * G + g] = o[i].float()
            local = torch.stack([outs[u] for u in mine]) if mine else \
                xt.new_zeros((0,) + unit_shape, dtype=torch.float32)
        return self.gather_grouped(local, P, G)

    # -- r04: the whole partitioned DDIM step as one launch sequence --------------------------------------------------
    def fused_layout(self, P: int, G: int):
        """'prompt' | 'pair' when the gathered buffer splits into per-rank blocks the update kernel can address directly
        ('prompt': rank r's block is [y of its P/W prompts | u of them]; 'pair': ONE prompt whose G = W = 2 branches sit on
        two ranks), else None (the eager gather + update path handles everything)."""
        W = self.world
        if G == 2 and self.layout(P, G) == "prompt":
            return "prompt"
        if P == 1 and G == 2 and W == 2:
            return "pair"
        return None

    def ddim_step(self, model, xt, t, model_kwargs, coef_tab, guide, mean_type, num_timesteps):
        """One classifier-free-guidance DDIM step (eta = 0) of P prompts over the ranks as ONE launch sequence on the
        compute stream:
            local units' UNet forward -> ONE all_gather_into_tensor of the unit outputs -> fused CFG + DDIM update of ALL
            prompts (redundantly on every rank, y / u read straight from the gathered buffer, coefficients gathered by t on
            the device) -> x_{t-1} [P, ...] and the local units' latent slots for the next step.
        r03 ran per step: graph replay, eager all-gather, index_select, ~10 tiny torch launches for the coefficient rows,
        eager update kernel.  With the nccl (= RCCL) backend the sequence is captured as ONE hipGraph; with gloo (CPU tests,
        the one-device functional test) the same sequence runs eagerly.  Returns (x_{t-1}, x0) — the session's own buffers,
        valid until the next step; passing that x_{t-1} back in skips the input copies — or None when the step does not fit (layout, kwargs, dtype):
        the caller then takes the eager path."""
        G = len(model_kwargs)
        P = xt.shape[0]
        kind = self.fused_layout(P, G)
        inner = getattr(model, "module", model)
        if kind is None or xt.dim() != 5 or not hasattr(inner, "_prepare_units") or t.dtype != torch.long:
            return None
        from . import ops
        W, dev = self.world, xt.device
        if kind == "prompt":
            ps = [p for p in range(P) if p % W == self.rank]
            idx, sub = self._local_kwargs(model_kwargs, ps, P, dev)
            sess = self.sessions.get(inner, (len(ps),) + tuple(xt.shape[1:]), dev, sub, t.dtype, num_timesteps)
            nloc = len(ps)
        else:       # "pair": session unit index g * P + p = g — cond on rank 0, uncond on rank 1
            idx, nloc = None, 1
            sess = self.sessions.get(inner, tuple(xt.shape), dev, list(model_kwargs), t.dtype, num_timesteps,
                                     units=[self.rank])
        if sess is None:
            return None
        S = self.slots(P * G)
        st = sess._static.get("pstep")
        if st is None:
            lat = (sess.C_lat, sess.F, sess.H, sess.W)
            f32 = dict(dtype=torch.float32, device=dev)
            st = dict(allb=torch.empty((W * S,) + tuple(sess.out.shape[1:]), **f32), x_all=torch.zeros((P,) + lat, **f32),
                      x0_blk=torch.empty((P,) + lat, **f32), t_ord=torch.zeros((P,), dtype=torch.long, device=dev))
            sess._static["pstep"] = st
        # ---- per-step inputs (outside the captured sequence): t in block order; x_t unless it is last step's x_{t-1} ----
        if kind == "prompt":
            st["t_ord"].view(W, nloc).copy_(t.view(nloc, W).t())        # t_ord[r, j] = t[r + W j]
            sess.t_units.view(G, nloc).copy_(st["t_ord"].view(W, nloc)[self.rank])
        else:
            st["t_ord"].copy_(t)
            sess.t_units.copy_(t)
        if xt is not st["x_all"]:
            st["x_all"].copy_(xt)
            if kind == "prompt":
                sess.x_units.view(G, nloc, *sess.x_units.shape[1:])[:, :, :sess.C_lat].copy_(xt.float().index_select(0, idx))
            else:
                sess.x_units[:, :sess.C_lat].copy_(xt.float())
        sess._last_out = None
        be = ops.backend()
        collective = self.world > 1 or _FORCE_COLLECTIVE
        nccl = dist.is_initialized() and dist.get_backend(self.group) == "nccl"

        def launches():
            sess._model_launches()
            if collective:
                dist.all_gather_into_tensor(st["allb"], sess.out, group=self.group)
                allb = st["allb"]
            else:
                allb = sess.out
            x_all, x0b, t_ord = st["x_all"], st["x0_blk"], st["t_ord"]
            # x_t -> x_{t-1} IN PLACE in x_all (every element is read and written by the same thread: the kernel's `rep` may
            # alias `xt`): x_all is the next step's x_t on every rank.  The local units' latent slots are written first,
            # by the same arithmetic on the still unmodified x_t.
            if kind == "prompt":
                # rank r's block of the gathered buffer = [y of prompts r, r + W, ... | u of the same]: the update of those
                # prompts reads x_t and writes x_{t-1} at row stride W (the update kernel addresses rows by stride)
                for r in range(W):
                    blk = allb[r * S:(r + 1) * S]
                    tr = t_ord[r * nloc:(r + 1) * nloc]
                    if r == self.rank:
                        be.ddim_update_strided(x_all[r::W], blk[:nloc], blk[nloc:2 * nloc], coef_tab, tr, guide, True,
                                               mean_type, rep_units=sess.x_units, G=G, C_lat=sess.C_lat)
                    be.ddim_update_strided(x_all[r::W], blk[:nloc], blk[nloc:2 * nloc], coef_tab, tr, guide, True, mean_type,
                                           out_rows=x_all[r::W], x0_out=x0b[r * nloc:(r + 1) * nloc])
            else:
                be.ddim_update_strided(x_all, allb[0:1], allb[1:2], coef_tab, t_ord, guide, True, mean_type,
                                       rep_units=sess.x_units, G=1, C_lat=sess.C_lat)
                be.ddim_update_strided(x_all, allb[0:1], allb[1:2], coef_tab, t_ord, guide, True, mean_type,
                                       out_rows=x_all, x0_out=x0b)

        key = ("pddim", id(coef_tab), float(guide), int(mean_type))
        sess._static[("tab",) + key] = coef_tab
        sess._run(key, launches, graph_ok=(nccl or not collective))
        if kind == "prompt" and W > 1:
            x0 = st["x0_blk"].view(W, nloc, *st["x0_blk"].shape[1:]).transpose(0, 1).reshape(st["x_all"].shape)
        else:
            x0 = st["x0_blk"]
        return st["x_all"], x0
