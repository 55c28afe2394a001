"""Sampling sessions: what stays fixed across the denoise steps of one prompt is computed once, and a step is ONE
hipGraph replay.

The reference's engines call `diffusion.ddim_sample_loop(noise, model, model_kwargs=[cond, uncond], ...)`
(tools/inferences/inference_text2video_entrance.py:200-206); every step then re-runs, per CFG branch, the condition
stems, `to_k / to_v` of the 16 cross-attention blocks on the unchanged context (tools/modules/unet/util.py:233-235,
after a x16 repeat_interleave, unet_t2v.py:255), the time-embedding MLP (unet_t2v.py:93-96,244-245) and ~12 torch
elementwise launches of the update (diffusion_ddim.py:157-162,194-197,230-240).  Between two steps only x_t and t
change.  A `UnitSession` therefore holds, for the G kwarg sets x B latents = G*B "units" of one step:

  x_units  [G*B, C_stem, F, H, W]  the UNet's stacked input; the condition-stem channels are written once, the
                                   latent channels of every unit slot are (re)written by the update kernel itself
  kv       K/V rows of all cross-attention blocks for every unit's context (one GEMM, once)
  t_units  [G*B] the timestep, the only per-step host input (plus optional noise)
  coefficient / time-embedding TABLES over all integer timesteps, gathered on the device by t

and captures   emb[t] -> UNet body -> fused CFG + DDIM update   as one graph (torch.cuda.graph over the C-ABI
launches, which only enqueue on the current stream).  Samplers that do their own update (DPM-Solver++, LCM) replay
the model-only graph and read the unit outputs.  On CPU tensors (the test-suite's ABI emulator) or with
VGEN_GRAPH=0 the same launch sequence runs eagerly — one code path, captured or not.

Sessions are cached on the diffusion object, keyed on the model (and its weights epoch), the latent shape and the
IDENTITY + version of every tensor in the kwarg sets (the engines pass the same tensors on every step).
"""
from __future__ import annotations

import os
import warnings
from typing import Optional, Sequence

import torch

from . import ops

_GRAPH_ON = os.environ.get("VGEN_GRAPH", "1") != "0"
_REBIND_ON = os.environ.get("VGEN_SESSION_REBIND", "1") != "0"    # switch: 0 = every new prompt builds + captures its own session


class Unkeyable(Exception):
    """A kwarg value whose identity cannot be keyed (the caller then runs without a session)."""


def _val_key(v):
    """Identity key of one kwarg value: tensors by (id, version, shape, dtype) — also inside lists / tuples / dicts (a
    repr() of a container of tensors elides its contents, so two conditionings could collide: ADVICE r02)."""
    if torch.is_tensor(v):
        # inference-mode tensors have no version counter: their identity + shape must do
        ver = -1 if v.is_inference() else v._version
        return ("T", id(v), ver, tuple(v.shape), v.dtype)
    if isinstance(v, (list, tuple)):
        return ("L", type(v).__name__) + tuple(_val_key(e) for e in v)
    if isinstance(v, dict):
        return ("D",) + tuple((str(k), _val_key(v[k])) for k in sorted(v, key=str))
    if v is None or isinstance(v, (bool, int, float, str)):
        return ("S", type(v).__name__, v)
    try:
        import numpy as np
        if isinstance(v, np.ndarray):
            return ("N", id(v), v.shape, str(v.dtype))
    except ImportError:                                     # pragma: no cover
        pass
    raise Unkeyable(type(v).__name__)


def _kw_key(kwargs_list):
    return tuple(tuple((k, _val_key(kw[k])) for k in sorted(kw)) for kw in kwargs_list)


def _val_struct(v):
    """_val_key without the identity: what two prompts' kwarg values must share for one captured launch sequence."""
    if torch.is_tensor(v):
        return ("T", tuple(v.shape), v.dtype, str(v.device))
    if isinstance(v, (list, tuple)):
        return ("L", type(v).__name__) + tuple(_val_struct(e) for e in v)
    if isinstance(v, dict):
        return ("D",) + tuple((str(k), _val_struct(v[k])) for k in sorted(v, key=str))
    if v is None or isinstance(v, (bool, int, float, str)):
        return ("S", type(v).__name__, v)
    return ("O", type(v).__name__)


def _kw_struct(kwargs_list):
    return tuple(tuple((k, _val_struct(kw[k])) for k in sorted(kw)) for kw in kwargs_list)


class UnitSession:
    def __init__(self, model, shape, device, kwargs_list, t_dtype=torch.long, num_timesteps=None,
                 units: Optional[Sequence[int]] = None):
        """`units`: subset of unit indices (g*B + b) evaluated here (vgen_amd.parallel partitions a step's units
        over the ranks); default all."""
        self.model = model
        self.G = len(kwargs_list)
        self.B, self.C_lat, self.F, self.H, self.W = shape
        self.device = torch.device(device)
        if model._packed is None:
            model.pack()
        U = self.G * self.B
        sel = list(range(U)) if units is None else list(units)
        self.units = sel
        self.full = units is None
        nU = len(sel)
        self._idx = torch.tensor(sel, dtype=torch.long, device=self.device)
        C_stem = model._stem_channels()
        self.x_units = torch.zeros((nU, C_stem, self.F, self.H, self.W), dtype=torch.float32, device=self.device)
        self.kv = self.fps = None
        self.per_frame, self.Lctx, self.shared = False, 0, 1
        if not self._bind(kwargs_list, first=True):
            raise ValueError("kwarg sets cannot share one UNet batch")
        self.t_units = torch.zeros((nU,), dtype=t_dtype, device=self.device)
        self.out = torch.empty((nU, model.out_dim, self.F, self.H, self.W), dtype=torch.float32, device=self.device)
        # time-embedding table: integer timesteps of a known schedule length, no fps term inside the SiLU
        self.emb_tab = None
        if num_timesteps and t_dtype == torch.long and self.fps is None:
            self.emb_tab = model.time_embedding_table(int(num_timesteps), self.device)
        self.use_graph = _GRAPH_ON and self.device.type == "cuda"
        # every graph of this session (model-only, DDIM kinds / strides / etas, ...) captures into ONE private memory
        # pool: a pool holds a whole forward's peak activations, and one pool per graph key kept several of them alive
        # per prompt (tens of GB at 704p over a multi-prompt run: ADVICE r02)
        self._pool = torch.cuda.graph_pool_handle() if self.use_graph else None
        self._graphs = {}
        self._static = {}
        self.xt_1 = torch.empty((self.B, self.C_lat, self.F, self.H, self.W), dtype=torch.float32, device=self.device)
        self.x0 = torch.empty_like(self.xt_1)
        self._last_out = None       # the tensor handed to the caller by the previous fused step
        self._bidx = None

    # -- the prompt-dependent state ------------------------------------------------------------------
    def _bind(self, kwargs_list, first=False):
        """Everything that depends on the PROMPT (condition-stem channels, the units' context -> K/V rows of every
        cross-attention block, fps) written into the session's static buffers.  first=False re-binds a live session to
        the kwarg sets of ANOTHER prompt of the same structure (r04): the captured graphs read these buffers by address,
        so a new prompt costs one K/V GEMM and two copies instead of an eager warm-up pass plus a re-capture of ~900
        launches — what a 4-step LCM loop or a multi-prompt serving run would otherwise pay per video.  Returns False
        (nothing modified that a graph reads) when the new sets do not fit the captured launch sequence."""
        model, dev = self.model, self.device
        prep = model._prepare_units((self.B, self.C_lat, self.F, self.H, self.W), dev, kwargs_list)
        if prep is None:
            return False
        U, nU, idx = self.G * self.B, len(self.units), self._idx
        C_stem = model._stem_channels()
        extra = prep["extra"]
        if (extra is None) != (C_stem == self.C_lat) or (extra is not None and extra.shape[1] != C_stem - self.C_lat):
            assert not first, (C_stem, self.C_lat)
            return False
        ctx = prep["ctx"].to(dev)
        per_frame = bool(prep["per_frame"])
        if per_frame:
            ctx = ctx.view(U, self.F, *ctx.shape[1:])[idx].reshape(nU * self.F, *ctx.shape[1:])
        else:
            ctx = ctx[idx]
        shared = model.shared_prefix_groups(prep, self.G, self.B) if self.full else 1
        fps = None if prep["fps"] is None else prep["fps"].to(dev)[idx].contiguous()
        if not first:
            # the launch sequence of the captured graphs is a function of these: a prompt that changes them needs its own session
            if per_frame != self.per_frame or ctx.shape[1] != self.Lctx or shared != self.shared or \
                    (fps is None) != (self.fps is None):
                return False
        # everything that can fail (allocations, the K/V GEMM) runs BEFORE the first write into a buffer a graph reads
        kv = model._context_kv(ctx.contiguous(), dev) if nU else None   # prompt constant: once per prompt
        if not first and kv is not None and kv.shape != self.kv.shape:
            return False
        if extra is not None:
            self.x_units[:, self.C_lat:] = extra.to(dev).float()[idx]
        if first:
            self.kv, self.fps = kv, fps
            self.per_frame, self.Lctx, self.shared = per_frame, ctx.shape[1], shared
        else:
            if kv is not None:
                self.kv.copy_(kv)
            if fps is not None:
                self.fps.copy_(fps)
        self.kwargs_ref = [dict(kw) for kw in kwargs_list]         # keeps the keyed tensors alive (ids stay unique)
        self._last_out = None
        return True

    # -- inputs --------------------------------------------------------------------------------------
    def load(self, xt, t):
        """x_t -> latent channels of every unit slot, t -> t_units (two broadcast copies)."""
        if len(self.units) == 0:
            return
        if self.full:
            if xt is not self._last_out:       # after a fused step the update kernel already wrote the slots
                self.x_units.view(self.G, self.B, *self.x_units.shape[1:])[:, :, :self.C_lat].copy_(xt)
            self.t_units.view(self.G, self.B).copy_(t.to(self.t_units.dtype))
        else:
            if self._bidx is None:                  # prompt index of every local unit (built once: no per-step H2D copy)
                self._bidx = torch.tensor([u % self.B for u in self.units], dtype=torch.long, device=self.device)
            self.x_units[:, :self.C_lat].copy_(xt.float().index_select(0, self._bidx))
            self.t_units.copy_(t.to(self.t_units.dtype).index_select(0, self._bidx))
        self._last_out = None

    # -- launch sequences ----------------------------------------------------------------------------
    def _model_launches(self):
        m = self.model
        nU = len(self.units)
        if self.emb_tab is not None:
            emb = ops.backend().gather_rows_f32(self.emb_tab, self.t_units)
        else:
            emb = m._embed(self.t_units, self.fps, nU, self.device)
        m._body(self.x_units, emb, self.kv, self.Lctx, self.per_frame, out=self.out, shared_groups=self.shared)

    def _run(self, key, launches, graph_ok=True):
        """Run `launches()` eagerly (first call: warms the allocator, JIT-free) and from then on as a graph.
        graph_ok=False: this launch sequence cannot be captured (a host-side collective inside it): always eager."""
        if not self.use_graph or not graph_ok:
            launches()
            return
        g = self._graphs.get(key)
        if g is None:
            st = self._static.setdefault(key, {"calls": 0})
            st["calls"] += 1
            if st["calls"] == 1:
                launches()                      # eager warm-up pass with the caller's inputs
                return
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            try:
                # thread_local: a watchdog thread of the process group may touch the runtime while we capture
                with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                    launches()
            except Exception as ex:             # noqa: BLE001 — capture is an optimisation: stay correct, say so
                warnings.warn(f"UnitSession: hipGraph capture failed ({type(ex).__name__}: {ex}); running eagerly")
                torch.cuda.synchronize(self.device)
                self.use_graph = False
                launches()
                return
            self._graphs[key] = g
        g.replay()

    # -- model only (samplers with their own update) ----------------------------------------------------
    def eval(self, xt, t):
        """-> tuple of G unit outputs [B, out_dim, F, H, W] (views of the session's static buffer: consume them
        before the next call).  Partitioned sessions return the local units' outputs stacked instead."""
        self.load(xt, t)
        if len(self.units):
            self._run("model", self._model_launches)
        if not self.full:
            return self.out
        return tuple(self.out.view(self.G, self.B, *self.out.shape[1:])[g] for g in range(self.G))

    # -- model + fused CFG / DDIM update ------------------------------------------------------------------
    def ddim_step(self, xt, t, coef_tab, guide, mean_type, noise=None, clone=True):
        """One denoise step.  coef_tab: [T+1, 7] fp32 table of the update's coefficients per integer timestep
        (vgen_cfg_ddim_step); t [B] long.  Returns (x_{t-1}, x0): fresh tensors, or with clone=False the session's
        own buffers (valid until the next step; passing that x_{t-1} back in skips the input copy)."""
        assert self.full and self.G in (1, 2)
        self.load(xt, t)
        use_guide = self.G == 2
        nz = None
        if noise is not None:
            nz = self._static.setdefault("noise", torch.empty_like(self.xt_1))
            nz.copy_(noise)
        key = ("ddim", id(coef_tab), float(guide), int(mean_type), noise is not None)
        self._static[("tab",) + key] = coef_tab                     # keep the table alive while a graph reads it
        outv = self.out.view(self.G, self.B, *self.out.shape[1:])

        def launches():
            self._model_launches()
            ops.backend().ddim_update_units(self.x_units, self.G, self.B, self.C_lat, outv[0],
                                            outv[1] if use_guide else None, nz, coef_tab, self.t_units[:self.B],
                                            guide, use_guide, mean_type, self.xt_1, self.x0, replicate=True)

        self._run(key, launches)
        if clone:
            return self.xt_1.clone(), self.x0.clone()
        self._last_out = self.xt_1
        return self.xt_1, self.x0


class SessionCache:
    """Small LRU of UnitSessions.  The engines build new kwarg tensors per prompt, so every prompt is a new key:
    capacity 2 = the prompt in flight plus one sibling (e.g. the inversion pass and the CFG pass of the SR600 stage);
    `clear()` frees both.  r04: when the cache is full and a new prompt has the STRUCTURE of a cached session (same
    model / weights epoch / latent shape / unit set, same kwarg names with tensors of the same shapes), the least
    recently used such session is RE-BOUND to the new prompt (UnitSession._bind) instead of evicted and rebuilt: its
    graphs and memory pool live on, the new prompt pays one K/V GEMM — not a warm-up pass and a capture."""

    def __init__(self, capacity=2):
        self.capacity = capacity
        self._items = {}
        self._struct = {}           # key -> structure key of its session
        self.rebinds = 0

    def get(self, model, shape, device, kwargs_list, t_dtype, num_timesteps, units=None):
        inner = getattr(model, "module", model)              # DistributedDataParallel wrapper of the engines
        if not hasattr(inner, "_prepare_units") or not hasattr(inner, "_body"):
            return None
        if getattr(inner, "_auto_cal", None):                 # `calibration: auto`: before anything is packed, staged or
            inner._maybe_auto_calibrate(tuple(shape), device, kwargs_list[0], t_dtype)      # captured (bumps inner._epoch)
        try:
            kkey = _kw_key(kwargs_list)
        except Unkeyable:
            return None                                      # no stable identity: evaluate without a session
        base = (id(inner), inner._epoch, tuple(shape), str(device), t_dtype, num_timesteps,
                None if units is None else tuple(units), ops.backend().name)
        key = base + (kkey,)
        s = self._items.pop(key, None)
        skey = self._struct.pop(key, None)
        if s is None:
            skey = base + (_kw_struct(kwargs_list),)
            if len(self._items) >= self.capacity and _REBIND_ON:
                for old in list(self._items):                   # insertion order = least recently used first
                    if self._struct.get(old) != skey:
                        continue
                    try:
                        ok = self._items[old]._bind(kwargs_list)
                    except Exception:
                        # _bind rewrites buffers the captured graphs read: a failure midway (OOM in the K/V GEMM, a shape
                        # mismatch in a copy) must not leave a half-rebound session filed under the OLD prompt's key
                        #  — the entry is dropped
                        self._items.pop(old, None)
                        self._struct.pop(old, None)
                        raise
                    if ok:
                        s = self._items.pop(old)
                        self._struct.pop(old, None)
                        self.rebinds += 1
                        break
            if s is None:
                try:
                    s = UnitSession(inner, tuple(shape), device, kwargs_list, t_dtype, num_timesteps, units)
                except ValueError:
                    return None
                while len(self._items) >= self.capacity:
                    old = next(iter(self._items))
                    self._items.pop(old)
                    self._struct.pop(old, None)
        self._items[key] = s
        self._struct[key] = skey
        return s

    def clear(self):
        self._items.clear()
        self._struct.clear()


def eval_units(cache, partition, model, xt, t, kwargs_list, num_timesteps=None):
    """Model evaluation shared by the samplers: G kwarg sets on the latent batch xt as units -> tuple of G outputs
    (partitioned over the ranks when `partition` is set, else one session replay), or None when `model` is not a
    vgen_amd unit model (the caller then calls it branch by branch)."""
    inner = getattr(model, "module", model)
    if partition is not None:
        return partition.run_units(inner, xt, t, list(kwargs_list), num_timesteps)
    if xt.dim() == 5 and cache is not None:
        nt = num_timesteps if t.dtype == torch.long else None
        sess = cache.get(inner, tuple(xt.shape), xt.device, list(kwargs_list), t.dtype, nt)
        if sess is not None:
            return sess.eval(xt, t)
    if hasattr(inner, "forward_units"):
        return inner.forward_units(xt, t, list(kwargs_list))
    return None
