"""Attribute the UNet's rel-L2 error vs the reference golden to the individual 16-bit rounding points.

TEST TOOLING (imports oracle/): the host logic of vgen_amd.unet runs on a CPU emulator of the C ABI
(oracle/abi_emulator.py) whose rounding points are switchable per category.  The model is packed in
fp32 and every category below is rounded to the 16-bit type only when its knob is on, so

    only-one     : variance contribution of that category alone (errors add in quadrature)
    leave-one-out: what removing that rounding would buy

    python tools/parity_attrib.py                 # tiny fixture, fp16, seconds
    python tools/parity_attrib.py --dtype bf16
    python tools/parity_attrib.py --full          # BASELINE config-2 shape (minutes per variant on CPU)
    python tools/parity_attrib.py --set gn_out,ln_out   # one custom knob set

Categories (where the HIP path rounds to 16 bit):
    w_conv, w_lin   packed weights of the 3x3 / temporal convs, of the linears
    gn_out, ln_out  GroupNorm(+SiLU) / LayerNorm outputs (GEMM A operands)
    qkv, q2, kv     attention projections emitted 16-bit (self q|k|v, cross q, cross k|v of the context)
    geglu, ff2      GEGLU output (A operand of ff2), FF output + token stream emitted 16-bit (A operand of proj_out)
    attn_p, attn_o  softmax probabilities before the PV product, attention output (A operand of the o-projection)
    cast            plain casts: Down/Upsample conv inputs, SiLU(emb), the ResBlock skip operand ("raw")
    temb, ctx, stem sinusoidal embedding, context tokens, im2col'd latent
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import torch_ref  # noqa: E402
from oracle.abi_emulator import EmuBackend, _strided  # noqa: E402
from vgen_amd import lib as L  # noqa: E402
from vgen_amd import ops  # noqa: E402

CATS = ["w_conv", "w_lin", "gn_out", "ln_out", "qkv", "q2", "kv", "geglu", "ff2", "attn_p", "attn_o",
        "cast", "temb", "ctx", "stem"]


class KnobEmu(EmuBackend):
    """EmuBackend whose 16-bit rounding points are individually switchable; tensors stay fp32."""
    name = "emu-knobs"

    def __init__(self, dt, on, scopes=None):
        self.dt = dt
        self.on = set(on)
        self.scopes = None if scopes is None else set(scopes)   # module names whose roundings are active
        self.scope = "glue"

    def r(self, x, cat):
        live = cat in self.on and (self.scopes is None or self.scope in self.scopes)
        return x.to(self.dt).float() if live else x.float()

    def groupnorm(self, x1, x2, nb, S, groups, eps, gamma, beta, silu, want_raw, dt):
        y, raw = super().groupnorm(x1, x2, nb, S, groups, eps, gamma, beta, silu, want_raw, torch.float32)
        return self.r(y, "gn_out"), (self.r(raw, "cast") if raw is not None else None)

    def layernorm(self, x, gamma, beta, eps, dt):
        return self.r(super().layernorm(x, gamma, beta, eps, torch.float32), "ln_out")

    def tapgemm(self, g):
        K = g.taps * g.C1 + g.C2
        W = self.r(g.W[: g.N, :K], "w_lin" if g.mode == L.TAP_LINEAR else "w_conv")
        acc = torch.zeros((g.M, g.N), dtype=torch.float32)
        for tap, rws in enumerate(self._src_rows(g)):
            a = g.A[:, : g.C1][rws.clamp(min=0)].float()
            a = torch.where((rws >= 0)[:, None], a, torch.zeros_like(a))
            acc += a @ W[:, tap * g.C1:(tap + 1) * g.C1].t()
        if g.C2:
            acc += g.A2[: g.M, : g.C2].float() @ W[:, g.taps * g.C1:].t()
        if g.bias is not None:
            acc += g.bias[: g.N]
        if g.rowbias is not None:
            acc += g.rowbias[torch.arange(g.M) // g.rows_per_rb][:, : g.N]
        n_out = g.N
        if g.epilogue == L.EPI_GEGLU:
            v = acc.view(g.M, g.N // 32, 2, 16)
            val, gate = v[:, :, 0], v[:, :, 1]
            acc = (val * (0.5 * gate * (1.0 + torch.erf(gate * 0.7071067811865476)))).reshape(g.M, g.N // 2)
            n_out = g.N // 2
        if g.residual is not None:
            acc += g.residual[:, :n_out]
        cat = getattr(g, "role", None)
        if cat is not None:
            acc = self.r(acc, cat)
        if g.out is not None:
            g.out[:, :n_out] = acc
            out = g.out
        else:
            out = acc
        if g.colstats:
            ns = (g.M + 63) // 64
            pad = torch.zeros((ns * 64, g.N), dtype=torch.float32)
            pad[: g.M] = acc
            pv = pad.view(ns, 64, g.N)
            out.vgen_cs = torch.stack([pv.sum(1), (pv * pv).sum(1)], 1).contiguous()
        return out

    def attention(self, g):
        no, ni = g.nbatch // g.inner, g.inner

        def seqs(t, s, n):
            rs, bo, bi = s
            return _strided(t, (no, ni, g.heads, n, 64), (bo, bi, 64, rs, 1)).float()

        q, k, v = seqs(g.q, g.q_s, g.nq), seqs(g.k, g.k_s, g.nk), seqs(g.v, g.v_s, g.nk)
        s = q @ k.transpose(-1, -2) * g.scale
        # the kernels round the UNNORMALISED probabilities exp(s - max) and divide by the fp32 row sum afterwards
        e = torch.exp(s - s.amax(-1, keepdim=True))
        o = (self.r(e, "attn_p") @ v) / e.sum(-1, keepdim=True)
        rs, bo, bi = g.o_s
        _strided(g.out, (no, ni, g.heads, g.nq, 64), (bo, bi, 64, rs, 1)).copy_(self.r(o, "attn_o"))
        return g.out

    def act_cast(self, x, act, dt):
        return self.r(super().act_cast(x, act, torch.float32), getattr(self, "_cast_cat", "cast"))

    def timestep_embedding(self, t, dim, dt):
        return self.r(super().timestep_embedding(t, dim, torch.float32), "temb")

    def im2col3x3_small(self, src, nimg, Fi, Cin, H, W, strides, Kpad, dt, split=False):
        return self.r(super().im2col3x3_small(src, nimg, Fi, Cin, H, W, strides, Kpad, torch.float32, split=split), "stem")


def tag_roles(model):
    """Label the 16-bit-output GEMMs of vgen_amd.unet by role (the model passes out_dtype=dt for exactly these)."""
    orig = model._linear

    def linear(A, Wb, M, **kw):
        W = Wb[0] if isinstance(Wb, tuple) else Wb
        role = None
        if "out_dtype" in kw:          # the model names an output dtype only for its 16-bit outputs
            n, k = W.shape[0], A.shape[1]
            if kw.get("epilogue", L.EPI_NONE) == L.EPI_GEGLU:
                role = "geglu"
            elif n == 3 * k:
                role = "qkv"
            elif k == 4 * n:
                role = "ff2"
            elif k == model.context_dim and n == model._packed["kv_width"]:
                role = "kv"
            else:
                role = "q2"
        kw["colstats"] = kw.get("colstats", False) and M >= ops.COLSTATS_MIN_ROWS
        b = Wb[1] if isinstance(Wb, tuple) else None
        g = ops.TapGemm(A=A, W=W, M=M, N=W.shape[0], C1=A.shape[1], bias=b, **kw)
        g.role = role
        return ops.backend().tapgemm(g)

    model._linear = linear
    return orig


def build(full, dtname):
    from conftest import gold
    from vgen_amd.unet import UNetSD_T2VBase
    g = gold("unet_t2v_full.pt" if full else "unet_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    m = UNetSD_T2VBase(**g["cfg"], compute_dtype=dtname, precision="fast").eval()
    m.load_state_dict(sd, strict=True)
    del sd
    if full:
        gen = torch.Generator("cpu").manual_seed(g["input_seed"])
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        y = torch.randn(1, 77, 1024, generator=gen)
    else:
        x, y = g["x"], g["y"]
    dt = m.compute_dtype
    m.compute_dtype = torch.float32          # pack fp32; KnobEmu decides what is rounded to `dt`
    tag_roles(m)
    return m, x, g["t"], y, g["out"].float(), dt


def scope_modules(m, be):
    """Make the emulator know which top-level module (ResBlock / Spatial / TemporalTransformer) is executing."""
    names = []
    for attr in ("_resblock", "_spatial_tx", "_temporal_tx"):
        orig = getattr(type(m), attr)

        def wrapped(mod, *a, _orig=orig, **k):
            be.scope = mod._pname
            try:
                return _orig(m, mod, *a, **k)
            finally:
                be.scope = "glue"
        setattr(m, attr, wrapped)
    for name, mod in m.named_modules():
        if type(mod).__name__ in ("_ResBlockP", "_SpatialTransformerP", "_TemporalTransformerP"):
            names.append(name)
    return names


def run(m, x, t, y, ref, dt, on, scopes=None):
    be = KnobEmu(dt, on, scopes)
    scope_modules(m, be)
    prev = ops.set_backend(be)
    try:
        # ctx cast goes through act_cast(…, 0): give it its own category for the duration of _trunk's prologue
        orig_cast = be.act_cast
        state = {"n": 0}

        def act_cast(xx, act, d):
            # call order in _trunk: SiLU(h) of time_embed [, fps], SiLU(e), ctx, then Down/Up casts
            state["n"] += 1
            be._cast_cat = "ctx" if (act == 0 and xx.dim() == 2 and xx.shape[-1] == m.context_dim) else "cast"
            return orig_cast(xx, act, d)

        be.act_cast = act_cast
        with torch.no_grad():
            out = m(x, t, y=y)
    finally:
        ops.set_backend(prev)
    return float((out - ref).norm() / ref.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--set", default=None, help="comma list of categories to round (one run)")
    ap.add_argument("--only", action="store_true", help="only-one sweep (default: only-one + leave-one-out)")
    ap.add_argument("--by-module", action="store_true", help="all categories, one module at a time")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    m, x, t, y, ref, dt = build(args.full, args.dtype)
    res = {"dtype": args.dtype, "shape": "full" if args.full else "tiny"}

    def go(name, on, scopes=None):
        t0 = time.time()
        e = run(m, x, t, y, ref, dt, on, scopes)
        res[name] = e
        print(f"{name:24s} {e:.4e}   ({time.time() - t0:.1f} s)", flush=True)
        return e

    if args.by_module:
        cats = [c for c in args.set.split(",") if c] if args.set else CATS      # --by-module --set w_lin,w_conv: weights only
        go("all", cats)
        names = scope_modules(m, KnobEmu(dt, []))
        go("module:glue (stem/head/down/up/emb)", cats, ["glue"])
        for n in names:
            go("module:" + n, cats, [n])
            if args.out:
                json.dump(res, open(args.out, "w"), indent=1)
    elif args.set is not None:
        go("set:" + args.set, [c for c in args.set.split(",") if c])
    else:
        go("none (fp32)", [])
        go("all", CATS)
        for c in CATS:
            go("only:" + c, [c])
        if not args.only:
            for c in CATS:
                go("without:" + c, [k for k in CATS if k != c])
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
