"""TEST TOOLING (imports oracle/): the (kind x level) frontier of two-term weight sets for precision="mixed" — accuracy side
only, on the ABI emulator (CPU; within ~1 % of the GPU on every fixture measured).  The full-size t2v model is packed in "high"
(every weight two-term) and every weight outside the candidate set is stripped of its W_lo operand, so any set is one
forward away.  Output: error energy (rel-L2 squared, 1e-8) each candidate move adds or removes relative to the default rule.

    python tools/mixed_frontier.py      -> profiles/r04_mixed_frontier.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import full_cases as fc  # noqa: E402
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402


def entries(m):
    """(level, side, kind, weight tensor) of every packed 16-bit weight of the trunk"""
    P, lv = m._packed, m._block_levels()
    out = []
    for name, d in P.items():
        if name == "kv_all":
            out.append((0, "glue", "kv", d))
            continue
        if name == "head_conv":
            out.append((0, "glue", "head", d[0]))
            continue
        parts = name.split(".")
        top = parts[0] if parts[0] == "middle_block" else ".".join(parts[:2])
        if top not in lv:
            continue
        side, level = lv[top]
        if isinstance(d, tuple):                                    # Down / Upsample conv
            out.append((level, side, "resample", d[0]))
            continue
        if not isinstance(d, dict):
            continue
        for k, v in d.items():
            if k == "tb":
                for kk, vv in v.items():
                    w = vv[0] if isinstance(vv, tuple) else vv
                    if torch.is_tensor(w) and w.dtype == torch.float16:
                        kind = {"qkv1": "qkv", "qkv2": "qkv", "o1": "o", "o2": "o"}.get(kk, kk)
                        out.append((level, side, kind, w))
            else:
                w = v[0] if isinstance(v, tuple) else v
                if torch.is_tensor(w) and w.dtype == torch.float16:
                    kind = "tconv" if k.startswith("tconv") else k
                    out.append((level, side, kind, w))
    return out


def main():
    g = fc.load("t2v")
    ops.set_backend(EmuBackend())
    m = fc.build("t2v", g, "high")
    m.pack()
    ents = [(l, s, k, w, w.vgen_dw) for l, s, k, w in entries(m) if getattr(w, "vgen_dw", None) is not None]
    x, kw = fc.inputs("t2v", g)
    ref = g["out"].float()
    SINGLE = ("ff1", "ff2", "q2")

    def default(l, s, k):
        return l == 0 and s in ("enc", "dec", "glue") and k not in SINGLE

    def run(keep):
        for l, s, k, w, dw in ents:
            if keep(l, s, k):
                w.vgen_dw = dw
            elif hasattr(w, "vgen_dw"):
                del w.vgen_dw
        with torch.no_grad():
            o = m(x, g["t"], **kw)
        return float((o - ref).norm() / ref.norm())

    base = run(default)
    print(f"default rule: {base:.4e}", flush=True)
    res = {"default_rule": base, "moves": {}}
    kinds1 = ["qkv", "o", "pin", "pout", "conv1", "conv2", "tconv", "ff1", "ff2", "resample"]
    moves = [(f"+{k}@L1", (lambda kk: lambda l, s, k: default(l, s, k) or (l == 1 and s != "mid" and k == kk))(k)) for k in kinds1]
    moves += [(f"+{k}@L2", (lambda kk: lambda l, s, k: default(l, s, k) or (l == 2 and s != "mid" and k == kk))(k))
              for k in ("qkv", "o", "conv2", "tconv")]
    moves += [(f"-{k}@L0", (lambda kk: lambda l, s, k: default(l, s, k) and k != kk)(k))
              for k in ("conv1", "o", "tconv", "conv2", "pin", "pout", "qkv", "resample", "kv", "head")]
    for name, keep in moves:
        e = run(keep)
        d = (e * e - base * base) * 1e8
        res["moves"][name] = {"rel_l2": e, "error_energy_delta_1e-8": round(d, 2)}
        print(f"{name:14s} {e:.4e}  ({d:+.2f}e-8)", flush=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r04_mixed_frontier.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
