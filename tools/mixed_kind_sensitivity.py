"""TEST TOOLING (imports oracle/): which LAYER KINDS of the level-0 two-term weight set of precision="mixed" buy how much —
the full-size t2v fixture on the ABI emulator (CPU; reproduces the GPU's roundings to ~1 %), the model packed in "mixed"
and then ONE kind at a time stripped of its W_lo term (its `.vgen_dw` operand removed: those launches run single-pass).
Error energy (rel-L2 squared, units 1e-8) added per kind, next to the per-step time the kind's dual-W launches cost on the
GPU (profiles/r04b_tapgemm_shapes_t2v_mixed.json vs r03b ..._fast.json).

    python tools/mixed_kind_sensitivity.py      -> profiles/r04_mixed_kind_sensitivity.json
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

KINDS = {"conv1": ("res", "conv1"), "conv2": ("res", "conv2"), "tconv": ("res", "tconv"), "pin": ("tx", "pin"),
         "pout": ("tx", "pout"), "qkv": ("tb", "qkv"), "q2": ("tb", "q2"), "o": ("tb", "o"), "ff1": ("tb", "ff1"),
         "ff2": ("tb", "ff2"), "glue(down/up/kv/head)": ("glue", "")}


def weights_of(P, kind):
    """the packed weight tensors of one kind across all modules"""
    where, key = KINDS[kind]
    out = []
    for name, d in P.items():
        if where == "glue":
            if name in ("kv_all",):
                out.append(d)
            elif name == "head_conv" or (isinstance(d, tuple) and len(d) == 2 and torch.is_tensor(d[0]) and d[0].dim() == 2
                                         and name not in ("te0", "te2", "fe0", "fe2", "emb_all", "conv_in")):
                out.append(d[0])
            continue
        if not isinstance(d, dict):
            continue
        src = d.get("tb", {}) if where == "tb" else d
        for k, v in src.items():
            if k.startswith(key) and (where != "tx" or k == key):
                out.append(v[0] if isinstance(v, tuple) else v)
    return [w for w in out if torch.is_tensor(w)]


def main():
    g = fc.load("t2v")
    ops.set_backend(EmuBackend())
    m = fc.build("t2v", g, "mixed")
    m.pack()
    P = m._packed
    x, kw = fc.inputs("t2v", g)
    ref = g["out"].float()

    def err():
        with torch.no_grad():
            o = m(x, g["t"], **kw)
        return float((o - ref).norm() / ref.norm())

    base = err()
    print(f"mixed, nothing stripped: {base:.4e}", flush=True)
    res = {"mixed": base, "kinds": {}}
    for kind in KINDS:
        ws = [w for w in weights_of(P, kind) if getattr(w, "vgen_dw", None) is not None]
        saved = [(w, w.vgen_dw) for w in ws]
        for w in ws:
            del w.vgen_dw
        e = err()
        for w, dw in saved:
            w.vgen_dw = dw
        res["kinds"][kind] = {"two_term_weights": len(ws), "rel_l2_without": e,
                              "error_energy_added_1e-8": round((e * e - base * base) * 1e8, 2)}
        print(f"{kind:24s} {len(ws):3d} weights  -> {e:.4e}  (+{(e * e - base * base) * 1e8:.2f}e-8)", flush=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r04_mixed_kind_sensitivity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
