"""Choose the layers that carry two-term (dual-W) weights in precision="mixed".

TEST / TUNING TOOLING (imports oracle/).  Inputs:
  * per-module weight-rounding sensitivity of the full-size t2v UNet (tools/parity_attrib.py --full --by-module
    --set w_lin,w_conv): rel-L2 vs the reference's fp32 forward with ONLY that module's packed weights rounded;
  * per-launch timings of the same step in both modes (bench.py --dump-shapes: profiles/*tapgemm_shapes*).
One recording pass of the host logic maps every module to its launch signatures; the extra time of running a module's
launches dual-W is its cost, the squared sensitivity its benefit (errors add in quadrature), and modules are taken
greedily by benefit / cost until the predicted remaining weight error is below --target.

    python tools/mixed_policy.py --sens profiles/r03_weight_sensitivity.json \
        --fast profiles/r02a_tapgemm_shapes_t2v.json --high profiles/r03a_tapgemm_shapes_t2v_high.json --target 0.45e-3
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def launches_by_module():
    import parity_attrib as pa
    from vgen_amd import ops
    m, x, t, y, ref, dt = pa.build(True, "fp16")
    be = pa.KnobEmu(dt, [])
    pa.scope_modules(m, be)
    rec = {}
    orig = be.tapgemm

    def tapgemm(g):
        K = g.taps * g.C1 + g.C2
        # the bench step evaluates 2 units: twice the rows of this single forward
        rec.setdefault(be.scope, []).append((g.mode, 2 * g.M, g.N, K, g.epilogue, "torch.float32" if g.out_dtype == torch.float32 else "torch.float16"))
        return orig(g)

    be.tapgemm = tapgemm
    prev = ops.set_backend(be)
    try:
        with torch.no_grad():
            m(x, t, y=y)
    finally:
        ops.set_backend(prev)
    return rec


def table(path, strip_dw=False):
    d = json.load(open(path))
    out = {}
    for sig, n, ms, _ in d["rows"]:
        sig = list(sig)
        sig[5] = sig[5].replace("+dw", "")
        out[tuple(sig)] = ms / n
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sens", required=True)
    ap.add_argument("--fast", required=True)
    ap.add_argument("--high", required=True)
    ap.add_argument("--target", type=float, default=0.45e-3, help="remaining weight-rounding rel-L2 to aim for")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sens = {k.split("module:")[1].split(" ")[0]: v for k, v in json.load(open(args.sens)).items() if k.startswith("module:")}
    fast, high = table(args.fast), table(args.high)
    rec = launches_by_module()
    rows = []
    for mod, e in sens.items():
        sigs = rec.get(mod, [])
        miss = [s for s in sigs if s not in fast or s not in high]
        cost = sum(high[s] - fast[s] for s in sigs if s in fast and s in high)
        rows.append(dict(module=mod, rel_l2=e, energy=e * e, launches=len(sigs), extra_ms=cost, unmatched=len(miss)))
    tot = sum(r["energy"] for r in rows)
    rows.sort(key=lambda r: -(r["energy"] / max(r["extra_ms"], 1e-3)))
    left, cost, chosen = tot, 0.0, []
    for r in rows:
        if left <= args.target ** 2:
            break
        chosen.append(r["module"])
        left -= r["energy"]
        cost += r["extra_ms"]
        r["chosen"] = True
    print(f"total weight-rounding energy {tot:.3e} (rel-L2 {tot ** 0.5:.3e}); chosen {len(chosen)} of {len(rows)} modules, "
          f"predicted remaining {max(left, 0) ** 0.5:.3e}, extra time {cost:.2f} ms of {sum(r['extra_ms'] for r in rows):.2f} ms (all modules)")
    for r in rows:
        print(f"{'*' if r.get('chosen') else ' '} {r['module']:36s} rel-L2 {r['rel_l2']:.3e}  extra {r['extra_ms']:.3f} ms  ({r['launches']} launches, {r['unmatched']} unmatched)")
    if args.out:
        json.dump(dict(target=args.target, chosen=sorted(chosen), rows=rows), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
