"""TEST TOOLING (imports oracle/ and, in the build container, the reference): how the rel-L2 of precision="mixed" moves along
the sampling trajectory — the full-size t2v UNet on the headline weights, one input per timestep, the reference's fp32 forward
(oracle/ref_import.py) against the host logic on the ABI emulator (which reproduces the GPU's roundings to ~1 %: tools/emu_parity.py).

    python tools/emu_tsweep.py [precision ...]        -> profiles/r04_emu_tsweep.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import full_cases as fc  # noqa: E402
from oracle import ref_import, torch_ref  # noqa: E402
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402


def main():
    precisions = sys.argv[1:] or ["mixed"]
    g = fc.load("t2v")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    R = ref_import.load()
    ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
    ref.load_state_dict(sd, strict=True)
    cases = []
    for tval in (1, 21, 261, 501, 741, 981):
        gen = torch.Generator("cpu").manual_seed(9000 + tval)
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        y = torch.randn(1, 77, 1024, generator=gen)
        t = torch.tensor([tval])
        with torch.no_grad():
            cases.append((tval, x, y, t, ref(x, t, y=y)))
    del ref
    ops.set_backend(EmuBackend())
    res = {}
    for pr in precisions:
        m = fc.build("t2v", g, pr)
        for tval, x, y, t, out in cases:
            t0 = time.time()
            with torch.no_grad():
                o = m(x, t, y=y)
            e = float((o - out).norm() / out.norm())
            res.setdefault(pr, {})[str(tval)] = e
            print(f"{pr} t={tval}: {e:.4e} ({time.time() - t0:.0f} s)", flush=True)
        del m
    out_path = os.path.join(ROOT, "profiles", "r04_emu_tsweep.json")
    if os.path.exists(out_path):                      # merge: earlier sweeps (other rules / modes) stay in the file
        old = json.load(open(out_path))["rel_l2"]
        res = {**old, **res}
    json.dump({"what": "emulated rel-L2 of the full-size t2v UNet (headline weights, seed 0) vs the reference's fp32 forward, one "
                       "input per timestep (input seed 9000 + t)", "rel_l2": res},
              open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
