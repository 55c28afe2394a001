"""TEST TOOLING (imports oracle/ and, in the build container, the reference): how the rel-L2 of precision="mixed" depends on the
WEIGHT DRAW — the full-size t2v UNet with fresh seeded weights per case (both recipes of vgen_amd/synth.py: Gaussian and the
heavy-tailed Student-t nu = 4), a fresh input and timestep per case, the reference's fp32 forward (oracle/ref_import.py) run
beside the host logic on the ABI emulator (which reproduces the GPU's roundings to ~1 %: tools/emu_parity.py).  VERDICT r03 weak
#1a: the 1e-3 claim rested on one (weights, input, t) triple; the three golden fixtures + this sweep are the answer.

    python tools/emu_seed_sweep.py [precision ...]        -> profiles/r04_emu_seed_sweep.json
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

CASES = [("gauss", 11, 941), ("gauss", 12, 621), ("gauss", 13, 301), ("gauss", 14, 61),
         ("student4", 21, 881), ("student4", 22, 441), ("student4", 23, 121)]


def main():
    precisions = sys.argv[1:] or ["mixed"]
    g = fc.load("t2v")
    R = ref_import.load()
    out_path = os.path.join(ROOT, "profiles", "r04_emu_seed_sweep.json")
    res = json.load(open(out_path))["rel_l2"] if os.path.exists(out_path) else {}
    ops.set_backend(EmuBackend())
    for recipe, seed, tval in CASES:
        sd = torch_ref.synth_state_dict(g["shapes"], seed=seed, recipe=recipe)
        ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
        ref.load_state_dict(sd, strict=True)
        gen = torch.Generator("cpu").manual_seed(7000 + seed)
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        y = torch.randn(1, 77, 1024, generator=gen)
        t = torch.tensor([tval])
        with torch.no_grad():
            out = ref(x, t, y=y)
        del ref
        for pr in precisions:
            gg = dict(g, seed=seed, recipe=recipe)
            m = fc.build("t2v", gg, pr)
            t0 = time.time()
            with torch.no_grad():
                o = m(x, t, y=y)
            e = float((o - out).norm() / out.norm())
            res.setdefault(pr, {})[f"{recipe}/seed{seed}/t{tval}"] = e
            print(f"{pr} {recipe} seed {seed} t={tval}: {e:.4e} (std {float(out.std()):.3f}, {time.time() - t0:.0f} s)", flush=True)
            del m
        json.dump({"what": "emulated rel-L2 of the full-size t2v UNet vs the reference's fp32 forward, fresh seeded weights "
                           "(recipe/seed), input (seed 7000 + weight seed) and timestep per case; 'mixed' = the committed default rule",
                   "rel_l2": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
