"""TEST TOOLING (imports oracle/): the PRODUCT's calibration pass (vgen_amd/calibrate.py::calibrate_single_pass) on the
full-size fixtures, with the ABI emulator as the op backend — what `precision="high"` + calibration is predicted to measure
on the GPU (the emulator is within 1 % of the GPU on every fixture both have run).  The calibration input is never the
fixture's: other noise, prompt and timestep.      python tools/emu_calibrated.py t2v [k_max]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import full_cases as fc  # noqa: E402
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402
from vgen_amd.calibrate import brief_report, calibrate_single_pass  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "t2v"
    k_max = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    ops.set_backend(EmuBackend())
    g = fc.load(name)
    m = fc.build(name, g, "high")
    x, kw = fc.inputs(name, g)
    gen = torch.Generator("cpu").manual_seed(424242)
    xc = torch.randn(x.shape, generator=gen)
    kwc = {k: (torch.randn(v.shape, generator=gen) if v.is_floating_point() else v) for k, v in kw.items()}
    tc = torch.full_like(g["t"], 637)
    t0 = time.time()
    rep = calibrate_single_pass(m, xc, tc, k_max=k_max, **kwc)
    print(f"{name}: calibrate_single_pass(k_max={k_max}) on seed 424242 / t = 637: {time.time() - t0:.0f} s  {brief_report(rep)}", flush=True)
    fixtures = [(name, g)] + ([("t2v_c", fc.load("t2v_c"))] if name == "t2v" else [])
    for fname, fg in fixtures:
        t0 = time.time()
        err, nr = fc.error(fc.forward(fname, m, fg), fg)
        print(f"{fname} fp16, precision high + calibrate_single_pass: emulated rel-L2 {err:.4e}  norm ratio {nr:.5f}  "
              f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
