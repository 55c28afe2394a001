"""TEST TOOLING (imports oracle/): predict, on the CPU, the rel-L2 of the HIP path against a full-size golden fixture by
running the host logic on the ABI emulator (oracle/abi_emulator.py reproduces the GPU's roundings to ~3 digits: full t2v
fp16 1.3721e-3 emulated vs 1.3746e-3 on MI355X).  Used to check the level rule of precision="mixed" on new fixtures
without spending GPU minutes.

    python tools/emu_parity.py t2v_b mixed [fast ...]      fixtures: tests/full_cases.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import full_cases as fc  # noqa: E402
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402


def main():
    name = sys.argv[1]
    g = fc.load(name)
    ops.set_backend(EmuBackend())
    for precision in sys.argv[2:] or ["mixed"]:
        m = fc.build(name, g, precision)
        t0 = time.time()
        err, nr = fc.error(fc.forward(name, m, g), g)
        print(f"{name} fp16/{precision}: emulated rel-L2 {err:.4e}  norm ratio {nr:.5f}  ({time.time() - t0:.0f} s)", flush=True)
        del m


if __name__ == "__main__":
    main()
