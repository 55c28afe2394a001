"""TEST TOOLING (imports oracle/): what would 16-bit GroupNorm INPUTS cost in parity, where the tensor has no other reader?

Inside a ResBlock (util.py:900-927 + TemporalConvBlock_v2 :1686-1697) six tensors feed a GroupNorm:
    0  x (the residual stream / skip concat)        fp32 needed (it is also the residual / skip operand)
    1  conv1 output h                                read ONLY by GroupNorm 2
    2  conv2 output (+ skip) = the block's h         fp32 needed (the temporal block adds it back at the end)
    3-5 temporal conv 1-3 outputs                    read ONLY by the next temporal GroupNorm
Four of the six could leave the producing tap-GEMM as 16-bit rows (half the epilogue's stores — the 10 B/clk/CU path of
DESIGN §3.1 — and half the norm's reads: 6 -> 3 B/element for those launches); the statistics would still come from the fp32
accumulators (column partials in the producer's epilogue), only the VALUE the norm normalises is rounded once more.
Emulated: those four GroupNorm inputs rounded to the compute type, their producer statistics kept, everything else unchanged.
    python tools/emu_gn16.py t2v [t2v_b ...]          (precision "mixed", fp16)
"""
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

ELIGIBLE = (1, 3, 4, 5)


class EmuGn16(EmuBackend):
    idx = None          # position of the next GroupNorm inside the running ResBlock (None: not inside one)
    rounded = 0
    elements = 0

    def groupnorm(self, x1, x2, nb, S, groups, eps, gamma, beta, silu, want_raw, dt):
        i = EmuGn16.idx
        if i is not None:
            EmuGn16.idx = i + 1
            if i in ELIGIBLE:
                assert x2 is None and not want_raw
                r = x1.to(dt).float()
                cs = getattr(x1, "vgen_cs", None)
                if cs is not None:
                    r.vgen_cs = cs                  # statistics: still the producer's fp32 column partials
                EmuGn16.rounded += 1
                EmuGn16.elements += x1.numel()
                x1 = r
        return super().groupnorm(x1, x2, nb, S, groups, eps, gamma, beta, silu, want_raw, dt)


def main():
    for name in sys.argv[1:] or ["t2v"]:
        g = fc.load(name)
        for label, be in (("fp32 GroupNorm inputs [product]", EmuBackend()), ("16-bit inputs where the norm is the only reader", EmuGn16())):
            ops.set_backend(be)
            EmuGn16.rounded = EmuGn16.elements = 0
            m = fc.build(name, g, "mixed")
            if isinstance(be, EmuGn16):
                inner = type(m)._resblock

                def rb(self, *a, _inner=inner, **k):
                    EmuGn16.idx = 0
                    try:
                        return _inner(self, *a, **k)
                    finally:
                        assert EmuGn16.idx == 6, EmuGn16.idx
                        EmuGn16.idx = None
                m._resblock = rb.__get__(m)
            t0 = time.time()
            err, _ = fc.error(fc.forward(name, m, g), g)
            extra = (f"  ({EmuGn16.rounded} norm inputs rounded, {EmuGn16.elements * 2 / 1e9:.2f} GB fewer bytes written "
                     f"and as many fewer read per forward)" if isinstance(be, EmuGn16) else "")
            print(f"{name} fp16/mixed, {label}: emulated rel-L2 {err:.4e}  ({time.time() - t0:.0f} s){extra}", flush=True)
            del m


if __name__ == "__main__":
    main()
