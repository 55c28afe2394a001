"""TEST TOOLING (imports oracle/): what would folding LayerNorm into its consumer GEMM cost in parity?  

    LN(x) . W^T = rstd_m . (x . (gamma o W)^T - mu_m . s) + c        s_n = sum_k gamma_k W_nk,  c_n = sum_k beta_k W_nk + b_n

The fused path would multiply a plain 16-bit copy of x (rounded BEFORE the mean is removed and the row is scaled) instead of
round16(LN(x)).  On the ABI emulator that is exactly: the consumer sees the UNROUNDED affine image of round16(x),
    a_eff = ((round16(x) - mu) * rstd) * gamma + beta        (mu, rstd from the fp32 row; fp32 carrier, no second rounding)
— the products against the 16-bit weights and the fp32 accumulation are unchanged (the rank-1 term and the row scale are fp32
epilogue arithmetic; gamma folded into the weight before ITS rounding instead of after is second order).  This tool runs the
full-size fixtures through the host logic on the emulator both ways and prints the rel-L2 against the reference's recorded
fp32 output:      python tools/emu_lnfold.py t2v [t2v_c i2vgen ...]      (precision = the benchmarked "mixed")
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


class _Carrier:
    """Stands in for a 16-bit operand whose numeric value is an fp32 tensor: EmuBackend.tapgemm only asks an operand for its
    dtype / strides, slices it and calls .float() on the slices."""
    def __init__(self, t16, t32):
        self.t16, self.t32 = t16, t32
        self.dtype, self.shape, self.device = t16.dtype, t16.shape, t16.device

    def stride(self, i=None):
        return self.t16.stride() if i is None else self.t16.stride(i)

    def __getitem__(self, idx):
        return _Carrier(self.t16[idx], self.t32[idx])

    def float(self):
        return self.t32


class EmuFold(EmuBackend):
    """LayerNorm hands its consumer the fold's effective operand (an fp32 carrier riding on the ordinary 16-bit result)."""
    folded = plain = norms = 0
    amp = 0.0

    def layernorm(self, x, gamma, beta, eps, dt):
        y = super().layernorm(x, gamma, beta, eps, dt)
        if dt in (torch.float16, torch.bfloat16):
            mean = x.mean(-1, keepdim=True)
            rstd = 1.0 / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
            y.vgen_fold32 = ((x.to(dt).float() - mean) * rstd) * gamma + beta
            # by how much the pre-normalisation rounding is amplified: rms(x) / sigma per row (1 = a zero-mean row)
            EmuFold.amp += float((x.pow(2).mean(-1).sqrt() * rstd.squeeze(-1)).mean())
            EmuFold.norms += 1
        return y

    def tapgemm(self, g):
        eff = getattr(g.A, "vgen_fold32", None)
        if eff is None:
            return super().tapgemm(g)
        if g.mode != 0 or g.C2:
            EmuFold.plain += 1
            return super().tapgemm(g)
        EmuFold.folded += 1
        a16 = g.A
        g.A = _Carrier(a16, eff)
        try:
            return super().tapgemm(g)
        finally:
            g.A = a16


def main():
    names = sys.argv[1:] or ["t2v"]
    for name in names:
        g = fc.load(name)
        for label, be in (("round16(LN(x)) [product]", EmuBackend()), ("LN folded into the consumer", EmuFold())):
            ops.set_backend(be)
            EmuFold.folded = EmuFold.plain = EmuFold.norms = 0
            EmuFold.amp = 0.0
            m = fc.build(name, g, "mixed")
            t0 = time.time()
            err, nr = fc.error(fc.forward(name, m, g), g)
            extra = ""
            if isinstance(be, EmuFold):
                extra = (f"  {EmuFold.norms} LayerNorms, consumers folded {EmuFold.folded} / not foldable {EmuFold.plain}, "
                         f"mean rms(x)/sigma {EmuFold.amp / max(EmuFold.norms, 1):.3f}")
            print(f"{name} fp16/mixed, {label}: emulated rel-L2 {err:.4e}  ({time.time() - t0:.0f} s){extra}", flush=True)
            del m


if __name__ == "__main__":
    main()
