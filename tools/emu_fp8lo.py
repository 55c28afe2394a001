"""TEST TOOLING (imports oracle/): what would running the W_lo pass of a dual-W launch on the block-scaled FP8 MFMA cost in
parity?  (`mixed` costs 1.13x `fast`; a dual-W launch is the A . W_hi^T product plus a second full-rate 16-bit
pass A . W_lo^T with W_lo = round16(W - W_hi) ~ 2^-11 |W|.)

gfx950's `v_mfma_scale_f32_16x16x128_f8f6f4` multiplies MX-format operands — 32-element blocks along K sharing one power-of-two
scale (E8M0), elements e4m3 — at twice the 16-bit rate, into the SAME fp32 accumulators.  The correction term tolerates 3-bit
significands: its rounding error is 2^-4 of a term that is itself 2^-11 of the product.  Emulated here exactly in that form:
    out = A . W_hi^T  +  mx8(A) . mx8(W_lo)^T          mx8(v): per 32-element K-block, s = 2^(floor(log2 max|v|) - 8),
                                                        elements round-to-nearest-even to e4m3 of v / s (OCP MX v1.0)
for every dual-W launch of the model (the operands of every other launch, the accumulation and the epilogues unchanged), and
compared with the product's own emulation (W_lo exact in 16 bits) on the full-size fixtures:
    python tools/emu_fp8lo.py t2v [t2v_b i2vgen ...]            (precision "mixed", fp16)
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
from vgen_amd import lib as L  # noqa: E402
from vgen_amd import ops  # noqa: E402


def mx8(v: torch.Tensor) -> torch.Tensor:
    """fp32 [R, K] (K % 32 == 0) -> the fp32 value of its MXFP8 (e4m3, 32-element blocks along K) encoding."""
    R, K = v.shape
    b = v.reshape(R, K // 32, 32)
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -120)))
    s = torch.exp2(e - 8.0)                                  # e4m3: largest binade 2^8 (max normal 448)
    q = (b / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s
    return torch.where(amax > 0, q, torch.zeros_like(q)).reshape(R, K)


class EmuFp8Lo(EmuBackend):
    launches = 0
    dw_launches = 0

    def tapgemm(self, g):
        EmuFp8Lo.launches += 1
        dw = getattr(g.W, "vgen_dw", None)
        if dw is None:
            return super().tapgemm(g)
        EmuFp8Lo.dw_launches += 1
        # the parent's arithmetic with the W_lo products taken on MXFP8 operands: run it once on W_hi alone (all
        # epilogue terms included), once on [mx8(A) | mx8(W_lo)] with no epilogue terms, and add in fp32
        from vgen_amd.ops import dw_terms
        dt = g.A.dtype
        K = g.taps * g.C1 + g.C2
        hi, lo = dw_terms(dw[: g.N])
        lo8 = mx8(lo.float())
        acc = torch.zeros((g.M, g.N), dtype=torch.float32)
        Whi = hi.float()
        for tap, r in enumerate(self._src_rows(g)):
            a = g.A[:, : g.C1][r.clamp(min=0)].float()
            a = torch.where((r >= 0)[:, None], a, torch.zeros_like(a))
            acc += a @ Whi[:, tap * g.C1:(tap + 1) * g.C1].t()
            acc += mx8(a) @ lo8[:, tap * g.C1:(tap + 1) * g.C1].t()
        if g.C2:
            a2 = g.A2[: g.M, : g.C2].float()
            acc += a2 @ Whi[:, g.taps * g.C1:].t()
            acc += mx8(a2) @ lo8[:, g.taps * g.C1:].t()
        # epilogue: the same order as EmuBackend.tapgemm
        if g.bias is not None:
            acc += g.bias[: g.N]
        if g.rowbias is not None:
            acc += g.rowbias[torch.arange(g.M) // g.rows_per_rb][:, : g.N]
        if g.epilogue == L.EPI_GEGLU:
            v = acc.view(g.M, g.N // 32, 2, 16)
            val, gate = v[:, :, 0], v[:, :, 1]
            acc = (val * (0.5 * gate * (1.0 + torch.erf(gate * 0.7071067811865476)))).reshape(g.M, g.N // 2)
            n_out = g.N // 2
        else:
            n_out = g.N
        if g.residual is not None:
            acc += g.residual[:, :n_out]
        split = bool(getattr(g, "split_out", False))
        out = g.out
        if out is None:
            out = torch.empty((g.M, 2 * n_out if split else n_out), dtype=g.out_dtype)
        out[:, :n_out] = acc.to(g.out_dtype)
        if split:
            out[:, n_out: 2 * n_out] = (acc - out[:, :n_out].float()).to(dt)
        if g.colstats:
            ns = (g.M + 63) // 64
            pad = torch.zeros((ns * 64, g.N), dtype=torch.float32)
            pad[: g.M] = acc
            pv = pad.view(ns, 64, g.N)
            out.vgen_cs = torch.stack([pv.sum(1), (pv * pv).sum(1)], 1).contiguous()
        return out


def main():
    for name in sys.argv[1:] or ["t2v"]:
        g = fc.load(name)
        for label, be in (("W_lo pass in 16 bits [product]", EmuBackend()), ("W_lo pass on MXFP8 operands", EmuFp8Lo())):
            ops.set_backend(be)
            EmuFp8Lo.launches = EmuFp8Lo.dw_launches = 0
            m = fc.build(name, g, "mixed")
            t0 = time.time()
            err, _ = fc.error(fc.forward(name, m, g), g)
            extra = f"  ({EmuFp8Lo.dw_launches} of {EmuFp8Lo.launches} tap-GEMM launches dual-W)" if isinstance(be, EmuFp8Lo) else ""
            print(f"{name} fp16/mixed, {label}: emulated rel-L2 {err:.4e}  ({time.time() - t0:.0f} s){extra}", flush=True)
            del m


if __name__ == "__main__":
    main()
