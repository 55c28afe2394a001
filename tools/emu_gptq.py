"""TEST TOOLING (imports oracle/): can a SINGLE-PASS 16-bit weight meet the 1e-3 tolerance if its rounding is chosen with
the layer's input statistics instead of to-nearest?  (`fast` lands at 1.33e-3, 0.98e-3 of it the to-nearest rounding of the
weights; `mixed` buys that back with two-term weights for 1.13x the step time.)

GPTQ / OBQ-style error feedback on the fp16 grid: for a layer out = A . W^T with H = A^T A from a CALIBRATION input,
quantise W one K-column at a time, each column's rounding error pushed onto the not-yet-rounded columns through the Cholesky
factor of H^-1 — the rounded matrix minimises ||A W^T - A Q^T|| over roundings instead of ||W - Q||.  It is still ONE fp16
matrix per layer (elements a few typical rounding errors from the original): the kernels and their single-pass launches are
unchanged.  Product form: vgen_amd/calibrate.py.

Procedure: the model is built with every weight two-term (W = W_hi + W_lo to 2^-22); a calibration forward on the emulator —
OTHER noise, timestep and prompt than the fixture's — visits every tap-GEMM launch, builds H from (a row sample of) its
gathered A operand, rounds, and continues with the rounded weight (sequential calibration); the fixture's forward then runs
every launch single-pass on the cached roundings.  Baselines in the same run: to-nearest single-pass, two-term everywhere.
    python tools/emu_gptq.py t2v [damp [min_rows]]
"""
import dataclasses
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


def gptq_round(W: torch.Tensor, H: torch.Tensor, dt, damp: float, block: int = 128) -> torch.Tensor:
    """W fp32 [N, K], H fp32 [K, K] -> Q [N, K] of dtype dt, every entry one of the two dt-neighbours' ... (error feedback
    can carry an entry further than 1 ulp only through the accumulated correction, which is what it is for)."""
    W = W.clone()
    K = W.shape[1]
    d = H.diagonal()
    dead = d <= 0
    H = H.clone()
    H[dead, dead] = 1.0
    W[:, dead] = 0.0
    H.diagonal().add_(damp * float(d.mean()))
    L = torch.linalg.cholesky(H.double())
    Hinv = torch.cholesky_inverse(L)
    U = torch.linalg.cholesky(Hinv, upper=True).float()
    Q = torch.empty_like(W)
    for i1 in range(0, K, block):
        i2 = min(i1 + block, K)
        W1 = W[:, i1:i2].clone()
        E1 = torch.zeros_like(W1)
        U1 = U[i1:i2, i1:i2]
        for i in range(i2 - i1):
            w = W1[:, i]
            q = w.to(dt).float()
            Q[:, i1 + i] = q
            e = (w - q) / U1[i, i]
            W1[:, i:] -= e[:, None] * U1[i, i:][None, :]
            E1[:, i] = e
        W[:, i2:] -= E1 @ U[i1:i2, i2:]
    return Q.to(dt)


class EmuGptq(EmuBackend):
    def __init__(self, damp):
        self.damp = damp
        self.phase = "calib"          # "calib" | "gptq" | "nearest" | "twoterm"
        self.cache = {}
        self.min_rows = 0             # launches with fewer rows keep to-nearest rounding (the low-resolution levels)
        self.stats = dict(layers=0, h_s=0.0, q_s=0.0)

    def _a_full(self, g, rows):
        parts = []
        for r in self._src_rows(g):
            rr = r[rows]
            a = g.A[:, : g.C1][rr.clamp(min=0)].float()
            parts.append(torch.where((rr >= 0)[:, None], a, torch.zeros_like(a)))
        if g.C2:
            parts.append(g.A2[: g.M, : g.C2][rows].float())
        return torch.cat(parts, 1)

    def tapgemm(self, g):
        dw = getattr(g.W, "vgen_dw", None)
        if dw is None or self.phase == "twoterm":
            return super().tapgemm(g)
        from vgen_amd.ops import dw_terms
        key = id(dw)
        K = g.taps * g.C1 + g.C2
        if self.phase == "nearest" or g.M < self.min_rows:
            Wq = dw_terms(dw[: g.N])[0].contiguous()          # to-nearest (also: layers below the row threshold)
        else:
            if key not in self.cache:
                assert self.phase == "calib", "launch not seen by the calibration forward"
                hi, lo = dw_terms(dw[: g.N])
                t0 = time.time()
                target = max(4 * K, 16384)
                stride = max(1, g.M // target)
                rows = torch.arange(0, g.M, stride)
                a = self._a_full(g, rows)
                H = a.t() @ a
                t1 = time.time()
                self.cache[key] = gptq_round(hi.float() + lo.float(), H, g.A.dtype, self.damp)
                self.stats["layers"] += 1
                self.stats["h_s"] += t1 - t0
                self.stats["q_s"] += time.time() - t1
            Wq = self.cache[key]
        return super().tapgemm(dataclasses.replace(g, W=Wq))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "t2v"
    damp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    min_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # calibrate only launches with >= this many rows
    g = fc.load(name)
    be = EmuGptq(damp)
    be.min_rows = min_rows
    ops.set_backend(be)
    m = fc.build(name, g, "high")
    # calibration input: other noise / context / timestep than the fixture's (fixture: input_seed, t in g)
    gen = torch.Generator("cpu").manual_seed(424242)
    xc = torch.randn(*fc.SHAPE.get(name, (1, 4, 16, 32, 56)), generator=gen)
    kwc = dict(y=torch.randn(1, 77, 1024, generator=gen))
    tc = torch.tensor([637])
    assert int(g["t"]) != 637
    t0 = time.time()
    with torch.no_grad():
        m(xc, tc, **kwc)
    print(f"{name}: calibration forward (t = 637, seed 424242) {time.time() - t0:.0f} s, {be.stats['layers']} weights rounded "
          f"(launches with >= {min_rows} rows), H {be.stats['h_s']:.0f} s, Cholesky + feedback {be.stats['q_s']:.0f} s, damp {damp}",
          flush=True)
    fixtures = [(name, g)]
    if name == "t2v":                                            # the same weights at another timestep / input
        fixtures.append(("t2v_c", fc.load("t2v_c")))
    thresholds = sorted({min_rows, max(min_rows, 14336), max(min_rows, 28672)})
    for fname, fg in fixtures:
        for phase, rows_, label in [("gptq", r, f"single-pass, error-feedback rounding where rows >= {r}") for r in thresholds] + \
                                   [("nearest", 0, "single-pass, to-nearest"), ("twoterm", 0, "two-term everywhere [precision high]")]:
            be.phase, be.min_rows = phase, rows_
            t0 = time.time()
            err, _ = fc.error(fc.forward(fname, m, fg), fg)
            print(f"{fname} fp16, {label}: emulated rel-L2 {err:.4e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
