"""Calibrated single-pass weights: choose the 16-bit rounding of a packed weight with its layer's input statistics.

Why.  The reference multiplies fp16 weights rounded to nearest (its autocast arithmetic): 0.98e-3 of the 1.33e-3 that the
single-pass path here lands from the reference's fp32 forward is that rounding, and `precision="mixed"` buys it
back with two-term weights (a second MFMA pass over `W_lo`) for 1.13x the step time.  To-nearest minimises `||W - Q||`; what
the output sees is `||A W^T - A Q^T||`, and the tap-GEMM operands of this UNet are far from white (GroupNorm + SiLU / LayerNorm
outputs, neighbouring conv taps).  GPTQ / OBQ-style error feedback (Frantar et al. 2022, "GPTQ"; restated here from the
paper's algorithm 1) rounds one K-column at a time and pushes each column's rounding error onto the columns not yet rounded
through the Cholesky factor of `H^-1`, `H = A^T A` from ONE calibration forward.  Every element stays within a few to-nearest
rounding errors of a typical element of its matrix (the kernels, their single-pass launches and the storage format are
unchanged: it is still one fp16 matrix per layer); the ABI emulator puts the
full-size t2v UNet at 8.3e-4 with EVERY weight single-pass (to-nearest: 1.21e-3; two-term everywhere: 6.8e-4) when calibrated
on a different noise / prompt / timestep than the one it is evaluated on (tools/emu_gptq.py, profiles/r05_emu_gptq.txt).

How.  Build the model with `precision="high"` (every packed weight carries `W_hi + W_lo`, i.e. the fp32 weight to 2^-22),
then
        calibrate_single_pass(model, x_cal, t_cal, **kwargs_cal)
runs ONE forward through a backend wrapper that, at every tap-GEMM launch that still has a two-term weight, forms H from (a
row sample of) the launch's gathered A operand ON THE DEVICE the operands live on, rounds on the HOST (LAPACK Cholesky in
fp64 + the column loop: device-independent, so the CPU tests cover it), writes the rounded matrix over `W_hi` IN PLACE and
drops the two-term operand — the launch itself and everything after it already run single-pass on the new weight
(sequential calibration: later layers see the activations the calibrated earlier layers produce).  Launches with K above
`k_max` keep to-nearest rounding (`W_hi` as it is): the 29 long-K convs of the low-resolution levels are 90 % of the
factorisation cost and a few per cent of the output's weight-rounding sensitivity.

This is pack-time work (one forward + ~1 minute of host linear algebra for the 1.4 G-parameter UNet), not part of the hot
path; nothing here runs inside a denoise step.
"""
from __future__ import annotations

import time

import torch

from . import lib as _lib
from . import ops


_HOST = [None, False]                      # [ctypes handle, looked for]


def host_lib():
    """libvgen_host.so (csrc/host_round.cpp, built by vgen_amd.build.build_host with g++): the column loop below as one call
    per column block, rows over threads.  None when it has not been built — the torch loop is the same arithmetic, ~50x
    slower on the full-size UNet (it is the tested restatement of the C++ loop, and both are HOST pack-time code)."""
    if not _HOST[1]:
        _HOST[1] = True
        import ctypes
        import os
        path = os.environ.get("VGEN_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvgen_host.so")
        if os.path.exists(path):
            h = ctypes.CDLL(path)
            i64, vp = ctypes.c_int64, ctypes.c_void_p
            h.vgen_host_gptq_block.argtypes = [vp, i64, i64, vp, i64, i64, i64, ctypes.c_int, vp, i64, vp, i64, ctypes.c_int]
            h.vgen_host_gptq_block.restype = ctypes.c_int
            h.vgen_host_abi_version.restype = ctypes.c_int
            if h.vgen_host_abi_version() != 1:
                raise RuntimeError(f"{path}: host ABI version {h.vgen_host_abi_version()} != 1 (rebuild: python -m vgen_amd.build)")
            _HOST[0] = h
    return _HOST[0]


def _round_block_torch(W, U, i1, i2, dt, Q, E1):
    """Columns [i1, i2): round column by column, feeding each column's error forward inside the block (fp32, one rounding per
    operation).  csrc/host_round.cpp is this loop in C++, bit for bit."""
    W1 = W[:, i1:i2].clone()
    U1 = U[i1:i2, i1:i2]
    # K x ~6 tiny tensor ops: on many threads each costs ~100 us of fork / join, on one ~15 us
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for i in range(i2 - i1):
            w = W1[:, i]
            q = w.to(dt).float()
            Q[:, i1 + i] = q
            e = (w - q) / U1[i, i]
            W1[:, i:] -= e[:, None] * U1[i, i:][None, :]
            E1[:, i] = e
    finally:
        torch.set_num_threads(threads)


def inverse_factor(H: torch.Tensor):
    """fp64 [K, K] SPD -> the upper-triangular U with H^-1 = U^T U (GPTQ's recipe: cholesky -> cholesky_inverse ->
    cholesky(upper), torch's threaded LAPACK), or None if H is numerically singular.  (One Cholesky of the index-reversed H
    plus a triangular inverse is the same matrix for half the flops, but scipy's dtrtri — the only trtri within reach — ran
    2.6x SLOWER than these three calls at K = 3840 on the build box: 0.91 s vs 0.35 s.)"""
    try:
        U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    except RuntimeError:                            # torch.linalg.LinAlgError is a RuntimeError
        return None
    return U if bool(torch.isfinite(U).all()) else None


def gptq_round(W: torch.Tensor, H: torch.Tensor, dt, damp: float = 0.01, block: int = 128, use_host_lib: bool = True) -> torch.Tensor:
    """W fp32 [N, K], H fp32 [K, K] (= A^T A of the layer's input rows) -> Q [N, K] of dtype `dt` minimising
    ||A W^T - A Q^T|| greedily over roundings to `dt`, K-column by K-column with error feedback.  Host tensors."""
    assert W.dim() == 2 and H.shape == (W.shape[1], W.shape[1]) and block <= 128
    assert dt in (torch.float16, torch.bfloat16)
    W = W.detach().to(device="cpu", dtype=torch.float32).contiguous().clone()
    H = H.detach().to(device="cpu", dtype=torch.float64).clone()
    N, K = W.shape
    d = H.diagonal()
    dead = d <= 0                                   # an input column that is identically zero: its weight cannot matter
    H[dead, dead] = 1.0
    W[:, dead] = 0.0
    mean_d = float(d.mean())
    U = None
    for attempt in range(3):                        # a numerically singular H: more damping; at worst, to-nearest
        H.diagonal().add_(damp * (10.0 ** attempt) * mean_d)
        U = inverse_factor(H)
        if U is not None:
            break
    if U is None:
        return W.to(dt)
    U = U.float().contiguous()                      # row i of U = how column i's error spreads over the later columns
    Q = torch.empty_like(W)
    hl = host_lib() if use_host_lib else None
    nthreads = max(1, min(torch.get_num_threads(), 64))
    for i1 in range(0, K, block):
        i2 = min(i1 + block, K)
        E1 = torch.zeros(N, i2 - i1)
        if hl is not None:
            rc = hl.vgen_host_gptq_block(W.data_ptr(), N, K, U.data_ptr(), K, i1, i2, 0 if dt == torch.float16 else 1,
                                         Q.data_ptr(), K, E1.data_ptr(), i2 - i1, nthreads)
            if rc != 0:
                raise RuntimeError(f"vgen_host_gptq_block rejected its arguments (rc {rc})")
        else:
            _round_block_torch(W, U, i1, i2, dt, Q, E1)
        if i2 < K:
            W[:, i2:] -= E1 @ U[i1:i2, i2:]
    return Q.to(dt)


def source_rows(g, rows: torch.Tensor):
    """For output rows `rows` of launch `g`: one index tensor per tap into g.A's rows, -1 where the tap reads padding
    (include/vgen_hip.h: the three tap modes of vgen_tapgemm)."""
    m = rows
    if g.mode == _lib.TAP_LINEAR:
        return [m]
    if g.mode == _lib.TAP_TEMPORAL3:
        f = (m // g.S) % g.F
        return [torch.where((f + dt_ >= 0) & (f + dt_ < g.F), m + dt_ * g.S, torch.full_like(m, -1)) for dt_ in (-1, 0, 1)]
    if g.mode == _lib.TAP_CONV3X3:
        hw = g.Ho * g.Wo
        img, rem = m // hw, m % hw
        oy, ox = rem // g.Wo, rem % g.Wo
        hv, wv = (g.Hi << g.ups) - 2 * g.crop_t, g.Wi << g.ups
        out = []
        for ky in range(3):
            for kx in range(3):
                iy, ix = oy * g.stride + ky - g.pad_t, ox * g.stride + kx - g.pad_l
                ok = (iy >= 0) & (iy < hv) & (ix >= 0) & (ix < wv)
                src = img * (g.Hi * g.Wi) + ((iy + g.crop_t) >> g.ups) * g.Wi + (ix >> g.ups)
                out.append(torch.where(ok, src, torch.full_like(src, -1)))
        return out
    raise ValueError(f"tap mode {g.mode}")


def gathered_operand(g, rows: torch.Tensor) -> torch.Tensor:
    """fp32 [len(rows), K]: the rows of the implicit-GEMM A operand of launch `g` (taps side by side, then the A2 segment),
    on the device of g.A."""
    parts = []
    for src in source_rows(g, rows):
        a = g.A[:, : g.C1][src.clamp(min=0)].float()
        parts.append(a * (src >= 0).unsqueeze(1).to(a.dtype))
    if g.C2:
        parts.append(g.A2[: g.M, : g.C2][rows].float())
    return torch.cat(parts, 1)


class CalibratingBackend:
    """Wraps an op backend: every tap-GEMM launch whose weight is still two-term gets its weight rounded (error feedback
    for K <= k_max, to-nearest above) and converted to single-pass IN PLACE before the launch is forwarded."""

    def __init__(self, inner, damp: float = 0.01, k_max: int = 9000, rows_per_k: int = 4, min_rows: int = 16384,
                 time_budget_s: float = 0.0, k_max_late: int = 3000):
        self.inner = inner
        self.damp, self.k_max, self.rows_per_k, self.min_rows = damp, k_max, rows_per_k, min_rows
        # time_budget_s > 0 bounds the pass.  The forward visits the full-resolution level — where the sensitivity sits, all of
        # it K <= 2880 — first and LAST, and the long-K layers of the low-resolution levels (K = 3840 / 5120 / 5760: two thirds
        # of the factorisation time, a tenth of the gain: k_max = 3000 alone reads 9.13e-4 on the emulator, 9000 8.36e-4) in
        # between.  So a pass that has used HALF its budget stops paying for K > k_max_late (those keep to-nearest), and only
        # a pass that has used all of it leaves everything that follows to-nearest.
        now = time.time()
        self.deadline = now + time_budget_s if time_budget_s > 0 else None
        self.half = now + 0.5 * time_budget_s if time_budget_s > 0 else None
        self.k_max_late = k_max_late
        self.report = dict(calibrated=0, nearest=0, over_budget=0, past_half_budget_long_k=0, seconds_h=0.0, seconds_round=0.0,
                           max_move=0.0)

    def __getattr__(self, name):            # every other op: the wrapped backend's
        return getattr(self.inner, name)

    def tapgemm(self, g):
        dw = getattr(g.W, "vgen_dw", None)
        if dw is not None:
            K = g.taps * g.C1 + g.C2
            dt = g.W.dtype
            assert dw.shape[1] == 2 * K and g.W.shape[1] == K and g.W.is_contiguous()
            now = time.time()
            late = self.deadline is not None and now > self.deadline
            k_lim = self.k_max
            if late:
                self.report["over_budget"] += 1
            elif self.half is not None and now > self.half and self.k_max_late < K <= self.k_max:
                self.report["past_half_budget_long_k"] += 1
                k_lim = self.k_max_late
            if K <= k_lim and g.M > 0 and not late:
                t0 = time.time()
                hi, lo = ops.dw_terms(dw)
                w32 = hi.float() + lo.float()                                  # the packed fp32 weight to 2^-22
                want = max(self.rows_per_k * K, self.min_rows)
                stride = max(1, g.M // want)
                rows = torch.arange(0, g.M, stride, device=g.A.device)
                a = gathered_operand(g, rows)
                H = a.t() @ a
                del a
                t1 = time.time()
                q = gptq_round(w32, H, dt, self.damp).to(g.W.device)
                del H
                # largest move of an element, in to-nearest rounding errors of a typical element of the matrix
                typical = float(w32.pow(2).mean().sqrt()) * (2.0 ** -11 if dt == torch.float16 else 2.0 ** -8)
                self.report["max_move"] = max(self.report["max_move"], float((q.float() - w32).abs().max()) / max(typical, 1e-30))
                g.W.copy_(q)
                self.report["calibrated"] += 1
                self.report["seconds_h"] += t1 - t0
                self.report["seconds_round"] += time.time() - t1
            else:
                self.report["nearest"] += 1                                    # W_hi IS the to-nearest rounding
            del g.W.vgen_dw
        return self.inner.tapgemm(g)


@torch.no_grad()
def calibrate_single_pass(model, x, t, damp: float = 0.01, k_max: int = 9000, time_budget_s: float = 0.0, **kwargs):
    """Turn a model packed with precision="high" into a single-pass model with calibrated roundings, in place.

    x, t, **kwargs: one calibration input for model.forward (any noise / timestep / conditioning of the shapes the model
    will be sampled at); time_budget_s > 0 bounds the pass (weights reached later keep to-nearest rounding).  Returns the
    report dict of the pass.  Afterwards `model.precision == "calibrated"`; repacking
    (loading other weights) returns the model to "high"."""
    if getattr(model, "precision", None) != "high":
        raise ValueError(f"calibrate_single_pass needs a model built with precision='high' (every packed weight two-term); "
                         f"got precision={getattr(model, 'precision', None)!r}")
    cb = CalibratingBackend(ops.backend(), damp=damp, k_max=k_max, time_budget_s=time_budget_s)
    prev = ops.set_backend(cb)
    # host linear algebra: LAPACK on a few tens of threads (on a 256-thread host the default thread count makes it slower)
    import os
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1, 32)))
    t0 = time.time()
    try:
        model(x, t, **kwargs)
    finally:
        ops.set_backend(prev)
        torch.set_num_threads(threads)
    left = sum(1 for w in _packed_tensors(model) if getattr(w, "vgen_dw", None) is not None)
    cb.report.update(seconds_total=time.time() - t0, two_term_left=left)
    model.precision = "calibrated"
    model._epoch = getattr(model, "_epoch", 0) + 1         # sampling sessions key on it: captured graphs are stale
    return cb.report


def _packed_tensors(model):
    def walk(o):
        if torch.is_tensor(o):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from walk(v)
        elif isinstance(o, (tuple, list)):
            for v in o:
                yield from walk(v)
    yield from walk(getattr(model, "_packed", None) or {})
