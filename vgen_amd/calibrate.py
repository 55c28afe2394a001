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

r06 — the pass is a pure function of (weights, calibration input, k_max, rows_per_k, damp):
  * which launches are calibrated is decided by RULE, per launch, from its own sizes: K <= k_max AND the calibration batch
    gives it at least `min_rows_per_k` x K operand rows (H = A^T A of fewer rows is rank-deficient; the rounding then fits
    one sample — ADVICE r05); everything else keeps to-nearest.  No wall clock: `time_budget_s` can only ABORT the pass
    (CalibrationTimeout), never change a bit of its result.  `report["layers"]` lists every launch with its decision.
  * the calibration input is a BATCH (`calibration_batch`: 8 noise / prompt draws at 8 timesteps spread over the sampling
    schedule, one forward): 8 x the operand rows per launch, and H averages over the trajectory instead of one t.
  * an all-zero input column ("dead": a temporal tap of a one-frame batch, an absent condition channel) is decoupled from
    the others and its weights are rounded to nearest — they are NOT zeroed (one forward cannot prove a column
    dead for every later input).
  * `save_calibrated(model, path)` / `load_calibrated(model, path)` persist the packed 16-bit matrices: a deployment
    calibrates once; `UNet: {precision: calibrated, calibration: <file>}` in a config reaches `load_calibrated` through the
    constructor (vgen_amd/unet.py::pack).

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
    # an input column that was identically zero in the calibration batch: its row / column of H is zero, so a unit diagonal
    # decouples it from the others and its weights come out rounded to nearest — NOT zeroed: one calibration batch cannot
    # prove the column dead for every later input (a temporal tap at F = 1, an optional condition channel)
    dead = d <= 0
    mean_d = float(d.mean())
    H[dead, dead] = mean_d if mean_d > 0 else 1.0
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


CALIBRATION_THREADS = 32


class CalibrationTimeout(RuntimeError):
    """The pass ran past `time_budget_s`.  It aborts — a budget never changes which weights are calibrated (r05 degraded
    silently: the packed bits then depended on the host's speed)."""


class CalibratingBackend:
    """Wraps an op backend: every tap-GEMM launch whose weight is still two-term gets its weight rounded and converted to
    single-pass IN PLACE before the launch is forwarded — by error feedback when the RULE says so, else to nearest:

        K <= k_max   and   rows of the launch (g.M) >= min_rows_per_k * K

    Both are properties of the launch, so the same model + calibration batch always gives the same bits."""

    def __init__(self, inner, damp: float = 0.01, k_max: int = 9000, rows_per_k: int = 4, min_rows: int = 16384,
                 time_budget_s: float = 0.0, min_rows_per_k: float = 2.0):
        self.inner = inner
        self.damp, self.k_max, self.rows_per_k, self.min_rows = damp, k_max, rows_per_k, min_rows
        self.min_rows_per_k = min_rows_per_k
        self.t_start = time.time()
        self.time_budget_s = time_budget_s
        self.report = dict(calibrated=0, nearest=0, nearest_long_k=0, nearest_few_rows=0, seconds_h=0.0, seconds_round=0.0,
                           max_move=0.0, layers=[],
                           rule=dict(k_max=k_max, min_rows_per_k=min_rows_per_k, rows_per_k=rows_per_k, min_rows=min_rows,
                                     damp=damp))

    def __getattr__(self, name):            # every other op: the wrapped backend's
        return getattr(self.inner, name)

    def decide(self, M: int, K: int) -> str:
        if K > self.k_max:
            return "nearest_long_k"
        if M < self.min_rows_per_k * K:
            return "nearest_few_rows"
        return "calibrated"

    def tapgemm(self, g):
        dw = getattr(g.W, "vgen_dw", None)
        if dw is not None:
            K = g.taps * g.C1 + g.C2
            dt = g.W.dtype
            assert dw.shape[1] == 2 * K and g.W.shape[1] == K and g.W.is_contiguous()
            if self.time_budget_s > 0 and time.time() - self.t_start > self.time_budget_s:
                raise CalibrationTimeout(f"calibration pass over its budget of {self.time_budget_s:.0f} s after "
                                         f"{len(self.report['layers'])} weights (nothing is degraded: re-run with a larger "
                                         f"budget or a smaller k_max)")
            what = self.decide(g.M, K)
            if what == "calibrated":
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
                self.report[what] += 1
            self.report["layers"].append((int(g.mode), int(g.M), int(g.N), int(K), what))
            del g.W.vgen_dw
        return self.inner.tapgemm(g)


def calibration_units(latent_shape) -> int:
    """Default size of the calibration batch: 8 units at the t2v latent (16 x 32 x 56 = 28 672 rows per unit at full
    resolution: the coarsest level then still has 3 584 rows for its K = 1 280 linears), proportionally fewer at larger
    latents, never fewer than 2 (two timesteps)."""
    C, F, H, W = latent_shape
    return max(2, min(8, round(8 * 28672 / float(F * H * W))))


def calibration_batch(latent_shape, n: int = None, seed: int = 424242, device="cpu", context=(77, 1024), num_timesteps: int = 1000):
    """(x [n, C, F, H, W], t [n] long, y [n, L, D]): the default calibration input — n (default calibration_units) seeded
    noise / prompt draws at n timesteps spread evenly over the sampling schedule (n = 8: t = 937, 812, ..., 62).  Generated on
    the CPU generator and moved, so the batch — and with it every calibrated bit — is the same on every box.  Family-specific
    conditioning (image tokens, local images, fps, condition maps) is the caller's to add with the same leading batch size."""
    C, F, H, W = latent_shape
    n = calibration_units(latent_shape) if n is None else n
    gen = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(n, C, F, H, W, generator=gen)
    y = torch.randn(n, context[0], context[1], generator=gen)
    t = torch.tensor([int(num_timesteps * (2 * (n - i) - 1) / (2 * n)) for i in range(n)], dtype=torch.long)
    return x.to(device), t.to(device), y.to(device)


@torch.no_grad()
def calibrate_single_pass(model, x, t=None, damp: float = 0.01, k_max: int = 9000, time_budget_s: float = 0.0,
                          min_rows_per_k: float = 2.0, forward=None, **kwargs):
    """Turn a model packed with precision="high" into a single-pass model with calibrated roundings, in place.

    x, t, **kwargs: the calibration input for model.forward — a batch (calibration_batch) of noise / timesteps / conditioning
    of the shapes the model will be sampled at.  forward: a list of callables run INSTEAD of model(x, t, **kwargs) under the
    calibrating backend (the AutoencoderKL: its decode and encode passes, calibrate_vae).  The result is a pure function of
    the weights, this input and (damp, k_max, min_rows_per_k); time_budget_s > 0 only bounds the pass: over it,
    CalibrationTimeout is raised and the model must be re-packed (model.invalidate()).  Returns the report dict of the pass
    (per-launch decisions under "layers").  Afterwards `model.precision == "calibrated"`; save_calibrated() persists the
    result; repacking (loading other weights) returns the model to "high"."""
    if getattr(model, "precision", None) != "high":
        raise ValueError(f"calibrate_single_pass needs a model built with precision='high' (every packed weight two-term); "
                         f"got precision={getattr(model, 'precision', None)!r}")
    cb = CalibratingBackend(ops.backend(), damp=damp, k_max=k_max, time_budget_s=time_budget_s, min_rows_per_k=min_rows_per_k)
    prev = ops.set_backend(cb)
    # host linear algebra: LAPACK on a FIXED thread count — min(32, the host's CPUs) — whatever the process was started
    # with.  More threads make it slower on a 256-thread host; fewer make it slow AND change the bits: under
    # torch.distributed.run (OMP_NUM_THREADS=1 by default) r06's first two-rank run took 247 s instead of 84 s and packed
    # different (equally good) roundings, because a blocked Cholesky rounds differently per thread count.  With the count
    # fixed the pass is a pure function of its inputs on a given class of host (profiles/r06_*: the same packed_digest from
    # the test suite's process and from bench.py's).
    import os
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, CALIBRATION_THREADS)))
    t0 = time.time()
    try:
        if forward is None:
            model(x, t, **kwargs)
        else:
            for fn in forward:
                fn()
    finally:
        ops.set_backend(prev)
        torch.set_num_threads(threads)
    # weights no calibration launch reached (a branch the calibration input does not exercise) keep to-nearest rounding
    unreached = 0
    for w in _packed_tensors(model):
        if getattr(w, "vgen_dw", None) is not None:
            del w.vgen_dw
            unreached += 1
    cb.report.update(seconds_total=time.time() - t0, two_term_left=0, unreached_to_nearest=unreached,
                     lapack_threads=max(1, min(os.cpu_count() or 1, CALIBRATION_THREADS)),
                     calibration_rows=int(x.shape[0]) if torch.is_tensor(x) else None)
    model.precision = "calibrated"
    model.calibration = None                               # in-memory result; save_calibrated() writes it out
    model._epoch = getattr(model, "_epoch", 0) + 1         # sampling sessions key on it: captured graphs are stale
    model._calibration_report = cb.report
    return cb.report


@torch.no_grad()
def calibrate_vae(vae, z, x=None, **opts):
    """The AutoencoderKL (vgen_amd/vae.py) built with precision="high" -> calibrated single-pass, in place: one eager decode
    pass over the latents `z` [n, 4, h, w] (already divided by the scale factor, as decode() takes them) and, when `x`
    [n, 3, H, W] is given, one encode pass — without `x` the encoder's weights keep to-nearest rounding.  Frames are rows
    here: 8 frames of 256 x 448 give the 32 x 56 level's K = 4608 convs the 2 K rows the rule asks for."""
    if hasattr(vae, "clear_graphs"):
        vae.clear_graphs()
    fns = [lambda: vae._decode_rows(z.float().contiguous())]
    if x is not None:
        fns.append(lambda: vae._encode_rows(x.float().contiguous()))
    if getattr(vae, "_packed", None) is None:
        vae.pack()
    rep = calibrate_single_pass(vae, z, forward=fns, **opts)
    if hasattr(vae, "clear_graphs"):
        vae.clear_graphs()
    return rep


def _named_packed(model):
    """(path, tensor) over the model's packed operands, paths like 'input_blocks.1.0/conv1/0'."""
    def walk(o, path):
        if torch.is_tensor(o):
            yield path, o
        elif isinstance(o, dict):
            for k, v in o.items():
                yield from walk(v, f"{path}/{k}" if path else str(k))
        elif isinstance(o, (tuple, list)):
            for i, v in enumerate(o):
                yield from walk(v, f"{path}/{i}")
    yield from walk(getattr(model, "_packed", None) or {}, "")


def _packed_tensors(model):
    for _, w in _named_packed(model):
        yield w


CALIBRATION_FORMAT = 1


def brief_report(rep: dict) -> dict:
    """The report without its per-launch list (for JSON lines): counts + the rule + a digest of the decisions."""
    import hashlib
    out = {k: v for k, v in rep.items() if k != "layers"}
    out["layers_digest"] = hashlib.sha256(repr(rep.get("layers", [])).encode()).hexdigest()[:16]
    for k in ("seconds_h", "seconds_round", "seconds_total", "max_move"):
        if k in out:
            out[k] = round(float(out[k]), 2)
    return out


def packed_digest(model) -> str:
    """sha256 over the 16-bit packed matrices (path order): two calibrations of the same model must agree on it."""
    import hashlib
    h = hashlib.sha256()
    for path, w in _named_packed(model):
        if w.dtype in (torch.float16, torch.bfloat16):
            h.update(path.encode())
            h.update(w.detach().contiguous().view(torch.int16).cpu().numpy().tobytes())
    return h.hexdigest()


def save_calibrated(model, path: str) -> dict:
    """Write the calibrated 16-bit operands of `model` (precision == "calibrated") to `path` (torch.save of plain tensors
    keyed by their place in the packed structure + the report).  Returns the header that was written."""
    if getattr(model, "precision", None) != "calibrated" or getattr(model, "_packed", None) is None:
        raise ValueError("save_calibrated needs a calibrated, packed model (calibrate_single_pass first)")
    tensors = {p: w.detach().cpu() for p, w in _named_packed(model) if w.dtype in (torch.float16, torch.bfloat16)}
    if any(getattr(w, "vgen_dw", None) is not None for w in _packed_tensors(model)):
        raise ValueError("save_calibrated: a two-term weight is left (the calibration pass did not reach every launch)")
    head = dict(format=CALIBRATION_FORMAT, model=type(model).__name__, compute_dtype=str(model.compute_dtype),
                report=getattr(model, "_calibration_report", None), count=len(tensors))
    torch.save(dict(head=head, tensors=tensors), path)
    return head


@torch.no_grad()
def load_calibrated(model, path: str, check: bool = True) -> dict:
    """Replace the 16-bit packed operands of `model` by the calibrated ones in `path`.  The model must be packed single-pass
    in the calibrated structure (vgen_amd/unet.py::pack does that for precision="calibrated" and calls this).  check: every
    matrix of the file must lie within a few rounding errors of the to-nearest rounding just packed from the model's OWN
    weights — a file made for another checkpoint is refused, not applied."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    head, tensors = blob["head"], blob["tensors"]
    if head.get("format") != CALIBRATION_FORMAT:
        raise ValueError(f"{path}: calibration format {head.get('format')} != {CALIBRATION_FORMAT}")
    if head.get("model") != type(model).__name__ or head.get("compute_dtype") != str(model.compute_dtype):
        raise ValueError(f"{path}: made for {head.get('model')} / {head.get('compute_dtype')}, "
                         f"not {type(model).__name__} / {model.compute_dtype}")
    mine = {p: w for p, w in _named_packed(model) if w.dtype in (torch.float16, torch.bfloat16)}
    if set(mine) != set(tensors):
        odd = sorted(set(mine) ^ set(tensors))[:4]
        raise ValueError(f"{path}: packed structure differs ({len(mine)} vs {len(tensors)} matrices; e.g. {odd})")
    for p, w in mine.items():
        q = tensors[p]
        if q.shape != w.shape or q.dtype != w.dtype:
            raise ValueError(f"{path}: {p} is {tuple(q.shape)} {q.dtype}, the model packs {tuple(w.shape)} {w.dtype}")
        if getattr(w, "vgen_dw", None) is not None:
            raise ValueError(f"load_calibrated: {p} is two-term — pack the model with precision='calibrated'")
        q = q.to(w.device)
        if check and w.numel():
            wf, qf = w.float(), q.float()
            typical = float(wf.pow(2).mean().sqrt()) * (2.0 ** -11 if w.dtype == torch.float16 else 2.0 ** -8)
            if float((qf - wf).abs().max()) > 64.0 * max(typical, 1e-30):
                raise ValueError(f"{path}: {p} is not a rounding of this model's weights (calibrated for another checkpoint?)")
        w.copy_(q)
    model._calibration_report = head.get("report")
    model._epoch = getattr(model, "_epoch", 0) + 1
    return head
