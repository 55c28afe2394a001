"""CPU-side checks of the C ABI's HOST logic (no launch happens: every call below returns from the argument checks or from the
pure-host planner): the rejection contract of vgen_tapgemm (include/vgen_hip.h: "VGEN_E_BADARG + vgen_last_error(), nothing
launched") and the legality of the plans make_plan picks for the launch signatures of the t2v UNet at its benchmark shape
(the committed per-shape profile) — tile width, split-K bound, workspace size, and the two shape exclusions (column statistics
never on the 128-row ping-pong shape, dual-W never on the two-blocks-per-CU shape)."""
import ctypes as C
import json
import os

import pytest

from vgen_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGEN_E_BADARG = -1         # include/vgen_hip.h
FAKE = 0x7f0000001000        # an aligned non-null address: the checks and the planner never dereference operands
SHAPE_PP, SHAPE_DUAL, SHAPE_PP128, SHAPE_PANEL = 0, 1, 2, 3


def good_linear():
    a = lib.TapGemmArgs()
    a.M, a.N, a.dtype = 512, 320, lib.VGEN_F16
    a.A, a.lda, a.C1, a.taps, a.mode = FAKE, 320, 320, 1, lib.TAP_LINEAR
    a.W, a.ldw = FAKE, 0
    a.out, a.ldo, a.out_dtype, a.epilogue = FAKE, 320, lib.VGEN_F32, lib.EPI_NONE
    return a


BAD = [
    ("dtype", lambda a: setattr(a, "dtype", lib.VGEN_F32), b"dtype"),
    ("N", lambda a: setattr(a, "N", 0), b"M/N"),
    ("C1", lambda a: setattr(a, "C1", 96), b"C1"),
    ("C2", lambda a: setattr(a, "C2", 32), b"C2"),
    ("dualw", lambda a: setattr(a, "dualw", 2), b"dualw"),
    ("lda", lambda a: setattr(a, "lda", 324), b"lda"),
    ("ldw_short", lambda a: setattr(a, "ldw", 256), b"ldw"),
    ("ldw_dualw", lambda a: (setattr(a, "dualw", 1), setattr(a, "ldw", 320)), b"ldw"),
    ("A_align", lambda a: setattr(a, "A", FAKE + 8), b"aligned"),
    ("A2_missing", lambda a: (setattr(a, "C2", 64), setattr(a, "lda2", 64)), b"aligned"),
    ("bias_align", lambda a: setattr(a, "bias", FAKE + 4), b"bias"),
    ("residual_align", lambda a: setattr(a, "residual", FAKE + 4), b"residual"),
    ("rowbias_rows", lambda a: (setattr(a, "rowbias", FAKE), setattr(a, "rows_per_rb", 0)), b"rowbias"),
    ("out_dtype", lambda a: setattr(a, "out_dtype", lib.VGEN_BF16), b"out_dtype"),
    ("ws_align", lambda a: setattr(a, "ws", FAKE + 4), b"workspace"),
    ("linear_taps", lambda a: setattr(a, "taps", 9), b"taps"),
    ("conv_geometry", lambda a: (setattr(a, "mode", lib.TAP_CONV3X3), setattr(a, "taps", 9)), b"conv3x3"),
    ("conv_M", lambda a: (setattr(a, "mode", lib.TAP_CONV3X3), setattr(a, "taps", 9), setattr(a, "stride", 1),
                          [setattr(a, k, 7) for k in ("Hi", "Wi", "Ho", "Wo")]), b"Ho*Wo"),
    ("temporal_geometry", lambda a: (setattr(a, "mode", lib.TAP_TEMPORAL3), setattr(a, "taps", 3), setattr(a, "F", 16),
                                     setattr(a, "S", 100)), b"temporal"),
    ("mode", lambda a: setattr(a, "mode", 7), b"mode"),
    ("K_long", lambda a: (setattr(a, "C1", 131072), setattr(a, "lda", 131072)), b"too long"),
    ("K_long_dualw", lambda a: (setattr(a, "C1", 65536), setattr(a, "lda", 65536), setattr(a, "dualw", 1)), b"too long"),
    ("geglu_N", lambda a: (setattr(a, "epilogue", lib.EPI_GEGLU), setattr(a, "N", 96)), b"GEGLU"),
    ("epilogue", lambda a: setattr(a, "epilogue", 9), b"epilogue"),
    ("colstats_16bit", lambda a: (setattr(a, "colstats", FAKE), setattr(a, "out_dtype", lib.VGEN_F16)), b"colstats"),
    ("split_out_fp32", lambda a: setattr(a, "split_out", 1), b"split_out"),
    ("split_out_ldo", lambda a: (setattr(a, "split_out", 1), setattr(a, "out_dtype", lib.VGEN_F16)), b"split_out"),
]


@pytest.mark.parametrize("name,mutate,needle", BAD, ids=[b[0] for b in BAD])
def test_tapgemm_rejects_before_launching(name, mutate, needle):
    l = lib.load()
    a = good_linear()
    mutate(a)
    assert l.vgen_tapgemm(C.byref(a), None) == VGEN_E_BADARG
    msg = l.vgen_last_error()
    assert needle in msg, (name, msg)


def test_null_args_are_rejected():
    l = lib.load()
    assert l.vgen_tapgemm(None, None) == VGEN_E_BADARG and b"null" in l.vgen_last_error()
    out3 = (C.c_int32 * 3)()
    assert l.vgen_tapgemm_query_plan(None, out3) == VGEN_E_BADARG
    assert l.vgen_tapgemm_ws_bytes(None) == 0


def _signatures():
    rows = json.load(open(os.path.join(ROOT, "profiles", "r04b_tapgemm_shapes_t2v_mixed.json")))["rows"]
    for (mode, M, N, K, epi, od), _n, _ms, _tf in rows:
        taps = {0: 1, 1: 9, 2: 3}[mode]
        if K % (64 * taps):                      # a skip-conv K segment rides along: K = taps * C1 + C2
            c2 = next(c for c in range(64, K, 64) if (K - c) % (64 * taps) == 0)
        else:
            c2 = 0
        yield mode, M, N, (K - c2) // taps, c2, taps, epi, "float32" in od, "+dw" in od


def test_plans_of_the_benchmark_launches_are_legal():
    l = lib.load()
    seen = set()
    for mode, M, N, C1, C2, taps, epi, f32, dw in _signatures():
        for colstats in (False, True):
            for residual in (False, True):
                if colstats and (not f32 or epi):
                    continue
                a = lib.TapGemmArgs()
                a.M, a.N, a.dtype, a.mode = M, N, lib.VGEN_F16, mode
                a.A, a.lda, a.C1, a.taps = FAKE, C1, C1, taps
                if C2:
                    a.A2, a.lda2, a.C2 = FAKE, C2, C2
                a.W, a.dualw = FAKE, int(dw)
                n_out = N // 2 if epi else N
                a.out, a.ldo, a.out_dtype, a.epilogue = FAKE, n_out, (lib.VGEN_F32 if f32 else lib.VGEN_F16), epi
                if residual:
                    a.residual, a.ldr = FAKE, n_out
                if colstats:
                    a.colstats = FAKE
                out3 = (C.c_int32 * 3)()
                assert l.vgen_tapgemm_query_plan(C.byref(a), out3) == 0
                shape, bn, sk = out3
                KT = taps * (C1 // 64) + C2 // 64
                sig = (mode, M, N, C1, C2, epi, f32, dw, colstats, residual)
                if shape == SHAPE_PANEL:
                    # r05, csrc/panelgemm.hip: the W-panel-resident shape takes the K = 320 linears of the full-resolution
                    # level and nothing else; 160-column panels, 80 with a two-term weight (64 under GEGLU), no split-K
                    assert mode == lib.TAP_LINEAR and C1 in (320, 640) and C2 == 0 and M >= 2048 and not colstats, sig
                    if C1 == 640:                        # the 16 x 28 level: 80-column single-pass panels, no GEGLU
                        assert bn == 80 and not dw and not epi, (sig, bn)
                    else:
                        assert bn == ((64 if epi else 80) if dw else 160), (sig, bn)
                    assert N % bn == 0 and sk == 1, (sig, bn, sk)
                    assert l.vgen_tapgemm_ws_bytes(C.byref(a)) == 0, sig
                    seen.add((shape, bn, False))
                    continue
                assert shape in (SHAPE_PP, SHAPE_DUAL, SHAPE_PP128), sig
                assert bn in (64, 128, 160) and (N % bn == 0 or bn == 64), (sig, bn)
                assert not (epi and bn == 160), sig                      # GEGLU tiles pair 16 value with 16 gate columns
                assert not (colstats and shape == SHAPE_PP128), sig
                assert not (dw and shape == SHAPE_DUAL), sig
                assert 1 <= sk <= max(1, min(KT // 4, 32)), (sig, sk)
                assert not (colstats and sk > 1), sig                    # slab statistics need the finished sums
                assert l.vgen_tapgemm_ws_bytes(C.byref(a)) == (sk * M * N * 4 if sk > 1 else 0), sig
                seen.add((shape, bn, sk > 1))
    # the sweep exercises the planner's whole range: every shape, every tile width, split-K and no split-K
    assert {s for s, _, _ in seen} == {SHAPE_PP, SHAPE_DUAL, SHAPE_PP128, SHAPE_PANEL}, seen
    assert {b for _, b, _ in seen} >= {128, 160} and {k for _, _, k in seen} == {False, True}, seen


def test_panel_shape_is_taken_exactly_where_it_is_legal():
    """r05, csrc/panelgemm.hip::vgen_panel_bn — the W-panel-resident shape has no row bias, no column statistics, no
    two-term output rows, no split-K and no N tails: every launch that needs one of those, or is too small to fill a CU's
    eight waves, must stay on the streaming shapes; every other K = 320 / 640 linear must take it."""
    l = lib.load()

    def plan(M=57344, N=320, C1=320, dw=0, epi=lib.EPI_NONE, f32=True, taps=1, mode=lib.TAP_LINEAR, **extra):
        a = lib.TapGemmArgs()
        a.M, a.N, a.dtype, a.mode = M, N, lib.VGEN_F16, mode
        a.A, a.lda, a.C1, a.taps = FAKE, C1, C1, taps
        a.W, a.dualw = FAKE, dw
        n_out = N // 2 if epi else N
        a.out, a.ldo, a.out_dtype, a.epilogue = FAKE, n_out, (lib.VGEN_F32 if f32 else lib.VGEN_F16), epi
        for k, v in extra.items():
            setattr(a, k, v)
        out3 = (C.c_int32 * 3)()
        assert l.vgen_tapgemm_query_plan(C.byref(a), out3) == 0
        return tuple(out3), l.vgen_tapgemm_ws_bytes(C.byref(a))

    # taken: the level-0 launches of the step, single-pass and dual-W, every epilogue; the K = 640 single-pass ones
    assert plan()[0] == (SHAPE_PANEL, 160, 1)
    assert plan(dw=1)[0] == (SHAPE_PANEL, 80, 1)
    assert plan(N=960, f32=False)[0] == (SHAPE_PANEL, 160, 1)
    assert plan(N=2560, f32=False, epi=lib.EPI_GEGLU)[0] == (SHAPE_PANEL, 160, 1)
    assert plan(N=2560, f32=False, epi=lib.EPI_GEGLU, dw=1)[0] == (SHAPE_PANEL, 64, 1)
    assert plan(residual=FAKE, ldr=320) == ((SHAPE_PANEL, 160, 1), 0)
    assert plan(M=14336, N=1920, C1=640, f32=False)[0] == (SHAPE_PANEL, 80, 1)
    # not taken
    for kw in (dict(rowbias=FAKE, rowbias_ld=320, rows_per_rb=1792), dict(colstats=FAKE), dict(M=2047), dict(N=336),
               dict(C1=384), dict(C1=640, dw=1, M=14336, N=640), dict(C1=640, M=14336, N=5120, f32=False, epi=lib.EPI_GEGLU),
               dict(f32=False, split_out=1, ldo=640), dict(N=2560, epi=lib.EPI_GEGLU, f32=True), dict(ldo=322),
               dict(f32=False, ldo=324), dict(residual=FAKE, ldr=322)):
        assert plan(**kw)[0][0] != SHAPE_PANEL, kw
    assert plan(C1=64, C2=0, mode=lib.TAP_TEMPORAL3, taps=3, F=16, S=3584)[0][0] != SHAPE_PANEL


# ---- the other hot entry points: same contract (argument checks come before any launch) ------------------------------------
def good_attn():
    a = lib.AttnArgs()
    a.q = a.k = a.v = a.out = FAKE
    a.dtype, a.heads, a.nq, a.nk, a.nbatch, a.inner = lib.VGEN_F16, 8, 1792, 1792, 32, 1
    for f in ("q_rs", "k_rs", "v_rs", "o_rs"):
        setattr(a, f, 512)
    a.scale = 0.125
    return a


BAD_ATTN = [
    ("dtype", lambda a: setattr(a, "dtype", lib.VGEN_F32), b"dtype"),
    ("sizes", lambda a: setattr(a, "nk", 0), b"sizes"),
    ("align", lambda a: setattr(a, "v", FAKE + 2), b"alignment"),
    ("stride", lambda a: setattr(a, "k_rs", 500), b"strides"),
    ("causal", lambda a: setattr(a, "causal", 3), b"causal"),
]


@pytest.mark.parametrize("name,mutate,needle", BAD_ATTN, ids=[b[0] for b in BAD_ATTN])
def test_attention_rejects_before_launching(name, mutate, needle):
    l = lib.load()
    a = good_attn()
    mutate(a)
    assert l.vgen_attention(C.byref(a), None) == VGEN_E_BADARG
    assert needle in l.vgen_last_error(), (name, l.vgen_last_error())
    assert l.vgen_attention(None, None) == VGEN_E_BADARG


def test_layernorm_rejects_before_launching():
    l = lib.load()
    ok = dict(x=FAKE, M=64, d=320, eps=1e-5, g=FAKE, b=FAKE, y=FAKE, dt=lib.VGEN_F16)
    call = lambda **kw: (lambda p: l.vgen_layernorm(p["x"], p["M"], p["d"], p["eps"], p["g"], p["b"], p["y"], p["dt"], None))({**ok, **kw})
    assert call(dt=5) == VGEN_E_BADARG and b"dtype" in l.vgen_last_error()
    assert call(d=322) == VGEN_E_BADARG and b"d=322" in l.vgen_last_error()
    assert call(d=1 << 20) == VGEN_E_BADARG
    assert call(x=FAKE + 4) == VGEN_E_BADARG and b"alignment" in l.vgen_last_error()
    assert call(M=0) == 0                                   # an empty batch is a no-op, not an error (nothing launched)
    assert call(M=1 << 33) == VGEN_E_BADARG and b"M too large" in l.vgen_last_error()


def test_groupnorm_rejects_before_launching():
    l = lib.load()
    ok = dict(x1=FAKE, C1=320, x2=None, C2=0, nb=2, S=1792, groups=32, eps=1e-5, g=FAKE, b=FAKE, silu=1, y=FAKE, raw=None,
              raw_split=0, dt=lib.VGEN_F16, ws=FAKE, ws_bytes=1 << 30)

    def call(**kw):
        p = {**ok, **kw}
        return l.vgen_groupnorm(p["x1"], p["C1"], p["x2"], p["C2"], p["nb"], p["S"], p["groups"], p["eps"], p["g"], p["b"],
                                p["silu"], p["y"], p["raw"], p["raw_split"], p["dt"], p["ws"], p["ws_bytes"], None)

    def call_cs(cs1=FAKE, cs2=None, **kw):
        p = {**ok, **kw}
        return l.vgen_groupnorm_cs(p["x1"], p["C1"], cs1, p["x2"], p["C2"], cs2, p["nb"], p["S"], p["groups"], p["eps"],
                                   p["g"], p["b"], p["silu"], p["y"], p["raw"], p["raw_split"], p["dt"], p["ws"],
                                   p["ws_bytes"], None)
    assert call(dt=lib.VGEN_F32) == VGEN_E_BADARG and b"dtype" in l.vgen_last_error()
    assert call(groups=7) == VGEN_E_BADARG and b"groups=7" in l.vgen_last_error()
    assert call(C1=322) == VGEN_E_BADARG
    assert call(C1=4096, groups=32) == VGEN_E_BADARG and b"3072" in l.vgen_last_error()
    assert call(C2=320) == VGEN_E_BADARG and b"x2 null" in l.vgen_last_error()
    assert call(y=FAKE + 8) == VGEN_E_BADARG and b"alignment" in l.vgen_last_error()
    assert call(nb=70000) == VGEN_E_BADARG and b"nb=70000" in l.vgen_last_error()
    assert call(ws_bytes=0) == -3 and b"workspace" in l.vgen_last_error()          # VGEN_E_WORKSPACE
    assert l.vgen_groupnorm_ws_bytes(2, 1792) > 0
    assert call_cs(cs1=None) == VGEN_E_BADARG and b"column statistics" in l.vgen_last_error()
    assert call_cs(S=1800) == VGEN_E_BADARG and b"64-row slab" in l.vgen_last_error()
    assert call_cs(C2=320, x2=FAKE, cs2=None, groups=32) == VGEN_E_BADARG


def test_sampler_update_entry_points_reject_before_launching():
    """The sampler algebra (a5 / a6 / a7 of SURVEY §8) and the decode epilogue (f3): null / inconsistent operands are refused,
    empty batches are no-ops — none of it needs a device."""
    l = lib.load()
    P = FAKE

    def ddim(xt=P, y=P, u=P, noise=None, coef=P, guide=9.0, use_guide=1, mean_type=0, B=1, per_b=1024, xt_1=P, x0=None):
        return l.vgen_cfg_ddim_step(xt, y, u, noise, coef, guide, use_guide, mean_type, B, per_b, xt_1, x0, None)
    assert ddim(mean_type=3) == VGEN_E_BADARG and b"mean_type" in l.vgen_last_error()
    assert ddim(u=None) == VGEN_E_BADARG and b"needs u" in l.vgen_last_error()
    assert ddim(coef=None) == VGEN_E_BADARG and b"null" in l.vgen_last_error()
    assert ddim(xt_1=None) == VGEN_E_BADARG and b"output" in l.vgen_last_error()
    assert ddim(B=0) == 0                                                         # nothing to do, nothing launched

    def units(nrep=1, rep=P, xt_1=None, B=1):
        return l.vgen_cfg_ddim_step_units(P, 1024, P, P, None, P, None, 9.0, 1, 0, B, 1024, xt_1, None, rep, nrep, 1024, 1024, None)
    assert units(nrep=-1) == VGEN_E_BADARG and b"replicas" in l.vgen_last_error()
    assert units(rep=None) == VGEN_E_BADARG and b"replicas" in l.vgen_last_error()
    assert units(nrep=0, rep=None) == VGEN_E_BADARG and b"no output" in l.vgen_last_error()
    assert units(B=0) == 0

    def dpm(x=P, den=P, out=P, n=1024):
        return l.vgen_dpmpp2m_sde_step(x, den, None, None, 1.0, 0.5, 0.0, 0.0, 0.0, 0.0, out, n, None)
    assert dpm(den=None) == VGEN_E_BADARG and b"null" in l.vgen_last_error()
    assert dpm(n=0) == 0
    assert dpm(n=1 << 40) == VGEN_E_BADARG and b"too large" in l.vgen_last_error()

    assert l.vgen_lincomb4(None, None, None, None, 1.0, 0.0, 0.0, 0.0, P, 16, None) == VGEN_E_BADARG
    assert l.vgen_lincomb4(P, None, None, None, 1.0, 0.0, 0.0, 0.0, P, 0, None) == 0

    frames = lambda x=P, rows=8, Cc=3, ldx=3: l.vgen_frames_u8(x, rows, Cc, ldx, P, P, P, None)
    assert frames(x=None) == VGEN_E_BADARG and b"frames_u8" in l.vgen_last_error()
    assert frames(ldx=2) == VGEN_E_BADARG                                          # row stride shorter than the channel count
    assert frames(rows=0) == 0


def test_splitk_parity_cases_really_split():
    """The GPU cases that are meant to exercise split-K (tests/kernel_cases.py::splitk_specs — the planner, not the test,
    decides) must be planned with split-K > 1, on a streaming shape, with the [split][M][N] fp32 workspace the header
    promises; the plan must not depend on whether the operands live on the host or the device (the planner never looks at
    them)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import kernel_cases as kc
    from vgen_amd import ops
    be = ops.HipBackend()
    for dt in (torch.float16, torch.bfloat16):
        for name, spec in kc.splitk_specs(dt).items():
            shape, bn, sk = be.tapgemm_plan(spec)
            assert shape in (SHAPE_PP, SHAPE_DUAL, SHAPE_PP128) and sk > 1, (name, shape, bn, sk)
            a = be._tapgemm_args(spec, alloc=False)
            assert a[0].ws_bytes == sk * spec.M * spec.N * 4 and a[1] is None and a[-1] is None, (name, sk, a[0].ws_bytes)
            tiles = -(-spec.M // (128 if shape == SHAPE_PP128 else 256)) * -(-spec.N // bn)
            assert tiles * sk <= 512, (name, tiles, sk)  # split-K exists to FILL the chip, not to oversubscribe it


def test_r06_shapes_are_reached_through_the_plan_table_and_only_where_legal():
    """pp256 (4) / q128 (5) are never proposed by the cost model; a plan-table row puts a launch on them — and is ignored
    where the shape is not legal (pp256: N % 256 == 0 and a 16-bit output, no column statistics); a table row also takes a
    K = 320 linear away from the panel shape (csrc/tapgemm.hip full_plan)."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import kernel_cases as kc
    from vgen_amd import ops
    be = ops.HipBackend()

    def planned(spec, plan):
        arr = (C.c_int64 * 12)(*(list(kc.plan_signature(spec)) + list(plan)))
        try:
            assert be.lib.vgen_tapgemm_set_plans(arr, 1) == 0
            return tuple(be.tapgemm_plan(spec))
        finally:
            be.lib.vgen_tapgemm_set_plans(None, -1)

    for dt in (torch.float16, torch.bfloat16):
        for name, (spec, plan) in kc.r06_shape_cases(dt).items():
            assert tuple(be.tapgemm_plan(spec))[0] not in (4, 5), name          # not without the table
            assert planned(spec, plan) == plan, name
            if plan[2] > 1:
                arr = (C.c_int64 * 12)(*(list(kc.plan_signature(spec)) + list(plan)))
                be.lib.vgen_tapgemm_set_plans(arr, 1)
                try:
                    assert be._tapgemm_args(spec, alloc=False)[0].ws_bytes == plan[2] * spec.M * spec.N * 4, name
                finally:
                    be.lib.vgen_tapgemm_set_plans(None, -1)
        f32 = kc.make_tapgemm(dt, 1000, 512, 640, residual=True)                 # fp32 output: pp256 has no epilogue for it ...
        assert planned(f32, (4, 256, 1))[0] != 4
        assert planned(f32, (4, 256, 2)) == (4, 256, 2)                          # ... but the split-K reducer has
        n320 = kc.make_tapgemm(dt, 1000, 320, 640, out_dtype=dt)                 # N % 256 != 0
        assert planned(n320, (4, 256, 1))[0] != 4
        pan = kc.make_tapgemm(dt, 9000, 2560, 320, epilogue=kc.L.EPI_GEGLU, out_dtype=dt)
        assert tuple(be.tapgemm_plan(pan))[0] == 3 and planned(pan, (4, 256, 1)) == (4, 256, 1)
