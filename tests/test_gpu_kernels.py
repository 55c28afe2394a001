"""-m gpu: every C-ABI kernel on the MI355X vs the CPU emulator of the ABI (identical seeded
inputs).  Tolerances (kernel_cases.py): 16-bit outputs TOL16_EMU[dtype] rel-L2 for the single-rounding kernels (SURVEY's
2e-3 in bf16), TOL16[dtype] where P is rounded inside the kernel or the reference is unrounded fp32, fp32 outputs 2e-5;
the fused CFG+DDIM update must be BIT-EXACT."""
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(res, dtname, out_is_16=None, tol16=None):
    for name, st in res.items():
        assert st["finite"], (name, st)
        tol = (tol16 or kc.TOL16)[dtname] if (out_is_16 is None or out_is_16) else kc.TOL32
        assert st["rel_l2"] <= tol, (name, st, tol)


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("case", kc.GN_CASES, ids=lambda c: "nb%d_S%d_C%d+%d" % c[:4])
def test_groupnorm(hip_backend, dtname, case):
    _check(kc.case_groupnorm(hip_backend, DEV, kc.DTS[dtname], *case), dtname, tol16=kc.TOL16_EMU)


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("case", kc.GN_CS_CASES, ids=lambda c: "nb%d_S%d_C%d+%d" % c[:4])
def test_groupnorm_from_colstats(hip_backend, dtname, case):
    _check(kc.case_groupnorm_cs(hip_backend, DEV, kc.DTS[dtname], *case), dtname, tol16=kc.TOL16_EMU)


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("case", kc.LN_CASES, ids=lambda c: "M%d_d%d" % c)
def test_layernorm(hip_backend, dtname, case):
    _check(kc.case_layernorm(hip_backend, DEV, kc.DTS[dtname], *case), dtname, tol16=kc.TOL16_EMU)


_TG = sorted(kc.tapgemm_cases(torch.bfloat16))


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("name", _TG)
def test_tapgemm(hip_backend, dtname, name):
    spec = kc.tapgemm_cases(kc.DTS[dtname])[name]
    res = kc.case_tapgemm(hip_backend, DEV, spec)
    cs = res.pop("colstats", None)
    _check(res, dtname, out_is_16=spec.out_dtype != torch.float32, tol16=kc.TOL16_EMU)
    if cs is not None:
        assert cs["finite"] and cs["rel_l2"] <= 2e-5, cs


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
def test_splitk_launches_are_repeatable(hip_backend, dtname):
    """Split-K launches (partial fp32 tiles to a caller workspace, `splitk_reduce_kernel` sums them in split order and
    applies the epilogue): every split launch of the list — the planner's own picks, incl. the 4 x 7 level of the
    benchmark step at its own shapes — is issued 12 times, interleaved with the others on one stream, into a workspace
    pre-filled with NaN (an element the main kernel did not write, or a reducer that ran ahead of it, would surface): each
    output must match the emulator and be BIT-identical to the first run of the same launch.  (r05 built the reduction
    into the main kernel — last block of a tile to arrive — and ran exactly this test on it: csrc/tapgemm.hip, the note
    above splitk_reduce_kernel, has what that measured and why the second launch stayed.)"""
    dt = kc.DTS[dtname]
    specs = kc.splitk_specs(dt)
    dev_specs, first = {}, {}
    for name, spec in specs.items():
        g = kc._clone_spec(spec, DEV)
        shape, bn, sk = hip_backend.tapgemm_plan(g)
        assert sk > 1, (name, shape, bn, sk)
        g.ws = torch.full((sk * g.M * g.N,), float("nan"), dtype=torch.float32, device=DEV)
        dev_specs[name] = g
    for rnd in range(12):
        for name, g in dev_specs.items():
            if rnd % 4 == 3:
                g.ws.fill_(float("nan"))
            out = hip_backend.tapgemm(g).clone()
            if rnd == 0:
                first[name] = out
                ref = kc.EMU.tapgemm(specs[name])
                st = kc.stats(out, ref)
                tol = kc.TOL32 if g.out_dtype == torch.float32 else kc.TOL16_EMU[dtname]
                assert st["finite"] and st["rel_l2"] <= tol, (name, st)
            else:
                assert torch.equal(out, first[name]), (name, rnd)


_R06 = sorted(kc.r06_shape_cases(torch.bfloat16))


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("name", _R06)
def test_tapgemm_r06_shapes(hip_backend, dtname, name):
    """pp256 / q128 (csrc/tapgemm.hip, r06): the launch is put on the shape by a one-row plan table installed through the
    PRODUCT ABI (vgen_tapgemm_set_plans — how tools/autotune_gemm.py's measured table reaches make_plan), the planner must
    confirm it, and the result must match the emulator like every other shape's."""
    import ctypes as C
    spec, plan = kc.r06_shape_cases(kc.DTS[dtname])[name]
    row = list(kc.plan_signature(spec)) + list(plan)
    arr = (C.c_int64 * 12)(*row)
    try:
        assert hip_backend.lib.vgen_tapgemm_set_plans(arr, 1) == 0
        assert tuple(hip_backend.tapgemm_plan(kc._clone_spec(spec, DEV))) == plan, name
        res = kc.case_tapgemm(hip_backend, DEV, spec)
    finally:
        hip_backend.lib.vgen_tapgemm_set_plans(None, -1)
    cs = res.pop("colstats", None)
    _check(res, dtname, out_is_16=spec.out_dtype != torch.float32, tol16=kc.TOL16_EMU)
    if cs is not None:
        assert cs["finite"] and cs["rel_l2"] <= 2e-5, cs


_TGDW = sorted(kc.tapgemm_dw_cases(torch.bfloat16))


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("name", _TGDW)
def test_tapgemm_dualw(hip_backend, dtname, name):
    """Dual-W launches (precision="high"): vs the emulator's A . (W_hi + W_lo)^T, and — the point of the mode — vs the
    product with the UNROUNDED fp32 weight: what is left of the weight rounding is 2^-22 (fp16) / 2^-16 (bf16)."""
    import torch_ops_ref as tr
    spec = kc.tapgemm_dw_cases(kc.DTS[dtname])[name]
    res = kc.case_tapgemm(hip_backend, DEV, spec)
    cs = res.pop("colstats", None)
    _check(res, dtname, out_is_16=spec.out_dtype != torch.float32, tol16=kc.TOL16_EMU)
    if cs is not None:
        assert cs["finite"] and cs["rel_l2"] <= 2e-5, cs
    out = hip_backend.tapgemm(kc._clone_spec(spec, DEV)).float().cpu()
    exact = spec.out_dtype == torch.float32
    if spec.split_out:                       # two-term rows: hi + lo reconstructs the fp32 value
        out, exact = out[:, : spec.N] + out[:, spec.N:], True
    err = kc.stats(out, tr.ref_tapgemm(spec, w32=spec.W.vgen_w32))["rel_l2"]
    tol = (2e-5 if dtname == "fp16" else 1.5e-4) if exact else kc.TOL16[dtname]
    assert err <= tol, (name, err)


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
def test_kernels_vs_plain_torch_operators(hip_backend, dtname):
    """The HIP kernels against torch's own fp32 operators on NCHW tensors (tests/torch_ops_ref.py: F.conv2d / conv1d /
    linear / group_norm / layer_norm / scaled_dot_product_attention), every tap-GEMM and attention case plus GroupNorm
    and LayerNorm — no layout convention of the ABI is shared with this reference."""
    import torch_ops_ref as tr
    dt = kc.DTS[dtname]
    tol16 = kc.TOL16[dtname]
    for name, spec in kc.tapgemm_cases(dt).items():
        out = hip_backend.tapgemm(kc._clone_spec(spec, DEV)).float().cpu()
        exact = spec.out_dtype == torch.float32
        if spec.split_out:
            out, exact = out[:, : spec.N] + out[:, spec.N:], dtname == "fp16"
        err = kc.stats(out, tr.ref_tapgemm(spec))["rel_l2"]
        assert err <= (kc.TOL32 if exact else tol16), (name, err)
    g = torch.Generator().manual_seed(3)
    for nb, S, C1, C2, silu in [(2, 1792, 320, 0, True), (3, 448, 640, 640, False), (16, 28, 1280, 0, True)]:
        x1 = torch.randn(nb * S, C1, generator=g) * 1.7 + 0.6
        x2 = torch.randn(nb * S, C2, generator=g) if C2 else None
        ga, be_ = 1 + 0.2 * torch.randn(C1 + C2, generator=g), 0.3 * torch.randn(C1 + C2, generator=g)
        y, _ = hip_backend.groupnorm(x1.to(DEV), None if x2 is None else x2.to(DEV), nb, S, 32, 1e-5, ga.to(DEV),
                                     be_.to(DEV), silu, False, dt)
        err = kc.stats(y, tr.ref_groupnorm(x1, x2, nb, S, 32, 1e-5, ga, be_, silu))["rel_l2"]
        assert err <= tol16, ("groupnorm", nb, S, err)
    x = torch.randn(3000, 320, generator=g) * 2 + 0.5
    ga, be_ = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    err = kc.stats(hip_backend.layernorm(x.to(DEV), ga.to(DEV), be_.to(DEV), 1e-5, dt), tr.ref_layernorm(x, ga, be_, 1e-5))["rel_l2"]
    assert err <= tol16, ("layernorm", err)
    for name, spec in kc.attn_cases(dt).items():
        dspec = kc._clone_spec(spec, DEV)
        hip_backend.attention(dspec)
        cspec = kc._clone_spec(spec, "cpu")
        cspec.out = dspec.out.cpu()
        ref, got = tr.ref_attention(cspec)
        err = kc.stats(got, ref)["rel_l2"]
        assert err <= 3 * tol16, (name, err)     # P is rounded to 16 bit before the PV product


_AT = sorted(kc.attn_cases(torch.bfloat16))


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
@pytest.mark.parametrize("name", _AT)
def test_attention(hip_backend, dtname, name):
    spec = kc.attn_cases(kc.DTS[dtname])[name]
    res = kc.case_attention(hip_backend, DEV, spec)
    # P is rounded to 16 bit before the PV product (like every flash kernel): 3x the plain bound
    for st in res.values():
        assert st["finite"] and st["rel_l2"] <= 3 * kc.TOL16[dtname], st


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
def test_softmax_rows(hip_backend, dtname):
    res = kc.case_softmax_rows(hip_backend, DEV, kc.DTS[dtname], 70, 200, 256)
    assert res["P"]["pad_untouched"]
    _check(res, dtname)


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
def test_small_kernels(hip_backend, dtname):
    dt = kc.DTS[dtname]
    _check(kc.case_act_cast(hip_backend, DEV, dt, 5000, 1), dtname)
    _check(kc.case_act_cast(hip_backend, DEV, dt, 777, 0), dtname)
    _check(kc.case_act_cast(hip_backend, DEV, dt, 4099, 2), dtname)          # exact GELU (text tower MLP)
    _check(kc.case_timestep_embedding(hip_backend, DEV, dt, 320), dtname)
    _check(kc.case_im2col(hip_backend, DEV, dt, "bcfhw"), dtname)
    _check(kc.case_im2col(hip_backend, DEV, dt, "rows"), dtname)


def test_pointwise_and_gaussian(hip_backend):
    _check(kc.case_pointwise(hip_backend, DEV), "fp16", out_is_16=False)
    _check(kc.case_gaussian(hip_backend, DEV), "fp16", out_is_16=False)


@pytest.mark.parametrize("mean_type", [0, 1, 2])
@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_cfg_ddim_step_bit_exact(hip_backend, mean_type, eta):
    res = kc.case_cfg_ddim(hip_backend, DEV, mean_type, eta)
    assert res["xt_1"]["bit_exact"] and res["x0"]["bit_exact"], res


def test_frames_u8_bit_exact(hip_backend):
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(4001, 3, generator=gen) * 1.3
    x[:7, 0] = torch.tensor([1.0, -1.0, 0.0, 0.999999, -0.9999999, 1.0000001, 0.00392])
    mean = torch.tensor([0.5, 0.4, 0.6])
    std = torch.tensor([0.5, 0.55, 0.45])
    ref = kc.EMU.frames_u8(x, mean, std)
    out = hip_backend.frames_u8(x.to(DEV), mean.to(DEV), std.to(DEV))
    assert out.dtype == torch.uint8 and torch.equal(out.cpu(), ref)
    xp = torch.randn(100, 8, generator=gen)                    # strided rows (ldx > C)
    out = hip_backend.frames_u8(xp.to(DEV)[:, :3], mean.to(DEV), std.to(DEV))
    assert torch.equal(out.cpu(), kc.EMU.frames_u8(xp[:, :3], mean, std))


def test_linear_f32_and_fp32_sinusoid(hip_backend):
    """vgen_linear_f32 vs float64, and its contract that an output element does not depend on how many rows are
    evaluated together or where the row sits (a session's time-embedding table row == the per-step evaluation)."""
    g = torch.Generator().manual_seed(11)
    for K, N in ((320, 1280), (1280, 20160), (64, 100)):
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        x = torch.randn(19, K, generator=g)
        add = torch.randn(19, N, generator=g)
        for act in (0, 1):
            out = hip_backend.linear_f32(x.to(DEV), W.to(DEV), b.to(DEV), act_in=act, add=add.to(DEV)).cpu()
            v = x.double()
            v = v * torch.sigmoid(v) if act else v
            ref = (v @ W.double().t() + b.double() + add.double()).float()
            assert float((out - ref).norm() / ref.norm()) < 2e-6
            one = hip_backend.linear_f32(x[5:6].to(DEV), W.to(DEV), b.to(DEV), act_in=act).cpu()
            for n, pos in ((2, 1), (8, 7), (9, 8), (1000, 601)):
                xs = torch.randn(n, K, generator=g)
                xs[pos] = x[5]
                o = hip_backend.linear_f32(xs.to(DEV), W.to(DEV), b.to(DEV), act_in=act).cpu()
                assert torch.equal(o[pos], one[0]), (K, N, act, n, pos)
    t = torch.tensor([0.0, 1.0, 601.0, 999.0, 12.5])
    s = hip_backend.timestep_embedding(t.to(DEV), 320, torch.float32).cpu()
    assert s.dtype == torch.float32 and float((s - kc.EMU.timestep_embedding(t, 320, torch.float32)).abs().max()) < 2e-4


def test_ddim_update_units_bit_exact(hip_backend):
    """vgen_cfg_ddim_step_units: table-indexed coefficients, x_t read from the unit slots, x_{t-1} replicated into
    them (in place) — the same bits as the plain kernel."""
    g = torch.Generator().manual_seed(3)
    G, B, Cs, Cl, F, H, W = 2, 3, 6, 4, 2, 5, 7
    xu = torch.randn(G * B, Cs, F, H, W, generator=g)
    xt = torch.randn(B, Cl, F, H, W, generator=g)
    xu.view(G, B, Cs, F, H, W)[:, :, :Cl] = xt
    y, u, nz = (torch.randn(B, Cl, F, H, W, generator=g) for _ in range(3))
    tab = torch.rand(50, 7, generator=g) * 0.8 + 0.1
    tab[:, 5] *= 0.3
    tidx = torch.tensor([7, 0, 49])
    ref, ref0 = kc.EMU.cfg_ddim_step(xt, y, u, nz, tab[tidx].contiguous(), 9.0, True, 1, True)
    xud = xu.to(DEV)
    o = torch.empty_like(xt, device=DEV)
    o0 = torch.empty_like(xt, device=DEV)
    hip_backend.ddim_update_units(xud, G, B, Cl, y.to(DEV), u.to(DEV), nz.to(DEV), tab.to(DEV), tidx.to(DEV), 9.0, True,
                                  1, o, o0, replicate=True)
    got, got0 = hip_backend.cfg_ddim_step(xt.to(DEV), y.to(DEV), u.to(DEV), nz.to(DEV), tab[tidx].contiguous().to(DEV),
                                          9.0, True, 1, True)
    assert torch.equal(o, got) and torch.equal(o0, got0)
    assert float((o.cpu() - ref).abs().max()) < 1e-5
    v = xud.view(G, B, Cs, F, H, W).cpu()
    assert torch.equal(v[0, :, :Cl], o.cpu()) and torch.equal(v[1, :, :Cl], o.cpu())
    assert torch.equal(v[:, :, Cl:], xu.view(G, B, Cs, F, H, W)[:, :, Cl:])          # stem channels untouched


def test_embed_tokens_and_fp32_layernorm(hip_backend):
    g = torch.Generator().manual_seed(2)
    table, pos = torch.randn(300, 128, generator=g), torch.randn(20, 128, generator=g)
    tok = torch.randint(0, 300, (3, 20), generator=g)
    out = hip_backend.embed_tokens(tok.to(DEV), table.to(DEV), pos.to(DEV)).cpu()
    assert torch.equal(out, kc.EMU.embed_tokens(tok, table, pos))
    for d in (128, 1024, 1280):
        x = torch.randn(77, d, generator=g) * 2 + 0.3
        ga, be_ = torch.randn(d, generator=g), torch.randn(d, generator=g)
        y = hip_backend.layernorm(x.to(DEV), ga.to(DEV), be_.to(DEV), 1e-5, torch.float32).cpu()
        ref = torch.nn.functional.layer_norm(x.double(), (d,), ga.double(), be_.double(), 1e-5).float()
        assert y.dtype == torch.float32 and float((y - ref).norm() / ref.norm()) < 2e-6


def test_condition_stem_kernels(hip_backend):
    """vgen_conv3x3_small / vgen_adaptive_avgpool2d / vgen_frame_transformer (fp32) vs torch on the same inputs."""
    g = torch.Generator().manual_seed(4)
    for n, cin, cout, H, W, stride, act in ((3, 4, 16, 11, 9, 1, 1), (2, 16, 4, 8, 8, 1, 0), (2, 32, 64, 32, 32, 2, 1),
                                            (1, 64, 1024, 16, 16, 2, 0), (2, 1, 32, 24, 40, 1, 1)):
        x = torch.randn(n, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
        b = torch.randn(cout, generator=g)
        y = hip_backend.conv3x3_small(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, act=act).cpu()
        ref = kc.EMU.conv3x3_small(x, w, b, stride=stride, act=act)
        assert y.shape == ref.shape and float((y - ref).norm() / ref.norm()) < 2e-6, (n, cin, cout, H, W, stride)
    for H, W, Ho, Wo in ((88, 160, 32, 32), (24, 40, 12, 20), (7, 5, 3, 4), (16, 16, 16, 16)):
        x = torch.randn(2, 5, H, W, generator=g)
        y = hip_backend.adaptive_avgpool2d(x.to(DEV), Ho, Wo).cpu()
        assert float((y - kc.EMU.adaptive_avgpool2d(x, Ho, Wo)).abs().max()) < 1e-6
    for B, F, d, H, W, heads, dh, hidden, has_out in ((2, 16, 4, 5, 7, 2, 4, 16, True), (1, 32, 8, 3, 4, 2, 8, 32, True),
                                                      (2, 3, 8, 2, 2, 1, 8, 32, False)):
        inner = heads * dh
        p = dict(ln_w=1 + 0.2 * torch.randn(d, generator=g), ln_b=0.2 * torch.randn(d, generator=g),
                 wqkv=torch.randn(3 * inner, d, generator=g) / d ** 0.5,
                 wout=torch.randn(d, inner, generator=g) / inner ** 0.5 if has_out else None,
                 bout=0.1 * torch.randn(d, generator=g) if has_out else None,
                 w1=torch.randn(hidden, d, generator=g) / d ** 0.5, b1=0.1 * torch.randn(hidden, generator=g),
                 w2=torch.randn(d, hidden, generator=g) / hidden ** 0.5, b2=0.1 * torch.randn(d, generator=g),
                 heads=heads, dim_head=dh, hidden=hidden)
        pd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in p.items()}
        x = torch.randn(B * F, d, H, W, generator=g)
        mid = hip_backend.frame_transformer(x.to(DEV), B, F, d, H * W, pd).cpu()
        assert float((mid - kc.EMU.frame_transformer(x, B, F, d, H * W, p)).norm() / mid.norm()) < 5e-6
        base = torch.randn(B * d * F * H * W, generator=g)
        out = hip_backend.frame_transformer(x.to(DEV), B, F, d, H * W, pd, out=base.clone().to(DEV), last=True,
                                            out_scale=2.0, accumulate=True).cpu()
        ref = kc.EMU.frame_transformer(x, B, F, d, H * W, p, out=base.clone(), last=True, out_scale=2.0, accumulate=True)
        assert float((out - ref).norm() / ref.norm()) < 5e-6


def test_step_graph_glue_kernels(hip_backend):
    """vgen_repeat_rows / vgen_gather_rows_f32 (the copies inside a captured step): bit-equal to torch's repeat / index_select,
    fp32 and 16-bit payloads, out-of-range indices clamped."""
    g = torch.Generator().manual_seed(0)
    for shape, dt in (((1792, 320), torch.float32), ((448, 640), torch.float16), ((28, 2, 320), torch.float32)):
        t = torch.randn(shape, generator=g).to(dt).to(DEV)
        for G in (1, 2, 3):
            assert torch.equal(hip_backend.repeat_rows(t, G), t.repeat((G,) + (1,) * (t.dim() - 1)))
    tab = torch.randn(1000, 20160, generator=g).to(DEV)
    idx = torch.tensor([981, 0, 999, 981, 1005, -3], dtype=torch.long, device=DEV)
    assert torch.equal(hip_backend.gather_rows_f32(tab, idx), tab[idx.clamp(0, 999)])


@pytest.mark.parametrize("dtname", ["bf16", "fp16"])
def test_cast_split_is_the_emulators_bits(hip_backend, dtname):
    """vgen_cast_split (two-term activations): hi and lo columns bit-equal to the emulator's, hi + lo within 2^-17 (bf16) /
    2^-22 (fp16) of the fp32 input, strided destination with a column offset (the ResBlock's two-source skip operand)."""
    dt = kc.DTS[dtname]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 320, generator=g) * 3
    ref = kc.EMU.cast_split(x, dt)
    out = hip_backend.cast_split(x.to(DEV), dt)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    rec = out[:, :320].float() + out[:, 320:].float()
    assert float((rec.cpu() - x).abs().max() / x.abs().max()) < (2e-5 if dtname == "bf16" else 1e-6)
    x2 = torch.randn(1000, 128, generator=g)
    buf_ref = torch.zeros(1000, 2 * 448, dtype=dt)
    kc.EMU.cast_split(x, dt, out=buf_ref, col=0, lo_off=448)
    kc.EMU.cast_split(x2, dt, out=buf_ref, col=320, lo_off=448)
    buf = torch.zeros(1000, 2 * 448, dtype=dt, device=DEV)
    hip_backend.cast_split(x.to(DEV), dt, out=buf, col=0, lo_off=448)
    hip_backend.cast_split(x2.to(DEV), dt, out=buf, col=320, lo_off=448)
    assert torch.equal(buf.cpu().view(torch.int16), buf_ref.view(torch.int16))
