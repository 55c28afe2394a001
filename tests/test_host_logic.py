"""CPU: the HOST side of the drop-ins (registry seam, state_dict parity, weight packing, layout /
stride bookkeeping, call order, sampler plumbing) run on the CPU emulator of the C ABI
(oracle/abi_emulator.py, a test double) against reference-generated fixtures.  The kernels
themselves are checked by the -m gpu tests."""
import pytest
import torch

from conftest import gold, rel_l2
from oracle import torch_ref


def _unet(dtname="fp16"):
    from vgen_amd.unet import UNetSD_T2VBase
    g = gold("unet_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    m = UNetSD_T2VBase(**g["cfg"], compute_dtype=dtname, precision="fast").eval()
    m.load_state_dict(sd, strict=True)
    return m, g, sd


def test_state_dict_keys_match_reference_fixture():
    m, g, _ = _unet()
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in g["shapes"].items()}
    assert "middle_block.0.temopral_conv.conv1.2.weight" in mine      # the reference's spelling
    gf = gold("unet_t2v_full.pt")
    from vgen_amd.unet import UNetSD_T2VBase
    with torch.device("meta"):
        full = UNetSD_T2VBase(**gf["cfg"])
    shapes = {k: tuple(v.shape) for k, v in full.state_dict().items()}
    assert shapes == {k: tuple(v) for k, v in gf["shapes"].items()}
    assert len(shapes) == 1480 and sum(torch.Size(s).numel() for s in shapes.values()) == 1411233860  # 1411.2 M


@pytest.mark.parametrize("dtname,tol", [("fp16", 3e-3), ("bf16", 2.5e-2)])
def test_unet_host_logic_vs_reference_golden(emu_backend, dtname, tol):
    m, g, _ = _unet(dtname)
    out = m(g["x"], g["t"], y=g["y"])
    assert out.shape == g["out"].shape
    assert rel_l2(out, g["out"]) < tol


def test_every_block_alone_vs_reference_golden(emu_backend):
    """Host logic of each block (operand packing, row / stride bookkeeping, skip-concat split, temporal layout) on the
    reference's own per-block inputs — see tests/block_cases.py; the kernels proper are the GPU twin of this test."""
    from block_cases import run_blocks
    res = run_blocks("fp16", "cpu")
    assert len(res) == 36 and {k for k, _ in res.values()} == {"ResBlock", "SpatialTransformer", "TemporalTransformer",
                                                               "Downsample", "Upsample"}
    bad = {n: v for n, v in res.items() if not v[1] < 2e-3}
    assert not bad, bad


def test_every_vae_block_alone_vs_reference_golden(emu_backend):
    from block_cases import run_vae_blocks
    res = run_vae_blocks("fp16", "cpu")
    assert len(res) == 24 and {k for k, _ in res.values()} == {"ResnetBlock", "AttnBlock", "Upsample", "Downsample"}
    bad = {n: v for n, v in res.items() if not v[1] < 2e-3}
    assert not bad, bad


def test_high_precision_mode_splits_every_packed_weight(emu_backend):
    """precision="high": the packers attach to every 16-bit weight operand its two-term form `.vgen_dw` = per 64-column
    K-tile [W_hi | W_lo] (W_hi + W_lo == the fp32 weight to ~2^-21); every launch that finds it is ONE dual-W launch
    (vgen_tapgemm_args.dualw) — linear, 3x3 and temporal gathers alike; the tiny UNet moves from 1.5e-3 to 1.1e-3 of
    the reference's fp32 forward."""
    from vgen_amd import ops
    m, g, sd = _unet("fp16")
    from vgen_amd.unet import UNetSD_T2VBase
    mh = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="high").eval()
    mh.load_state_dict(sd, strict=True)
    P = mh.pack()
    name = next(n for n, mod in mh.named_modules() if type(mod).__name__ == "_ResBlockP" and isinstance(mod.skip_connection, torch.nn.Conv2d))
    for w in (P[name]["conv1"][0], P[name]["conv2"][0], P[name]["tconv1"][0], P["kv_all"],
              P[next(n for n, mod in mh.named_modules() if type(mod).__name__ == "_SpatialTransformerP")]["tb"]["ff1"][0]):
        assert w.vgen_dw.shape == (w.shape[0], 2 * w.shape[1]) and w.vgen_dw.dtype == w.dtype
        hi, lo = ops.dw_terms(w.vgen_dw)
        assert torch.equal(hi, w)
    rb = mh.get_submodule(name)
    wsk = rb.skip_connection.weight.detach().reshape(rb.cout, -1)
    # out-conv | skip | skip: the skip conv's operand is a two-term pair [raw_hi | raw_lo] in this mode
    full = torch.cat([rb.out_layers[3].weight.detach().permute(0, 2, 3, 1).reshape(rb.cout, -1), wsk, wsk], 1)
    w = P[name]["conv2"][0]
    hi, lo = ops.dw_terms(w.vgen_dw)
    assert float(((hi.float() + lo.float()) - full).abs().max() / full.abs().max()) < 2e-6
    e_fast = rel_l2(m(g["x"], g["t"], y=g["y"]), g["out"])
    e_high = rel_l2(mh(g["x"], g["t"], y=g["y"]), g["out"])
    assert e_high < 0.8 * e_fast and e_high < 1.2e-3, (e_fast, e_high)
    # every tap-GEMM of the high-precision forward is one launch on a two-term weight (the stem's own [hi | lo | hi]
    # split and the cross-attention K/V aside, nothing runs on a plain 16-bit weight)
    seen = []
    be = ops.backend()
    orig = be.tapgemm
    be.tapgemm = lambda spec: (seen.append((spec.mode, getattr(spec.W, "vgen_dw", None) is not None)), orig(spec))[1]
    try:
        mh(g["x"], g["t"], y=g["y"])
    finally:
        del be.tapgemm
    assert {md for md, _ in seen} == {0, 1, 2}
    plain = [md for md, dw in seen if not dw]
    assert plain == [0], plain                                      # the stem conv (its own 3-segment split)


def test_mixed_precision_splits_by_resolution_level(emu_backend):
    """precision="mixed": two-term weights exactly in the blocks of the resolution levels named by MIXED_LEVELS (the
    constructor's own scale bookkeeping replayed), plain 16-bit weights elsewhere; the error lands between the two pure
    modes."""
    from vgen_amd.unet import UNetSD_T2VBase
    m, g, sd = _unet("fp16")
    mm = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="mixed").eval()
    mm.MIXED_LEVELS = {"enc": (0,), "mid": (), "dec": (0,)}
    mm.load_state_dict(sd, strict=True)
    lv = mm._block_levels()
    assert lv["input_blocks.0"] == ("enc", 0) and lv["middle_block"][0] == "mid"
    assert max(l for _, l in lv.values()) == len(g["cfg"]["dim_mult"]) - 1
    assert {k.split(".")[0] for k in lv} == {"input_blocks", "middle_block", "output_blocks"}
    assert len([k for k in lv if k.startswith("output_blocks")]) == len(mm.output_blocks)
    assert len([k for k in lv if k.startswith("input_blocks")]) == len(mm.input_blocks)
    P = mm.pack()

    def first_weight(d):
        for v in d.values():
            v = v[0] if isinstance(v, tuple) else v
            if isinstance(v, dict):
                return first_weight(v)
            if torch.is_tensor(v) and v.dtype == torch.float16:
                return v
        raise KeyError

    n_split = n_plain = 0
    for name, mod in mm.named_modules():
        if type(mod).__name__ in ("_ResBlockP", "_SpatialTransformerP", "_TemporalTransformerP"):
            top = "middle_block" if name.startswith("middle_block") else ".".join(name.split(".")[:2])
            side, level = lv[top]
            has = getattr(first_weight(P[name]), "vgen_dw", None) is not None
            # first_weight: conv1 of a ResBlock, proj_in of a transformer — level 0: every kind; level 1: the r04 extra
            # kinds (conv2 / proj_in / proj_out: MIXED_EXTRA_KINDS), so proj_in yes, conv1 no
            want = (level == 0 and side != "mid") or (level == 1 and side != "mid" and type(mod).__name__ != "_ResBlockP")
            assert has == want, (name, side, level, has)
            n_split += has
            n_plain += not has
    assert n_split > 0 and n_plain > 0
    # r04: inside the two-term levels the FeedForward pair and the cross-attention query stay single-pass
    # (UNetSD_T2VBase.MIXED_SINGLE_KINDS); "mixed:...:all" and "high" keep them two-term
    dw = lambda w: getattr(w[0] if isinstance(w, tuple) else w, "vgen_dw", None) is not None
    lvl = lambda n: lv["middle_block" if n.startswith("middle_block") else ".".join(n.split(".")[:2])]
    st0 = next(n for n, mod in mm.named_modules() if type(mod).__name__ == "_SpatialTransformerP" and lvl(n) == ("enc", 0))
    tb = P[st0]["tb"]
    assert dw(tb["qkv1"]) and dw(tb["o1"]) and dw(tb["o2"]) and dw(P[st0]["pin"]) and dw(P[st0]["pout"])
    assert not dw(tb["ff1"]) and not dw(tb["ff2"]) and not dw(tb["q2"])
    # ... and at level 1 exactly the extra kinds are two-term: the ResBlock out-conv, proj_in, proj_out
    assert mm.MIXED_EXTRA_KINDS == {1: ("conv2", "pin", "pout")}
    rb1 = next(n for n, mod in mm.named_modules() if type(mod).__name__ == "_ResBlockP" and lvl(n) == ("enc", 1))
    st1 = next(n for n, mod in mm.named_modules() if type(mod).__name__ == "_SpatialTransformerP" and lvl(n) == ("dec", 1))
    assert dw(P[rb1]["conv2"]) and not dw(P[rb1]["conv1"]) and not dw(P[rb1]["tconv1"])
    assert dw(P[st1]["pin"]) and dw(P[st1]["pout"]) and not dw(P[st1]["tb"]["qkv1"]) and not dw(P[st1]["tb"]["o1"])
    rbm = next(n for n, mod in mm.named_modules() if type(mod).__name__ == "_ResBlockP" and n.startswith("middle_block"))
    assert not dw(P[rbm]["conv2"])
    ma = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="mixed:e0d0:all").eval()
    ma.load_state_dict(sd, strict=True)
    assert ma.precision == "mixed" and ma.MIXED_SINGLE_KINDS == () and ma.MIXED_EXTRA_KINDS == {} and ma.MIXED_LEVELS["enc"] == (0,)
    Pa = ma.pack()
    tba = Pa[st0]["tb"]
    assert dw(tba["ff1"]) and dw(tba["ff2"]) and dw(tba["q2"]) and not dw(Pa[rb1]["conv2"]) and not dw(Pa[st1]["pin"])
    mn = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="mixed:e0d0:noextra").eval()   # level 0 minus ff1 / ff2 / q2
    mn.load_state_dict(sd, strict=True)
    assert mn.MIXED_SINGLE_KINDS == ("ff1", "ff2", "q2") and mn.MIXED_EXTRA_KINDS == {} and mn.MIXED_LEVELS["dec"] == (0,)
    Pn = mn.pack()
    assert dw(Pn[st0]["tb"]["qkv1"]) and not dw(Pn[st0]["tb"]["ff1"]) and not dw(Pn[rb1]["conv2"]) and not dw(Pn[st1]["pin"])
    e_fast = rel_l2(m(g["x"], g["t"], y=g["y"]), g["out"])
    mh = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="high").eval()
    mh.load_state_dict(sd, strict=True)
    e_high = rel_l2(mh(g["x"], g["t"], y=g["y"]), g["out"])
    e_mixed = rel_l2(mm(g["x"], g["t"], y=g["y"]), g["out"])
    assert e_high <= e_mixed * 1.02 and e_mixed < e_fast, (e_fast, e_mixed, e_high)


def test_dual_w_launch_semantics_vs_fp32_weights(emu_backend):
    """The dual-W cases of the GPU suite on the emulator: A . (W_hi + W_lo)^T reproduces the product with the unrounded
    fp32 weight to the residual of the split (2^-22 fp16 / 2^-17 bf16), through plain-torch operators on NCHW tensors."""
    import kernel_cases as kc
    import torch_ops_ref as tr
    for dtname, tol in (("fp16", 3e-6), ("bf16", 6e-5)):
        dt = kc.DTS[dtname]
        for name, spec in kc.tapgemm_dw_cases(dt).items():
            if spec.M * spec.N * (spec.taps * spec.C1 + spec.C2) > 3e10:
                continue                                            # the big ones run on the GPU
            out = kc.EMU.tapgemm(spec).float()
            exact = spec.out_dtype == torch.float32
            if spec.split_out:
                out, exact = out[:, : spec.N] + out[:, spec.N:], True
            ref = tr.ref_tapgemm(spec, w32=spec.W.vgen_w32)
            err = kc.stats(out, ref)["rel_l2"]
            lim = tol if exact else kc.TOL16[dtname]
            assert err <= lim, (dtname, name, err)


def test_unet_other_shapes_vs_oracle(emu_backend):
    m, g, sd = _unet("fp16")
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(1, 4, 3, 8, 12, generator=gen)         # odd F, non-square
    y = torch.randn(1, 5, 1024, generator=gen)             # short context
    t = torch.tensor([3])
    ref = torch_ref.unet_forward(sd, x, t, y, g["cfg"]["dim"])
    assert rel_l2(m(x, t, y=y), ref) < 3e-3


def test_forward_units_equals_two_forwards(emu_backend, one_thread):
    m, g, _ = _unet("fp16")
    y2 = torch.roll(g["y"], 1, 0)
    a, b = m.forward_units(g["x"], g["t"], [dict(y=g["y"]), dict(y=y2)])
    assert rel_l2(a, m(g["x"], g["t"], y=g["y"])) < 1e-5
    assert rel_l2(b, m(g["x"], g["t"], y=y2)) < 1e-5


def test_repack_after_load_state_dict(emu_backend):
    m, g, sd = _unet("fp16")
    o1 = m(g["x"], g["t"], y=g["y"])
    sd2 = torch_ref.synth_state_dict(g["shapes"], seed=99)
    m.load_state_dict(sd2, strict=True)
    o2 = m(g["x"], g["t"], y=g["y"])
    assert rel_l2(o2, torch_ref.unet_forward(sd2, g["x"], g["t"], g["y"], g["cfg"]["dim"])) < 3e-3
    assert rel_l2(o1, o2) > 0.1


@pytest.mark.parametrize("dtname,tol", [("fp16", 3e-3), ("bf16", 2e-2)])
def test_vae_host_logic_vs_reference_golden(emu_backend, dtname, tol):
    from vgen_amd.vae import AutoencoderKL
    g = gold("vae_tiny.pt")
    v = AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=4, compute_dtype=dtname).eval()
    assert {k: tuple(p.shape) for k, p in v.state_dict().items()} == {k: tuple(s) for k, s in g["shapes"].items()}
    v.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    assert rel_l2(v.decode(g["z"]), g["dec"]) < tol
    assert rel_l2(v.encode(g["img"]).parameters, g["moments"]) < tol
    torch.manual_seed(g["sample_seed"])
    assert rel_l2(v.encode_firsr_stage(g["img"], 0.18215), g["z_sample"]) < tol
    gf = gold("vae_sd_full.pt")
    with torch.device("meta"):
        full = AutoencoderKL(ddconfig=gf["ddconfig"], embed_dim=4)
    assert {k: tuple(p.shape) for k, p in full.state_dict().items()} == {k: tuple(s) for k, s in gf["shapes"].items()}


def test_vae_odd_latent_size(emu_backend):
    """h*w not a multiple of 64 exercises the padded P / V^T buffers of the mid attention."""
    from vgen_amd.vae import AutoencoderKL
    g = gold("vae_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    v = AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=4, compute_dtype="fp16").eval()
    v.load_state_dict(sd, strict=True)
    z = torch.randn(1, 4, 5, 6, generator=torch.Generator().manual_seed(2))
    assert rel_l2(v.decode(z), torch_ref.vae_decode(sd, z)) < 3e-3


def test_ddim_sampler_bit_exact_vs_reference_golden(emu_backend):
    from oracle.make_golden import dummy_model
    from vgen_amd.diffusion import DiffusionDDIM
    g = gold("ddim.pt")
    d = DiffusionDDIM(**g["cfg"])
    out = d.ddim_sample_loop(g["noise"].clone(), dummy_model, g["kw"], guide_scale=9.0, ddim_timesteps=50, eta=0.0)
    assert torch.equal(out, g["out50"])
    inv = d.ddim_reverse_sample_loop(g["x0"].clone(), dummy_model, g["kw"][0], guide_scale=None, ddim_timesteps=20)
    assert torch.equal(inv, g["inv20"])
    xt1, x0 = d.ddim_sample(g["noise"].clone(), g["step_t"], dummy_model, g["kw"], guide_scale=9.0,
                            ddim_timesteps=50, eta=0.0)
    assert torch.equal(xt1, g["step_xt1"]) and torch.equal(x0, g["step_x0"])


def test_ancestral_sampler_and_q_helpers_bit_exact_vs_reference_golden(emu_backend):
    """p_sample / p_sample_loop / p_mean_variance (both fixed variance types), the closed-form q(.) helpers, q_sample
    with offset noise and a stochastic (eta = 0.7) DDIM step against the reference's outputs (oracle/make_golden.py
    make_ddpm; the samplers draw their own noise, so the RNG stream must match draw for draw)."""
    from oracle.make_golden import dummy_model
    from vgen_amd.diffusion import DiffusionDDIM
    g = gold("ddpm.pt")
    same = lambda a, b: all(torch.equal(x, y) for x, y in zip(a, b)) and len(a) == len(b)
    for vt in ("fixed_small", "fixed_large"):
        d = DiffusionDDIM(**dict(g["cfg"], var_type=vt))
        torch.manual_seed(3)
        for tt, want in zip((torch.tensor([999, 500]), torch.tensor([1, 0])), g[vt]["steps"]):
            assert same(d.p_sample(g["noise"].clone(), tt, dummy_model, g["kw"], guide_scale=9.0), want), vt
        assert same(d.p_mean_variance(g["noise"].clone(), torch.tensor([999, 500]), dummy_model, g["kw"], guide_scale=9.0),
                    g[vt]["pmv"]), vt
        torch.manual_seed(4)
        assert torch.equal(d.p_sample_loop(g["noise"].clone(), dummy_model, g["kw"][0], guide_scale=None), g[vt]["loop"])
    d = DiffusionDDIM(**g["cfg"])
    t = torch.tensor([981, 21])
    torch.manual_seed(5)
    assert same(d.ddim_sample(g["noise"].clone(), t, dummy_model, g["kw"], guide_scale=9.0, ddim_timesteps=50, eta=0.7),
                g["ddim_eta"])
    assert same(d.q_mean_variance(g["x0"], t), g["q_mean_variance"])
    assert same(d.q_posterior_mean_variance(g["x0"], g["noise"], t), g["q_posterior"])
    torch.manual_seed(6)
    assert torch.equal(d.q_sample(g["x0"], t), g["q_sample"])
    # the reference's default var_type is 'learned_range' (diffusion_ddim.py:34): construction works as in the reference
    # (the learned-variance branches themselves: test_sampler_options_unused_by_the_configs_vs_reference_golden)
    cfg = {k: v for k, v in g["cfg"].items() if k != "var_type"}
    dl = DiffusionDDIM(**cfg)
    assert dl.var_type == "learned_range"


def test_sampler_options_unused_by_the_configs_vs_reference_golden(emu_backend):
    """Learned / learned_range variances, mean_type 'x_{t-1}', clamp / percentile and classifier guidance against the
    reference's own outputs (tests/golden/ddim_branches.pt, oracle/make_golden.py make_ddim_branches): p_mean_variance,
    p_sample, a stochastic ddim_sample and ddim_reverse_sample each.  fp32 torch elementwise ops in the reference's
    order: bit-exact."""
    from oracle.make_golden import dummy_model2, dummy_model4
    from vgen_amd.diffusion import DiffusionDDIM
    g = gold("ddim_branches.pt")
    same = lambda a, b: len(a) == len(b) and all(torch.equal(x.float(), y.float()) for x, y in zip(a, b))
    cond = lambda x, t, **k: 0.3 * torch.tanh(x)
    noise, kw, t = g["noise"], g["kw"], g["t"]
    for name, c in g["cases"].items():
        d = DiffusionDDIM(**g["cfg"], var_type=c["var_type"], mean_type=c["mean_type"])
        mdl = dummy_model2 if c["model"] == 2 else dummy_model4
        opt = dict(clamp=c.get("clamp"), percentile=c.get("percentile"))
        cf = cond if c.get("cond") else None
        assert same(d.p_mean_variance(noise.clone(), t, mdl, kw, guide_scale=9.0, **opt), c["pmv"]), name
        torch.manual_seed(7)
        assert same(d.p_sample(noise.clone(), t, mdl, kw if cf is None else kw[0], condition_fn=cf,
                               guide_scale=9.0 if cf is None else None, **opt), c["p_sample"]), name
        torch.manual_seed(8)
        assert same(d.ddim_sample(noise.clone(), t, mdl, kw if cf is None else kw[0], condition_fn=cf,
                                  guide_scale=9.0 if cf is None else None, ddim_timesteps=50, eta=0.5, **opt), c["ddim"]), name
        assert same(d.ddim_reverse_sample(noise.clone(), t, mdl, kw, guide_scale=9.0, ddim_timesteps=50, **opt),
                    c["reverse"]), name


def test_ddim_call_pattern_and_rng_parity(emu_backend):
    from oracle.make_golden import dummy_model
    from vgen_amd.diffusion import DiffusionDDIM
    g = gold("ddim.pt")
    d = DiffusionDDIM(**g["cfg"])
    calls = []

    def model(x, t, **kw):
        calls.append(int(t[0]))
        return dummy_model(x, t, **kw)

    torch.manual_seed(5)
    d.ddim_sample_loop(g["noise"].clone(), model, g["kw"], guide_scale=9.0, ddim_timesteps=50, eta=0.0)
    after = torch.rand(1)
    assert len(calls) == 100 and calls[:4] == [981, 981, 961, 961] and calls[-1] == 1
    # the reference draws randn_like(xt) every step even when eta == 0 (diffusion_ddim.py:237)
    torch.manual_seed(5)
    for _ in range(50):
        torch.randn_like(g["noise"])
    assert torch.equal(after, torch.rand(1))


def test_sampler_uses_batched_units_when_available(emu_backend):
    from vgen_amd.diffusion import DiffusionDDIM
    m, g, _ = _unet("fp16")
    d = DiffusionDDIM(**gold("ddim.pt")["cfg"])
    kw = [dict(y=g["y"]), dict(y=torch.zeros_like(g["y"]))]
    t = torch.tensor([981, 981])
    a, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    seq = lambda x, tt, **k: m(x, tt, **k)                 # plain callable -> two sequential calls
    b, _ = d.ddim_sample(g["x"], t, seq, kw, guide_scale=9.0, ddim_timesteps=50)
    assert rel_l2(a, b) < 1e-5


# ---- UNetSD_SR600 (trunk variant) ----------------------------------------------------
def test_sr600_oracle_and_host_logic_vs_reference_golden(emu_backend):
    from vgen_amd.unet import UNetSD_SR600
    g = gold("unet_sr600_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    assert rel_l2(torch_ref.unet_sr600_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"]), g["out"]) < 2e-5
    m = UNetSD_SR600(**g["cfg"], compute_dtype="fp16", precision="fast").eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], g["y"], x_lr=None)
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3


# ---- UNetSD_I2VGen (trunk variant with condition stems) -------------------------------
def test_i2vgen_oracle_and_host_logic_vs_reference_golden(emu_backend):
    from vgen_amd.unet_i2vgen import UNetSD_I2VGen
    g = gold("unet_i2vgen_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    ref = torch_ref.unet_i2vgen_forward(sd, g["x"], g["t"], g["y"], g["image"], g["local_image"], g["fps"],
                                        g["cfg"]["dim"])
    assert rel_l2(ref, g["out"]) < 2e-5                      # the restatement is pinned to the reference's output
    m = UNetSD_I2VGen(**g["cfg"], compute_dtype="fp16", precision="fast").eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    kw = dict(y=g["y"], image=g["image"], local_image=g["local_image"], fps=g["fps"])
    out = m(g["x"], g["t"], **kw)
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3
    # stems are prompt constants: cached per conditioning tensor object, recomputed when it changes
    stems = lambda li: m.condition_stems(li, g["image"], *g["x"].shape[:1], *g["x"].shape[2:])
    c0 = stems(g["local_image"])[0]
    m(g["x"] * 0.5, g["t"], **kw)
    assert stems(g["local_image"])[0] is c0
    li2 = g["local_image"] * 1.5                                  # another tensor: no aliasing by address
    assert stems(li2)[0] is not c0 and rel_l2(stems(li2)[0], c0) > 1e-3
    g["local_image"].mul_(1.0)                                    # in-place edit bumps _version -> recompute
    assert stems(g["local_image"])[0] is not c0
    # CFG pair as one batch == two calls
    kw2 = dict(kw, y=torch.roll(g["y"], 1, 1))
    a, b = m.forward_units(g["x"], g["t"], [kw, kw2])
    assert rel_l2(a, m(g["x"], g["t"], **kw)) < 1e-5 and rel_l2(b, m(g["x"], g["t"], **kw2)) < 1e-5


# ---- f3: decode epilogue to displayable bytes + engine glue a21 ---------------------------------------------
def test_decode_to_uint8_and_decode_video_match_the_engine_post_processing(emu_backend):
    from vgen_amd.vae import AutoencoderKL
    g = gold("vae_tiny.pt")
    v = AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=4, compute_dtype="fp16").eval()
    v.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    z = g["z"] * 3.0                                         # push some pixels past the clamp
    dec = v.decode(z)
    # utils/video_op.py:181-188 on the decoded frames, restated: mul_(std).add_(mean).clamp_(0,1)*255 -> uint8
    ref = ((dec * 0.5 + 0.5).clamp(0, 1) * 255.0).permute(0, 2, 3, 1).contiguous().numpy().astype("uint8")
    u8 = v.decode_to_uint8(z)
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == ref.shape
    assert (u8.numpy() == ref).all()
    assert 0 in u8 and 255 in u8
    # a21: latents [B, 4, F, h, w] -> per-video frames, chunked by decoder_bs
    lat = (z * 0.18215).view(1, 2, 4, *z.shape[2:]).permute(0, 2, 1, 3, 4).contiguous()
    vid = v.decode_video(lat, scale_factor=0.18215, decoder_bs=1)
    assert tuple(vid.shape) == (1, 2) + ref.shape[1:]
    assert (vid[0].numpy().astype(int) - ref.astype(int)).__abs__().max() <= 1     # 1/s * (s * z) is not exactly z
    f32 = v.decode_video(lat, scale_factor=0.18215, decoder_bs=2, to_uint8=False)
    assert tuple(f32.shape) == (1, 3, 2) + tuple(dec.shape[2:]) and rel_l2(f32[0].permute(1, 0, 2, 3), dec) < 1e-3


# ---- UNetSD_VideoLCM, text-only composition ------------------------------------------------
def test_videolcm_text_oracle_and_host_logic_vs_reference_golden(emu_backend):
    import types
    from vgen_amd.unet_videolcm import UNetSD_VideoLCM
    g = gold("unet_videolcm_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    ref = torch_ref.unet_videolcm_text_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"], g["cfg"]["concat_dim"])
    assert rel_l2(ref, g["out"]) < 2e-5
    cfg = types.SimpleNamespace(video_compositions=["text"], resolution=[64, 128])
    m = UNetSD_VideoLCM(config=cfg, **g["cfg"], compute_dtype="fp16", precision="fast").eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], y=g["y"])                       # float timesteps, like the LCM engine
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3
    with pytest.raises(ValueError):                           # not one of this model's compositions
        m(g["x"], g["t"], y=g["y"], depth=torch.zeros(1, 1, 4, 128, 64))
    with pytest.raises(NotImplementedError):                  # unknown composition names are refused loudly
        UNetSD_VideoLCM(config=types.SimpleNamespace(video_compositions=["text", "zebra"], resolution=[64, 128]),
                        **g["cfg"])


def test_tft2v_text_image_oracle_and_host_logic_vs_reference_golden(emu_backend):
    import types
    from vgen_amd.unet_videolcm import UNetSD_TFT2V
    g = gold("unet_tft2v_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    ref = torch_ref.unet_videolcm_text_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"], g["cfg"]["concat_dim"],
                                               image=g["image"])
    assert rel_l2(ref, g["out"]) < 2e-5
    cfg = types.SimpleNamespace(video_compositions=["text", "image"], resolution=[64, 128])
    m = UNetSD_TFT2V(config=cfg, **g["cfg"], compute_dtype="fp16", precision="fast").eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], y=g["y"], image=g["image"])
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3
    kw = dict(y=g["y"], image=g["image"])
    kw2 = dict(y=torch.roll(g["y"], 1, 1), image=g["image"] * 0.5)
    a, b = m.forward_units(g["x"], g["t"], [kw, kw2])
    assert rel_l2(a, out) < 1e-5 and rel_l2(b, m(g["x"], g["t"], **kw2)) < 1e-5


def test_vcomposer_spatial_stems_oracle_and_host_logic_vs_reference_golden(emu_backend):
    """UNetSD_TFT2V with the composition list of configs/tft2v_vcomposer_infer.yaml:74 (mask, depthmap, sketch, motion,
    image, local_image, single_sketch)."""
    import types
    from vgen_amd.unet_videolcm import UNetSD_TFT2V, UNetSD_VideoLCM
    g = gold("unet_vcomposer_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    conds = {k: v.float() for k, v in g["conds"].items()}
    ref = torch_ref.unet_composer_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"], g["cfg"]["concat_dim"],
                                          g["resolution"], image=g["image"], **conds)
    assert rel_l2(ref, g["out"]) < 2e-5
    cfg = types.SimpleNamespace(video_compositions=g["comps"], resolution=g["resolution"])
    for cls in (UNetSD_TFT2V, UNetSD_VideoLCM):            # same trunk, same parameter set
        m = cls(config=cfg, **g["cfg"], compute_dtype="fp16", precision="fast").eval()
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], y=g["y"], image=g["image"], **conds)
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3
    n = len(m._stem_cache)
    m(g["x"] * 0.9, g["t"], y=g["y"], image=g["image"], **conds)          # next denoise step: stems come from the cache
    assert len(m._stem_cache) == n == 6
    # CFG pair in one batch: conditional branch with all maps, "unconditional" with other text and a scaled depth map
    kw1 = dict(y=g["y"], image=g["image"], **conds)
    kw2 = dict(kw1, y=torch.roll(g["y"], 1, 1), depth=conds["depth"] * 0.5)
    a, b = m.forward_units(g["x"], g["t"], [kw1, kw2])
    # batched vs one-at-a-time: a different GEMM M changes the fp32 summation order, which 16-bit activations
    # amplify up to their rounding-noise floor (see test_unet_forward_units_matches_sequential on the GPU)
    assert rel_l2(a, out) < 3e-3 and rel_l2(b, m(g["x"], g["t"], **kw2)) < 3e-3
    assert abs(rel_l2(a, g["out"]) - rel_l2(out, g["out"])) < 5e-4
    with pytest.raises(ValueError):
        m(g["x"], g["t"], y=g["y"], histogram=torch.zeros(1, 3, 156))  # not in this model's compositions
    with pytest.raises(ValueError):
        m(g["x"], g["t"], y=g["y"], canny=conds["depth"])               # 'canny' is not in this model's compositions


def test_histogram_per_frame_context_oracle_and_host_logic_vs_reference_golden(emu_backend):
    """video_compositions ['text', 'histogram', 'canny']: one extra context token PER FRAME -> K/V projected per
    (prompt, frame) and addressed frame-major by the cross-attention strides."""
    import types
    from vgen_amd.unet_videolcm import UNetSD_VideoLCM
    g = gold("unet_histogram_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    canny = g["canny"].float()
    ref = torch_ref.unet_composer_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"], g["cfg"]["concat_dim"],
                                          g["resolution"], histogram=g["histogram"], canny=canny)
    assert rel_l2(ref, g["out"]) < 2e-5
    cfg = types.SimpleNamespace(video_compositions=g["comps"], resolution=g["resolution"])
    m = UNetSD_VideoLCM(config=cfg, **g["cfg"], compute_dtype="fp16", precision="fast").eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in g["shapes"].items()}
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], y=g["y"], histogram=g["histogram"], canny=canny)
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 3e-3
    kw1 = dict(y=g["y"], histogram=g["histogram"], canny=canny)
    kw2 = dict(y=torch.roll(g["y"], 1, 1), histogram=g["histogram"] * 0.5, canny=canny)
    a, b = m.forward_units(g["x"], g["t"], [kw1, kw2])       # per-frame contexts of two units stacked frame-major
    assert rel_l2(a, out) < 3e-3 and rel_l2(b, m(g["x"], g["t"], **kw2)) < 3e-3
    assert abs(rel_l2(a, g["out"]) - rel_l2(out, g["out"])) < 5e-4


def test_vae_attention_query_blocks(emu_backend):
    """The mid-block attention forms its scores per block of queries (720p latents: hw x hw would not fit);
    several ragged blocks must give the single-block result."""
    from vgen_amd.vae import AutoencoderKL
    g = gold("vae_tiny.pt")
    v = AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=4, compute_dtype="fp16").eval()
    v.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    one = v.decode(g["z"])
    v._attn_qb = 12                                        # hw = 32 -> blocks of 12, 12, 8 queries
    # the CPU BLAS sums a 12-row block in a different order than the 32-row one; where that flips a 16-bit rounding
    # the difference is amplified to the rounding-noise floor downstream (on the GPU the per-row sums do not depend on
    # the block: tests/test_gpu_model.py::test_vae_odd_frame_vs_reference_golden asserts 1e-6 there)
    assert rel_l2(v.decode(g["z"]), one) < 2e-3
    with pytest.raises(NotImplementedError):
        AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=8)



def test_composer_trunks_default_to_high_precision_with_spatial_condition_stems():
    """r04: precision=None is "mixed" for every trunk — except composer trunks whose composition list carries spatial
    condition stems: the full-width vcomposer fixture measures 1.01e-3 in "mixed", 8.6e-4 in "high" (DESIGN §4.1).  An
    explicit keyword wins."""
    import types
    from vgen_amd.unet_videolcm import UNetSD_TFT2V, UNetSD_VideoLCM
    g = gold("unet_vcomposer_tiny.pt")
    mk = lambda cls, comps, **kw: cls(**g["cfg"], config=types.SimpleNamespace(video_compositions=comps, resolution=g["resolution"]), **kw)
    with torch.device("meta"):
        assert mk(UNetSD_TFT2V, ["text", "image"]).precision == "mixed"
        assert mk(UNetSD_VideoLCM, ["text"]).precision == "mixed"
        assert mk(UNetSD_TFT2V, g["comps"]).precision == "high"
        assert mk(UNetSD_VideoLCM, ["text", "histogram", "canny"]).precision == "high"
        assert mk(UNetSD_TFT2V, g["comps"], precision="fast").precision == "fast"
        assert mk(UNetSD_TFT2V, g["comps"], precision="mixed").precision == "mixed"
