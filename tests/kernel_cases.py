"""Kernel-level parity cases: every C-ABI op on the GPU (HipBackend) against the CPU emulator of
the ABI (oracle/abi_emulator.py) on identical seeded inputs.  Shared by tests/test_gpu_kernels.py
and tests/gpu_diag.py (which dumps per-tile error maps for debugging)."""
from __future__ import annotations

import torch

from oracle.abi_emulator import EmuBackend
from vgen_amd import lib as L
from vgen_amd.ops import Attn, TapGemm

EMU = EmuBackend()
DTS = {"bf16": torch.bfloat16, "fp16": torch.float16}
# rel-L2 tolerances of 16-bit-output kernels; fp32-output kernels are held to 2e-5.
#   TOL16      a 16-bit output against an UNROUNDED fp32 reference (tests/torch_ops_ref.py, the per-block goldens): one
#              rounding of the output is 2^-9 / 2^-12 relative at worst, ~1.7e-3 / 2.1e-4 rel-L2 on random data — and the
#              attention kernels against the emulator, whose P = softmax(S) the kernel rounds to 16 bits before the P.V
#              MFMAs like every flash kernel (measured on the MI355X, tests/gpu_diag.py: 2.0 - 2.7e-3 bf16, 2.6 - 3.3e-4
#              fp16; their tests allow 3 x this bound).
#   TOL16_EMU  a 16-bit output against the emulator of the same arithmetic, for the kernels whose output is ONE rounding of
#              an fp32 value that differs from the emulator's only by summation order (tap-GEMM incl. GEGLU / dual-W /
#              split_out, GroupNorm, LayerNorm): the two roundings disagree on a few elements per million — measured
#              worst cases 3.0e-5 (bf16) / 1.3e-5 (fp16) — so SURVEY's 2e-3 per kernel is met with > 60x to spare and
#              the bound is set there (r02 held every kernel to the attention bound: VERDICT r02, item 8).
TOL16 = {"bf16": 4e-3, "fp16": 6e-4}
TOL16_EMU = {"bf16": 2e-3, "fp16": 3e-4}
TOL32 = 2e-5


def _g(seed):
    return torch.Generator("cpu").manual_seed(seed)


def _to_dev(obj, dev):
    if isinstance(obj, torch.Tensor):
        return obj.to(dev)
    return obj


def _clone_spec(g, dev):
    """Deep-copy a TapGemm/Attn spec to a device, preserving views' (offset, strides)."""
    kw = {}
    for k, v in g.__dict__.items():
        if isinstance(v, torch.Tensor):
            base = v._base if v._base is not None else v
            nb = base.to(dev)
            if v._base is not None:
                nb = torch.as_strided(nb, v.shape, v.stride(), v.storage_offset())
            if getattr(v, "vgen_dw", None) is not None:          # two-term weight: the dual-W operand travels along
                nb.vgen_dw = v.vgen_dw.to(dev)
            kw[k] = nb
        else:
            kw[k] = v
    return type(g)(**kw)


def stats(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    d = out - ref
    return dict(rel_l2=float(d.norm() / ref.norm().clamp_min(1e-30)), max_abs=float(d.abs().max()),
                ref_absmax=float(ref.abs().max()), finite=bool(torch.isfinite(out).all()))


# ---------------------------------------------------------------------------------------------
def case_groupnorm(be, dev, dt, nb, S, C1, C2, silu, raw, eps=1e-5, seed=0):
    g = _g(seed)
    x1 = torch.randn(nb * S, C1, generator=g) * 1.7 + 0.6
    x2 = torch.randn(nb * S, C2, generator=g) * 0.5 - 1.0 if C2 else None
    C = C1 + C2
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    y_ref, r_ref = EMU.groupnorm(x1, x2, nb, S, 32, eps, gamma, beta, silu, raw, dt)
    y, r = be.groupnorm(x1.to(dev), _to_dev(x2, dev), nb, S, 32, eps, gamma.to(dev), beta.to(dev), silu, raw, dt)
    res = {"y": stats(y, y_ref)}
    if raw:
        res["raw"] = stats(r, r_ref)
    return res


def case_groupnorm_cs(be, dev, dt, nb, S, C1, C2, silu, seed=0):
    """GroupNorm fed by producer column statistics (vgen_groupnorm_cs): x1 / x2 are tap-GEMM outputs."""
    assert S % 64 == 0
    g1 = make_tapgemm(dt, nb * S, C1, 64, residual=True, colstats=True, seed=seed)
    g2 = make_tapgemm(dt, nb * S, C2, 128, colstats=True, seed=seed + 1) if C2 else None
    gen = _g(seed + 2)
    C = C1 + C2
    gamma = 1 + 0.2 * torch.randn(C, generator=gen)
    beta = 0.3 * torch.randn(C, generator=gen)
    x1r = EMU.tapgemm(g1)
    x2r = EMU.tapgemm(g2) if g2 else None
    y_ref, _ = EMU.groupnorm(x1r, x2r, nb, S, 32, 1e-5, gamma, beta, silu, False, dt)
    x1 = be.tapgemm(_clone_spec(g1, dev))
    x2 = be.tapgemm(_clone_spec(g2, dev)) if g2 else None
    assert getattr(x1, "vgen_cs", None) is not None
    y, _ = be.groupnorm(x1, x2, nb, S, 32, 1e-5, gamma.to(dev), beta.to(dev), silu, False, dt)
    # and the statistics path must agree with the plain path on the same device tensors
    x1p = x1.clone()
    x2p = x2.clone() if x2 is not None else None
    y_plain, _ = be.groupnorm(x1p, x2p, nb, S, 32, 1e-5, gamma.to(dev), beta.to(dev), silu, False, dt)
    return {"y": stats(y, y_ref), "y_vs_plain": stats(y, y_plain)}


GN_CS_CASES = [(2, 12032, 320, 0, True), (3, 1792, 640, 640, True), (2, 448, 1280, 0, False), (4, 4096, 256, 128, True)]


def case_layernorm(be, dev, dt, M, d, seed=0):
    g = _g(seed)
    x = torch.randn(M, d, generator=g) * 2 + 0.5
    gamma = 1 + 0.2 * torch.randn(d, generator=g)
    beta = 0.3 * torch.randn(d, generator=g)
    ref = EMU.layernorm(x, gamma, beta, 1e-5, dt)
    y = be.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, dt)
    return {"y": stats(y, ref)}


def make_tapgemm(dt, M, N, C1, mode=L.TAP_LINEAR, C2=0, bias=True, rowbias=0, residual=False,
                 out_dtype=torch.float32, epilogue=L.EPI_NONE, a_pad=0, w_pad=0, seed=0, dualw=False, **geom):
    """Build a TapGemm spec on CPU.  a_pad: extra columns in A's storage (lda > C1) to exercise
    views; w_pad likewise for W (ldw > K).  dualw: W is a two-term weight (ops.split_weight of an fp32 matrix, kept as
    `.vgen_w32`): the launch is a dual-W one."""
    g = _g(seed)
    taps = {L.TAP_LINEAR: 1, L.TAP_CONV3X3: 9, L.TAP_TEMPORAL3: 3}[mode]
    if mode == L.TAP_CONV3X3:
        nimg = geom.pop("nimg")
        src_rows = nimg * geom["Hi"] * geom["Wi"]
        assert M == nimg * geom["Ho"] * geom["Wo"]
    else:
        src_rows = M
    A = (torch.randn(src_rows, C1 + a_pad, generator=g)).to(dt)[:, :C1]
    K = taps * C1 + C2
    if dualw:
        from vgen_amd.ops import split_weight
        w32 = torch.randn(N, K, generator=g) / (K ** 0.5)
        W = split_weight(w32, dt)
        W.vgen_w32 = w32
    else:
        W = (torch.randn(N, K + w_pad, generator=g) / (K ** 0.5)).to(dt)[:, :K]
    spec = dict(A=A, W=W, M=M, N=N, C1=C1, mode=mode, taps=taps, out_dtype=out_dtype, epilogue=epilogue)
    spec.update(geom)
    if C2:
        spec.update(A2=torch.randn(M, C2, generator=g).to(dt), C2=C2)
    if bias:
        spec["bias"] = torch.randn(N, generator=g)
    if rowbias:
        nbat = (M + rowbias - 1) // rowbias
        spec.update(rowbias=torch.randn(nbat, N + 8, generator=g)[:, 4:4 + N], rows_per_rb=rowbias)
    n_out = N // 2 if epilogue == L.EPI_GEGLU else N
    if residual:
        spec["residual"] = torch.randn(M, n_out, generator=g)
    return TapGemm(**spec)



def splitk_specs(dt):
    """Launches the planner splits along K (tests/test_abi_contract.py checks on the CPU that it does): the named parity
    cases above plus the 4 x 7 level of the benchmark step at its own shapes — the Conv3d(3,1,1) of the ResBlocks
    (util.py:1665-1680; 28 launches per step), a 3 x 3 conv over the 2560-channel concat (util.py:848) and the FeedForward
    down-projection with a 16-bit output (util.py:737)."""
    c = tapgemm_cases(dt)
    s = {k: c[k] for k in ("conv_splitk_L3", "lin_splitk_geglu", "lin_splitk_out16", "temporal_splitk")}
    s["L3_temporal_896x1280x3840"] = make_tapgemm(dt, 2 * 16 * 28, 1280, 1280, mode=L.TAP_TEMPORAL3, F=16, S=28,
                                                  residual=True, seed=3)
    s["L3_conv_896x1280x11520"] = make_tapgemm(dt, 32 * 4 * 7, 1280, 1280, mode=L.TAP_CONV3X3, nimg=32, Hi=4, Wi=7, Ho=4,
                                               Wo=7, stride=1, pad_t=1, pad_l=1, ups=0, rowbias=16 * 28, seed=4)
    s["L3_ff2_896x1280x5120_out16"] = make_tapgemm(dt, 896, 1280, 5120, out_dtype=dt, residual=True, seed=5)
    return s


def case_tapgemm(be, dev, spec, want_map=False):
    ref = EMU.tapgemm(spec)
    out = be.tapgemm(_clone_spec(spec, dev))
    res = {"out": stats(out, ref)}
    if spec.colstats:
        res["colstats"] = stats(out.vgen_cs, ref.vgen_cs)
    if want_map:
        d = (out.float().cpu() - ref.float()).abs()
        M, N = d.shape
        # coarse error map over (64-row, 16-col) tiles to localise layout bugs
        tm, tn = (M + 63) // 64, (N + 15) // 16
        emap = torch.zeros(tm, tn)
        for i in range(tm):
            for j in range(tn):
                emap[i, j] = d[i * 64:(i + 1) * 64, j * 16:(j + 1) * 16].max()
        res["emap"] = emap.tolist()
        res["sample_out"] = out.float().cpu()[:4, :8].tolist()
        res["sample_ref"] = ref.float()[:4, :8].tolist()
    return res


def make_attn(dt, kind, seed=0, **p):
    g = _g(seed)
    heads = p["heads"]
    d = heads * 64
    if kind == "spatial":        # packed qkv [nb*N, 3d]
        nb, N = p["nb"], p["N"]
        qkv = (torch.randn(nb * N, 3 * d, generator=g) * p.get("amp", 1.0)).to(dt)
        out = torch.zeros(nb * N, d, dtype=dt)
        ld = 3 * d
        return Attn(q=qkv, k=qkv[:, d:], v=qkv[:, 2 * d:], out=out, heads=heads, nq=N, nk=N, nbatch=nb,
                    inner=1, q_s=(ld, N * ld, 0), k_s=(ld, N * ld, 0), v_s=(ld, N * ld, 0),
                    o_s=(d, N * d, 0), scale=0.125, causal=p.get("causal", False))
    if kind == "cross":          # q [B*F*N, d]; kv [B*Lc, kw] slices
        B, F, N, Lc, off, kw = p["B"], p["F"], p["N"], p["Lc"], p["off"], p["kw"]
        q = torch.randn(B * F * N, d, generator=g).to(dt)
        kv = torch.randn(B * Lc, kw, generator=g).to(dt)
        out = torch.zeros(B * F * N, d, dtype=dt)
        return Attn(q=q, k=kv[:, off:off + d], v=kv[:, off + d: off + 2 * d], out=out, heads=heads, nq=N,
                    nk=Lc, nbatch=B * F, inner=F, q_s=(d, F * N * d, N * d), k_s=(kw, Lc * kw, 0),
                    v_s=(kw, Lc * kw, 0), o_s=(d, F * N * d, N * d), scale=0.125)
    if kind == "temporal":       # packed qkv [B*F*S, 3d], sequences over frames
        B, F, S = p["B"], p["F"], p["S"]
        qkv = (torch.randn(B * F * S, 3 * d, generator=g) * p.get("amp", 1.0)).to(dt)
        out = torch.zeros(B * F * S, d, dtype=dt)
        ld = 3 * d
        s3 = (S * ld, F * S * ld, ld)
        return Attn(q=qkv, k=qkv[:, d:], v=qkv[:, 2 * d:], out=out, heads=heads, nq=F, nk=F, nbatch=B * S,
                    inner=S, q_s=s3, k_s=s3, v_s=s3, o_s=(S * d, F * S * d, d), scale=0.125)
    raise ValueError(kind)


def case_attention(be, dev, spec):
    ref = EMU.attention(_clone_spec(spec, "cpu")).clone()
    out = be.attention(_clone_spec(spec, dev))
    return {"out": stats(out, ref)}


def case_softmax_rows(be, dev, dt, rows, cols, ldp, seed=0):
    g = _g(seed)
    base = torch.randn(rows, cols + 3, generator=g) * 4
    S = base[:, :cols]
    ref = EMU.softmax_rows(S, cols, 0.37, dt)
    P = torch.zeros(rows, ldp, dtype=dt, device=dev)
    Sd = base.to(dev)[:, :cols]
    be.softmax_rows(Sd, cols, 0.37, dt, out=P)
    pad_ok = bool((P[:, cols:] == 0).all())
    r = stats(P[:, :cols], ref)
    r["pad_untouched"] = pad_ok
    return {"P": r}


def case_act_cast(be, dev, dt, n, act, seed=0):
    x = torch.randn(n, generator=_g(seed)) * 3
    return {"y": stats(be.act_cast(x.to(dev), act, dt), EMU.act_cast(x, act, dt))}


def case_timestep_embedding(be, dev, dt, dim):
    t = torch.tensor([981.0, 1.0, 400.0, 0.0])
    return {"y": stats(be.timestep_embedding(t.to(dev), dim, dt), EMU.timestep_embedding(t, dim, dt))}


def case_im2col(be, dev, dt, layout, seed=0):
    g = _g(seed)
    if layout == "bcfhw":
        B, C, F, H, W = 2, 4, 3, 6, 5
        src = torch.randn(B, C, F, H, W, generator=g)
        s = (C * F * H * W, H * W, F * H * W, W, 1)
        nimg, Fi = B * F, F
    else:                        # rows [n*H*W, C]
        n, C, H, W = 3, 3, 5, 7
        src = torch.randn(n * H * W, C, generator=g)
        s = (H * W * C, 0, 1, W * C, C)
        nimg, Fi = n, 1
    ref = EMU.im2col3x3_small(src, nimg, Fi, C, H, W, s, 64, dt)
    out = be.im2col3x3_small(src.to(dev), nimg, Fi, C, H, W, s, 64, dt)
    # split layout [hi | lo | hi]: must be the emulator's bits (the lo segment is an exact fp32 subtraction)
    ref3 = EMU.im2col3x3_small(src, nimg, Fi, C, H, W, s, 128, dt, split=True)
    out3 = be.im2col3x3_small(src.to(dev), nimg, Fi, C, H, W, s, 128, dt, split=True)
    r = stats(out3, ref3)
    r["finite"] = r["finite"] and bool(torch.equal(out3.cpu().view(torch.int16), ref3.view(torch.int16)))
    return {"col": stats(out, ref), "col_split": r}


def case_pointwise(be, dev, seed=0):
    g = _g(seed)
    n, C, H, W, Co = 2, 4, 5, 6, 3
    src = torch.randn(n, C, H, W, generator=g)
    Wm = torch.randn(Co, C, generator=g)
    b = torch.randn(Co, generator=g)
    s = (C * H * W, 0, H * W, W, 1)
    d = (H * W * Co, 0, 1, W * Co, Co)
    ref = EMU.pointwise_small(src, n, 1, C, H, W, s, Wm, b, Co, torch.zeros(n * H * W, Co), d)
    out = be.pointwise_small(src.to(dev), n, 1, C, H, W, s, Wm.to(dev), b.to(dev), Co,
                             torch.zeros(n * H * W, Co, device=dev), d)
    return {"out": stats(out, ref)}


def case_cfg_ddim(be, dev, mean_type, eta, seed=0):
    g = _g(seed)
    B = 2
    shp = (B, 4, 3, 8, 8)
    xt, y, u = (torch.randn(shp, generator=g) for _ in range(3))
    noise = torch.randn(shp, generator=g) if eta else None
    ac = torch.tensor([0.3, 0.9])
    acp = torch.tensor([0.5, 0.97])
    sig = eta * torch.sqrt((1 - acp) / (1 - ac) * (1 - ac / acp))
    coef = torch.stack([ac.sqrt(), (1 - ac).sqrt(), (1 / ac).sqrt(), (1 / ac - 1).sqrt(), acp, sig,
                        torch.tensor([1.0, 0.0])], 1).contiguous()
    r_ref, x0_ref = EMU.cfg_ddim_step(xt, y, u, noise, coef, 9.0, True, mean_type, True)
    r, x0 = be.cfg_ddim_step(xt.to(dev), y.to(dev), u.to(dev), _to_dev(noise, dev), coef.to(dev), 9.0, True,
                             mean_type, True)
    res = {"xt_1": stats(r, r_ref), "x0": stats(x0, x0_ref)}
    res["xt_1"]["bit_exact"] = bool(torch.equal(r.cpu(), r_ref))
    res["x0"]["bit_exact"] = bool(torch.equal(x0.cpu(), x0_ref))
    return res


def case_gaussian(be, dev, seed=0):
    g = _g(seed)
    n, zc, H, W = 2, 4, 5, 6
    mom = torch.randn(n * H * W, 2 * zc, generator=g) * 3
    noise = torch.randn(n, zc, H, W, generator=g)
    ref = EMU.gaussian_sample(mom, noise, n, zc, H * W, 0.18215)
    out = be.gaussian_sample(mom.to(dev), noise.to(dev), n, zc, H * W, 0.18215)
    return {"z": stats(out, ref)}


# --- the case table -----------------------------------------------------------------------------
GN_CASES = [  # nb, S, C1, C2, silu, raw
    (16, 128, 320, 0, True, False), (2, 96, 1280, 640, True, True), (1, 256, 64, 0, False, False),
    (3, 50, 128, 0, True, False), (1, 37, 2560, 0, True, False), (2, 1792, 320, 0, True, False),
    (2, 33, 640, 320, True, True),
    # two-term raw copy [hi | lo] (vgen_groupnorm raw_split): single-launch LDS kernel, register-resident kernel, streaming pass
    (2, 96, 1280, 640, True, "split"), (2, 448, 1280, 1280, True, "split"), (3, 1001, 640, 640, True, "split"),
    (2, 12000, 320, 0, True, "split"),
    # group slice > 64 KiB: the three-launch streaming path (smaller slices take the single-launch kernel)
    (2, 1792, 1280, 0, True, False), (3, 1001, 640, 640, True, True), (2, 12000, 320, 0, True, False),
    # single-launch kernel: slice just under the LDS bound, two-source rows, odd row counts
    (5, 409, 1280, 0, True, False), (3, 271, 640, 1280, False, True),
    # register-resident single launch (24 K < slice <= 72 K elements): the 8x14-level 5-D norm, two sources with the raw
    # copy, ragged last iteration, the upper bound of the path (36 iterations)
    (2, 1792, 1280, 0, True, False), (2, 448, 1280, 1280, True, True), (1, 1003, 1920, 0, False, False),
    (2, 1836, 1280, 0, True, False),
    # just above it (37 iterations): back to the streaming pipeline
    (2, 1837, 1280, 0, True, False),
]
LN_CASES = [(100, 320), (7, 1280), (300, 64), (5, 512), (1, 2048), (40003, 320), (9001, 640), (3, 1024), (700, 1280), (50, 192)]


def tapgemm_cases(dt):
    c = {}
    c["lin_300x320x320_b64"] = make_tapgemm(dt, 300, 320, 320)
    c["lin_1000x256x640_b128"] = make_tapgemm(dt, 1000, 256, 640, residual=True)
    c["lin_77x128x1024_out16"] = make_tapgemm(dt, 77, 128, 1024, out_dtype=dt, bias=False)
    c["lin_500x4x576_smallN"] = make_tapgemm(dt, 500, 4, 576)
    c["lin_130x3x128_nonvec"] = make_tapgemm(dt, 130, 3, 128, residual=True)
    c["lin_2x1280x320_tinyM"] = make_tapgemm(dt, 2, 1280, 320)
    c["lin_views_lda_ldw"] = make_tapgemm(dt, 200, 192, 128, a_pad=64, w_pad=128, residual=True)
    c["lin_rowbias"] = make_tapgemm(dt, 384, 128, 64, rowbias=96, residual=True)
    c["lin_geglu"] = make_tapgemm(dt, 200, 512, 64, epilogue=L.EPI_GEGLU, out_dtype=dt)
    c["lin_geglu_b64"] = make_tapgemm(dt, 150, 320, 128, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True)
    c["conv_s1"] = make_tapgemm(dt, 3 * 9 * 7, 128, 64, mode=L.TAP_CONV3X3, nimg=3, Hi=9, Wi=7, Ho=9, Wo=7,
                                stride=1, pad_t=1, pad_l=1, ups=0, rowbias=63)
    c["conv_s2_pad1"] = make_tapgemm(dt, 2 * 5 * 4, 64, 128, mode=L.TAP_CONV3X3, nimg=2, Hi=10, Wi=8, Ho=5,
                                     Wo=4, stride=2, pad_t=1, pad_l=1, ups=0)
    c["conv_s2_odd"] = make_tapgemm(dt, 2 * 4 * 4, 64, 64, mode=L.TAP_CONV3X3, nimg=2, Hi=7, Wi=7, Ho=4,
                                    Wo=4, stride=2, pad_t=1, pad_l=1, ups=0)
    c["conv_ups"] = make_tapgemm(dt, 2 * 10 * 12, 192, 64, mode=L.TAP_CONV3X3, nimg=2, Hi=5, Wi=6, Ho=10,
                                 Wo=12, stride=1, pad_t=1, pad_l=1, ups=1)
    c["conv_ups_crop"] = make_tapgemm(dt, 2 * 10 * 10, 64, 64, mode=L.TAP_CONV3X3, nimg=2, Hi=6, Wi=5, Ho=10,
                                      Wo=10, stride=1, pad_t=1, pad_l=1, ups=1, crop_t=1)     # SR600 Upsample: 2x, crop 1 row each side
    c["conv_vae_down"] = make_tapgemm(dt, 2 * 4 * 3, 64, 64, mode=L.TAP_CONV3X3, nimg=2, Hi=8, Wi=6, Ho=4,
                                      Wo=3, stride=2, pad_t=0, pad_l=0, ups=0)
    c["conv_skipseg"] = make_tapgemm(dt, 2 * 6 * 6, 128, 128, mode=L.TAP_CONV3X3, nimg=2, Hi=6, Wi=6, Ho=6,
                                     Wo=6, stride=1, pad_t=1, pad_l=1, ups=0, C2=192)
    c["conv_big"] = make_tapgemm(dt, 4 * 32 * 56, 320, 320, mode=L.TAP_CONV3X3, nimg=4, Hi=32, Wi=56,
                                 Ho=32, Wo=56, stride=1, pad_t=1, pad_l=1, ups=0, residual=True)
    c["conv_splitk_L3"] = make_tapgemm(dt, 16 * 4 * 7, 1280, 1280, mode=L.TAP_CONV3X3, nimg=16, Hi=4, Wi=7,
                                       Ho=4, Wo=7, stride=1, pad_t=1, pad_l=1, ups=0, residual=True, rowbias=28 * 8)
    c["lin_splitk_geglu"] = make_tapgemm(dt, 100, 512, 1024, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True)
    c["lin_splitk_out16"] = make_tapgemm(dt, 300, 256, 2048, out_dtype=dt)
    c["temporal_splitk"] = make_tapgemm(dt, 2 * 4 * 28, 640, 640, mode=L.TAP_TEMPORAL3, F=4, S=28, residual=True)
    c["temporal"] = make_tapgemm(dt, 2 * 5 * 24, 64, 64, mode=L.TAP_TEMPORAL3, F=5, S=24, residual=True)
    # column statistics for the consuming GroupNorm: ragged M (last slab partial), every shape family
    c["cs_lin_1000x320"] = make_tapgemm(dt, 1000, 320, 320, residual=True, colstats=True)
    c["cs_conv_rowbias"] = make_tapgemm(dt, 6 * 24 * 20, 128, 64, mode=L.TAP_CONV3X3, nimg=6, Hi=24, Wi=20, Ho=24, Wo=20,
                                        rowbias=3 * 24 * 20, colstats=True)
    c["cs_temporal_K2"] = make_tapgemm(dt, 2 * 8 * 96, 640, 640, mode=L.TAP_TEMPORAL3, F=8, S=96, residual=True,
                                       colstats=True)
    c["cs_lin_wide_dual"] = make_tapgemm(dt, 9000, 1280, 128, colstats=True)
    c["temporal_b128"] = make_tapgemm(dt, 1 * 16 * 28, 128, 128, mode=L.TAP_TEMPORAL3, F=16, S=28)
    c["lin_154x4096x1024_textmlp"] = make_tapgemm(dt, 154, 4096, 1024)
    c["lin_ff2_split_out_640"] = make_tapgemm(dt, 3000, 640, 2560, out_dtype=dt, residual=True, split_out=True)
    c["lin_split_out_shortK_dual"] = make_tapgemm(dt, 9000, 320, 320, out_dtype=dt, residual=True, split_out=True)
    c.update(panel_cases(dt))
    return c


def r06_shape_cases(dt):
    """name -> (spec, (shape, bn, splitk)): launches forced (through the product ABI's plan table, vgen_tapgemm_set_plans) onto
    the two r06 shapes of csrc/tapgemm.hip — "pp256" (4: 256 x 256 x 32 ping-pong, 16-bit outputs) and "q128" (5: 4-wave
    128 x BN x 32, several blocks per CU) — over every epilogue / tap mode / tail they are legal for."""
    c = {}
    c["pp256_lin_out16_res"] = (make_tapgemm(dt, 1000, 512, 640, out_dtype=dt, residual=True, seed=21), (4, 256, 1))
    c["pp256_lin_out16_raggedM_3tiles"] = (make_tapgemm(dt, 777, 768, 320, out_dtype=dt, bias=False, seed=22), (4, 256, 1))
    c["pp256_geglu"] = (make_tapgemm(dt, 700, 1024, 320, epilogue=L.EPI_GEGLU, out_dtype=dt, seed=23), (4, 256, 1))
    c["pp256_geglu_res"] = (make_tapgemm(dt, 515, 512, 1280, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True, seed=24), (4, 256, 1))
    c["pp256_splitk2_out16"] = (make_tapgemm(dt, 300, 256, 2048, out_dtype=dt, seed=25), (4, 256, 2))
    c["pp256_geglu_splitk2"] = (make_tapgemm(dt, 260, 512, 1024, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True, seed=26), (4, 256, 2))
    c["pp256_temporal_out16"] = (make_tapgemm(dt, 2 * 5 * 24, 256, 128, mode=L.TAP_TEMPORAL3, F=5, S=24, out_dtype=dt, seed=27), (4, 256, 1))
    c["pp256_views_rowbias"] = (make_tapgemm(dt, 384, 256, 192, a_pad=64, w_pad=128, rowbias=96, out_dtype=dt, seed=28), (4, 256, 1))
    c["pp256_f32_res_splitk3_conv"] = (make_tapgemm(dt, 16 * 4 * 7, 1280, 1280, mode=L.TAP_CONV3X3, nimg=16, Hi=4, Wi=7, Ho=4, Wo=7,
                                                    stride=1, pad_t=1, pad_l=1, ups=0, residual=True, rowbias=28 * 8, seed=29), (4, 256, 3))
    c["q128_lin_res_b128"] = (make_tapgemm(dt, 896, 1280, 1280, residual=True, seed=31), (5, 128, 1))
    c["q128_lin_res_b128_sk2"] = (make_tapgemm(dt, 896, 1280, 1280, residual=True, seed=31), (5, 128, 2))
    c["q128_lin_b160_raggedM"] = (make_tapgemm(dt, 1000, 320, 640, residual=True, seed=32), (5, 160, 1))
    c["q128_lin_b64_out16"] = (make_tapgemm(dt, 300, 192, 256, out_dtype=dt, seed=33), (5, 64, 1))
    c["q128_temporal_res_sk3"] = (make_tapgemm(dt, 2 * 16 * 28, 1280, 1280, mode=L.TAP_TEMPORAL3, F=16, S=28, residual=True, seed=34), (5, 128, 3))
    c["q128_conv_rowbias_b160"] = (make_tapgemm(dt, 32 * 4 * 7, 320, 256, mode=L.TAP_CONV3X3, nimg=32, Hi=4, Wi=7, Ho=4, Wo=7,
                                                stride=1, pad_t=1, pad_l=1, ups=0, rowbias=16 * 28, seed=35), (5, 160, 1))
    c["q128_conv_s2_skipseg"] = (make_tapgemm(dt, 2 * 5 * 4, 128, 128, mode=L.TAP_CONV3X3, nimg=2, Hi=10, Wi=8, Ho=5, Wo=4,
                                              stride=2, pad_t=1, pad_l=1, ups=0, C2=192, seed=36), (5, 128, 1))
    c["q128_geglu_b128"] = (make_tapgemm(dt, 200, 512, 320, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True, seed=37), (5, 128, 1))
    c["q128_geglu_b64"] = (make_tapgemm(dt, 150, 320, 128, epilogue=L.EPI_GEGLU, out_dtype=dt, seed=38), (5, 64, 1))
    c["q128_colstats_b160"] = (make_tapgemm(dt, 9000, 320, 320, residual=True, colstats=True, seed=39), (5, 160, 1))
    c["q128_colstats_b128_rowbias"] = (make_tapgemm(dt, 6 * 24 * 20, 128, 64, mode=L.TAP_CONV3X3, nimg=6, Hi=24, Wi=20, Ho=24,
                                                    Wo=20, rowbias=3 * 24 * 20, colstats=True, seed=40), (5, 128, 1))
    c["q128_split_out"] = (make_tapgemm(dt, 3000, 640, 2560, out_dtype=dt, residual=True, split_out=True, seed=41), (5, 128, 1))
    return c


def plan_signature(g):
    """The 9 integers a plan-table row is keyed on (csrc/tapgemm.hip PlanEntry)."""
    from vgen_amd.ops import _ENUM
    flags = (1 if g.residual is not None else 0) | (2 if g.rowbias is not None else 0) | (4 if g.colstats else 0)
    return (g.mode, g.M, g.N, g.C1, g.C2, g.taps, g.epilogue, _ENUM[g.out_dtype], flags)


def panel_cases(dt, dualw=False):
    """r05: launches the W-panel-resident shape takes (csrc/panelgemm.hip: linear, K = 320, M >= 2048, N a multiple of the
    panel width) — every epilogue, ragged M (last slice partial, slices not a multiple of the 8 waves or of the row
    ranges), A / W / out / residual views with padded leading dimensions, one / several / many column panels."""
    px = "dw_panel_" if dualw else "panel_"
    c = {}
    c[px + "o_proj_res_f32"] = make_tapgemm(dt, 9000, 320, 320, residual=True, dualw=dualw)
    c[px + "f32_nores_raggedM"] = make_tapgemm(dt, 2049, 320, 320, dualw=dualw)
    c[px + "qkv_out16"] = make_tapgemm(dt, 5003, 960, 320, out_dtype=dt, bias=False, dualw=dualw)
    c[px + "q_out16_res"] = make_tapgemm(dt, 4100, 320, 320, out_dtype=dt, residual=True, dualw=dualw)
    c[px + "geglu"] = make_tapgemm(dt, 4099, 2560, 320, epilogue=L.EPI_GEGLU, out_dtype=dt, dualw=dualw)
    c[px + "geglu_res"] = make_tapgemm(dt, 2500, 640, 320, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True, dualw=dualw)
    c[px + "views"] = make_tapgemm(dt, 3000, 320, 320, a_pad=64, w_pad=0 if dualw else 128, residual=True, dualw=dualw)
    c[px + "wide_N"] = make_tapgemm(dt, 2100, 1600 if not dualw else 1280, 320, out_dtype=dt, dualw=dualw)
    if not dualw:
        # K = 640 (the 16 x 28 level): 80-column single-pass panels, 20 A chunks per slice
        c[px + "k640_o_proj_res_f32"] = make_tapgemm(dt, 4500, 640, 640, residual=True)
        c[px + "k640_qkv_out16"] = make_tapgemm(dt, 3001, 1920, 640, out_dtype=dt, bias=False)
        c[px + "k640_views"] = make_tapgemm(dt, 2050, 320, 640, a_pad=128, w_pad=64, out_dtype=dt, residual=True)
    return c


def tapgemm_dw_cases(dt):
    """Dual-W launches (vgen_tapgemm_args.dualw, the models' precision="high"): every row map, both K segments, the
    epilogues, column statistics, split-K, N tails of the three column tiles, 128- and 256-row blocks, K of one tile."""
    c = {}
    c["dw_lin_300x320x320"] = make_tapgemm(dt, 300, 320, 320, dualw=True)
    c["dw_lin_1000x256x640_res"] = make_tapgemm(dt, 1000, 256, 640, residual=True, dualw=True)
    c["dw_lin_K64"] = make_tapgemm(dt, 700, 128, 64, dualw=True)
    c["dw_lin_out16_nobias"] = make_tapgemm(dt, 77, 128, 1024, out_dtype=dt, bias=False, dualw=True)
    c["dw_lin_smallN"] = make_tapgemm(dt, 500, 4, 576, dualw=True)
    c["dw_lin_nonvec"] = make_tapgemm(dt, 130, 3, 128, residual=True, dualw=True)
    c["dw_lin_tinyM"] = make_tapgemm(dt, 2, 1280, 320, dualw=True)
    c["dw_lin_bigM_qkv"] = make_tapgemm(dt, 9000, 960, 320, out_dtype=dt, bias=False, dualw=True)
    c["dw_lin_rowbias"] = make_tapgemm(dt, 384, 128, 64, rowbias=96, residual=True, dualw=True)
    c["dw_lin_geglu"] = make_tapgemm(dt, 200, 512, 64, epilogue=L.EPI_GEGLU, out_dtype=dt, dualw=True)
    c["dw_lin_geglu_res"] = make_tapgemm(dt, 4100, 2560, 320, epilogue=L.EPI_GEGLU, out_dtype=dt, residual=True, dualw=True)
    c["dw_conv_s1_rowbias"] = make_tapgemm(dt, 3 * 9 * 7, 128, 64, mode=L.TAP_CONV3X3, nimg=3, Hi=9, Wi=7, Ho=9, Wo=7,
                                           stride=1, pad_t=1, pad_l=1, ups=0, rowbias=63, dualw=True)
    c["dw_conv_s2"] = make_tapgemm(dt, 2 * 5 * 4, 64, 128, mode=L.TAP_CONV3X3, nimg=2, Hi=10, Wi=8, Ho=5, Wo=4,
                                   stride=2, pad_t=1, pad_l=1, ups=0, dualw=True)
    c["dw_conv_ups_crop"] = make_tapgemm(dt, 2 * 10 * 10, 64, 64, mode=L.TAP_CONV3X3, nimg=2, Hi=6, Wi=5, Ho=10, Wo=10,
                                         stride=1, pad_t=1, pad_l=1, ups=1, crop_t=1, dualw=True)
    c["dw_conv_skipseg"] = make_tapgemm(dt, 2 * 6 * 6, 128, 128, mode=L.TAP_CONV3X3, nimg=2, Hi=6, Wi=6, Ho=6, Wo=6,
                                        stride=1, pad_t=1, pad_l=1, ups=0, C2=192, dualw=True)
    c["dw_conv_big_cs"] = make_tapgemm(dt, 4 * 32 * 56, 320, 320, mode=L.TAP_CONV3X3, nimg=4, Hi=32, Wi=56, Ho=32, Wo=56,
                                       stride=1, pad_t=1, pad_l=1, ups=0, residual=True, colstats=True, dualw=True)
    c["dw_conv_splitk_L3"] = make_tapgemm(dt, 16 * 4 * 7, 1280, 1280, mode=L.TAP_CONV3X3, nimg=16, Hi=4, Wi=7, Ho=4, Wo=7,
                                          stride=1, pad_t=1, pad_l=1, ups=0, residual=True, rowbias=28 * 8, dualw=True)
    c["dw_conv_wideK"] = make_tapgemm(dt, 2 * 8 * 14, 1280, 2560, mode=L.TAP_CONV3X3, nimg=2, Hi=8, Wi=14, Ho=8, Wo=14,
                                      stride=1, pad_t=1, pad_l=1, ups=0, C2=2560, dualw=True)      # K = 25600: r02's bound was 32704 / 2
    c["dw_temporal"] = make_tapgemm(dt, 2 * 5 * 24, 64, 64, mode=L.TAP_TEMPORAL3, F=5, S=24, residual=True, dualw=True)
    c["dw_temporal_cs"] = make_tapgemm(dt, 2 * 8 * 96, 640, 640, mode=L.TAP_TEMPORAL3, F=8, S=96, residual=True,
                                       colstats=True, dualw=True)
    c["dw_temporal_splitk"] = make_tapgemm(dt, 2 * 4 * 28, 640, 640, mode=L.TAP_TEMPORAL3, F=4, S=28, residual=True, dualw=True)
    c["dw_lin_splitk_out16"] = make_tapgemm(dt, 300, 256, 2048, out_dtype=dt, dualw=True)
    # two-term OUTPUT rows [hi | lo] (the FF output feeding proj_out): 160- and 128-wide column tiles, ragged M
    c["dw_ff2_split_out_320"] = make_tapgemm(dt, 4000, 320, 1280, out_dtype=dt, residual=True, split_out=True, dualw=True)
    c["dw_ff2_split_out_1280"] = make_tapgemm(dt, 900, 1280, 5120, out_dtype=dt, residual=True, split_out=True, dualw=True)
    c.update(panel_cases(dt, dualw=True))
    return c


def attn_cases(dt):
    c = {}
    c["spatial_200"] = make_attn(dt, "spatial", heads=2, nb=3, N=200)
    c["spatial_28"] = make_attn(dt, "spatial", heads=3, nb=2, N=28)
    c["spatial_448_peaky"] = make_attn(dt, "spatial", heads=1, nb=1, N=448, amp=3.0)
    c["cross_77"] = make_attn(dt, "cross", heads=2, B=2, F=3, N=130, Lc=77, off=256, kw=640)
    c["cross_1"] = make_attn(dt, "cross", heads=1, B=1, F=2, N=40, Lc=1, off=0, kw=128)
    c["temporal_16"] = make_attn(dt, "temporal", heads=2, B=2, F=16, S=30)
    c["temporal_4"] = make_attn(dt, "temporal", heads=1, B=1, F=4, S=9, amp=2.0)
    c["temporal_32_flash"] = make_attn(dt, "temporal", heads=1, B=1, F=32, S=6)
    # causal mask (CLIP text tower): 77 tokens (2 KV tiles, ragged), 200 (several Q tiles), 12 (<= 16: must not take
    # the maskless temporal kernel)
    c["causal_77"] = make_attn(dt, "spatial", heads=2, nb=3, N=77, causal=True)
    c["causal_200"] = make_attn(dt, "spatial", heads=1, nb=2, N=200, causal=True, amp=2.0)
    c["causal_12"] = make_attn(dt, "spatial", heads=1, nb=2, N=12, causal=True)
    return c
