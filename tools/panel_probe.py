"""GPU probe (r05): the W-panel-resident tap-GEMM shape (csrc/panelgemm.hip) against the streaming shapes on the level-0
launches of the t2v step — same process, same buffers, the TUNING library's switch VGEN_TAPGEMM_PANEL flipped between the
two (VGEN_HIP_LIB must point at libvgen_hip_tuning.so).  Per shape: both results compared with each other (they differ
by fp32 summation order only) and timed over a hot loop AND over a loop that walks ROT distinct buffer sets (so that the
operands of a launch do not sit in the 256 MiB Infinity Cache from the launch before, as they mostly do not in the model).

usage: VGEN_HIP_LIB=vgen_amd/libvgen_hip_tuning.so python tools/panel_probe.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vgen_amd import lib as L, ops  # noqa: E402
from vgen_amd.ops import TapGemm, split_weight  # noqa: E402

DEV = "cuda:0"
ROT = 6


def make(dt, M, N, K, res=False, out16=False, geglu=False, dw=False, seed=0):
    g = torch.Generator("cpu").manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w32 = torch.randn(N, K, generator=g) / K ** 0.5
    W = split_weight(w32.to(DEV), dt) if dw else w32.to(dt).to(DEV)
    kw = dict(bias=torch.randn(N, generator=g).to(DEV))
    if geglu:
        kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
    elif out16:
        kw["out_dtype"] = dt
    if res:
        kw["residual"] = torch.randn(M, N // 2 if geglu else N, generator=g).to(DEV)
    return TapGemm(A=A, W=W, M=M, N=N, C1=K, **kw)


def clone(spec, seed):
    import dataclasses
    g = torch.Generator("cpu").manual_seed(1000 + seed)
    kw = {f.name: getattr(spec, f.name) for f in dataclasses.fields(spec)}
    kw["A"] = torch.randn(spec.A.shape, generator=g).to(spec.A.dtype).to(DEV)
    if spec.residual is not None:
        kw["residual"] = torch.randn(spec.residual.shape, generator=g).to(DEV)
    return TapGemm(**kw)


def timeit(specs, outs, iters):
    be = ops.backend()
    for i in range(len(specs)):
        specs[i].out = outs[i]
        be.tapgemm(specs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        be.tapgemm(specs[i % len(specs)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dt = torch.float16
    M = 57344
    shapes = [
        ("o-proj 57344x320x320 +res f32", dict(N=320, res=True)),
        ("o-proj 57344x320x320 +res f32 dw", dict(N=320, res=True, dw=True)),
        ("proj_in 57344x320x320 f32", dict(N=320)),
        ("proj_in 57344x320x320 f32 dw", dict(N=320, dw=True)),
        ("q 57344x320x320 out16", dict(N=320, out16=True)),
        ("qkv 57344x960x320 out16", dict(N=960, out16=True)),
        ("qkv 57344x960x320 out16 dw", dict(N=960, out16=True, dw=True)),
        ("geglu 57344x2560x320", dict(N=2560, geglu=True)),
        ("geglu 57344x2560x320 dw", dict(N=2560, geglu=True, dw=True)),
        ("L1 qkv 14336x1920x640 out16", dict(N=1920, out16=True, M=14336, K=640)),
        ("L1 o-proj 14336x640x640 +res f32", dict(N=640, res=True, M=14336, K=640)),
        ("L1 q 14336x640x640 out16", dict(N=640, out16=True, M=14336, K=640)),
        ("i2vgen q 450560x320x320 out16", dict(N=320, out16=True, M=450560)),
        ("i2vgen qkv 450560x960x320 out16 dw", dict(N=960, out16=True, dw=True, M=450560)),
    ]
    rows = []
    be = ops.backend()
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
    if only:
        shapes = [sh for sh in shapes if any(o in sh[0] for o in only)]
    for name, s in shapes:
        s = dict(s)
        m = s.pop("M", M)
        kdim = s.pop("K", 320)
        rot = ROT if m <= M else 2
        specs = [make(dt, m, K=kdim, **s)]
        specs += [clone(specs[0], i) for i in range(rot - 1)]
        n_out = specs[0].N // 2 if specs[0].epilogue == L.EPI_GEGLU else specs[0].N
        outs = [torch.empty((m, n_out), dtype=specs[0].out_dtype, device=DEV) for _ in range(rot)]
        res = {"shape": name}
        ref = {}
        for mode in (("1",) if "--panel-only" in sys.argv else ("0", "1")):
            os.environ["VGEN_TAPGEMM_PANEL"] = mode
            pl = "panel" if mode == "1" else "stream"
            res[pl + "_hot_us"] = round(timeit(specs[:1], outs[:1], 30), 2)
            res[pl + "_rot_us"] = round(timeit(specs, outs, 5 * rot), 2)
            specs[0].out = None
            ref[pl] = be.tapgemm(specs[0]).float()
        fl = 2.0 * m * specs[0].N * kdim
        res["panel_TFLOPs_rot"] = round(fl / res["panel_rot_us"] / 1e6, 1)
        res["finite"] = bool(torch.isfinite(ref["panel"]).all())
        if "stream" in ref:
            d = (ref["panel"] - ref["stream"]).norm() / ref["stream"].norm()
            res["panel_vs_stream_rel_l2"] = float(d)
            res["stream_TFLOPs_rot"] = round(fl / res["stream_rot_us"] / 1e6, 1)
        if "--stagger-scan" in sys.argv:
            # head start of waves 0-3 over their SIMD partners (x 64 cycles; tuning build: VGEN_PANEL_STAGGER)
            os.environ["VGEN_TAPGEMM_PANEL"] = "1"
            scan = {}
            for sg in (0, 16, 32, 48, 64, 96, 128):
                os.environ["VGEN_PANEL_STAGGER"] = str(sg)
                scan[sg] = [round(timeit(specs[:1], outs[:1], 30), 2), round(timeit(specs, outs, 5 * rot), 2)]
            os.environ.pop("VGEN_PANEL_STAGGER")
            res["stagger_scan_hot_rot_us"] = scan
        if "--stamps" in sys.argv:
            # per-wave s_memtime sums per segment of the slice loop (tuning build, csrc/panelgemm.hip PANEL_STAMP)
            os.environ["VGEN_TAPGEMM_PANEL"] = "1"
            st = torch.zeros(4096 * 8, dtype=torch.int64, device=DEV)
            specs[0].out, specs[0].ws = outs[0], st
            be.tapgemm(specs[0])
            torch.cuda.synchronize()
            specs[0].ws = None
            v = st.view(-1, 8).cpu().double()
            v = v[v[:, 5] > 0]
            names = ["residual_issue", "wait_all_landed", "mfma_loop", "next_A_issue", "epilogue"]
            per_slice = (v[:, :5].sum(0) / v[:, 5].sum()).tolist()
            res["stamp_cycles_per_slice"] = {n: round(c) for n, c in zip(names, per_slice)}
            res["stamp_slices_per_wave_max"] = int(v[:, 5].max())
            res["stamp_wave_total_cycles_max_mean"] = [round(float(v[:, :5].sum(1).max())), round(float(v[:, :5].sum(1).mean()))]
        rows.append(res)
        print(json.dumps(res), flush=True)
        del specs, outs, ref
        torch.cuda.empty_cache()
    outp = [a for a in sys.argv[1:] if not a.startswith("--")]
    if outp:
        json.dump(rows, open(outp[0], "w"), indent=1)


if __name__ == "__main__":
    main()
