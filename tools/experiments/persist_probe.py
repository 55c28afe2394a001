# (needs tapgemm_persist.patch and a `record_plan` hook in HipBackend.tapgemm that stores vgen_tapgemm_query_plan in `last_plan`)
"""GPU probe: the resident tile loop (plan shape 3) against the other tap-GEMM plans on the hot launch signatures of
the t2v step.  Times each forced plan with back-to-back launches (HIP events on the launch stream)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vgen_amd import ops, lib as L
from vgen_amd.ops import TapGemm

dev = "cuda:0"; dt = torch.float16
be = ops.backend()
be.record_plan = True


def lin(M, N, K, res=False, geglu=False, out16=False):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** .5).to(dt)
    kw = {}
    n_out = N // 2 if geglu else N
    if res: kw["residual"] = torch.randn(M, n_out, device=dev)
    if geglu: kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
    elif out16: kw["out_dtype"] = dt
    return TapGemm(A=A, W=W, M=M, N=N, C1=K, bias=torch.randn(N, device=dev), **kw)


def temporal(M, N, C, F, S):
    A = torch.randn(M, C, device=dev).to(dt)
    W = (torch.randn(N, 3 * C, device=dev) / (3 * C) ** .5).to(dt)
    return TapGemm(A=A, W=W, M=M, N=N, C1=C, mode=L.TAP_TEMPORAL3, taps=3, F=F, S=S, bias=torch.randn(N, device=dev),
                   residual=torch.randn(M, N, device=dev))


cases = {
    "oproj 57344x320x320 +res f32": (lin(57344, 320, 320, res=True), [160]),
    "geglu 57344x2560x320": (lin(57344, 2560, 320, geglu=True), [128]),
    "qkv 57344x960x320 out16": (lin(57344, 960, 320, out16=True), [160, 128]),
    "ff2 57344x320x1280 +res out16": (lin(57344, 320, 1280, res=True, out16=True), [160]),
    "q 57344x320x320 out16": (lin(57344, 320, 320, out16=True), [160]),
    "lin 14336x640x640 +res f32": (lin(14336, 640, 640, res=True), [128, 160]),
    "geglu 14336x5120x640": (lin(14336, 5120, 640, geglu=True), [128]),
    "qkv 14336x1920x640 out16": (lin(14336, 1920, 640, out16=True), [128, 160]),
    "ff2 14336x640x2560 +res out16": (lin(14336, 640, 2560, res=True, out16=True), [128, 160]),
    "geglu 3584x10240x1280": (lin(3584, 10240, 1280, geglu=True), [128]),
    "qkv 3584x3840x1280 out16": (lin(3584, 3840, 1280, out16=True), [128, 160]),
    "temporal 57344x320x960 +res": (temporal(57344, 320, 320, 16, 1792), [160]),
    "temporal 14336x640x1920 +res": (temporal(14336, 640, 640, 16, 448), [128, 160]),
}


def bench(s, iters=40):
    for _ in range(3): be.tapgemm(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): be.tapgemm(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


res = {}
for name, (s, bns) in cases.items():
    row = {}
    os.environ.pop("VGEN_TAPGEMM_PLAN", None)
    os.environ["VGEN_TAPGEMM_PERSIST"] = "0"
    t = bench(s)
    row["model_plan"] = [list(be.last_plan), round(t, 1)]
    for shape in (0, 1, 3):
        for bn in bns:
            os.environ["VGEN_TAPGEMM_PLAN"] = f"{shape},{bn},1"
            t = bench(s)
            if tuple(be.last_plan) == (shape, bn, 1):
                row[f"{shape},{bn}"] = round(t, 1)
    os.environ.pop("VGEN_TAPGEMM_PLAN", None)
    res[name] = row
    print(f"{name:34s}", row, flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "persist_probe.json")
json.dump(res, open(out, "w"), indent=1)
