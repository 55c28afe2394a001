"""GPU probe: is the tap-GEMM K loop bound by the fabric (L2 misses -> MALL / HBM) or by the CU?  VGEN_TAPGEMM_ABLATE
bit 3 makes every block stage tile (0, 0)'s operands (all DMA traffic hits the L2); bit 1 drops the epilogue stores."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the phase-ablation / plan-override switches exist only in the TUNING build of the library (-DVGEN_TUNING,
# vgen_amd/build.py build(tuning=True) -> libvgen_hip_tuning.so; build it in the container before gpurun ships the tree)
os.environ.setdefault("VGEN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "libvgen_hip_tuning.so"))
import torch
from vgen_amd import ops, lib as L
from vgen_amd.ops import TapGemm

dev = "cuda:0"; dt = torch.float16
be = ops.backend()


def lin(M, N, K, res=False, geglu=False, out16=False):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** .5).to(dt)
    kw = {}
    if res: kw["residual"] = torch.randn(M, N // 2 if geglu else N, device=dev)
    if geglu: kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
    elif out16: kw["out_dtype"] = dt
    return TapGemm(A=A, W=W, M=M, N=N, C1=K, bias=torch.randn(N, device=dev), **kw)


def conv(nimg, H, Wd, C, N, res=True):
    M = nimg * H * Wd
    A = torch.randn(M, C, device=dev).to(dt)
    W = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** .5).to(dt)
    return TapGemm(A=A, W=W, M=M, N=N, C1=C, mode=L.TAP_CONV3X3, taps=9, Hi=H, Wi=Wd, Ho=H, Wo=Wd,
                   bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev) if res else None)


cases = {
    "conv 57344x320x2880": (conv(32, 32, 56, 320, 320), ["0,160,1"]),
    "conv 14336x640x5760": (conv(32, 16, 28, 640, 640), ["0,160,1", "0,128,1"]),
    "conv 3584x1280x11520": (conv(32, 8, 14, 1280, 1280), [None]),
    "qkv 57344x960x320 out16": (lin(57344, 960, 320, out16=True), ["0,160,1", "1,160,1"]),
    "geglu 57344x2560x320": (lin(57344, 2560, 320, geglu=True), ["0,128,1", "1,128,1"]),
    "ff2 57344x320x1280 +res out16": (lin(57344, 320, 1280, res=True, out16=True), ["0,160,1", "1,160,1"]),
    "lin 3584x1280x5120 out16": (lin(3584, 1280, 5120, out16=True), [None]),
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
for name, (s, plans) in cases.items():
    for plan in plans:
        if plan: os.environ["VGEN_TAPGEMM_PLAN"] = plan
        else: os.environ.pop("VGEN_TAPGEMM_PLAN", None)
        row = {}
        for tag, v in (("full", 0), ("l2hit", 8), ("nostore", 2), ("l2hit_nostore", 10)):
            os.environ["VGEN_TAPGEMM_ABLATE"] = str(v)
            row[tag] = round(bench(s), 1)
        os.environ["VGEN_TAPGEMM_ABLATE"] = "0"
        fl = 2.0 * s.M * s.N * (s.taps * s.C1)
        row["TF/s full"] = round(fl / row["full"] / 1e6, 0)
        row["TF/s l2hit_nostore"] = round(fl / row["l2hit_nostore"] / 1e6, 0)
        res[f"{name} [{plan or 'model plan'}]"] = row
        print(f"{name:34s} {plan or 'model':9s}", row, flush=True)
os.environ.pop("VGEN_TAPGEMM_PLAN", None)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "l2_probe.json")
json.dump(res, open(out, "w"), indent=1)
