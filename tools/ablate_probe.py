"""GPU probe: phase decomposition of tap-GEMM launches (VGEN_TAPGEMM_ABLATE): full, no K loop, no epilogue, neither."""
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

def spec(M, N, K, res=False, geglu=False, out16=False, mode=0, geom=None):
    A = torch.randn(M, K if mode == 0 else geom["C"], device=dev).to(dt)
    kw = {}
    if mode == 0:
        W = (torch.randn(N, K, device=dev) / K ** .5).to(dt)
        if res: kw["residual"] = torch.randn(M, N, device=dev)
        if geglu: kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
        elif out16: kw["out_dtype"] = dt
        return TapGemm(A=A, W=W, M=M, N=N, C1=K, bias=torch.randn(N, device=dev), **kw)
    C = geom["C"]
    W = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** .5).to(dt)
    return TapGemm(A=A, W=W, M=M, N=N, C1=C, mode=L.TAP_CONV3X3, taps=9, Hi=geom["H"], Wi=geom["W"], Ho=geom["H"], Wo=geom["W"],
                   bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev))

cases = {"qkv 57344x960x320 out16": spec(57344, 960, 320, out16=True),
         "geglu 57344x2560x320": spec(57344, 2560, 320, geglu=True),
         "oproj 57344x320x320 +res f32": spec(57344, 320, 320, res=True),
         "ff2 57344x320x1280 +res out16": TapGemm(A=torch.randn(57344, 1280, device=dev).to(dt), W=(torch.randn(320, 1280, device=dev) / 36).to(dt), M=57344, N=320, C1=1280, bias=torch.randn(320, device=dev), residual=torch.randn(57344, 320, device=dev), out_dtype=dt),
         "lin 14336x640x640 +res": spec(14336, 640, 640, res=True),
         "lin 3584x1280x1280 +res": spec(3584, 1280, 1280, res=True),
         "conv 57344x320x2880 +res": spec(57344, 320, 0, mode=1, geom=dict(C=320, H=32, W=56))}
# cold-ish: rotate through several copies of the streams so the Infinity Cache does not hold them
def bench(s, iters=30):
    for _ in range(3): be.tapgemm(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): be.tapgemm(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {}
for name, s in cases.items():
    row = {}
    for tag, v in (("full", 0), ("no_kloop", 1), ("no_epilogue", 2), ("neither", 3)):
        os.environ["VGEN_TAPGEMM_ABLATE"] = str(v)
        row[tag] = round(bench(s), 1)
    os.environ["VGEN_TAPGEMM_ABLATE"] = "0"
    pl = (__import__("ctypes").c_int32 * 3)()
    res[name] = row
    print(f"{name:36s}", row, flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ablation.json"), "w"), indent=1)
