"""GPU probe: time vgen_tapgemm on a few shapes (TFLOP/s), for kernel tuning."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vgen_amd import ops, lib as L
from vgen_amd.ops import TapGemm

def bench(spec, iters=20):
    be = ops.backend()
    for _ in range(3): be.tapgemm(spec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): be.tapgemm(spec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * spec.M * spec.N * (spec.taps * spec.C1 + spec.C2)
    return ms, fl / ms / 1e9

def main():
    dev = "cuda:0"; dt = torch.bfloat16
    shapes = [("lin 4096^3", dict(M=4096, N=4096, K=4096)),
              ("lin 8192x4096x4096", dict(M=8192, N=4096, K=4096)),
              ("lin 57344x320x320 f32out+res", dict(M=57344, N=320, K=320, res=True)),
              ("lin 57344x2560x320 geglu", dict(M=57344, N=2560, K=320, geglu=True)),
              ("lin 14336x5120x640 geglu", dict(M=14336, N=5120, K=640, geglu=True)),
              ("lin 3584x1280x1280 res", dict(M=3584, N=1280, K=1280, res=True)),
              ("lin 14336x640x640 res", dict(M=14336, N=640, K=640, res=True)),
              ("lin 57344x960x320 qkv", dict(M=57344, N=960, K=320)),
              ("lin 57344x320x1280 ff2", dict(M=57344, N=320, K=1280, res=True)),
              ("lin 3584x10240x1280 geglu", dict(M=3584, N=10240, K=1280, geglu=True)),
              ("lin 3584x3840x1280 qkv", dict(M=3584, N=3840, K=1280)),
              ("lin 3584x1280x5120 ff2", dict(M=3584, N=1280, K=5120, res=True)),
              ("lin 896x1280x1280 res", dict(M=896, N=1280, K=1280, res=True)),
              ("tconv 3584x1280x3840", dict(temporal=(2, 16, 112), C=1280)),
              ("tconv 896x1280x3840", dict(temporal=(2, 16, 28), C=1280)),
              ("conv 896x1280x(9*1280)", dict(conv=(32, 4, 7), C=1280, N=1280)),
              ("conv 57344x320x(9*320)", dict(conv=(32, 32, 56), C=320, N=320)),
              ("conv 14336x640x(9*640)", dict(conv=(32, 16, 28), C=640, N=640)),
              ("conv 3584x1280x(9*1280)", dict(conv=(32, 8, 14), C=1280, N=1280)),
              ("conv 57344x640x(9*640)", dict(conv=(32, 32, 56), C=640, N=640)),
              ("tconv 57344x320x960", dict(temporal=(2, 16, 1792), C=320)),
              ("tconv 14336x640x1920", dict(temporal=(2, 16, 448), C=640)),
              ]
    only = sys.argv[1:] 
    for name, s in shapes:
        if only and not any(o in name for o in only): continue
        if "conv" in s:
            nimg, H, W = s["conv"]; C = s["C"]; N = s["N"]
            A = torch.randn(nimg * H * W, C, device=dev).to(dt); Wt = (torch.randn(N, 9 * C, device=dev) / (9*C) ** .5).to(dt)
            spec = TapGemm(A=A, W=Wt, M=nimg * H * W, N=N, C1=C, mode=L.TAP_CONV3X3, taps=9, Hi=H, Wi=W, Ho=H, Wo=W,
                           bias=torch.randn(N, device=dev))
        elif "temporal" in s:
            B, F, S = s["temporal"]; C = s["C"]
            A = torch.randn(B * F * S, C, device=dev).to(dt); Wt = (torch.randn(C, 3 * C, device=dev) / (3*C) ** .5).to(dt)
            spec = TapGemm(A=A, W=Wt, M=B * F * S, N=C, C1=C, mode=L.TAP_TEMPORAL3, taps=3, F=F, S=S, bias=torch.randn(C, device=dev))
        else:
            M, N, K = s["M"], s["N"], s["K"]
            A = torch.randn(M, K, device=dev).to(dt); Wt = (torch.randn(N, K, device=dev) / K ** .5).to(dt)
            kw = {}
            if s.get("res"): kw["residual"] = torch.randn(M, N, device=dev)
            if s.get("geglu"): kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
            elif not s.get("res"): kw["out_dtype"] = dt
            spec = TapGemm(A=A, W=Wt, M=M, N=N, C1=K, bias=torch.randn(N, device=dev), **kw)
        ms, tf = bench(spec)
        print(f"{name:40s} {ms*1e3:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
