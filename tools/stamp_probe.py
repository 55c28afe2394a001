"""GPU probe: where a ping-pong tap-GEMM wave spends its cycles (s_memtime segment sums of the -DVGEN_STAMP build).

    python -m vgen_amd.build --variant=stamp -DVGEN_STAMP          (in the container; the .so travels with gpurun)
    VGEN_HIP_LIB=vgen_amd/libvgen_hip_stamp.so python tools/stamp_probe.py [tag]

Per launch: the mean over the first 64 blocks of the per-K-step cycles of each segment, leaders (waves 0-3) and followers
(waves 4-7, one phase behind) apart.  Segments of a K-step: R = read phase issued and its fragments landed (18 ds_read_b128
+ the DMA pieces of tile t+2), V = counted vmcnt wait for tile t+1, B1 = barrier before the matrix phase, M = matrix phase
issued (40 MFMAs), B2 = DMA pointer step + barrier after it.  Dual-W launches report the even and the odd step of a pair."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGEN_HIP_LIB", os.path.join(ROOT, "vgen_amd", "libvgen_hip_stamp.so"))
import torch
from vgen_amd import ops, lib as L
from vgen_amd.ops import TapGemm

dev = "cuda:0"; dt = torch.float16
be = ops.backend()
lib = L.load()
lib.vgen_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.vgen_debug_stamps.restype = ctypes.c_int
SLOTS, BLOCKS = 12, 64


def lin(M, N, K, res=False, geglu=False, out16=False, dw=False):
    A = torch.randn(M, K, device=dev).to(dt)
    w32 = torch.randn(N, K, device=dev) / K ** .5
    W = ops.split_weight(w32, dt) if dw else w32.to(dt)
    kw = {}
    if res: kw["residual"] = torch.randn(M, N, device=dev)
    if geglu: kw.update(epilogue=L.EPI_GEGLU, out_dtype=dt)
    elif out16: kw["out_dtype"] = dt
    return TapGemm(A=A, W=W, M=M, N=N, C1=K, bias=torch.randn(N, device=dev), **kw)


def conv(M, N, C, H, Wd, dw=False):
    A = torch.randn(M, C, device=dev).to(dt)
    w32 = torch.randn(N, 9 * C, device=dev) / (9 * C) ** .5
    W = ops.split_weight(w32, dt) if dw else w32.to(dt)
    return TapGemm(A=A, W=W, M=M, N=N, C1=C, mode=L.TAP_CONV3X3, taps=9, Hi=H, Wi=Wd, Ho=H, Wo=Wd,
                   bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev))


cases = {
    "conv 57344x320x2880 +res": conv(57344, 320, 320, 32, 56),
    "conv 57344x320x5760 +res": conv(57344, 320, 640, 32, 56),
    "conv 57344x320x2880 +res dual-W": conv(57344, 320, 320, 32, 56, dw=True),
    "conv 14336x640x5760 +res": conv(14336, 640, 640, 16, 28),
    "ff2 57344x320x1280 +res": lin(57344, 320, 1280, res=True),
    "geglu 57344x2560x320": lin(57344, 2560, 320, geglu=True),
    "geglu 57344x2560x320 dual-W": lin(57344, 2560, 320, geglu=True, dw=True),
    "qkv 57344x960x320 out16": lin(57344, 960, 320, out16=True),
}
NAMES = ["R", "V", "B1", "M", "B2"]


def timed(s, iters=20):
    for _ in range(3): be.tapgemm(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): be.tapgemm(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def stamps():
    buf = torch.zeros(BLOCKS * 8 * SLOTS, dtype=torch.int64, device=dev)
    rc = lib.vgen_debug_stamps(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return buf.cpu().to(torch.float64).view(BLOCKS, 8, SLOTS)


out = {}
prev = None
for name, s in cases.items():
    us = timed(s)
    torch.cuda.synchronize()
    st = stamps()
    if prev is not None and torch.equal(st, prev):            # not a ping-pong launch (dual shape): nothing was stored
        out[name] = {"us": round(us, 1), "note": "not a ping-pong launch"}
        print(f"{name:36s}", json.dumps(out[name]), flush=True)
        continue
    prev = st
    nk = st[:, :, 10]
    live = nk[:, 0] > 0
    if not bool(live.any()):
        out[name] = {"us": round(us, 1), "note": "no counters stored", "nonzero": int((st != 0).sum().item())}
        print(f"{name:36s}", json.dumps(out[name]), flush=True)
        continue
    st = st[live]
    row = {"us": round(us, 1), "k_steps": int(st[0, 0, 10].item()), "blocks": int(live.sum().item())}
    pairs = bool((st[:, :, 5:10].sum() > 0).item())
    steps = st[:, :, 10] / (2 if pairs else 1)                 # K-steps (pairs for dual-W) per wave
    for role, sl in (("leaders", slice(0, 4)), ("followers", slice(4, 8))):
        seg = {}
        for i, n in enumerate(NAMES):
            seg[n] = round((st[:, sl, i] / steps[:, sl]).mean().item(), 1)
            if pairs:
                seg[n + "_odd"] = round((st[:, sl, 5 + i] / steps[:, sl]).mean().item(), 1)
        seg["step" if not pairs else "pair"] = round((st[:, sl, 11] / steps[:, sl]).mean().item(), 1)
        row[role] = seg
    out[name] = row
    print(f"{name:36s}", json.dumps(row), flush=True)
tag = sys.argv[1] if len(sys.argv) > 1 else "stamp"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"stamps_{tag}.json"), "w"), indent=1)
