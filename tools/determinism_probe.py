"""GPU probe: run the full-size t2v UNet forward several times on identical inputs; report bitwise
differences / non-finite outputs (a race or an uninitialised read shows up as run-to-run drift)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import UNET_T2V, randomize_
from vgen_amd import ops
from vgen_amd.unet import UNetSD_T2VBase

dev = torch.device("cuda", 0)
ops.set_backend(None)
with torch.device(dev):
    m = UNetSD_T2VBase(**UNET_T2V, compute_dtype="bf16")
m.eval(); randomize_(m, 0); m.pack()
g = torch.Generator(device=dev).manual_seed(8888)
x = torch.randn(2, 4, 16, 32, 56, generator=g, device=dev)
y = torch.randn(2, 77, 1024, generator=g, device=dev)
t = torch.tensor([981, 981], device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ref = None
bad = 0
for i in range(n):
    # poison the caching allocator's free blocks so that reads of never-written memory differ run to run
    junk = torch.full((64 * 1024 * 1024,), float("nan") if i % 2 else 1e30, device=dev); del junk
    out = m(x, t, y=y)
    torch.cuda.synchronize()
    fin = bool(torch.isfinite(out).all())
    if ref is None:
        ref = out.clone()
        print("run 0 finite", fin, "absmax", float(out.abs().max()))
    else:
        d = (out - ref)
        same = torch.equal(out, ref)
        bad += (not same) or (not fin)
        print("run", i, "finite", fin, "bitwise equal", same, "max|diff|", float(d.abs().nan_to_num(nan=1e38).max()))
print("RESULT", "DRIFT" if bad else "deterministic")
