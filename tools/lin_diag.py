import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vgen_amd import ops
be = ops.backend()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for K, N in ((320, 1280), (1280, 1280), (1280, 20160)):
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    x1 = torch.randn(1, K, device=dev, generator=g)
    for act in (0, 1):
        ref = be.linear_f32(x1, W, b, act_in=act)
        for n in (2, 4, 8, 9, 16, 1000):
            xs = torch.randn(n, K, device=dev, generator=g)
            for pos in sorted({0, 1, min(3, n - 1), n - 1, min(601, n - 1)}):
                xx = xs.clone(); xx[pos] = x1[0]
                o = be.linear_f32(xx, W, b, act_in=act)
                if not torch.equal(o[pos], ref[0]):
                    print("MISMATCH", K, N, act, n, pos, float((o[pos] - ref[0]).abs().max()))
t = torch.arange(1000, dtype=torch.float32, device=dev)
s_all = be.timestep_embedding(t, 320, torch.float32)
s_one = be.timestep_embedding(torch.full((4,), 601.0, device=dev), 320, torch.float32)
print("sinusoid equal:", torch.equal(s_all[601], s_one[0]), float((s_all[601] - s_one[0]).abs().max()))
print("done")
