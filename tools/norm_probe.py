"""GPU probe: GroupNorm / LayerNorm / attention kernels on the UNet's main shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vgen_amd import ops
from vgen_amd.ops import Attn
be = ops.backend(); dev = "cuda:0"; dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nb, S, C in [(2, 28672, 320), (32, 1792, 320), (2, 7168, 640), (2, 1792, 1280), (2, 448, 1280), (32, 28, 1280), (32, 448, 640), (32, 112, 1280), (32, 112, 2560), (32, 28, 2560)]:
    x = torch.randn(nb * S, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    us = timeit(lambda: be.groupnorm(x, None, nb, S, 32, 1e-5, g, b, True, False, dt))
    print(f"gn nb={nb} S={S} C={C}: {us:.1f} us  {nb*S*C*10/us/1e3:.0f} GB/s")
for M, d in [(57344, 320), (14336, 640), (3584, 1280)]:
    x = torch.randn(M, d, device=dev); g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
    us = timeit(lambda: be.layernorm(x, g, b, 1e-5, dt))
    print(f"ln M={M} d={d}: {us:.1f} us  {M*d*6/us/1e3:.0f} GB/s")
for nb, heads, N in [(32, 5, 1792), (32, 10, 448)]:
    d = heads * 64
    qkv = torch.randn(nb * N, 3 * d, device=dev).to(dt); out = torch.empty(nb * N, d, device=dev, dtype=dt)
    ld = 3 * d
    a = Attn(q=qkv, k=qkv[:, d:], v=qkv[:, 2 * d:], out=out, heads=heads, nq=N, nk=N, nbatch=nb, inner=1,
             q_s=(ld, N * ld, 0), k_s=(ld, N * ld, 0), v_s=(ld, N * ld, 0), o_s=(d, N * d, 0), scale=0.125)
    us = timeit(lambda: be.attention(a))
    print(f"attn nb={nb} h={heads} N={N}: {us:.1f} us  {4*nb*heads*N*N*64/us/1e6:.0f} TFLOP/s")
