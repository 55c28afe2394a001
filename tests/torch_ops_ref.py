"""Plain PyTorch fp32 references of the ops behind the C ABI, written with the framework's own operators
(F.conv2d / F.conv1d / F.linear / F.group_norm / F.layer_norm / F.scaled_dot_product_attention / F.gelu /
F.interpolate) on NCHW / [batch, heads, seq, 64] tensors — i.e. independent of the ABI's row layouts, tap order,
folded upsampling and packed GEGLU columns, which is exactly what they cross-check.  Used twice: CPU, against the
ABI emulator (oracle/abi_emulator.py) so the test double itself is pinned to torch's operators; GPU, against the
HIP kernels."""
import torch
import torch.nn.functional as Fn

from vgen_amd import lib as L


def ref_groupnorm(x1, x2, nb, S, groups, eps, gamma, beta, silu):
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    C = x.shape[1]
    y = Fn.group_norm(x.float().view(nb, S, C).transpose(1, 2), groups, gamma, beta, eps)   # [nb, C, S]
    if silu:
        y = Fn.silu(y)
    return y.transpose(1, 2).reshape(nb * S, C)


def ref_layernorm(x, gamma, beta, eps):
    return Fn.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps)


def ref_tapgemm(g, w32=None):
    """fp32 result of a TapGemm spec ([M, n_out]) from NCHW convolutions.  w32: use this fp32 weight instead of the
    16-bit operand (dual-W launches are checked against the unrounded weight)."""
    C1, N = g.C1, g.N
    A = g.A[:, :C1].float()
    W = g.W[:N].float() if w32 is None else w32[:N].float()
    if g.mode == L.TAP_LINEAR:
        acc = Fn.linear(A[: g.M], W[:, :C1])
    elif g.mode == L.TAP_CONV3X3:
        nimg = g.M // (g.Ho * g.Wo)
        x = A[: nimg * g.Hi * g.Wi].view(nimg, g.Hi, g.Wi, C1).permute(0, 3, 1, 2)
        if g.ups:
            x = Fn.interpolate(x, scale_factor=2, mode="nearest")
        if g.crop_t:
            x = x[:, :, g.crop_t: x.shape[2] - g.crop_t]
        w = W[:, : 9 * C1].view(N, 3, 3, C1).permute(0, 3, 1, 2)                 # [N, C, ky, kx]
        # explicit zero padding: (pad_l, pad_t) in front, whatever the output extent needs behind
        need_h = (g.Ho - 1) * g.stride + 3 - g.pad_t - x.shape[2]
        need_w = (g.Wo - 1) * g.stride + 3 - g.pad_l - x.shape[3]
        x = Fn.pad(x, (g.pad_l, max(need_w, 0), g.pad_t, max(need_h, 0)))
        y = Fn.conv2d(x, w, stride=g.stride)[:, :, : g.Ho, : g.Wo]
        acc = y.permute(0, 2, 3, 1).reshape(g.M, N)
    elif g.mode == L.TAP_TEMPORAL3:
        B = g.M // (g.F * g.S)
        x = A[: g.M].view(B, g.F, g.S, C1).permute(0, 2, 3, 1).reshape(B * g.S, C1, g.F)
        w = W[:, : 3 * C1].view(N, 3, C1).permute(0, 2, 1)                       # [N, C, kt]
        y = Fn.conv1d(x, w, padding=1)                                           # Conv3d (3,1,1), padding (1,0,0)
        acc = y.view(B, g.S, N, g.F).permute(0, 3, 1, 2).reshape(g.M, N)
    else:
        raise ValueError(g.mode)
    if g.C2:
        acc = acc + Fn.linear(g.A2[: g.M, : g.C2].float(), W[:, g.taps * C1: g.taps * C1 + g.C2])
    if g.bias is not None:
        acc = acc + g.bias[:N]
    if g.rowbias is not None:
        acc = acc + g.rowbias[torch.arange(g.M) // g.rows_per_rb][:, :N]
    if g.epilogue == L.EPI_GEGLU:
        # packed projection: 16 value columns, then their 16 gate columns (include/vgen_hip.h VGEN_EPI_GEGLU)
        v = acc.view(g.M, N // 32, 2, 16)
        acc = (v[:, :, 0] * Fn.gelu(v[:, :, 1])).reshape(g.M, N // 2)
    if g.residual is not None:
        acc = acc + g.residual[:, : acc.shape[1]]
    return acc


def ref_attention(g):
    """-> [no, ni, heads, nq, 64] fp32 and a function that scatters such a tensor into the spec's output layout."""
    no, ni = g.nbatch // g.inner, g.inner

    def seqs(t, s, n):
        rs, bo, bi = s
        return torch.as_strided(t, (no, ni, g.heads, n, 64), (bo, bi, 64, rs, 1), t.storage_offset()).float()

    q, k, v = seqs(g.q, g.q_s, g.nq), seqs(g.k, g.k_s, g.nk), seqs(g.v, g.v_s, g.nk)
    o = Fn.scaled_dot_product_attention(q, k, v, is_causal=bool(getattr(g, "causal", False)), scale=g.scale)
    rs, bo, bi = g.o_s
    got = torch.as_strided(g.out, (no, ni, g.heads, g.nq, 64), (bo, bi, 64, rs, 1), g.out.storage_offset())
    return o, got
