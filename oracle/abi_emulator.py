"""TEST INFRASTRUCTURE — CPU emulation of the libvgen_hip.so C ABI (include/vgen_hip.h).

Implements the same op set as vgen_amd.ops.HipBackend in plain torch so the test-suite can run
the HOST logic (weight packing, layouts, strides, call order of vgen_amd.unet / vae / diffusion)
on CPU against the reference.  Each method follows the semantics written in the header — row
gathers, packed weight layouts, strided sequences — not the structure of the reference model.
16-bit rounding points are reproduced (operands are rounded to `dt`, accumulation in fp32).

Never imported by product code: tests install it with `vgen_amd.ops.set_backend(EmuBackend())`.
"""
from __future__ import annotations

import math

import torch

from vgen_amd import lib as L


def _strided(t, sizes, strides):
    return torch.as_strided(t, sizes, strides, t.storage_offset())


class EmuBackend:
    name = "emu"

    # -- norms ---------------------------------------------------------------------------------
    def groupnorm(self, x1, x2, nb, S, groups, eps, gamma, beta, silu, want_raw, dt):
        from vgen_amd.ops import CS_ROWS, colstats_of
        x = x1 if x2 is None else torch.cat([x1, x2], 1)
        C = x.shape[1]
        assert C % 4 == 0 and C % groups == 0 and x1.shape[1] % 4 == 0
        v = x.view(nb, S, groups, C // groups).float()
        cs1 = colstats_of(x1, nb * S)
        cs2 = colstats_of(x2, nb * S) if x2 is not None else None
        if S % CS_ROWS == 0 and cs1 is not None and (x2 is None or cs2 is not None):
            # vgen_groupnorm_cs: statistics come from the producers' column partials, not from x — a stale
            # or mis-plumbed partial shows up as a parity failure of the host-logic tests
            cs = cs1 if cs2 is None else torch.cat([cs1, cs2], 2)
            t = cs.double().view(nb, S // CS_ROWS, 2, groups, C // groups).sum(dim=(1, 4))   # [nb, 2, groups]
            n = S * (C // groups)
            mean64 = t[:, 0] / n
            var64 = (t[:, 1] / n - mean64 * mean64).clamp_min(0)
            mean = mean64.float().view(nb, 1, groups, 1)
            var = var64.float().view(nb, 1, groups, 1)
        else:
            mean = v.mean(dim=(1, 3), keepdim=True)
            var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((v - mean) / torch.sqrt(var + eps)).view(nb * S, C) * gamma + beta
        if silu:
            y = y * torch.sigmoid(y)
        raw = None
        if want_raw == "split":                          # two-term rows [hi | lo] (vgen_groupnorm raw_split)
            hi = x.to(dt)
            raw = torch.cat([hi, (x.float() - hi.float()).to(dt)], 1)
        elif want_raw:
            raw = x.to(dt)
        return y.to(dt), raw

    def layernorm(self, x, gamma, beta, eps, dt):
        mean = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        return (((x - mean) / torch.sqrt(var + eps)) * gamma + beta).to(dt)

    # -- tap GEMM --------------------------------------------------------------------------------
    @staticmethod
    def _src_rows(g):
        m = torch.arange(g.M)
        rows = []
        if g.mode == L.TAP_LINEAR:
            rows.append(m)
        elif g.mode == L.TAP_CONV3X3:
            hw = g.Ho * g.Wo
            img, rem = m // hw, m % hw
            oy, ox = rem // g.Wo, rem % g.Wo
            Hv, Wv = (g.Hi << g.ups) - 2 * g.crop_t, g.Wi << g.ups
            for tap in range(9):
                iy = oy * g.stride + tap // 3 - g.pad_t
                ix = ox * g.stride + tap % 3 - g.pad_l
                ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv)
                r = img * g.Hi * g.Wi + ((iy + g.crop_t) >> g.ups) * g.Wi + (ix >> g.ups)
                rows.append(torch.where(ok, r, torch.full_like(r, -1)))
        elif g.mode == L.TAP_TEMPORAL3:
            f = (m // g.S) % g.F
            for tap in range(3):
                f2 = f + tap - 1
                ok = (f2 >= 0) & (f2 < g.F)
                rows.append(torch.where(ok, m + (tap - 1) * g.S, torch.full_like(m, -1)))
        else:
            raise ValueError(g.mode)
        return rows

    def tapgemm(self, g):
        assert g.C1 % 64 == 0 and g.C2 % 64 == 0
        assert g.A.stride(1) == 1 and g.A.stride(0) % 8 == 0
        dt = g.A.dtype
        assert dt in (torch.bfloat16, torch.float16) and g.W.dtype == dt
        K = g.taps * g.C1 + g.C2
        dw = getattr(g.W, "vgen_dw", None)
        if dw is not None:                                # dual-W launch (vgen_tapgemm_args.dualw): A (W_hi + W_lo)^T
            from vgen_amd.ops import dw_terms
            assert dw.dtype == dt and dw.shape[1] == 2 * K
            hi, lo = dw_terms(dw[: g.N])
            W = hi.float() + lo.float()
        else:
            W = g.W[: g.N, :K].float()
        acc = torch.zeros((g.M, g.N), dtype=torch.float32)
        for tap, r in enumerate(self._src_rows(g)):
            a = g.A[:, : g.C1][r.clamp(min=0)].float()
            a = torch.where((r >= 0)[:, None], a, torch.zeros_like(a))
            acc += a @ W[:, tap * g.C1:(tap + 1) * g.C1].t()
        if g.C2:
            assert g.A2.dtype == dt
            acc += g.A2[: g.M, : g.C2].float() @ W[:, g.taps * g.C1:].t()
        if g.bias is not None:
            acc += g.bias[: g.N]
        if g.rowbias is not None:
            idx = torch.arange(g.M) // g.rows_per_rb
            acc += g.rowbias[idx][:, : g.N]
        if g.epilogue == L.EPI_GEGLU:
            assert g.N % 64 == 0 and g.rowbias is None
            v = acc.view(g.M, g.N // 32, 2, 16)
            val, gate = v[:, :, 0], v[:, :, 1]
            acc = (val * (0.5 * gate * (1.0 + torch.erf(gate * 0.7071067811865476)))).reshape(g.M, g.N // 2)
            n_out = g.N // 2
        else:
            n_out = g.N
        if g.residual is not None:
            acc += g.residual[:, :n_out]
        out = g.out
        split = bool(getattr(g, "split_out", False))
        if out is None:
            out = torch.empty((g.M, 2 * n_out if split else n_out), dtype=g.out_dtype)
        assert out.dtype == g.out_dtype and out.dtype in (torch.float32, dt)
        out[:, :n_out] = acc.to(g.out_dtype)
        if split:                                           # two-term rows: [round16(v) | round16(v - hi)]
            assert g.out_dtype == dt and g.epilogue == L.EPI_NONE and not g.colstats
            out[:, n_out: 2 * n_out] = (acc - out[:, :n_out].float()).to(dt)
        if g.colstats:
            assert g.out_dtype == torch.float32 and g.epilogue == L.EPI_NONE and g.N % 4 == 0
            ns = (g.M + 63) // 64
            pad = torch.zeros((ns * 64, g.N), dtype=torch.float32)
            pad[: g.M] = acc
            pv = pad.view(ns, 64, g.N)
            out.vgen_cs = torch.stack([pv.sum(1), (pv * pv).sum(1)], 1).contiguous()
        return out

    # -- attention -------------------------------------------------------------------------------
    def attention(self, g):
        assert g.nbatch % g.inner == 0
        no, ni = g.nbatch // g.inner, g.inner

        def seqs(t, s, n):
            rs, bo, bi = s
            return _strided(t, (no, ni, g.heads, n, 64), (bo, bi, 64, rs, 1)).float()

        q, k, v = seqs(g.q, g.q_s, g.nq), seqs(g.k, g.k_s, g.nk), seqs(g.v, g.v_s, g.nk)
        rs, bo, bi = g.o_s
        ov = _strided(g.out, (no, ni, g.heads, g.nq, 64), (bo, bi, 64, rs, 1))
        # sequences are independent: evaluate the outer batch in chunks of <= ~2 GB of fp32 scores (the 14 400-token
        # self-attention of a 32-frame 720p latent is 133 GB at once)
        per = ni * g.heads * g.nq * g.nk * 4
        step = max(1, int(2e9 // max(per, 1)))
        for a in range(0, no, step):
            sc = q[a:a + step] @ k[a:a + step].transpose(-1, -2) * g.scale
            if getattr(g, "causal", False):
                sc = sc.masked_fill(torch.ones(g.nq, g.nk, dtype=torch.bool).triu(1), float("-inf"))
            ov[a:a + step].copy_((torch.softmax(sc, dim=-1) @ v[a:a + step]).to(g.out.dtype))
        return g.out

    def softmax_rows(self, S, cols, scale, dt, out=None):
        p = torch.softmax(S[:, :cols].float() * scale, dim=-1).to(dt)
        if out is None:
            return p
        out[:, :cols] = p
        return out

    # -- small kernels ---------------------------------------------------------------------------
    def act_cast(self, x, act, dt):
        if act == 2:
            return (0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))).to(dt)
        return (x * torch.sigmoid(x) if act == 1 else x).to(dt)

    def cast_split(self, x, dt, out=None, col=0, lo_off=None):
        M, Cc = x.shape
        lo_off = Cc if lo_off is None else lo_off
        if out is None:
            out = torch.zeros((M, col + lo_off + Cc), dtype=dt)
        hi = x.to(dt)
        out[:, col: col + Cc] = hi
        out[:, col + lo_off: col + lo_off + Cc] = (x.float() - hi.float()).to(dt)
        return out

    def conv3x3_small(self, x, w, b, stride=1, act=0):
        y = torch.nn.functional.conv2d(x, w, b, stride=stride, padding=1)
        return y * torch.sigmoid(y) if act == 1 else y

    def adaptive_avgpool2d(self, x, Ho, Wo):
        return torch.nn.functional.adaptive_avg_pool2d(x, (Ho, Wo))

    def frame_transformer(self, x, B, F, d, HW, p, out=None, last=False, out_scale=1.0, accumulate=False):
        Fn = torch.nn.functional
        t = x.reshape(B, F, d, HW).permute(0, 3, 1, 2)                      # [B, HW, F, d]
        n = Fn.layer_norm(t, (d,), p["ln_w"], p["ln_b"], 1e-5)
        h, dh = p["heads"], p["dim_head"]
        qkv = (n @ p["wqkv"].t()).view(B, HW, F, 3, h, dh).permute(3, 0, 1, 4, 2, 5)
        w = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * dh ** -0.5, dim=-1)
        o = (w @ qkv[2]).permute(0, 1, 3, 2, 4).reshape(B, HW, F, h * dh)
        if p["wout"] is not None:
            o = o @ p["wout"].t() + p["bout"]
        t = t + o
        t = Fn.linear(Fn.gelu(Fn.linear(t, p["w1"], p["b1"])), p["w2"], p["b2"]) + t
        if not last:
            r = t.permute(0, 2, 3, 1).reshape(x.shape)                         # back to frames [B*F, d, H, W]
            if out is None:
                return r.contiguous()
            out.view_as(r).copy_(r)
            return out
        r = (t.permute(0, 3, 2, 1) * out_scale).reshape(-1)                    # [B, d, F, HW]
        if out is None:
            return r.contiguous()
        if accumulate:
            out.view(-1).add_(r)
        else:
            out.view(-1).copy_(r)
        return out

    def embed_tokens(self, tokens, table, pos):
        B, Lk = tokens.shape
        return (table[tokens.clamp(0, table.shape[0] - 1)] + pos[None]).reshape(B * Lk, -1).contiguous()

    def linear_f32(self, x, W, b, act_in=0, add=None):
        # float64 accumulation, rounded once: the result of a row does not depend on how many rows are evaluated
        # together (the kernel's fixed per-element summation order has the same property)
        v = x.double()
        if act_in == 1:
            v = v * torch.sigmoid(v)
        o = v.float().double() @ W.double().t()
        if b is not None:
            o = o + b.double()
        o = o.float()
        return o if add is None else o + add

    def timestep_embedding(self, t, dim, dt):
        half = dim // 2
        w = torch.pow(torch.tensor(10000.0), -(torch.arange(half).float() / half))
        a = t[:, None] * w[None, :]
        e = torch.cat([torch.cos(a), torch.sin(a)], 1)
        if dim % 2:
            e = torch.cat([e, torch.zeros_like(e[:, :1])], 1)
        return e.to(dt)

    @staticmethod
    def _view5(t, nimg, Fi, C, H, W, s):
        s_bo, s_fi, s_c, s_y, s_x = s
        return _strided(t, (nimg // Fi, Fi, C, H, W), (s_bo, s_fi, s_c, s_y, s_x))

    def im2col3x3_small(self, src, nimg, Fi, Cin, H, W, strides, Kpad, dt, split=False):
        x = self._view5(src, nimg, Fi, Cin, H, W, strides).reshape(nimg, Cin, H, W)
        xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
        col = torch.zeros((nimg, H, W, 9 * Cin), dtype=torch.float32)
        for tap in range(9):
            ky, kx = tap // 3, tap % 3
            col[..., tap * Cin:(tap + 1) * Cin] = xp[:, :, ky:ky + H, kx:kx + W].permute(0, 2, 3, 1)
        out = torch.zeros((nimg, H, W, Kpad), dtype=dt)
        hi = col.to(dt)
        out[..., : 9 * Cin] = hi
        if split:                                   # [hi | lo | hi], lo = what the 16-bit rounding lost
            out[..., 9 * Cin: 18 * Cin] = (col - hi.float()).to(dt)
            out[..., 18 * Cin: 27 * Cin] = hi
        return out.reshape(nimg * H * W, Kpad)

    def pointwise_small(self, src, nimg, Fi, Cin, H, W, s_strides, Wm, b, Cout, dst, d_strides):
        x = self._view5(src, nimg, Fi, Cin, H, W, s_strides)
        o = torch.einsum("oc,bfchw->bfohw", Wm, x)
        if b is not None:
            o = o + b.view(1, 1, -1, 1, 1)
        self._view5(dst, nimg, Fi, Cout, H, W, d_strides).copy_(o)
        return dst

    def cfg_ddim_step(self, xt, y, u, noise, coef, guide, use_guide, mean_type, want_x0):
        shape = (xt.shape[0],) + (1,) * (xt.ndim - 1)
        c = [coef[:, i].view(shape) for i in range(7)]
        out = y
        if use_guide:
            out = u + torch.tensor(guide, dtype=torch.float32) * (y - u)
        x0 = out if mean_type == 2 else c[0] * xt - c[1] * out
        eps = (c[2] * xt - x0) / c[3]
        r = torch.sqrt(c[4]) * x0 + torch.sqrt(1.0 - c[4] - c[5] * c[5]) * eps
        if noise is not None:
            r = r + c[6] * c[5] * noise
        return r, (x0.clone() if want_x0 else None)

    def ddim_update_units(self, x_units, G, B, C_lat, y, u, noise, coef_tab, t_idx, guide, use_guide, mean_type,
                          xt_1, x0, replicate=True):
        xt = x_units[:B, :C_lat].clone()
        coef = coef_tab.view(-1, 7)[t_idx] if t_idx is not None else coef_tab.view(-1, 7)[:B]
        r, x0v = self.cfg_ddim_step(xt, y.view_as(xt), None if u is None else u.view_as(xt),
                                    None if noise is None else noise.view_as(xt), coef.contiguous(), guide, use_guide,
                                    mean_type, True)
        if xt_1 is not None:
            xt_1.view_as(xt).copy_(r)
        if x0 is not None:
            x0.view_as(xt).copy_(x0v)
        if replicate:
            x_units.view(G, B, *x_units.shape[1:])[:, :, :C_lat] = r
        return xt_1, x0

    def ddim_update_strided(self, xt_rows, y, u, coef_tab, t_idx, guide, use_guide, mean_type, out_rows=None, x0_out=None,
                            rep_units=None, G=0, C_lat=0):
        xt = xt_rows.clone()
        coef = coef_tab.view(-1, 7)[t_idx]
        r, x0v = self.cfg_ddim_step(xt, y.view_as(xt), None if u is None else u.view_as(xt), None, coef.contiguous(),
                                    guide, use_guide, mean_type, True)
        if x0_out is not None:
            x0_out.view_as(xt).copy_(x0v)
        if out_rows is not None:
            out_rows.copy_(r)
        else:
            B = xt.shape[0]
            rep_units.view(G, B, *rep_units.shape[1:])[:, :, :C_lat] = r

    def lowfreq_filter(self, x, nimg, H, W, scale):
        # header formula, evaluated directly (independent of torch.fft)
        C = x.shape[1]
        v = x.view(nimg, H, W, C).double()
        th = 2 * math.pi * torch.arange(H, dtype=torch.float64) / H
        tw = 2 * math.pi * torch.arange(W, dtype=torch.float64) / W
        acc = torch.zeros_like(v)
        for u in (0, -1):
            for w_ in (0, -1):
                ph = (u * th)[:, None] + (w_ * tw)[None, :]                       # [H, W]
                X = (v * torch.exp(-1j * ph)[None, :, :, None]).sum(dim=(1, 2))      # [nimg, C] complex
                acc += (X[:, None, None, :] * torch.exp(1j * ph)[None, :, :, None]).real
        return (v + (scale - 1.0) / (H * W) * acc).float().view(nimg * H * W, C)

    def frames_u8(self, x, mean, std):
        v = x * std[None, :]
        v = v + mean[None, :]
        v = v.clamp(0.0, 1.0) * 255.0
        return v.to(torch.int32).to(torch.uint8)          # truncation, like numpy astype('uint8') on [0, 255]

    def scale_channels(self, x, c0, c1, s):
        from vgen_amd.ops import drop_colstats
        x[:, c0:c1] *= s
        drop_colstats(x)
        return x

    def gauss_denoise(self, xt, y, u, guide, rescale, coef, pred_type, want_eps):
        shape = (xt.shape[0],) + (1,) * (xt.ndim - 1)
        alpha, sigma = coef[:, 0].view(shape), coef[:, 1].view(shape)
        out = y if u is None else u + torch.tensor(guide, dtype=torch.float32) * (y - u)
        if rescale is not None:
            ratio = (y.flatten(1).std(dim=1) / (out.flatten(1).std(dim=1) + 1e-12)).view(shape)
            out = out * (rescale * ratio + (1 - rescale) * 1.0)
        if pred_type == 2:
            x0 = out
        elif pred_type == 0:
            x0 = (xt - sigma * out) / alpha
        else:
            x0 = alpha * xt - sigma * out
        eps = (xt - alpha * x0) / sigma if want_eps else None
        return x0, eps

    def lincomb4(self, a, b, c, d, ca, cb, cc, cd):
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        r = f(ca) * a
        for t, cf in ((b, cb), (c, cc), (d, cd)):
            if t is not None:
                r = r + f(cf) * t
        return r

    def repeat_rows(self, t, G):
        return t.repeat((G,) + (1,) * (t.dim() - 1))

    def gather_rows_f32(self, table, idx):
        return table.index_select(0, idx.clamp(0, table.shape[0] - 1))

    def dpmpp2m_sde_step(self, x, denoised, old, noise, ca, cb, cc, cn):
        # vgen_dpmpp2m_sde_step: the three tensor statements of diffusion_gauss.py:122-139 in the reference's own
        # rounding order (0-dim scalars folded on the host, tensor products rounded one by one)
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        r = f(ca) * x + f(cb) * denoised
        if old is not None:
            r = r + f(cc) * (denoised - old)
        if noise is not None:
            r = r + noise * f(cn[0]) * f(cn[1]) * f(cn[2])
        return r

    def gaussian_sample(self, moments, noise, nimg, zc, HW, scale):
        m = moments.view(nimg, HW, 2 * zc)
        mean = m[..., :zc].permute(0, 2, 1).reshape(noise.shape)
        logvar = m[..., zc:].permute(0, 2, 1).reshape(noise.shape).clamp(-30.0, 20.0)
        return scale * (mean + torch.exp(0.5 * logvar) * noise)
