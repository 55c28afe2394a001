"""UNetSD_I2VGen drop-in (reference: tools/modules/unet/unet_i2vgen.py:20-346).

The trunk — time/fps embedding, encoder / middle / decoder of ResBlock + temporal conv, Spatial and
Temporal transformers, head — is the t2v trunk (`UNetSD_T2VBase._trunk`, all hand-written HIP); the
differences are the stem conv's extra `concat_dim` input channels and the longer cross-attention
context (77 text tokens + 64 local-image tokens + `num_tokens` global-image tokens).

The condition stems that produce those extras depend only on the conditioning image, not on x or t:
the reference recomputes them on every denoise step (unet_i2vgen.py:280-321), here they are evaluated
once per conditioning input and cached.  They are tiny (4..64-channel convs, an 8x8-token pooling
pyramid, a 4-wide one-layer transformer over frames) and run on the C ABI's stem kernels
(vgen_conv3x3_small / vgen_adaptive_avgpool2d / vgen_frame_transformer / vgen_linear_f32, fp32);
the nn.Modules below only hold the parameters.
Parameter names / shapes equal the reference's, so stock checkpoints load strict.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F_

from . import ops
from .unet import UNetSD_T2VBase


# ------------------------------------------------------------------------------------------
# native condition stems: the nn.Modules below are parameter containers; these helpers run them
# on the C ABI (vgen_conv3x3_small / vgen_adaptive_avgpool2d / vgen_frame_transformer / vgen_linear_f32)
# ------------------------------------------------------------------------------------------
def _p32(t):
    return t.detach().float().contiguous()


def conv_stack(x, seq):
    """nn.Sequential of Conv2d(3x3, padding 1, stride 1|2) / SiLU / AdaptiveAvgPool2d on NCHW fp32 frames."""
    be = ops.backend()
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            assert m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride[0] == m.stride[1]
            act = 1 if i + 1 < len(mods) and isinstance(mods[i + 1], nn.SiLU) else 0
            x = be.conv3x3_small(x, _p32(m.weight), _p32(m.bias), stride=m.stride[0], act=act)
            i += 2 if act else 1
        elif isinstance(m, nn.AdaptiveAvgPool2d):
            ho, wo = m.output_size
            x = be.adaptive_avgpool2d(x, ho, wo)
            i += 1
        else:
            raise TypeError(type(m))
    return x


def frame_transformer(tf, frames, B, F, out=None, out_scale=1.0, accumulate=False):
    """`tf` (_FrameTransformer) over the frame axis of every pixel: frames [(B F), d, H, W] -> [B, d, F, H, W]."""
    be = ops.backend()
    d, H, W = frames.shape[1:]
    n = len(tf.layers)
    x = frames
    for li, (attn_pre, mlp) in enumerate(tf.layers):
        attn = attn_pre.fn
        inner = attn.to_qkv.weight.shape[0] // 3
        has_out = not isinstance(attn.to_out, nn.Identity)
        p = dict(ln_w=_p32(attn_pre.norm.weight), ln_b=_p32(attn_pre.norm.bias), wqkv=_p32(attn.to_qkv.weight),
                 wout=_p32(attn.to_out[0].weight) if has_out else None, bout=_p32(attn.to_out[0].bias) if has_out else None,
                 w1=_p32(mlp.net[0][0].weight), b1=_p32(mlp.net[0][0].bias), w2=_p32(mlp.net[2].weight), b2=_p32(mlp.net[2].bias),
                 heads=attn.heads, dim_head=inner // attn.heads, hidden=mlp.net[0][0].weight.shape[0])
        last = li == n - 1
        x = be.frame_transformer(x, B, F, d, H * W, p, out=out if last else None, last=last,
                                 out_scale=out_scale if last else 1.0, accumulate=accumulate and last)
    return x.view(B, d, F, H, W)


def mlp_f32(x, seq):
    """Linear - SiLU - Linear (context / histogram embeddings) on vgen_linear_f32; x [n, K] fp32."""
    be = ops.backend()
    l0, l2 = seq[0], seq[2]
    h = be.linear_f32(x.float().contiguous(), _p32(l0.weight), _p32(l0.bias))
    return be.linear_f32(h, _p32(l2.weight), _p32(l2.bias), act_in=1)


class _FrameAttention(nn.Module):
    """util.py:1396-1424 — multi-head softmax attention over the frame axis, bias-free qkv."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_qkv = nn.Linear(dim, 3 * inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim)) if not (heads == 1 and dim_head == dim) else nn.Identity()

    def forward(self, x):                                   # [n, f, dim]
        n, f, _ = x.shape
        q, k, v = self.to_qkv(x).view(n, f, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        o = F_.scaled_dot_product_attention(q, k, v)        # scale = dim_head ** -0.5
        return self.to_out(o.transpose(1, 2).reshape(n, f, -1))


class _PreNorm(nn.Module):
    """util.py:1426-1432."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x)) + x


class _MLP(nn.Module):
    """util.py:724-741 with glu=False: Linear -> GELU -> Linear (parameter paths net.0.0 / net.2)."""

    def __init__(self, dim, dim_out, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Sequential(nn.Linear(dim, dim * mult), nn.GELU()), nn.Identity(),
                                 nn.Linear(dim * mult, dim_out))

    def forward(self, x):
        return self.net(x)


class _FrameTransformer(nn.Module):
    """util.py:1434-1453 (TransformerV2)."""

    def __init__(self, heads, dim, dim_head, mlp_dim, depth):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _FrameAttention(dim, heads, dim_head)),
                                                    _MLP(dim, mlp_dim)]) for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x)
            x = ff(x) + x
        return x


class UNetSD_I2VGen(UNetSD_T2VBase):
    @staticmethod
    def _extra_stem_channels(kwargs):
        return kwargs["_i2v_concat"]

    def __init__(self, config=None, in_dim=7, dim=512, y_dim=512, context_dim=512, hist_dim=156, concat_dim=8,
                 dim_condition=4, out_dim=6, num_tokens=4, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64,
                 num_res_blocks=3, attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1,
                 temporal_attn_times=1, temporal_attention=True, use_checkpoint=False, use_image_dataset=False,
                 use_sim_mask=False, training=True, inpainting=True, p_all_zero=0.1, p_all_keep=0.1, zero_y=None,
                 adapter_transformer_layers=1, compute_dtype=None, **kwargs):
        kwargs.pop("use_fps_condition", None)            # yaml carries it; the fps embedding is unconditional here
        cc = in_dim                                      # unet_i2vgen.py:81: the stems use in_dim channels
        super().__init__(config=config, in_dim=in_dim, dim=dim, y_dim=y_dim, context_dim=context_dim,
                         hist_dim=hist_dim, dim_condition=dim_condition, out_dim=out_dim, num_tokens=num_tokens,
                         dim_mult=dim_mult, num_heads=num_heads, head_dim=head_dim, num_res_blocks=num_res_blocks,
                         attn_scales=attn_scales, use_scale_shift_norm=use_scale_shift_norm, dropout=dropout,
                         temporal_attn_times=temporal_attn_times, temporal_attention=temporal_attention,
                         use_checkpoint=use_checkpoint, use_image_dataset=use_image_dataset, use_sim_mask=use_sim_mask,
                         training=training, inpainting=inpainting, use_fps_condition=True, p_all_zero=p_all_zero,
                         p_all_keep=p_all_keep, zero_y=zero_y, adapter_transformer_layers=adapter_transformer_layers,
                         compute_dtype=compute_dtype, _i2v_concat=cc, **kwargs)
        if concat_dim != in_dim:
            raise ValueError("UNetSD_I2VGen: concat_dim must equal in_dim (the reference adds an in_dim-channel "
                             "local-image map into a concat_dim-channel buffer, unet_i2vgen.py:81,281-295)")
        self.concat_dim, self.num_tokens = concat_dim, num_tokens
        embed_dim = dim * 4
        self.context_embedding = nn.Sequential(nn.Linear(y_dim, embed_dim), nn.SiLU(),
                                               nn.Linear(embed_dim, context_dim * num_tokens))
        self.local_image_concat = nn.Sequential(nn.Conv2d(4, cc * 4, 3, padding=1), nn.SiLU(),
                                                nn.Conv2d(cc * 4, cc * 4, 3, padding=1), nn.SiLU(),
                                                nn.Conv2d(cc * 4, cc, 3, padding=1))
        self.local_temporal_encoder = _FrameTransformer(heads=2, dim=cc, dim_head=cc, mlp_dim=cc,
                                                        depth=adapter_transformer_layers)
        self.local_image_embedding = nn.Sequential(nn.Conv2d(4, cc * 8, 3, padding=1), nn.SiLU(),
                                                   nn.AdaptiveAvgPool2d((32, 32)),
                                                   nn.Conv2d(cc * 8, cc * 16, 3, stride=2, padding=1), nn.SiLU(),
                                                   nn.Conv2d(cc * 16, 1024, 3, stride=2, padding=1))
        self._stem_cache = None

    def invalidate(self):
        super().invalidate()
        self._stem_cache = None

    # -- condition stems (prompt constants) ----------------------------------------------------------
    @torch.no_grad()
    def condition_stems(self, local_image, image, B, F, H, W):
        """-> (concat [B, concat_dim, F, H, W] fp32, extra context [B, 64 (+ num_tokens), context_dim] fp32).
        Cached per conditioning TENSOR OBJECT (the engine passes the same tensors on every denoise step); the
        cache holds references to them, so a recycled allocation can never alias a stale entry, and their
        `_version` counters catch in-place edits."""
        key = (id(local_image), local_image._version, None if image is None else (id(image), image._version),
               B, F, H, W)
        cache = self._stem_cache if self._stem_cache is not None else {}
        hit = cache.get(key)
        if hit is not None and hit[0] is local_image and hit[1] is image:
            return hit[2]
        li = local_image.float()
        li = li[:, :, :1] if li.dim() == 5 else li.unsqueeze(2)                    # [B, 4, 1, H, W]
        # frame 0 = the image latent, frames 1.. = their relative time (t+1)/(F-1) in every channel (:282-288)
        frames = [li]
        for tpos in range(F - 1):
            frames.append(torch.full_like(li, (tpos + 1) / (F - 1)))
        xi = torch.cat(frames, 2).permute(0, 2, 1, 3, 4).reshape(B * F, li.shape[1], H, W).contiguous()
        xi = conv_stack(xi, self.local_image_concat)                               # [(B F), cc, H, W]
        # TransformerV2 over the frames of every pixel, written straight into the stem-channel layout; the reference
        # adds the map twice (:294-295), kept: scale 2
        concat = frame_transformer(self.local_temporal_encoder, xi, B, F, out_scale=2.0)
        lc = conv_stack(li[:, :, 0].contiguous(), self.local_image_embedding)      # [B, 1024, 8, 8]
        extra = lc.flatten(2).transpose(1, 2)                                      # [B, 64, 1024]
        if image is not None:
            ic = mlp_f32(image.reshape(-1, image.shape[-1]), self.context_embedding).view(-1, self.num_tokens, self.context_dim)
            extra = torch.cat([extra, ic], 1)
        out = (concat, extra.contiguous())
        if len(cache) >= 8:                              # a handful of prompts in flight; never grows unbounded
            cache.clear()
        cache[key] = (local_image, image, out)
        self._stem_cache = cache
        return out

    def _with_stems(self, x, t, y, concat, extra, fps):
        B = x.shape[0]
        ctx = y if y is not None else self.zero_y.repeat(B, 1, 1)[:, :1, :]
        ctx = torch.cat([ctx.float(), extra.to(ctx.device)], 1)                    # text | local | global tokens
        return self._trunk(torch.cat([x.float(), concat], 1), t, ctx, fps)

    @torch.no_grad()
    def forward(self, x, t, y=None, image=None, local_image=None, masked=None, fps=None, video_mask=None,
                focus_present_mask=None, prob_focus_present=0., mask_last_frame_num=0, **kwargs):
        if local_image is None or fps is None:
            raise ValueError("UNetSD_I2VGen.forward needs local_image and fps (unet_i2vgen.py:262-265,298)")
        self._maybe_auto_calibrate(tuple(x.shape), x.device, dict(y=y, image=image, local_image=local_image, fps=fps), t.dtype)
        B, C, F, H, W = x.shape
        concat, extra = self.condition_stems(local_image, image, B, F, H, W)
        return self._with_stems(x, t, y, concat, extra, fps)

    def _prepare_units(self, shape, device, kwargs_list):
        """CFG units of one latent batch (see UNetSD_T2VBase._prepare_units).  The stems are evaluated (and cached)
        per unit on the caller's own conditioning tensors, then stacked."""
        if any(kw.get("y") is None or kw.get("local_image") is None or kw.get("fps") is None for kw in kwargs_list):
            return None
        B, C, F, H, W = shape
        stems = [self.condition_stems(kw["local_image"], kw.get("image"), B, F, H, W) for kw in kwargs_list]
        if len({s[1].shape[1] for s in stems}) != 1:
            return None
        extra = torch.cat([s[1].to(device) for s in stems], 0)
        y = torch.cat([kw["y"].to(device=device, dtype=torch.float32) for kw in kwargs_list], 0)
        return dict(extra=torch.cat([s[0].to(device) for s in stems], 0), ctx=torch.cat([y, extra], 1), per_frame=False,
                    fps=torch.cat([kw["fps"].reshape(-1).to(device) for kw in kwargs_list], 0))
