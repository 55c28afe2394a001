"""TEST INFRASTRUCTURE (oracle) — CPU restatement, in plain fp32 torch, of the reference's
algorithm for the sampling hot path.  Functional style over a state_dict with the reference's key
names; every function cites the reference file:line it follows (paths relative to the reference
root).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY PIN: the reference ships no golden vectors / tests for this path, so the
oracle is pinned against the reference's OWN modules executed on CPU in the build container
(oracle/ref_import.py): tests/test_oracle_vs_reference.py checks every function here against
them, and oracle/make_golden.py stores reference-generated fixtures under tests/golden/ that
travel to the GPU box.  Third-party arithmetic restated mathematically (not pinned by any
reference test): xformers.ops.memory_efficient_attention (xformers==0.0.13) = softmax(QK^T/sqrt(d))V.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------
# deterministic synthetic weights (pretrained checkpoints are not available offline)
# ------------------------------------------------------------------------------------------


def synth_state_dict(shapes: dict, seed: int = 0, gain: float = 0.8, recipe: str = "gauss") -> dict:
    """Seeded re-randomisation of EVERY parameter (the reference zero-initialises the ResBlock
    out-conv, temporal conv4, proj_out, head conv — util.py:873-875,1683-1684,351,1229,
    unet_t2v.py:208 — which would make a parity check vacuous; SURVEY.md §8c "Trap").
    The recipe lives in vgen_amd/synth.py (bench.py times the very weights the golden fixtures were made with)."""
    from vgen_amd.synth import seeded_state_dict
    return seeded_state_dict(shapes, seed, gain, recipe)


def shapes_of(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


# ------------------------------------------------------------------------------------------
# UNet blocks
# ------------------------------------------------------------------------------------------


def sinusoidal_embedding(timesteps, dim):
    # unet/util.py:178-190
    half = dim // 2
    timesteps = timesteps.float()
    freqs = torch.pow(10000, -torch.arange(half).to(timesteps).div(half))
    s = torch.outer(timesteps, freqs)
    x = torch.cat([torch.cos(s), torch.sin(s)], dim=1)
    if dim % 2 != 0:
        x = torch.cat([x, torch.zeros_like(x[:, :1])], dim=1)
    return x


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def attention(sd, p, x, context, heads):
    # MemoryEfficientCrossAttention.forward, unet/util.py:231-269: q/k/v Linear (no bias),
    # heads split to [B*h, M, 64], softmax(q k^T * 64^-0.5) v, merge heads, to_out Linear.
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, _ = q.shape
    d = q.shape[-1] // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    w = torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def transformer_block(sd, p, x, context, heads):
    # BasicTransformerBlock.forward, unet/util.py:700-704; GEGLU :712-714; FeedForward :724-741
    def ln(i, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.norm{i}.weight"], sd[f"{p}.norm{i}.bias"], 1e-5)

    x = attention(sd, p + ".attn1", ln(1, x), None, heads) + x
    x = attention(sd, p + ".attn2", ln(2, x), context, heads) + x
    h = F.linear(ln(3, x), sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    return F.linear(h, sd[p + ".ff.net.2.weight"], sd[p + ".ff.net.2.bias"]) + x


def spatial_transformer(sd, p, x, context):
    # SpatialTransformer.forward (use_linear=True), unet/util.py:354-373
    b, c, h, w = x.shape
    inner = sd[p + ".proj_in.weight"].shape[0]
    t = _gn(sd, p + ".norm", x, 1e-6)
    t = t.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = F.linear(t, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    t = transformer_block(sd, p + ".transformer_blocks.0", t, context, inner // 64)
    t = F.linear(t, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def temporal_transformer(sd, p, x):
    # TemporalTransformer.forward (use_linear=False, only_self_att=True), unet/util.py:1240-1286:
    # GN over the 5-D tensor, tokens = frames of one pixel, both attentions are self-attention.
    b, c, f, h, w = x.shape
    inner = sd[p + ".proj_in.weight"].shape[0]
    t = _gn(sd, p + ".norm", x, 1e-6)
    t = t.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f)
    t = F.conv1d(t, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    t = t.permute(0, 2, 1)
    t = transformer_block(sd, p + ".transformer_blocks.0", t, None, inner // 64)
    t = t.permute(0, 2, 1)
    t = F.conv1d(t, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return t.reshape(b, h, w, c, f).permute(0, 3, 4, 1, 2) + x


def temporal_conv_block(sd, p, x):
    # TemporalConvBlock_v2.forward, unet/util.py:1686-1697 (dropout inactive in eval)
    h = x
    for i in (1, 2, 3, 4):
        ci = 2 if i == 1 else 3
        h = F.silu(_gn(sd, f"{p}.conv{i}.0", h, 1e-5))
        h = F.conv3d(h, sd[f"{p}.conv{i}.{ci}.weight"], sd[f"{p}.conv{i}.{ci}.bias"], padding=(1, 0, 0))
    return x + h


def resblock(sd, p, x, emb, batch):
    # ResBlock._forward, unet/util.py:900-927 (use_scale_shift_norm False, no up/down)
    h = F.conv2d(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"],
                 sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"],
                 sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    h = x + h
    bf, c, hh, ww = h.shape
    h5 = h.reshape(batch, bf // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
    return h5.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


def _kind(sd, p):
    if p + ".in_layers.0.weight" in sd:
        return "res"
    if p + ".transformer_blocks.0.attn1.to_q.weight" in sd:
        return "temporal" if sd[p + ".proj_in.weight"].dim() == 3 else "spatial"
    if p + ".op.weight" in sd:
        return "down"
    if p + ".conv.weight" in sd:
        return "up"
    return None


def fourier_filter(x, threshold, scale):
    # Fourier_filter, unet/unet_sr600.py:30-49 (the reference hard-codes .cuda() for the mask)
    xf = torch.fft.fftshift(torch.fft.fftn(x.float(), dim=(-2, -1)), dim=(-2, -1))
    B, C, H, W = xf.shape
    mask = torch.ones((B, C, H, W))
    crow, ccol = H // 2, W // 2
    mask[..., crow - threshold:crow + threshold, ccol - threshold:ccol + threshold] = scale
    return torch.fft.ifftn(torch.fft.ifftshift(xf * mask, dim=(-2, -1)), dim=(-2, -1)).real


def unet_sr600_forward(sd, x, t, y, dim):
    """UNetSD_SR600.forward, unet/unet_sr600.py:220-299: the t2v trunk with Downsample padding (2,1)
    (:151), UpsampleSR600's crop (util.py:801) and the FreeU-style tweaks of decoder blocks 0/1."""
    return unet_forward(sd, x, t, y, dim, down_padding=(2, 1), up_crop=1, freeu=((1.1, 0.6), (1.2, 0.4)))


def i2vgen_stems(sd, local_image, image, b, f, h, w, num_tokens, context_dim):
    """UNetSD_I2VGen condition stems, unet_i2vgen.py:262-265 (first frame of the local image),
    :280-295 (local-image map: conv stack -> one TransformerV2 layer over frames, added TWICE into the
    concat buffer), :310-321 (64 local-image tokens from the pooled conv pyramid, num_tokens global tokens)."""
    li = local_image[:, :, :1] if local_image.dim() == 5 else local_image.unsqueeze(2)
    frames = [li] + [torch.ones_like(li) * ((tp + 1) / (f - 1)) for tp in range(f - 1)]
    xi = torch.cat(frames, 2).permute(0, 2, 1, 3, 4).reshape(b * f, li.shape[1], h, w)
    p = "local_image_concat"
    xi = F.conv2d(xi, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1)
    xi = F.conv2d(F.silu(xi), sd[p + ".2.weight"], sd[p + ".2.bias"], padding=1)
    xi = F.conv2d(F.silu(xi), sd[p + ".4.weight"], sd[p + ".4.bias"], padding=1)
    cc = xi.shape[1]
    s = xi.reshape(b, f, cc, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, f, cc)
    j = 0
    while f"local_temporal_encoder.layers.{j}.0.norm.weight" in sd:      # TransformerV2, util.py:1434-1453
        q = f"local_temporal_encoder.layers.{j}"
        n = F.layer_norm(s, (cc,), sd[q + ".0.norm.weight"], sd[q + ".0.norm.bias"])
        qkv = F.linear(n, sd[q + ".0.fn.to_qkv.weight"])
        inner = qkv.shape[-1] // 3
        heads = 2
        qh, kh, vh = [u.reshape(b * h * w, f, heads, inner // heads).transpose(1, 2) for u in qkv.split(inner, -1)]
        att = torch.softmax(qh @ kh.transpose(-1, -2) * (inner // heads) ** -0.5, -1) @ vh
        att = att.transpose(1, 2).reshape(b * h * w, f, inner)
        s = F.linear(att, sd[q + ".0.fn.to_out.0.weight"], sd[q + ".0.fn.to_out.0.bias"]) + s
        m = F.gelu(F.linear(s, sd[q + ".1.net.0.0.weight"], sd[q + ".1.net.0.0.bias"]))
        s = F.linear(m, sd[q + ".1.net.2.weight"], sd[q + ".1.net.2.bias"]) + s
        j += 1
    concat = 2.0 * s.reshape(b, h, w, f, cc).permute(0, 4, 3, 1, 2)
    p = "local_image_embedding"
    lc = F.silu(F.conv2d(li[:, :, 0], sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1))
    lc = F.adaptive_avg_pool2d(lc, (32, 32))
    lc = F.silu(F.conv2d(lc, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2, padding=1))
    lc = F.conv2d(lc, sd[p + ".5.weight"], sd[p + ".5.bias"], stride=2, padding=1)
    extra = lc.flatten(2).transpose(1, 2)
    if image is not None:
        ic = F.linear(F.silu(F.linear(image, sd["context_embedding.0.weight"], sd["context_embedding.0.bias"])),
                      sd["context_embedding.2.weight"], sd["context_embedding.2.bias"])
        extra = torch.cat([extra, ic.reshape(-1, num_tokens, context_dim)], 1)
    return concat, extra


def unet_i2vgen_forward(sd, x, t, y, image, local_image, fps, dim, num_tokens=4, context_dim=1024):
    """UNetSD_I2VGen.forward, unet_i2vgen.py:243-346: stems, then the shared trunk on cat([x, concat]) with
    context = text | local-image | global-image tokens."""
    b, c, f, h, w = x.shape
    concat, extra = i2vgen_stems(sd, local_image, image, b, f, h, w, num_tokens, context_dim)
    return unet_forward(sd, torch.cat([x, concat], 1), t, torch.cat([y, extra], 1), dim, fps=fps)


_COMPOSER_STEMS = (("depth", "depth_embedding", "depth_embedding_after"), ("local_image", "local_image_embedding", "local_image_embedding_after"),
                   ("motion", "motion_embedding", "motion_embedding_after"), ("canny", "canny_embedding", "canny_embedding_after"),
                   ("sketch", "sketch_embedding", "sketch_embedding_after"),
                   ("single_sketch", "single_sketch_embedding", "single_sketch_embedding_after"),
                   ("masked", "masked_embedding", "mask_embedding_after"))


def _frame_transformer(sd, q0, s, heads=2):
    """TransformerV2 / Transformer_v2 (util.py:1434-1453, unet_videolcm.py:121-141) on [n, f, c]."""
    n, f, cc = s.shape
    j = 0
    while f"{q0}.layers.{j}.0.norm.weight" in sd:
        q = f"{q0}.layers.{j}"
        nn_ = F.layer_norm(s, (cc,), sd[q + ".0.norm.weight"], sd[q + ".0.norm.bias"])
        qkv = F.linear(nn_, sd[q + ".0.fn.to_qkv.weight"])
        inner = qkv.shape[-1] // 3
        qh, kh, vh = [u.reshape(n, f, heads, inner // heads).transpose(1, 2) for u in qkv.split(inner, -1)]
        att = torch.softmax(qh @ kh.transpose(-1, -2) * (inner // heads) ** -0.5, -1) @ vh
        att = att.transpose(1, 2).reshape(n, f, inner)
        s = F.linear(att, sd[q + ".0.fn.to_out.0.weight"], sd[q + ".0.fn.to_out.0.bias"]) + s
        m = F.gelu(F.linear(s, sd[q + ".1.net.0.0.weight"], sd[q + ".1.net.0.0.bias"]))
        s = F.linear(m, sd[q + ".1.net.2.weight"], sd[q + ".1.net.2.bias"]) + s
        j += 1
    return s


def composer_concat(sd, conds, resolution, b):
    """Sum of the spatial composition stems, unet_videolcm.py:598-699 (eval: misc_dropout is the identity), in
    the reference's order of accumulation."""
    concat = None
    for kwarg, stem, after in _COMPOSER_STEMS:
        c = conds.get(kwarg)
        if c is None:
            continue
        bc, ch, f, hh, ww = c.shape
        z = c.permute(0, 2, 1, 3, 4).reshape(bc * f, ch, hh, ww)
        z = F.silu(F.conv2d(z, sd[stem + ".0.weight"], sd[stem + ".0.bias"], padding=1))
        z = F.adaptive_avg_pool2d(z, (resolution[1] // 2, resolution[0] // 2))
        z = F.silu(F.conv2d(z, sd[stem + ".3.weight"], sd[stem + ".3.bias"], stride=2, padding=1))
        z = F.conv2d(z, sd[stem + ".5.weight"], sd[stem + ".5.bias"], stride=2, padding=1)
        cd, h, w = z.shape[1:]
        sq = z.reshape(b, f, cd, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, f, cd)
        sq = _frame_transformer(sd, after, sq)
        z5 = sq.reshape(b, h, w, f, cd).permute(0, 4, 3, 1, 2)
        concat = z5 if concat is None else concat + z5
    return concat


def unet_composer_forward(sd, x, t, y, dim, concat_dim, resolution, image=None, num_tokens=4, context_dim=1024,
                          histogram=None, **conds):
    """UNetSD_VideoLCM / UNetSD_TFT2V.forward with any composition list (unet_videolcm.py:541-784)."""
    b, c, f, h, w = x.shape
    concat = composer_concat(sd, conds, resolution, b)
    if concat is None:
        concat = x.new_zeros(b, concat_dim, f, h, w)
    ctx = y
    if image is not None:
        ic = F.linear(F.silu(F.linear(image, sd["pre_image_condition.0.weight"], sd["pre_image_condition.0.bias"])),
                      sd["pre_image_condition.2.weight"], sd["pre_image_condition.2.bias"])
        ctx = torch.cat([ctx, ic.reshape(-1, num_tokens, context_dim)], 1)
    if histogram is not None:                            # one more token per frame (:747-755)
        hc = F.linear(F.silu(F.linear(histogram, sd["hist_context_embedding.0.weight"], sd["hist_context_embedding.0.bias"])),
                      sd["hist_context_embedding.2.weight"], sd["hist_context_embedding.2.bias"])
        ctx = torch.cat([ctx.repeat_interleave(repeats=f, dim=0), hc.reshape(b * f, 1, context_dim)], 1)
        return unet_forward(sd, torch.cat([x, concat], 1), t, ctx, dim, context_per_frame=True)
    return unet_forward(sd, torch.cat([x, concat], 1), t, ctx, dim)


def unet_videolcm_text_forward(sd, x, t, y, dim, concat_dim, image=None, num_tokens=4, context_dim=1024):
    """UNetSD_VideoLCM / UNetSD_TFT2V.forward with video_compositions within ['text', 'image']
    (unet_videolcm.py:598, 702-705, 709-784; unet_tf2tv.py likewise): zero concat buffer, identity pre_image,
    context = text tokens (+ num_tokens global-image tokens from pre_image_condition, :743-745), shared trunk."""
    b, c, f, h, w = x.shape
    ctx = y
    if image is not None:
        ic = F.linear(F.silu(F.linear(image, sd["pre_image_condition.0.weight"], sd["pre_image_condition.0.bias"])),
                      sd["pre_image_condition.2.weight"], sd["pre_image_condition.2.bias"])
        ctx = torch.cat([ctx, ic.reshape(-1, num_tokens, context_dim)], 1)
    return unet_forward(sd, torch.cat([x, x.new_zeros(b, concat_dim, f, h, w)], 1), t, ctx, dim)


def unet_forward(sd, x, t, y, dim, down_padding=1, up_crop=0, freeu=None, fps=None, context_per_frame=False):
    """UNetSD_T2VBase.forward / _forward_single, unet/unet_t2v.py:210-348 (y given; `fps` adds the
    fps embedding, :244-245 / unet_i2vgen.py:298).  The block structure is recovered from the state_dict keys."""
    b, c, f, h, w = x.shape
    emb = F.linear(sinusoidal_embedding(t, dim), sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if fps is not None:
        e2 = F.linear(sinusoidal_embedding(fps, dim), sd["fps_embedding.0.weight"], sd["fps_embedding.0.bias"])
        emb = emb + F.linear(F.silu(e2), sd["fps_embedding.2.weight"], sd["fps_embedding.2.bias"])
    emb = emb.repeat_interleave(repeats=f, dim=0)
    context = y if context_per_frame else y.repeat_interleave(repeats=f, dim=0)     # [(b f), L, D]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)

    def run(p, x):
        kind = _kind(sd, p)
        if kind == "res":
            return resblock(sd, p, x, emb, b)
        if kind == "spatial":
            return spatial_transformer(sd, p, x, context)
        if kind == "temporal":
            bf, cc, hh, ww = x.shape
            x5 = x.reshape(b, bf // b, cc, hh, ww).permute(0, 2, 1, 3, 4)
            x5 = temporal_transformer(sd, p, x5)
            return x5.permute(0, 2, 1, 3, 4).reshape(bf, cc, hh, ww)
        if kind == "down":
            return F.conv2d(x, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=down_padding)
        if kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if up_crop:
                x = x[..., up_crop:-up_crop, :]
            return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        raise KeyError(p)

    def run_list(p, x):
        if _kind(sd, p) is not None:
            return run(p, x)
        j = 0
        while _kind(sd, f"{p}.{j}") is not None:
            x = run(f"{p}.{j}", x)
            j += 1
        return x

    xs = []
    x = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    x = run("input_blocks.0.1", x)
    xs.append(x)
    i = 1
    while any(k.startswith(f"input_blocks.{i}.") for k in sd):
        x = run_list(f"input_blocks.{i}", x)
        xs.append(x)
        i += 1
    x = run_list("middle_block", x)
    i = 0
    while any(k.startswith(f"output_blocks.{i}.") for k in sd):
        skip = xs.pop()
        if freeu is not None and i < len(freeu):        # unet_sr600.py:274-287
            x = x.clone()
            x[:, : x.shape[1] // 2] = x[:, : x.shape[1] // 2] * freeu[i][0]
            skip = fourier_filter(skip, 1, freeu[i][1])
        x = torch.cat([x, skip], dim=1)
        x = run_list(f"output_blocks.{i}", x)
        i += 1
    x = F.conv2d(F.silu(_gn(sd, "out.0", x, 1e-5)), sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return x.reshape(b, f, -1, h, w).permute(0, 2, 1, 3, 4)


# ------------------------------------------------------------------------------------------
# AutoencoderKL
# ------------------------------------------------------------------------------------------


def _swish(x):
    return x * torch.sigmoid(x)      # autoencoder.py:11-13


def vae_resnet(sd, p, x):
    # ResnetBlock.forward (temb None), autoencoder.py:315-335
    h = F.conv2d(_swish(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def vae_attn(sd, p, x):
    # AttnBlock.forward, autoencoder.py:418-442
    h = _gn(sd, p + ".norm", x, 1e-6)
    q = F.conv2d(h, sd[p + ".q.weight"], sd[p + ".q.bias"])
    k = F.conv2d(h, sd[p + ".k.weight"], sd[p + ".k.bias"])
    v = F.conv2d(h, sd[p + ".v.weight"], sd[p + ".v.bias"])
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def _vae_mid(sd, p, h):
    h = vae_resnet(sd, p + ".mid.block_1", h)
    h = vae_attn(sd, p + ".mid.attn_1", h)
    return vae_resnet(sd, p + ".mid.block_2", h)


def vae_decode(sd, z):
    # AutoencoderKL.decode -> Decoder.forward, autoencoder.py:100-103, 653-686
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _vae_mid(sd, "decoder", h)
    nlev = 0
    while f"decoder.up.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lvl in reversed(range(nlev)):
        j = 0
        while f"decoder.up.{lvl}.block.{j}.norm1.weight" in sd:
            h = vae_resnet(sd, f"decoder.up.{lvl}.block.{j}", h)
            j += 1
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lvl}.upsample.conv.weight"],
                         sd[f"decoder.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_gn(sd, "decoder.norm_out", h, 1e-6))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def vae_encode_moments(sd, x):
    # AutoencoderKL.encode -> Encoder.forward + quant_conv, autoencoder.py:79-83, 549-578
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nlev = 0
    while f"encoder.down.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lvl in range(nlev):
        j = 0
        while f"encoder.down.{lvl}.block.{j}.norm1.weight" in sd:
            h = vae_resnet(sd, f"encoder.down.{lvl}.block.{j}", h)
            j += 1
        if lvl != nlev - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)     # autoencoder.py:476-478
            h = F.conv2d(h, sd[f"encoder.down.{lvl}.downsample.conv.weight"],
                         sd[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2)
    h = _vae_mid(sd, "encoder", h)
    h = _swish(_gn(sd, "encoder.norm_out", h, 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def gaussian_sample(moments, noise, scale):
    # DiagonalGaussianDistribution + get_first_stage_encoding, autoencoder.py:212-225, 19-27
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return scale * (mean + torch.exp(0.5 * logvar) * noise)


# ------------------------------------------------------------------------------------------
# DDIM sampler (fp32 algebra with fp64 tables cast per use, like `_i`)
# ------------------------------------------------------------------------------------------


def ddim_tables(betas64):
    # DiffusionDDIM.__init__, diffusions/diffusion_ddim.py:54-70
    ac = torch.cumprod(1 - betas64, dim=0)
    return dict(ac=ac, sqrt_ac=torch.sqrt(ac), sqrt_1m=torch.sqrt(1.0 - ac),
                sqrt_recip=torch.sqrt(1.0 / ac), sqrt_recipm1=torch.sqrt(1.0 / ac - 1))


def _i(tab, t, x):
    # diffusion_ddim.py:10-16
    return tab[t].view((x.size(0),) + (1,) * (x.ndim - 1)).to(x)


def ddim_step(tabs, xt, t, y_out, u_out, guide_scale, mean_type, stride, eta=0.0, noise=None,
              num_timesteps=1000):
    """p_mean_variance (CFG + x0) and ddim_sample, diffusion_ddim.py:157-162, 187-197, 230-240."""
    out = y_out if guide_scale is None else u_out + guide_scale * (y_out - u_out)
    if mean_type == "v":
        x0 = _i(tabs["sqrt_ac"], t, xt) * xt - _i(tabs["sqrt_1m"], t, xt) * out
    elif mean_type == "eps":
        x0 = _i(tabs["sqrt_recip"], t, xt) * xt - _i(tabs["sqrt_recipm1"], t, xt) * out
    else:
        x0 = out
    eps = (_i(tabs["sqrt_recip"], t, xt) * xt - x0) / _i(tabs["sqrt_recipm1"], t, xt)
    alphas = _i(tabs["ac"], t, xt)
    alphas_prev = _i(tabs["ac"], (t - stride).clamp(0), xt)
    sigmas = eta * torch.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if noise is None:
        noise = torch.zeros_like(xt)
    direction = torch.sqrt(1 - alphas_prev - sigmas ** 2) * eps
    mask = t.ne(0).float().view(-1, *((1,) * (xt.ndim - 1)))
    xt_1 = torch.sqrt(alphas_prev) * x0 + direction + mask * sigmas * noise
    return xt_1, x0


def ddim_timesteps(num_timesteps, ddim_steps):
    # diffusion_ddim.py:250
    return (1 + torch.arange(0, num_timesteps, num_timesteps // ddim_steps)).clamp(0, num_timesteps - 1).flip(0)


def ddim_sample_loop(betas64, noise, model, model_kwargs, guide_scale, ddim_steps, mean_type="v", eta=0.0):
    # DiffusionDDIM.ddim_sample_loop, diffusion_ddim.py:243-254 (two sequential model calls per step)
    tabs = ddim_tables(betas64)
    T = len(betas64)
    xt = noise
    for step in ddim_timesteps(T, ddim_steps):
        t = torch.full((noise.size(0),), int(step), dtype=torch.long)
        y_out = model(xt, t, **model_kwargs[0])
        u_out = model(xt, t, **model_kwargs[1])
        xt, _ = ddim_step(tabs, xt, t, y_out, u_out, guide_scale, mean_type, T // ddim_steps, eta,
                          torch.randn_like(xt) if eta else None, T)
    return xt


# ------------------------------------------------------------------------------------------
# GaussianDiffusion: denoise + DPM-Solver++(2M) SDE + DDIM inversion (fp32)
# ------------------------------------------------------------------------------------------


def gauss_tables(sigmas64):
    # GaussianDiffusion.__init__, diffusions/diffusion_gauss.py:147-152
    return dict(sigmas=sigmas64.float(), alphas=torch.sqrt(1 - sigmas64 ** 2).float())


def gauss_x0_eps(tabs, xt, t, y_out, u_out, guide_scale, guide_rescale, prediction_type):
    """GaussianDiffusion.denoise, diffusion_gauss.py:182-183 (tables), :198-218 (CFG + guide_rescale),
    :220-243 (x0, eps)."""
    shape = (xt.size(0),) + (1,) * (xt.ndim - 1)
    sigmas, alphas = tabs["sigmas"][t].view(shape), tabs["alphas"][t].view(shape)
    out = y_out
    if u_out is not None:
        out = u_out + guide_scale * (y_out - u_out)
        if guide_rescale is not None:
            ratio = (y_out.flatten(1).std(dim=1) / (out.flatten(1).std(dim=1) + 1e-12)).view(shape)
            out = out * (guide_rescale * ratio + (1 - guide_rescale) * 1.0)
    if prediction_type == "x0":
        x0 = out
    elif prediction_type == "eps":
        x0 = (xt - sigmas * out) / alphas
    else:
        x0 = alphas * xt - sigmas * out
    return x0, (xt - alphas * x0) / sigmas


def gauss_log_sigmas(tabs):
    return torch.sqrt(tabs["sigmas"] ** 2 / (1 - tabs["sigmas"] ** 2)).log()


def gauss_sigma_to_t(tabs, sigma):
    # GaussianDiffusion._sigma_to_t, diffusion_gauss.py:436-456
    ls = gauss_log_sigmas(tabs).to(sigma)
    log_sigma = sigma.log()
    dists = log_sigma - ls[:, None]
    low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=ls.shape[0] - 2)
    high_idx = low_idx + 1
    low, high = ls[low_idx], ls[high_idx]
    w = ((low - log_sigma) / (low - high)).clamp(0, 1)
    t = ((1 - w) * low_idx + w * high_idx).view(sigma.shape)
    return t.unsqueeze(0) if t.ndim == 0 else t


def gauss_t_to_sigma(tabs, t):
    # GaussianDiffusion._t_to_sigma, diffusion_gauss.py:458-464
    t = t.float()
    low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
    ls = gauss_log_sigmas(tabs).to(t)
    log_sigma = (1 - w) * ls[low_idx] + w * ls[high_idx]
    log_sigma[torch.isnan(log_sigma) | torch.isinf(log_sigma)] = float("inf")
    return log_sigma.exp()


def gauss_trailing_sigmas(tabs, steps, t_max, t_min=0):
    # GaussianDiffusion.sample with discretization='trailing', discard_penultimate_step=True
    # (diffusion_gauss.py:318-364): steps+1 trailing timesteps, sigma(t), append 0, drop penultimate
    steps = steps + 1
    ts = torch.arange(t_max, t_min - 1, -((t_max - t_min + 1) / steps)).clamp_(t_min, t_max)
    sig = gauss_t_to_sigma(tabs, torch.as_tensor(ts, dtype=torch.float32))
    sig = torch.cat([sig, sig.new_zeros([1])])
    return torch.cat([sig[:-2], sig[-1:]])


def dpmpp_2m_sde(tabs, noise, model, model_kwargs, sigmas, guide_scale, guide_rescale, prediction_type,
                 eta=1.0, s_noise=1.0, noise_fn=None):
    """sample_dpmpp_2m_sde (midpoint), diffusion_gauss.py:85-142, with model_fn of :303-316 inlined;
    `noise_fn(sigma, sigma_next)` supplies the (third-party torchsde) Brownian increment / sqrt(dt)."""
    x = noise * sigmas[0]
    old_denoised, h_last = None, None
    for i in range(len(sigmas) - 1):
        c_in = 1 / (sigmas[i] ** 2 + 1.) ** 0.5
        t = gauss_sigma_to_t(tabs, sigmas[i]).repeat(len(x)).round().long()
        xin = x * c_in
        y_out = model(xin, t=t, **model_kwargs[0])
        u_out = model(xin, t=t, **model_kwargs[1])
        denoised, _ = gauss_x0_eps(tabs, xin, t, y_out, u_out, guide_scale, guide_rescale, prediction_type)
        if sigmas[i + 1] == 0:
            x = denoised
            h = None
        else:
            tt, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - tt
            eta_h = eta * h
            x = sigmas[i + 1] / sigmas[i] * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * denoised
            if old_denoised is not None:
                r = h_last / h
                x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / r) * (denoised - old_denoised)
            if eta and noise_fn is not None:
                x = x + noise_fn(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise
        old_denoised, h_last = denoised, h
    return x


def gauss_ddim_reverse_loop(tabs, x0, model, model_kwargs, prediction_type, ddim_timesteps, reverse_steps):
    # GaussianDiffusion.ddim_reverse_sample(_loop), diffusion_gauss.py:375-434 (no guidance)
    xt = x0
    stride = reverse_steps // ddim_timesteps
    for step in torch.arange(0, reverse_steps, stride):
        t = torch.full((x0.size(0),), int(step), dtype=torch.long)
        out = model(xt, t=t, **model_kwargs)
        px0, eps = gauss_x0_eps(tabs, xt, t, out, None, None, None, prediction_type)
        s = (t + stride).clamp(0, reverse_steps - 1)
        shape = (xt.size(0),) + (1,) * (xt.ndim - 1)
        a_s = tabs["alphas"][s].view(shape)
        xt = a_s * px0 + torch.sqrt(1 - a_s ** 2) * eps
    return xt


# ------------------------------------------------------------------------------------------
# LCM multistep consistency sampler — PARITY UNPINNED: `diffusers.LCMScheduler` is not under
# /root/reference and not version-pinned; restated from the published algorithm with the engine's ctor
# arguments (inference_videolcm_entrance.py:171) and loop (:228-257).
# ------------------------------------------------------------------------------------------


def lcm_alphas_cumprod(T=1000, beta_start=0.00085, beta_end=0.012, zero_snr=True):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    if zero_snr:
        a = (1.0 - betas).cumprod(0).sqrt()
        a0, aT = a[0].clone(), a[-1].clone()
        a = (a - aT) * (a0 / (a0 - aT))
        ab = a ** 2
        betas = 1.0 - torch.cat([ab[:1], ab[1:] / ab[:-1]])
    return torch.cumprod(1.0 - betas, 0)


def lcm_timesteps(n, T=1000, original_steps=50):
    import numpy as np
    k = T // original_steps
    origin = (np.arange(1, original_steps + 1) * k - 1)[::-1].copy()
    idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
    return [int(v) for v in origin[idx]]


def lcm_sample_loop(ac, timesteps, noise, model, model_kwargs, guidance_scale, step_noise, prediction_type="v_prediction",
                    timestep_scaling=10.0):
    x = noise.clone()
    for i, t in enumerate(timesteps):
        tt = torch.full((x.shape[0],), float(t))
        v = model(x, tt, **model_kwargs[0])
        if guidance_scale is not None:
            u = model(x, tt, **model_kwargs[1])
            v = u + torch.tensor(guidance_scale, dtype=torch.float32) * (v - u)
        a_t = ac[t]
        prev_t = timesteps[i + 1] if i + 1 < len(timesteps) else t
        a_prev = ac[prev_t]
        s = t * timestep_scaling
        c_skip = 0.25 / (s ** 2 + 0.25)
        c_out = s / (s ** 2 + 0.25) ** 0.5
        if prediction_type == "v_prediction":
            x0 = a_t.sqrt() * x - (1 - a_t).sqrt() * v
        else:
            x0 = (x - (1 - a_t).sqrt() * v) / a_t.sqrt()
        den = c_out * x0 + c_skip * x
        x = a_prev.sqrt() * den + (1 - a_prev).sqrt() * step_noise[i] if i != len(timesteps) - 1 else den
    return x


# ------------------------------------------------------------------------------------------------
# OpenCLIP text tower (third-party `open_clip`, not vendored / not pinned by the reference: PARITY UNPINNED against
# the package itself; pinned instead on the independent implementation of the same architecture that IS installed,
# transformers' CLIPTextModelWithProjection, on identical weights: tests/test_oracle.py, <= 2e-6).  Restated from its published architecture around torch.nn.MultiheadAttention — the module
# open_clip's ResidualAttentionBlock wraps — following the reference's call sequence
# (tools/modules/clip_embedder.py:154-161 encode_with_transformer, :55-64 text_transformer_forward).
def clip_text_forward(sd, tokens, heads, layer_idx=1, prefix="model."):
    """tokens int64 [B, L] -> (x [B, L, width] after ln_final, xt [B, embed] = x[eot] @ text_projection)."""
    import torch.nn.functional as F
    g = lambda k: sd[prefix + k].float()
    x = g("token_embedding.weight")[tokens] + g("positional_embedding")             # :155-156
    L = tokens.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)                                # open_clip build_attention_mask
    nl = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(prefix + "transformer.resblocks."))
    x = x.permute(1, 0, 2)                                                           # NLD -> LND (:157)
    d = x.shape[-1]
    for i in range(nl - layer_idx):                                                  # penultimate: skip the last block
        p = f"transformer.resblocks.{i}."
        n = F.layer_norm(x, (d,), g(p + "ln_1.weight"), g(p + "ln_1.bias"), 1e-5)
        a, _ = F.multi_head_attention_forward(n, n, n, d, heads, g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias"),
                                              None, None, False, 0.0, g(p + "attn.out_proj.weight"),
                                              g(p + "attn.out_proj.bias"), training=False, need_weights=False,
                                              attn_mask=mask)
        x = x + a
        n = F.layer_norm(x, (d,), g(p + "ln_2.weight"), g(p + "ln_2.bias"), 1e-5)
        h = F.gelu(F.linear(n, g(p + "mlp.c_fc.weight"), g(p + "mlp.c_fc.bias")))
        x = x + F.linear(h, g(p + "mlp.c_proj.weight"), g(p + "mlp.c_proj.bias"))
    x = x.permute(1, 0, 2)
    x = F.layer_norm(x, (d,), g("ln_final.weight"), g("ln_final.bias"), 1e-5)
    xt = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ g("text_projection")
    return x, xt
