"""TEST INFRASTRUCTURE — generates tests/golden/*.pt by running the REFERENCE's own modules
(imported from /root/reference via oracle/ref_import.py) on seeded synthetic weights/inputs.

The reference has no golden vectors of its own; these fixtures are what pins the
oracle (oracle/torch_ref.py) and the HIP path on the GPU box, where the reference tree does not
exist.  Weights are NOT stored: they are regenerated from `seed` by torch_ref.synth_state_dict
(deterministic CPU generator over the stored key/shape list).

    python -m oracle.make_golden            # tiny fixtures (seconds)
    python -m oracle.make_golden --full     # + full-size UNet forward / VAE frame (minutes, ~12 GB RAM)
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import, torch_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

UNET_TINY = dict(in_dim=4, dim=64, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4],
                 num_heads=2, head_dim=64, num_res_blocks=1, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                 temporal_attention=True, temporal_attn_times=1, use_checkpoint=False,
                 use_fps_condition=False, use_sim_mask=False)
# configs/t2v_train.yaml:32-51 over tools/modules/config.py:96-114
UNET_T2V = dict(in_dim=4, dim=320, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
                num_heads=8, head_dim=64, num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                temporal_attention=True, temporal_attn_times=1, use_checkpoint=False,
                use_fps_condition=False, use_sim_mask=False)
VAE_TINY = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64,
                ch_mult=[1, 2, 4, 4], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
# tools/modules/config.py:118-135
VAE_SD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0,
              video_kernel_size=[3, 1, 1])
DDIM_T2V = dict(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
                mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False,
                noise_strength=0.1)


def _inputs(seed, B, F, H, W, L=77, ctx=1024):
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(B, 4, F, H, W, generator=g)
    y = torch.randn(B, L, ctx, generator=g)
    return x, y


def dummy_model(x, t, y=None, **kw):
    """Cheap deterministic stand-in for the UNet in sampler fixtures."""
    w = torch.tensor([[0.6, -0.2, 0.1, 0.0], [0.1, 0.5, -0.3, 0.2], [-0.2, 0.1, 0.7, 0.1], [0.0, 0.3, -0.1, 0.4]])
    return torch.einsum("oc,bcfhw->bofhw", w.to(x), x) + 0.05 * y.float().mean() \
        + 0.001 * t.float().view(-1, 1, 1, 1, 1)


def dummy_model4(x, t, y=None, **kw):
    """dummy_model for 4-D "image" latents."""
    w = torch.tensor([[0.6, -0.2, 0.1, 0.0], [0.1, 0.5, -0.3, 0.2], [-0.2, 0.1, 0.7, 0.1], [0.0, 0.3, -0.1, 0.4]])
    return torch.einsum("oc,bchw->bohw", w.to(x), x) + 0.05 * y.float().mean() + 0.001 * t.float().view(-1, 1, 1, 1)


def dummy_model2(x, t, y=None, **kw):
    """2 C-channel stand-in (mean prediction | variance fraction) for the learned-variance branches."""
    o = dummy_model4(x, t, y=y)
    return torch.cat([o, torch.tanh(0.7 * o.flip(1)) * 0.8], dim=1)


def make_ddim_branches(R):
    """The sampler options no inference yaml uses, from the reference itself (diffusion_ddim.py:116-241): learned /
    learned_range variances, mean_type 'x_{t-1}', clamp / percentile on x0, classifier guidance (condition_fn) — one
    p_mean_variance, p_sample, ddim_sample and ddim_reverse_sample each, 4-D "image" latents (the reference's percentile
    branch reshapes with view(-1, 1, 1, 1))."""
    g = torch.Generator("cpu").manual_seed(31)
    noise = torch.randn(2, 4, 8, 8, generator=g)
    kw = [dict(y=torch.randn(2, 7, 16, generator=g)), dict(y=torch.randn(2, 7, 16, generator=g))]
    t = torch.tensor([981, 21])
    cond = lambda x, t, **k: 0.3 * torch.tanh(x)
    base = {k: v for k, v in DDIM_T2V.items() if k not in ("var_type", "mean_type")}
    res = dict(cfg=base, noise=noise, kw=kw, t=t, cases={})
    cases = {
        "learned_range_eps": dict(var_type="learned_range", mean_type="eps", model=2),
        "learned_v": dict(var_type="learned", mean_type="v", model=2),
        "xtm1_fixed_small": dict(var_type="fixed_small", mean_type="x_{t-1}", model=1),
        "v_clamp": dict(var_type="fixed_small", mean_type="v", model=1, clamp=0.8),
        "v_percentile": dict(var_type="fixed_large", mean_type="v", model=1, percentile=0.9),
        "v_condfn": dict(var_type="fixed_small", mean_type="v", model=1, cond=True),
    }
    for name, c in cases.items():
        d = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **base, var_type=c["var_type"], mean_type=c["mean_type"]))
        mdl = dummy_model2 if c["model"] == 2 else dummy_model4
        opt = dict(clamp=c.get("clamp"), percentile=c.get("percentile"))
        cf = cond if c.get("cond") else None
        out = {}
        out["pmv"] = [v.clone() for v in d.p_mean_variance(noise.clone(), t, mdl, kw, guide_scale=9.0, **opt)]
        torch.manual_seed(7)
        out["p_sample"] = [v.clone() for v in d.p_sample(noise.clone(), t, mdl, kw if cf is None else kw[0],
                                                         condition_fn=cf, guide_scale=9.0 if cf is None else None, **opt)]
        torch.manual_seed(8)
        out["ddim"] = [v.clone() for v in d.ddim_sample(noise.clone(), t, mdl, kw if cf is None else kw[0], condition_fn=cf,
                                                         guide_scale=9.0 if cf is None else None, ddim_timesteps=50, eta=0.5, **opt)]
        out["reverse"] = [v.clone() for v in d.ddim_reverse_sample(noise.clone(), t, mdl, kw, guide_scale=9.0,
                                                                   ddim_timesteps=50, **opt)]
        res["cases"][name] = dict(c, **out)
    torch.save(res, os.path.join(GOLD, "ddim_branches.pt"))
    print("ddim_branches", list(res["cases"]))


I2V_TINY = dict(in_dim=4, dim=64, y_dim=1024, context_dim=1024, concat_dim=4, out_dim=4, dim_mult=[1, 2, 4],
                num_heads=2, head_dim=64, num_res_blocks=1, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                temporal_attention=True, temporal_attn_times=1, use_checkpoint=False, use_fps_condition=False,
                use_sim_mask=False, training=False)


def make_i2vgen(R):
    """tiny UNetSD_I2VGen: local-image concat branch, 64 local + 4 global image tokens, fps embedding."""
    ref = R["MODEL"].build(dict(type="UNetSD_I2VGen", **I2V_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=4), strict=True)
    g = torch.Generator("cpu").manual_seed(11)
    x = torch.randn(2, 4, 4, 16, 8, generator=g)
    y = torch.randn(2, 77, 1024, generator=g)
    image = torch.randn(2, 1, 1024, generator=g)
    local_image = torch.randn(2, 4, 16, 8, generator=g)
    fps = torch.tensor([8, 16])
    t = torch.tensor([981, 401])
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # mask_pos hard-codes .cuda() (unet_i2vgen.py:284)
    try:
        with torch.no_grad():
            out = ref(x, t, y=y, image=image, local_image=local_image, fps=fps)
    finally:
        torch.Tensor.cuda = _cuda
    torch.save(dict(cfg=I2V_TINY, seed=4, shapes=shapes, x=x, t=t, y=y, image=image, local_image=local_image,
                    fps=fps, out=out), os.path.join(GOLD, "unet_i2vgen_tiny.pt"))
    print("unet_i2vgen_tiny", tuple(out.shape), float(out.std()))


LCM_TINY = dict(in_dim=4, dim=64, y_dim=1024, context_dim=1024, concat_dim=8, out_dim=4, dim_mult=[1, 2, 4],
                num_heads=2, head_dim=64, num_res_blocks=1, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                temporal_attention=True, temporal_attn_times=1, use_checkpoint=False, use_fps_condition=False,
                use_sim_mask=False, num_tokens=4, training=False)


def make_videolcm(R):
    """tiny UNetSD_VideoLCM, text-only composition (configs/videolcm_t2v_infer.yaml:67), float timesteps."""
    import types
    cfg = types.SimpleNamespace(video_compositions=["text"], resolution=[64, 128])
    ref = R["MODEL"].build(dict(type="UNetSD_VideoLCM", config=cfg, **LCM_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=5), strict=True)
    x, y = _inputs(13, 2, 4, 16, 8)
    t = torch.tensor([759.0, 259.0])                      # the LCM engine passes float timesteps
    with torch.no_grad():
        out = ref(x, t, y=y)
    torch.save(dict(cfg=LCM_TINY, seed=5, shapes=shapes, x=x, t=t, y=y, out=out),
               os.path.join(GOLD, "unet_videolcm_tiny.pt"))
    print("unet_videolcm_tiny", tuple(out.shape), float(out.std()))


def make_tft2v(R):
    """tiny UNetSD_TFT2V with the compositions of configs/tft2v_t2v_infer.yaml:65 (['text', 'image'])."""
    import types
    cfg = types.SimpleNamespace(video_compositions=["text", "image"], resolution=[64, 128])
    ref = R["MODEL"].build(dict(type="UNetSD_TFT2V", config=cfg, **LCM_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=6), strict=True)
    x, y = _inputs(17, 2, 4, 16, 8)
    image = torch.randn(2, 1, 1024, generator=torch.Generator("cpu").manual_seed(18))
    t = torch.tensor([981, 401])
    with torch.no_grad():
        out = ref(x, t, y=y, image=image)
    torch.save(dict(cfg=LCM_TINY, seed=6, shapes=shapes, x=x, t=t, y=y, image=image, out=out),
               os.path.join(GOLD, "unet_tft2v_tiny.pt"))
    print("unet_tft2v_tiny", tuple(out.shape), float(out.std()))


def make_histogram(R):
    """tiny UNetSD_VideoLCM with a per-frame context token: video_compositions ['text', 'histogram', 'canny']."""
    import types
    comps = ["text", "histogram", "canny"]
    cfg = types.SimpleNamespace(video_compositions=comps, resolution=[64, 128])
    ref = R["MODEL"].build(dict(type="UNetSD_VideoLCM", config=cfg, **LCM_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=8), strict=True)
    x, y = _inputs(23, 2, 3, 16, 8)
    g = torch.Generator("cpu").manual_seed(24)
    hist = torch.rand(2, 3, 156, generator=g)
    canny = torch.randn(2, 1, 3, 128, 64, generator=g).half().float()
    t = torch.tensor([759.0, 259.0])
    with torch.no_grad():
        out = ref(x, t, y=y, histogram=hist, canny=canny)
    torch.save(dict(cfg=LCM_TINY, comps=comps, resolution=[64, 128], seed=8, shapes=shapes, x=x, t=t, y=y,
                    histogram=hist, canny=canny.half(), out=out), os.path.join(GOLD, "unet_histogram_tiny.pt"))
    print("unet_histogram_tiny", tuple(out.shape), float(out.std()))


VCOMPOSER = ["text", "mask", "depthmap", "sketch", "motion", "image", "local_image", "single_sketch"]


def make_vcomposer(R):
    """tiny UNetSD_TFT2V with the composition list of configs/tft2v_vcomposer_infer.yaml:74."""
    import types
    cfg = types.SimpleNamespace(video_compositions=list(VCOMPOSER), resolution=[64, 128])
    ref = R["MODEL"].build(dict(type="UNetSD_TFT2V", config=cfg, **LCM_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=7), strict=True)
    x, y = _inputs(19, 1, 3, 16, 8)
    g = torch.Generator("cpu").manual_seed(20)
    mk = lambda c: torch.randn(1, c, 3, 128, 64, generator=g).half().float()     # stored as fp16: exactly representable
    conds = dict(depth=mk(1), sketch=mk(1), single_sketch=mk(1), motion=mk(2), local_image=mk(3), masked=mk(4))
    image = torch.randn(1, 1, 1024, generator=g)
    t = torch.tensor([601])
    with torch.no_grad():
        out = ref(x, t, y=y, image=image, **conds)
    torch.save(dict(cfg=LCM_TINY, comps=list(VCOMPOSER), resolution=[64, 128], seed=7, shapes=shapes, x=x, t=t, y=y,
                    image=image, conds={k: v.half() for k, v in conds.items()}, out=out),
               os.path.join(GOLD, "unet_vcomposer_tiny.pt"))
    print("unet_vcomposer_tiny", tuple(out.shape), float(out.std()))


def vcomposer_conds(seed, F, H, W):
    """the six spatial conditions of the vcomposer list at pixel resolution, regenerated from a seed (the fixture cannot
    carry 350 MB of maps); values are fp16-representable.  Order = the order tests/full_cases.py regenerates them in."""
    g = torch.Generator("cpu").manual_seed(seed)
    mk = lambda c: torch.randn(1, c, F, H, W, generator=g).half().float()
    conds = dict(depth=mk(1), sketch=mk(1), single_sketch=mk(1), motion=mk(2), local_image=mk(3), masked=mk(4))
    image = torch.randn(1, 1, 1024, generator=g)
    return conds, image


def make_vcomposer_full(R):
    """FULL-WIDTH UNetSD_TFT2V with the composition list of configs/tft2v_vcomposer_infer.yaml:74 at BASELINE config 5's
    first-stage shape: 32 frames 896 x 512, latent [1,4,32,64,112] (78 TFLOP on the CPU) — six spatial condition stems at
    pixel resolution summed into the concat channels, image token, 32-frame temporal attention (the flash kernel's path:
    more than 16 frames)."""
    import time
    import types
    cfgm = dict(UNET_T2V, concat_dim=8, num_tokens=4, training=False)
    cfg = types.SimpleNamespace(video_compositions=list(VCOMPOSER), resolution=[896, 512])
    ref = R["MODEL"].build(dict(type="UNetSD_TFT2V", config=cfg, **cfgm)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=0), strict=True)
    g = torch.Generator("cpu").manual_seed(8896)
    x = torch.randn(1, 4, 32, 64, 112, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    conds, image = vcomposer_conds(8897, 32, 512, 896)
    t = torch.tensor([601])
    t0 = time.time()
    with torch.no_grad():
        out = ref(x, t, y=y, image=image, **conds)
    print("unet_vcomposer_full: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    torch.save(dict(cfg=cfgm, comps=list(VCOMPOSER), resolution=[896, 512], seed=0, shapes=shapes, input_seed=8896,
                    cond_seed=8897, t=t, out_sub=_sub(out), out_norm=float(out.norm())),
               os.path.join(GOLD, "unet_vcomposer_full.pt"))


@torch.no_grad()
def make_vae_full2(R):
    """Full-size SD AutoencoderKL fixtures the r02 set lacked: `encode` moments + the stochastic
    `encode_firsr_stage` sample at 256x448, and one 720x1280 frame through encode and decode (the SR600 stage runs 32
    such frames, inference_tft2v_sr600_entrance.py:118; i2vgen encodes one, inference_i2vgen_entrance.py:193).
    720p tensors are stored sub-sampled with their norms."""
    import time
    vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=VAE_SD, embed_dim=4)).eval()
    vshapes = torch_ref.shapes_of(vae)
    vae.load_state_dict(torch_ref.synth_state_dict(vshapes, seed=0), strict=True)
    out = dict(ddconfig=VAE_SD, seed=0, shapes=vshapes)
    with torch.no_grad():
        g = torch.Generator("cpu").manual_seed(8890)
        img = torch.randn(1, 3, 256, 448, generator=g).clamp(-1, 1)
        t0 = time.time()
        mom = vae.encode(img).parameters
        torch.manual_seed(5)
        zs = vae.encode_firsr_stage(img, 0.18215)
        print("vae 256x448 encode x2: %.1f s" % (time.time() - t0))
        out.update(enc256_seed=8890, enc256_moments=mom.contiguous(), enc256_sample_seed=5, enc256_z=zs.contiguous())
        g = torch.Generator("cpu").manual_seed(8891)
        img = torch.randn(1, 3, 720, 1280, generator=g).clamp(-1, 1)
        t0 = time.time()
        mom = vae.encode(img).parameters                                        # [1, 8, 90, 160]
        print("vae 720x1280 encode: %.1f s" % (time.time() - t0))
        out.update(enc720_seed=8891, enc720_moments_sub=mom[:, :, ::2, ::2].contiguous(), enc720_norm=float(mom.norm()))
        z = torch.randn(1, 4, 90, 160, generator=g)
        t0 = time.time()
        dec = vae.decode(z)                                                      # [1, 3, 720, 1280]
        print("vae 720x1280 decode: %.1f s" % (time.time() - t0))
        out.update(dec720_sub=dec[:, :, ::8, ::8].contiguous(), dec720_norm=float(dec.norm()))
    torch.save(out, os.path.join(GOLD, "vae_sd_full2.pt"))


def make_i2vgen_full(R):
    """The reference's FULL-WIDTH UNetSD_I2VGen (dim 320, 1420 M parameters) at BASELINE config 3's real latent
    [1,4,16,88,160] (i2vgen_xl_infer.yaml: 16 frames 1280x704), 77 + 64 + 4 context tokens, local-image stem channels,
    fps embedding: 88 TFLOP on the CPU.  Output stored sub-sampled (frames ::2, rows / cols ::4) with its norm."""
    import time
    cfg = dict(UNET_T2V, concat_dim=4, upper_len=128, default_fps=8, training=False)      # i2vgen_xl_train.yaml:32-51
    ref = R["MODEL"].build(dict(type="UNetSD_I2VGen", **cfg)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=0), strict=True)
    g = torch.Generator("cpu").manual_seed(8892)
    x = torch.randn(1, 4, 16, 88, 160, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    image = torch.randn(1, 1, 1024, generator=g)
    local_image = torch.randn(1, 4, 88, 160, generator=g)
    fps = torch.tensor([8])
    t = torch.tensor([601])
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # mask_pos hard-codes .cuda() (unet_i2vgen.py:284)
    try:
        with torch.no_grad():
            t0 = time.time()
            out = ref(x, t, y=y, image=image, local_image=local_image, fps=fps)
            print("unet_i2vgen full forward: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    finally:
        torch.Tensor.cuda = _cuda
    torch.save(dict(cfg=cfg, seed=0, shapes=shapes, input_seed=8892, t=t, fps=fps,
                    out_sub=out[:, :, ::2, ::4, ::4].contiguous(), out_norm=float(out.norm())),
               os.path.join(GOLD, "unet_i2vgen_full.pt"))


def make_i2vgen_full_b(R):
    """r05 (VERDICT r04 weak #1b: the I2VGen margin rested on ONE (weights, input, t) triple): a second full-width
    UNetSD_I2VGen fixture at the same real latent [1,4,16,88,160] — heavy-tailed (Student-t, nu = 4) weights of seed 1,
    another input, t = 301, fps = 16.  Same storage as make_i2vgen_full (output sub-sampled + its norm)."""
    import time
    cfg = dict(UNET_T2V, concat_dim=4, upper_len=128, default_fps=8, training=False)      # i2vgen_xl_train.yaml:32-51
    ref = R["MODEL"].build(dict(type="UNetSD_I2VGen", **cfg)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=1, recipe="student4"), strict=True)
    g = torch.Generator("cpu").manual_seed(8894)
    x = torch.randn(1, 4, 16, 88, 160, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    image = torch.randn(1, 1, 1024, generator=g)
    local_image = torch.randn(1, 4, 88, 160, generator=g)
    fps = torch.tensor([16])
    t = torch.tensor([301])
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # mask_pos hard-codes .cuda() (unet_i2vgen.py:284)
    try:
        with torch.no_grad():
            t0 = time.time()
            out = ref(x, t, y=y, image=image, local_image=local_image, fps=fps)
            print("unet_i2vgen full_b forward: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    finally:
        torch.Tensor.cuda = _cuda
    torch.save(dict(cfg=cfg, seed=1, recipe="student4", shapes=shapes, input_seed=8894, t=t, fps=fps,
                    out_sub=out[:, :, ::2, ::4, ::4].contiguous(), out_norm=float(out.norm())),
               os.path.join(GOLD, "unet_i2vgen_full_b.pt"))


def _sub(out):
    """full-size outputs are stored sub-sampled (frames ::2, rows / cols ::4) with their norm"""
    return out[:, :, ::2, ::4, ::4].contiguous()


def make_t2v_extra(R):
    """r04: two more full-size t2v fixtures next to unet_t2v_full.pt (seed 0, Gaussian weights, t = 981) so that the 1e-3
    claim of precision="mixed" does not rest on one (weights, input, t) triple:
    unet_t2v_full_b.pt — a SECOND weight seed with heavy-tailed (Student-t, nu = 4) matrices, t = 741;
    unet_t2v_full_c.pt — the headline weights at a mid-trajectory t = 501 on another input."""
    import time
    ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **UNET_T2V)).eval()
    shapes = torch_ref.shapes_of(ref)
    for tag, seed, recipe, input_seed, tval in (("b", 1, "student4", 8890, 741), ("c", 0, "gauss", 8891, 501)):
        ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=seed, recipe=recipe), strict=True)
        g = torch.Generator("cpu").manual_seed(input_seed)
        x = torch.randn(1, 4, 16, 32, 56, generator=g)
        y = torch.randn(1, 77, 1024, generator=g)
        t = torch.tensor([tval])
        t0 = time.time()
        with torch.no_grad():
            out = ref(x, t, y=y)
        print("unet_t2v_full_%s: %.1f s, std %.4f" % (tag, time.time() - t0, float(out.std())))
        torch.save(dict(cfg=UNET_T2V, seed=seed, recipe=recipe, shapes=shapes, input_seed=input_seed, t=t, out=out,
                        out_norm=float(out.norm())), os.path.join(GOLD, "unet_t2v_full_%s.pt" % tag))


def make_videolcm_full(R):
    """FULL-WIDTH UNetSD_VideoLCM (text-only composition, configs/videolcm_t2v_infer.yaml:67) at BASELINE config 4's
    latent [1,4,16,32,56], float timestep as the LCM engine passes it (unet_videolcm.py:541-784)."""
    import time
    import types
    cfgm = dict(UNET_T2V, concat_dim=8, num_tokens=4, training=False)
    cfg = types.SimpleNamespace(video_compositions=["text"], resolution=[448, 256])
    ref = R["MODEL"].build(dict(type="UNetSD_VideoLCM", config=cfg, **cfgm)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=0), strict=True)
    g = torch.Generator("cpu").manual_seed(8893)
    x = torch.randn(1, 4, 16, 32, 56, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    t = torch.tensor([759.0])
    t0 = time.time()
    with torch.no_grad():
        out = ref(x, t, y=y)
    print("unet_videolcm_full: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    torch.save(dict(cfg=cfgm, comps=["text"], resolution=[448, 256], seed=0, shapes=shapes, input_seed=8893, t=t,
                    out=out, out_norm=float(out.norm())), os.path.join(GOLD, "unet_videolcm_full.pt"))


def make_tft2v_full(R):
    """FULL-WIDTH UNetSD_TFT2V with the compositions of configs/tft2v_t2v_infer.yaml:65 (['text', 'image']) at BASELINE
    config 5's first-stage latent [1,4,16,64,112] (16 x 896 x 512; unet_tf2tv.py:538-777): 39 TFLOP on the CPU."""
    import time
    import types
    cfgm = dict(UNET_T2V, concat_dim=8, num_tokens=4, training=False)
    cfg = types.SimpleNamespace(video_compositions=["text", "image"], resolution=[896, 512])
    ref = R["MODEL"].build(dict(type="UNetSD_TFT2V", config=cfg, **cfgm)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=0), strict=True)
    g = torch.Generator("cpu").manual_seed(8894)
    x = torch.randn(1, 4, 16, 64, 112, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    image = torch.randn(1, 1, 1024, generator=g)
    t = torch.tensor([401])
    t0 = time.time()
    with torch.no_grad():
        out = ref(x, t, y=y, image=image)
    print("unet_tft2v_full: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    torch.save(dict(cfg=cfgm, comps=["text", "image"], resolution=[896, 512], seed=0, shapes=shapes, input_seed=8894,
                    t=t, out_sub=_sub(out), out_norm=float(out.norm())), os.path.join(GOLD, "unet_tft2v_full.pt"))


def make_sr600_full(R):
    """FULL-WIDTH UNetSD_SR600 at BASELINE config 5's second-stage latent [1,4,32,90,160] (32 x 1280 x 720;
    unet_sr600.py:220-299): 90 -> 45 -> 23 -> 12 rows, 14 400-token spatial attention over 32 frames, FreeU +
    Fourier skip filter; 186 TFLOP on the CPU."""
    import time
    cfgm = dict(UNET_T2V, use_scale_shift_norm=True, inpainting=True)
    cfgm.pop("use_fps_condition", None)
    ref = R["MODEL"].build(dict(type="UNetSD_SR600", **cfgm)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=0), strict=True)
    g = torch.Generator("cpu").manual_seed(8895)
    x = torch.randn(1, 4, 32, 90, 160, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    t = torch.tensor([699])
    t0 = time.time()
    with _no_cuda(), torch.no_grad():
        out = ref(x.clone(), t, y)
    print("unet_sr600_full: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
    torch.save(dict(cfg=cfgm, seed=0, shapes=shapes, input_seed=8895, t=t, out_sub=_sub(out),
                    out_norm=float(out.norm())), os.path.join(GOLD, "unet_sr600_full.pt"))


def _no_cuda():
    """context: the reference hard-codes .cuda() in a few places (unet_i2vgen.py:284, unet_sr600.py:38)"""
    import contextlib

    @contextlib.contextmanager
    def cm():
        _cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            yield
        finally:
            torch.Tensor.cuda = _cuda
    return cm()


def make_odd(R):
    """BASELINE configs 3-5 shapes at reduced width: the 720p latent grid of the sr600 / i2vgen stages (H = 90:
    90 -> 45 -> 23 -> 12 rows through pad-(2,1) downsamples and cropped upsamples, 14400-token spatial attention,
    S % 64 != 0 at the lower levels; 88 x 160 with a 145-token context for i2vgen) and an odd-sized VAE frame."""
    SR = dict(UNET_TINY, dim_mult=[1, 2, 4, 4], use_scale_shift_norm=True, inpainting=True)
    SR.pop("use_fps_condition", None)
    sr = R["MODEL"].build(dict(type="UNetSD_SR600", **SR)).eval()
    shapes = torch_ref.shapes_of(sr)
    sr.load_state_dict(torch_ref.synth_state_dict(shapes, seed=13), strict=True)
    g = torch.Generator("cpu").manual_seed(19)
    x = torch.randn(1, 4, 4, 90, 160, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    t = torch.tensor([699])
    with _no_cuda(), torch.no_grad():
        out = sr(x.clone(), t, y)
    torch.save(dict(cfg=SR, seed=13, shapes=shapes, x=x, t=t, y=y, out=out), os.path.join(GOLD, "unet_sr600_odd.pt"))
    print("unet_sr600_odd", tuple(out.shape), float(out.std()))

    ref = R["MODEL"].build(dict(type="UNetSD_I2VGen", **I2V_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    ref.load_state_dict(torch_ref.synth_state_dict(shapes, seed=14), strict=True)
    g = torch.Generator("cpu").manual_seed(23)
    x = torch.randn(1, 4, 4, 88, 160, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    image = torch.randn(1, 1, 1024, generator=g)
    local_image = torch.randn(1, 4, 88, 160, generator=g)
    fps = torch.tensor([8])
    t = torch.tensor([981])
    with _no_cuda(), torch.no_grad():
        out = ref(x, t, y=y, image=image, local_image=local_image, fps=fps)
    torch.save(dict(cfg=I2V_TINY, seed=14, shapes=shapes, x=x, t=t, y=y, image=image, local_image=local_image,
                    fps=fps, out=out), os.path.join(GOLD, "unet_i2vgen_odd.pt"))
    print("unet_i2vgen_odd", tuple(out.shape), float(out.std()))

    vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=VAE_TINY, embed_dim=4)).eval()
    vshapes = torch_ref.shapes_of(vae)
    vae.load_state_dict(torch_ref.synth_state_dict(vshapes, seed=15), strict=True)
    g = torch.Generator("cpu").manual_seed(29)
    z = torch.randn(1, 4, 15, 25, generator=g)             # 375 latent pixels: the attention's 64-column padding
    img = torch.randn(1, 3, 120, 200, generator=g)
    with torch.no_grad():
        dec = vae.decode(z)
        mom = vae.encode(img).parameters
    torch.save(dict(ddconfig=VAE_TINY, seed=15, shapes=vshapes, z=z, img=img, dec=dec, moments=mom),
               os.path.join(GOLD, "vae_odd.pt"))
    print("vae_odd", tuple(dec.shape), tuple(mom.shape))


def make_trajectory(R, full):
    """a whole 50-step DDIM CFG trajectory of the reference (its own DiffusionDDIM driving its own
    UNetSD_T2VBase, fp32 CPU) on the tiny fixture — final x0 and snapshots of x_t along the way, so the HIP path's
    per-step drift can be reported — and, with --full, ONE full-size CFG step (2 forwards of the 1411 M model)."""
    g = torch.load(os.path.join(GOLD, "unet_tiny.pt"), weights_only=False)
    m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **dict(DDIM_T2V, noise_strength=0.0)))
    gen = torch.Generator("cpu").manual_seed(8888)
    noise = torch.randn(g["x"].shape, generator=gen)
    y_u = torch.randn(g["y"].shape, generator=gen)
    kw = [dict(y=g["y"]), dict(y=y_u)]
    steps = (1 + torch.arange(0, 1000, 20)).clamp(0, 999).flip(0)
    snaps, xt = {}, noise.clone()
    with torch.no_grad():
        for i, step in enumerate(steps):
            t = torch.full((noise.shape[0],), int(step), dtype=torch.long)
            xt, _ = diff.ddim_sample(xt, t, m, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
            if i in (0, 4, 9, 19, 29, 39, 49):
                snaps[i] = xt.clone()
    # the same trajectory under the reference's own production arithmetic (amp.autocast fp16 / bf16 around the model,
    # fp32 sampler state: inference_text2video_entrance.py:197-206) — the drift yardstick
    yard = {}
    for dn, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        xa, dev_ = noise.clone(), {}
        with torch.no_grad():
            for i, step in enumerate(steps):
                t = torch.full((noise.shape[0],), int(step), dtype=torch.long)
                with torch.autocast("cpu", dtype=dt):
                    xa, _ = diff.ddim_sample(xa, t, m, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
                xa = xa.float()
                if i in snaps:
                    dev_[i] = float((xa - snaps[i]).norm() / snaps[i].norm())
        yard[dn] = dev_
        print("autocast", dn, dev_, flush=True)
    torch.save(dict(noise=noise, y_u=y_u, snaps=snaps, guide_scale=9.0, ddim_timesteps=50, cfg=dict(DDIM_T2V, noise_strength=0.0),
                    autocast_drift=yard),
               os.path.join(GOLD, "ddim_traj_tiny.pt"))
    print("ddim_traj_tiny", {k: float(v.std()) for k, v in snaps.items()})
    if full:
        g = torch.load(os.path.join(GOLD, "unet_t2v_full.pt"), weights_only=False)
        m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
        m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
        gen = torch.Generator("cpu").manual_seed(g["input_seed"])
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        y = torch.randn(1, 77, 1024, generator=gen)
        y_u = torch.randn(1, 77, 1024, generator=gen)
        t = torch.tensor([981])
        with torch.no_grad():
            xt1, x0 = diff.ddim_sample(x, t, m, [dict(y=y), dict(y=y_u)], guide_scale=9.0, ddim_timesteps=50, eta=0.0)
        torch.save(dict(input_seed=g["input_seed"], t=t, xt1=xt1, x0=x0), os.path.join(GOLD, "ddim_step_full.pt"))
        print("ddim_step_full", float(xt1.std()), float(x0.std()))


def make_t2v_traj6(R):
    """r05 (VERDICT r04 missing #4 / SURVEY §8c "per-step drift"): the FIRST SIX steps of the 50-step CFG DDIM trajectory
    of the reference at full size — its own DiffusionDDIM (diffusion_ddim.py:208-254) driving its own UNetSD_T2VBase
    (1411 M parameters, headline weights of unet_t2v_full.pt) in fp32 on the CPU, guidance 9, eta 0, from seeded noise:
    12 forwards.  x_t after every step is stored whole (459 KB each), so the GPU test can report the drift of the
    benchmarked precision mode step by step through the public ddim_sample_loop-style API."""
    g = torch.load(os.path.join(GOLD, "unet_t2v_full.pt"), weights_only=False)
    m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **dict(DDIM_T2V, noise_strength=0.0)))
    gen = torch.Generator("cpu").manual_seed(8888)                  # the engines' seed (t2v_infer.yaml:14)
    noise = torch.randn(1, 4, 16, 32, 56, generator=gen)
    y = torch.randn(1, 77, 1024, generator=gen)
    y_u = torch.randn(1, 77, 1024, generator=gen)
    steps = (1 + torch.arange(0, 1000, 20)).clamp(0, 999).flip(0)[:6]
    xs, x0s, xt = [], [], noise.clone()
    import time
    with torch.no_grad():
        for step in steps:
            t0 = time.time()
            t = torch.full((1,), int(step), dtype=torch.long)
            xt, x0 = diff.ddim_sample(xt, t, m, [dict(y=y), dict(y=y_u)], guide_scale=9.0, ddim_timesteps=50, eta=0.0)
            xs.append(xt.clone())
            x0s.append(x0.clone())
            print("traj6 step t=%d: x_t std %.4f, x0 std %.4f (%.0f s)" % (int(step), float(xt.std()), float(x0.std()),
                                                                          time.time() - t0), flush=True)
    torch.save(dict(noise_seed=8888, steps=steps, xt=torch.stack(xs), x0=torch.stack(x0s), guide_scale=9.0, ddim_timesteps=50,
                    model_fixture="unet_t2v_full.pt", cfg=dict(DDIM_T2V, noise_strength=0.0)),
               os.path.join(GOLD, "ddim_traj6_full.pt"))


def make_ddpm(R):
    """The reference's ancestral sampler and closed-form q(.) helpers (diffusion_ddim.py:99-145) with the dummy model:
    three p_sample steps (CFG, t = 999 / 500 / 0), a whole 1000-step p_sample_loop, one stochastic DDIM step (eta = 0.7),
    both variance types.  RNG: torch.manual_seed on CPU — the sampler draws randn_like(xt) itself."""
    g = torch.Generator("cpu").manual_seed(31)
    noise = torch.randn(2, 4, 4, 8, 8, generator=g)
    x0 = torch.randn(2, 4, 4, 8, 8, generator=g)
    kw = [dict(y=torch.randn(2, 77, 16, generator=g)), dict(y=torch.randn(2, 77, 16, generator=g))]
    out = dict(noise=noise, x0=x0, kw=kw, cfg=DDIM_T2V)
    for vt in ("fixed_small", "fixed_large"):
        diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **dict(DDIM_T2V, var_type=vt)))
        t = torch.tensor([999, 500])
        torch.manual_seed(3)
        steps = [diff.p_sample(noise.clone(), tt, dummy_model, kw, guide_scale=9.0)
                 for tt in (torch.tensor([999, 500]), torch.tensor([1, 0]))]
        pmv = diff.p_mean_variance(noise.clone(), t, dummy_model, kw, guide_scale=9.0)
        torch.manual_seed(4)
        loop = diff.p_sample_loop(noise.clone(), dummy_model, kw[0], guide_scale=None)
        out[vt] = dict(steps=steps, pmv=pmv, loop=loop)
    diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **DDIM_T2V))
    t = torch.tensor([981, 21])
    torch.manual_seed(5)
    out["ddim_eta"] = diff.ddim_sample(noise.clone(), t, dummy_model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.7)
    out["q_mean_variance"] = diff.q_mean_variance(x0, t)
    out["q_posterior"] = diff.q_posterior_mean_variance(x0, noise, t)
    torch.manual_seed(6)
    out["q_sample"] = diff.q_sample(x0, t)
    torch.save(out, os.path.join(GOLD, "ddpm.pt"))
    print("ddpm", {k: float(v["loop"].std()) for k, v in out.items() if isinstance(v, dict) and "loop" in v})


def make_blocks(R):
    """Per-block goldens ("no per-block (ResBlock / ST / TT) GPU golden"): forward hooks on EVERY
    ResBlock (incl. its TemporalConvBlock_v2), SpatialTransformer, TemporalTransformer, Downsample and Upsample of the
    reference's tiny UNetSD_T2VBase record each block's input and output.  The fixture stores every output once (as
    [B*F, C, H, W] fp32) plus, per block, where its input came from: the previous block's output, or — decoder
    ResBlocks — that concatenated with an earlier output (unet_t2v.py:262-264).  A test can then feed each block of
    the HIP model the REFERENCE's input and compare with the reference's output: errors do not accumulate along the
    depth, a wrong block is named."""
    g = torch.load(os.path.join(GOLD, "unet_tiny.pt"), weights_only=False)
    m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    B, F, H, W = 2, 4, 8, 8
    gen = torch.Generator("cpu").manual_seed(77)
    x = torch.randn(B, 4, F, H, W, generator=gen)
    y = torch.randn(B, 77, 1024, generator=gen)
    t = torch.tensor([741, 61])
    kinds = ("ResBlock", "SpatialTransformer", "TemporalTransformer", "Downsample", "Upsample")
    canon = lambda v: v.permute(0, 2, 1, 3, 4).reshape(-1, v.shape[1], *v.shape[3:]) if v.dim() == 5 else v
    recs, outs = [], []

    def hook(name):
        def fn(mod, args, out):
            xin, o = canon(args[0]).float(), canon(out).float()
            src = None
            for j in range(len(outs) - 1, -1, -1):                      # whole input = an earlier output
                if outs[j].shape == xin.shape and torch.equal(outs[j], xin):
                    src = (j,)
                    break
            if src is None and outs:                                    # cat(previous output, an earlier output)
                c1 = outs[-1].shape[1]
                if xin.shape[1] > c1 and outs[-1].shape[2:] == xin.shape[2:] and torch.equal(outs[-1], xin[:, :c1]):
                    for j in range(len(outs) - 1, -1, -1):
                        if outs[j].shape == xin[:, c1:].shape and torch.equal(outs[j], xin[:, c1:]):
                            src = (len(outs) - 1, j)
                            break
            recs.append(dict(name=name, kind=type(mod).__name__, src=src, x=None if src else xin.clone(),
                             H=xin.shape[2], W=xin.shape[3]))
            outs.append(o.clone())
        return fn

    hs = [mod.register_forward_hook(hook(n)) for n, mod in m.named_modules() if type(mod).__name__ in kinds]
    with torch.no_grad():
        out = m(x, t, y=y)
    for h in hs:
        h.remove()
    assert sum(r["x"] is not None for r in recs) == 1, [r["name"] for r in recs if r["x"] is not None]
    torch.save(dict(x=x, y=y, t=t, out=out, recs=recs, outs=outs, B=B, F=F), os.path.join(GOLD, "unet_blocks_tiny.pt"))
    print("blocks", len(recs), [(r["name"], r["kind"], r["src"]) for r in recs][:8], "...")


def make_vae_blocks(R):
    """Per-block goldens of the reference's tiny AutoencoderKL (encode of the fixture image + decode of the fixture
    latent): every ResnetBlock, AttnBlock, Upsample and Downsample with its input and output — all blocks are
    sequential, so a block's input is the previous record's output (or the stem's output, stored)."""
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=g["ddconfig"], embed_dim=4)).eval()
    vae.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    kinds = ("ResnetBlock", "AttnBlock", "Upsample", "Downsample")
    recs, outs = [], []

    def hook(name):
        def fn(mod, args, out):
            xin = args[0].float()
            src = len(outs) - 1 if outs and outs[-1].shape == xin.shape and torch.equal(outs[-1], xin) else None
            recs.append(dict(name=name, kind=type(mod).__name__, src=src, x=None if src is not None else xin.clone()))
            outs.append(out.float().clone())
        return fn

    hs = [m.register_forward_hook(hook(n)) for n, m in vae.named_modules() if type(m).__name__ in kinds]
    gen = torch.Generator("cpu").manual_seed(5)
    img = torch.randn(1, 3, 48, 32, generator=gen)          # small on purpose: 24 blocks x (in, out) stay ~2 MB
    z = torch.randn(1, 4, 6, 4, generator=gen)
    with torch.no_grad():
        vae.encode(img)
        vae.decode(z)
    for h in hs:
        h.remove()
    torch.save(dict(recs=recs, outs=outs, img=img, z=z), os.path.join(GOLD, "vae_blocks_tiny.pt"))
    print("vae blocks", len(recs), [(r["name"], r["kind"], r["src"]) for r in recs if r["src"] is None])


def make_yardstick(R, full):
    """How far the reference's OWN mixed-precision arithmetic (amp.autocast, the mode its engines run:
    `use_fp16: True`, inference_text2video_entrance.py:197) lands from its fp32 forward on the fixtures' inputs —
    the yardstick the 16-bit HIP path is held to (tests/test_gpu_model.py)."""
    import json
    res = {}

    def measure(name, model, ref, call):
        for dn, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            with torch.no_grad(), torch.autocast("cpu", dtype=dt):
                o = call(model)
            res[f"{name}/{dn}"] = float((o.float() - ref).norm() / ref.norm())
            print(name, dn, res[f"{name}/{dn}"], flush=True)

    g = torch.load(os.path.join(GOLD, "unet_tiny.pt"), weights_only=False)
    m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    measure("unet_tiny", m, g["out"].float(), lambda mm: mm(g["x"], g["t"], y=g["y"]))
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    v = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=g["ddconfig"], embed_dim=4)).eval()
    v.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
    measure("vae_tiny_decode", v, g["dec"].float(), lambda vv: vv.decode(g["z"]))
    if full:
        g = torch.load(os.path.join(GOLD, "unet_t2v_full.pt"), weights_only=False)
        m = R["MODEL"].build(dict(type="UNetSD_T2VBase", **g["cfg"])).eval()
        m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True)
        gen = torch.Generator("cpu").manual_seed(g["input_seed"])
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        y = torch.randn(1, 77, 1024, generator=gen)
        measure("unet_t2v_full", m, g["out"].float(), lambda mm: mm(x, g["t"], y=y))
    path = os.path.join(GOLD, "autocast_yardstick.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(res)
    json.dump(old, open(path, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default=None, help="regenerate a single fixture family (i2vgen)")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    R = ref_import.load()
    if args.only == "i2vgen":
        make_i2vgen(R)
        return
    if args.only == "videolcm":
        make_videolcm(R)
        return
    if args.only == "tft2v":
        make_tft2v(R)
        return
    if args.only == "vcomposer":
        make_vcomposer(R)
        return
    if args.only == "histogram":
        make_histogram(R)
        return
    if args.only == "odd":
        make_odd(R)
        return
    if args.only == "trajectory":
        make_trajectory(R, args.full)
        return
    if args.only == "yardstick":
        make_yardstick(R, args.full)
        return
    if args.only == "ddpm":
        make_ddpm(R)
        return
    if args.only == "blocks":
        make_blocks(R)
        return
    if args.only == "vae_blocks":
        make_vae_blocks(R)
        return
    if args.only == "ddim_branches":
        make_ddim_branches(R)
        return
    if args.only == "vae_full2":
        make_vae_full2(R)
        return
    if args.only == "i2vgen_full":
        make_i2vgen_full(R)
        return
    if args.only == "i2vgen_full_b":
        make_i2vgen_full_b(R)
        return
    extra = dict(t2v_traj6=make_t2v_traj6, t2v_extra=make_t2v_extra, videolcm_full=make_videolcm_full, tft2v_full=make_tft2v_full,
                 sr600_full=make_sr600_full, vcomposer_full=make_vcomposer_full)
    if args.only in extra:
        extra[args.only](R)
        return
    torch.manual_seed(0)

    # ---- schedules + sampler --------------------------------------------------------------
    rs = R["schedules"]
    sched = {
        "cosine_zts": rs.beta_schedule("cosine", 1000, zero_terminal_snr=True, cosine_s=0.008),
        "linear_sd_zts": rs.beta_schedule("linear_sd", 1000, zero_terminal_snr=True, init_beta=0.00085, last_beta=0.012),
        "quadratic": rs.beta_schedule("quadratic", 1000, init_beta=None, last_beta=None),
        "sigma_logsnr_cosine_interp_zts": rs.sigma_schedule("logsnr_cosine_interp", 1000, zero_terminal_snr=True,
                                                            scale_min=2.0, scale_max=4.0, logsnr_min=-15.0, logsnr_max=15.0),
        "sigma_cosine_zts": rs.sigma_schedule("cosine", 1000, zero_terminal_snr=True, cosine_s=0.008),
    }
    diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **DDIM_T2V))
    g = torch.Generator("cpu").manual_seed(11)
    noise = torch.randn(2, 4, 4, 8, 8, generator=g)
    kw = [dict(y=torch.randn(2, 77, 16, generator=g)), dict(y=torch.randn(2, 77, 16, generator=g))]
    out50 = diff.ddim_sample_loop(noise.clone(), dummy_model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
    x0 = torch.randn(2, 4, 4, 8, 8, generator=g)
    inv20 = diff.ddim_reverse_sample_loop(x0.clone(), dummy_model, kw[0], guide_scale=None, ddim_timesteps=20)
    t = torch.tensor([981, 21])
    xt1, x0p = diff.ddim_sample(noise.clone(), t, dummy_model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
    torch.save(dict(schedules=sched, cfg=DDIM_T2V, noise=noise, kw=kw, out50=out50, x0=x0, inv20=inv20,
                    step_t=t, step_xt1=xt1, step_x0=x0p,
                    tables=dict(alphas_cumprod=diff.alphas_cumprod, sqrt_recipm1=diff.sqrt_recipm1_alphas_cumprod)),
               os.path.join(GOLD, "ddim.pt"))

    # ---- GaussianDiffusion (sr600 forward/reverse diffusions, sr600_infer.yaml) ----------------------
    G = R["diffusion_gauss"]
    sig_fwd = rs.sigma_schedule("logsnr_cosine_interp", 1000, zero_terminal_snr=True, scale_min=2.0,
                                scale_max=4.0, logsnr_min=-15.0, logsnr_max=15.0)
    sig_rev = rs.sigma_schedule("cosine", 1000, zero_terminal_snr=True, cosine_s=0.008)
    gm = lambda x, t=None, y=None, **k: dummy_model(x, t, y=y)
    fwd = G.GaussianDiffusion(sigmas=sig_fwd, prediction_type="v")
    rev = G.GaussianDiffusion(sigmas=sig_rev, prediction_type="v")
    gg = torch.Generator("cpu").manual_seed(21)
    gnoise = torch.randn(2, 4, 4, 8, 8, generator=gg)
    gkw = [dict(y=torch.randn(2, 7, 16, generator=gg)), dict(y=torch.randn(2, 7, 16, generator=gg))]
    torch.manual_seed(1)
    g_s0 = fwd.sample(noise=gnoise.clone(), model=gm, model_kwargs=gkw, guide_scale=9.0, guide_rescale=0.3,
                      solver="dpmpp_2m_sde", steps=30, t_max=699, t_min=0, discretization="trailing", eta=0.0)
    g_inv = rev.ddim_reverse_sample_loop(gnoise.clone(), gm, gkw[1], guide_scale=None, ddim_timesteps=30,
                                         reverse_steps=700)
    gt = torch.tensor([500, 20])
    g_den = fwd.denoise(gnoise, gt, None, gm, gkw, guide_scale=7.5, guide_rescale=0.3)
    torch.save(dict(sig_fwd=sig_fwd, sig_rev=sig_rev, noise=gnoise, kw=gkw, sample_eta0=g_s0, inv30=g_inv,
                    den_t=gt, den=[v.contiguous() for v in g_den],
                    sigma_to_t=fwd._sigma_to_t(torch.tensor(3.7)),
                    t_to_sigma=fwd._t_to_sigma(torch.tensor([10.5, 699.0]))), os.path.join(GOLD, "gauss.pt"))

    # ---- tiny UNet ---------------------------------------------------------------------------
    ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **UNET_TINY)).eval()
    shapes = torch_ref.shapes_of(ref)
    sd = torch_ref.synth_state_dict(shapes, seed=1)
    ref.load_state_dict(sd, strict=True)
    x, y = _inputs(5, 2, 4, 16, 8)
    t = torch.tensor([981, 401])
    out = ref(x, t, y=y)
    torch.save(dict(cfg=UNET_TINY, seed=1, shapes=shapes, x=x, t=t, y=y, out=out),
               os.path.join(GOLD, "unet_tiny.pt"))
    print("unet_tiny", tuple(out.shape), float(out.std()))

    # ---- tiny UNetSD_SR600 ---------------------------------------------------------------------
    SR_TINY = dict(UNET_TINY, dim_mult=[1, 2, 4, 4], use_scale_shift_norm=True, inpainting=True)
    for k in ("use_fps_condition",):
        SR_TINY.pop(k, None)
    sr = R["MODEL"].build(dict(type="UNetSD_SR600", **SR_TINY)).eval()
    srshapes = torch_ref.shapes_of(sr)
    sr.load_state_dict(torch_ref.synth_state_dict(srshapes, seed=3), strict=True)
    gsr = torch.Generator("cpu").manual_seed(9)
    xsr = torch.randn(1, 4, 3, 18, 16, generator=gsr)       # 18 -> 10 -> 6 -> 4 rows: exercises pad (2,1) + crop
    ysr = torch.randn(1, 77, 1024, generator=gsr)
    tsr = torch.tensor([500])
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # Fourier_filter hard-codes .cuda() (unet_sr600.py:38)
    try:
        osr = sr(xsr.clone(), tsr, ysr)
    finally:
        torch.Tensor.cuda = _cuda
    torch.save(dict(cfg=SR_TINY, seed=3, shapes=srshapes, x=xsr, t=tsr, y=ysr, out=osr),
               os.path.join(GOLD, "unet_sr600_tiny.pt"))
    print("unet_sr600_tiny", tuple(osr.shape), float(osr.std()))
    make_i2vgen(R)
    make_videolcm(R)
    make_tft2v(R)
    make_vcomposer(R)
    make_histogram(R)
    make_odd(R)

    # ---- tiny VAE ------------------------------------------------------------------------------
    vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=VAE_TINY, embed_dim=4)).eval()
    vshapes = torch_ref.shapes_of(vae)
    vsd = torch_ref.synth_state_dict(vshapes, seed=2)
    vae.load_state_dict(vsd, strict=True)
    g = torch.Generator("cpu").manual_seed(6)
    z = torch.randn(2, 4, 8, 16, generator=g)
    img = torch.randn(2, 3, 64, 32, generator=g)
    dec = vae.decode(z)
    mom = vae.encode(img).parameters
    torch.manual_seed(3)
    zs = vae.encode_firsr_stage(img, 0.18215)
    torch.save(dict(ddconfig=VAE_TINY, seed=2, shapes=vshapes, z=z, img=img, dec=dec, moments=mom,
                    sample_seed=3, z_sample=zs), os.path.join(GOLD, "vae_tiny.pt"))
    print("vae_tiny", tuple(dec.shape), float(dec.std()))

    if args.full:
        # ---- full-size t2v UNet forward (config 2 of BASELINE.json), B = 1 ----------------------
        import time
        ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **UNET_T2V)).eval()
        shapes = torch_ref.shapes_of(ref)
        sd = torch_ref.synth_state_dict(shapes, seed=0)
        ref.load_state_dict(sd, strict=True)
        del sd
        g = torch.Generator("cpu").manual_seed(8888)          # cfg seed, t2v_infer.yaml:14
        x = torch.randn(1, 4, 16, 32, 56, generator=g)
        y = torch.randn(1, 77, 1024, generator=g)
        t = torch.tensor([981])
        t0 = time.time()
        out = ref(x, t, y=y)
        print("unet_t2v full forward: %.1f s, std %.4f" % (time.time() - t0, float(out.std())))
        torch.save(dict(cfg=UNET_T2V, seed=0, shapes=shapes, input_seed=8888, t=t, out=out,
                        out_norm=float(out.norm())), os.path.join(GOLD, "unet_t2v_full.pt"))
        del ref
        vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=VAE_SD, embed_dim=4)).eval()
        vshapes = torch_ref.shapes_of(vae)
        vae.load_state_dict(torch_ref.synth_state_dict(vshapes, seed=0), strict=True)
        g = torch.Generator("cpu").manual_seed(8889)
        z = torch.randn(1, 4, 32, 56, generator=g)
        t0 = time.time()
        dec = vae.decode(z)
        print("vae full decode: %.1f s" % (time.time() - t0))
        # keep the fixture small: store a strided sub-sample + norms
        torch.save(dict(ddconfig=VAE_SD, seed=0, shapes=vshapes, input_seed=8889,
                        dec_sub=dec[:, :, ::4, ::4].contiguous(), dec_norm=float(dec.norm())),
                   os.path.join(GOLD, "vae_sd_full.pt"))


if __name__ == "__main__":
    main()
