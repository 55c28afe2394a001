"""Full-width fixtures of the model families (r04): one place that knows how to rebuild each model on
the fixture's seeded weights, regenerate its input from the stored seed and compare with the reference's recorded fp32
output.  Used by the -m gpu tests (device "cuda:0", HIP backend) and by tools/emu_parity.py (CPU, ABI emulator).

fixture (oracle/make_golden.py --only ...)   model / shape                                   reference forward
  t2v      unet_t2v_full.pt      UNetSD_T2VBase  [1,4,16,32,56]  Gaussian weights seed 0, t = 981   unet_t2v.py:210-277
  t2v_b    unet_t2v_full_b.pt    same, Student-t (nu = 4) weights seed 1, t = 741
  t2v_c    unet_t2v_full_c.pt    same, the headline weights, t = 501, other input
  videolcm unet_videolcm_full.pt UNetSD_VideoLCM ['text'] [1,4,16,32,56], float t = 759            unet_videolcm.py:541-784
  tft2v    unet_tft2v_full.pt    UNetSD_TFT2V ['text','image'] [1,4,16,64,112], t = 401            unet_tf2tv.py:538-777
  sr600    unet_sr600_full.pt    UNetSD_SR600 [1,4,32,90,160], t = 699                              unet_sr600.py:220-299
  i2vgen   unet_i2vgen_full.pt   UNetSD_I2VGen [1,4,16,88,160], t = 601 (r03)                       unet_i2vgen.py:243-262
  i2vgen_b unet_i2vgen_full_b.pt same, Student-t (nu = 4) weights seed 1, another input, t = 301, fps 16 (r05)
  vcomposer unet_vcomposer_full.pt UNetSD_TFT2V, the 8-entry vcomposer composition list, [1,4,32,64,112] (32 frames 896x512),
                                 six pixel-resolution condition maps regenerated from a seed, t = 601   unet_tf2tv.py:538-777
"""
import os
import types

import torch

from oracle import torch_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = {"t2v": "unet_t2v_full.pt", "t2v_b": "unet_t2v_full_b.pt", "t2v_c": "unet_t2v_full_c.pt",
       "videolcm": "unet_videolcm_full.pt", "tft2v": "unet_tft2v_full.pt", "sr600": "unet_sr600_full.pt",
       "i2vgen": "unet_i2vgen_full.pt", "i2vgen_b": "unet_i2vgen_full_b.pt", "vcomposer": "unet_vcomposer_full.pt"}
SHAPE = {"tft2v": (1, 4, 16, 64, 112), "sr600": (1, 4, 32, 90, 160), "i2vgen": (1, 4, 16, 88, 160), "i2vgen_b": (1, 4, 16, 88, 160),
         "vcomposer": (1, 4, 32, 64, 112)}


def load(name):
    return torch.load(os.path.join(GOLD, FIX[name]), map_location="cpu", weights_only=False)


def build(name, g, precision, dev="cpu", dtname="fp16"):
    from vgen_amd.unet import UNetSD_SR600, UNetSD_T2VBase
    from vgen_amd.unet_i2vgen import UNetSD_I2VGen
    from vgen_amd.unet_videolcm import UNetSD_TFT2V, UNetSD_VideoLCM
    kw = dict(g["cfg"])
    cls = {"videolcm": UNetSD_VideoLCM, "tft2v": UNetSD_TFT2V, "vcomposer": UNetSD_TFT2V, "sr600": UNetSD_SR600,
           "i2vgen": UNetSD_I2VGen, "i2vgen_b": UNetSD_I2VGen}.get(name, UNetSD_T2VBase)
    if name in ("videolcm", "tft2v", "vcomposer"):
        kw["config"] = types.SimpleNamespace(video_compositions=g["comps"], resolution=g["resolution"])
    with torch.device("meta"):
        m = cls(**kw, compute_dtype=dtname, precision=precision)
    m = m.to_empty(device="cpu").eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"], recipe=g.get("recipe", "gauss")),
                      strict=True, assign=True)
    return m.to(dev)


def inputs(name, g):
    """(x, kwargs) in the generator order of the fixture's generator function."""
    gen = torch.Generator("cpu").manual_seed(g["input_seed"])
    x = torch.randn(*SHAPE.get(name, (1, 4, 16, 32, 56)), generator=gen)
    kw = dict(y=torch.randn(1, 77, 1024, generator=gen))
    if name in ("tft2v", "i2vgen", "i2vgen_b"):
        kw["image"] = torch.randn(1, 1, 1024, generator=gen)
    if name in ("i2vgen", "i2vgen_b"):
        kw["local_image"] = torch.randn(1, 4, 88, 160, generator=gen)
        kw["fps"] = g["fps"]
    if name == "vcomposer":
        # the condition maps are regenerated from their own seed, in the generator order of oracle/make_golden.py::vcomposer_conds
        cg = torch.Generator("cpu").manual_seed(g["cond_seed"])
        F_, (W_, H_) = SHAPE[name][2], g["resolution"]
        mk = lambda c: torch.randn(1, c, F_, H_, W_, generator=cg).half().float()
        kw.update(depth=mk(1), sketch=mk(1), single_sketch=mk(1), motion=mk(2), local_image=mk(3), masked=mk(4))
        kw["image"] = torch.randn(1, 1, 1024, generator=cg)
    return x, kw


def calibration_inputs(name, g, dev="cpu", n=None, seed=424242):
    """(x, t, kwargs): the calibration batch of vgen_amd.calibrate.calibration_batch for this fixture's model family — other
    noise, prompts and timesteps than the fixture's, plus the family's own conditioning drawn from the same seed with the
    same leading batch size."""
    from vgen_amd.calibrate import calibration_batch
    shape = SHAPE.get(name, (1, 4, 16, 32, 56))[1:]
    x, t, y = calibration_batch(shape, n=n, seed=seed)
    n = x.shape[0]
    assert int(g["t"].reshape(-1)[0]) not in t.tolist()
    gen = torch.Generator("cpu").manual_seed(seed + 1)
    kw = dict(y=y)
    if name in ("tft2v", "i2vgen", "i2vgen_b", "vcomposer"):
        kw["image"] = torch.randn(n, 1, 1024, generator=gen)
    if name in ("i2vgen", "i2vgen_b"):
        kw["local_image"] = torch.randn(n, 4, 88, 160, generator=gen)
        kw["fps"] = g["fps"].reshape(-1)[:1].repeat(n)
    if name == "vcomposer":
        F_, (W_, H_) = shape[1], g["resolution"]
        mk = lambda c: torch.randn(n, c, F_, H_, W_, generator=gen).half().float()
        kw.update(depth=mk(1), sketch=mk(1), single_sketch=mk(1), motion=mk(2), local_image=mk(3), masked=mk(4))
    if g["t"].is_floating_point():
        t = t.to(g["t"].dtype)
    return x.to(dev), t.to(dev), {k: v.to(dev) for k, v in kw.items()}


def calibrate(name, m, g, dev="cpu", **opts):
    """calibrate_single_pass on the family's calibration batch; returns the report."""
    from vgen_amd.calibrate import calibrate_single_pass
    x, t, kw = calibration_inputs(name, g, dev)
    if name == "sr600":
        return calibrate_single_pass(m, x, t, y=kw["y"], **opts)
    return calibrate_single_pass(m, x, t, **kw, **opts)


def forward(name, m, g, dev="cpu"):
    x, kw = inputs(name, g)
    kw = {k: v.to(dev) for k, v in kw.items()}
    with torch.no_grad():
        if name == "sr600":
            return m(x.to(dev), g["t"].to(dev), kw["y"])
        return m(x.to(dev), g["t"].to(dev), **kw)


def error(out, g):
    """(rel-L2 vs the recorded reference output — sub-sampled for the big fixtures —, norm ratio)"""
    out = out.float().cpu()
    if "out" in g:
        ref = g["out"].float()
        err = float((out - ref).norm() / ref.norm())
    else:
        sub = out[:, :, ::2, ::4, ::4]
        err = float((sub - g["out_sub"]).norm() / g["out_sub"].norm())
    return err, float(out.norm()) / g["out_norm"]
