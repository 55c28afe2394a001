"""bench.py — denoise-steps/sec of the sampling hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one classifier-free-guidance DDIM denoise step of the t2v configuration
(BASELINE.json configs[1]: latent [1,4,16,32,56] = 16 frames 448x256, 77x1024 context, guide 9,
UNetSD_T2VBase 1411 M parameters): two UNet forwards (evaluated as one batch of 2 units) + the
fused CFG/DDIM update, with all inputs resident in HBM.  Synthetic latents/context, seeded
random-init weights (no checkpoints offline) — for the t2v configuration the very weights of the golden
fixture tests/golden/unet_t2v_full.pt (vgen_amd/synth.py), so the model that is timed is the model whose
output is compared with the reference's.

The timed region calls the PUBLIC sampler API the engines use, step by step:
`DiffusionDDIM.ddim_sample(xt, t, model, [cond, uncond], guide_scale=9, ddim_timesteps=50, eta=0)`
(`ddim_sample_loop` is a loop over exactly this call; tools/inferences/inference_text2video_entrance.py:200-206),
each call's x_{t-1} fed to the next, t walking the 50-step DDIM schedule.  Underneath, a cached sampling
session (vgen_amd/session.py) replays one hipGraph per step; the untimed setup does two steps (eager
warm-up + capture), like weight loading.

HEADLINE MODE (r06): fp16, precision "calibrated" (vgen_amd/calibrate.py) on EVERY leg of this script (the timed steps, e2e,
scaling_model, --partition, --gpus N > 1, the other --config shapes): every packed weight is ONE 16-bit matrix — single-pass
launches only, the kernels of precision="fast" — whose rounding was chosen at pack time by error feedback from one forward on
a seeded calibration batch (calibrate.calibration_batch: 8 noise / prompt draws at 8 timesteps spread over the schedule, none
of them timed or parity-checked).  Which launches are calibrated is a rule of their sizes (K <= 9000 and rows >= 2 K); there is
no wall clock in the pass, so the same weights always pack the same bits (`calibration.packed_digest` on the line).  Pack-time
work (like weight loading): `calibration.seconds` reports it, the timed region starts after it.  With N > 1 rank 0 calibrates
and saves (calibrate.save_calibrated), the other ranks build `precision="calibrated", calibration=<file>` — the persistent
route a deployment uses.
`parity.unet_rel_l2` is COMPUTED IN THIS RUN (max over the three t2v fixtures): the timed model evaluates the golden
fixtures' inputs and is compared with the reference's recorded fp32 outputs (the third fixture has its own weights: its model
is calibrated the same way); a value outside the tolerance sets `parity_exceeds_tolerance` on the line and warns on stderr.
`variants` carries timed steps + parity (this run, same process) for fp16/mixed (two-term weights at the full-resolution
level: the r03-r05 default, 1.13x the launches' MFMA work), fp16/high (two-term weights everywhere: the largest margin),
fp16/fast (weights to nearest: the reference's own autocast arithmetic) and bf16/fast (BASELINE.json's literal "bf16") — the
last two outside 1e-3.

N > 1 (weak scaling): P = N prompts in flight -> 2N units spread over the ranks, ONE all-gather of the unit
outputs per step (RCCL), every rank applies the cheap update for all prompts.  value = prompts * steps /
max-over-ranks time.  `--partition` runs that code path at N = 1.

`--config {i2vgen,sr600,tft2v896,tft2v32f,videolcm}` runs the other BASELINE.json shapes through the
same API (random-init weights of that architecture; see CONFIGS).  `--config sr600` times the SR600 stage's own
samplers: `GaussianDiffusion.sample(solver='dpmpp_2m_sde')` CFG steps (value) and the DDIM-inversion forwards
(`inversion` object), tools/inferences/inference_tft2v_sr600_entrance.py:284-308.

Objects on the JSON line — every number is measured in this run unless its key says `committed`:
  parity       — see above.
  calibration  — the report of the pack-time calibration pass of the timed model (counts per decision, the rule, host
                 seconds, a digest of the per-launch decisions and of the packed bits).
  roofline     — dominant kernel (tap-GEMM, MFMA-bound): algorithmic FLOP per launch (2 M N K of the product each
                 launch computes; a dual-W launch executes twice the MFMAs for it) / average launch duration, measured
                 with HIP events on the launch stream in an instrumented eager pass of the same step; peak = 2.5
                 PFLOP/s dense 16-bit MFMA.  `committed` sub-object: HBM-side bytes per launch and MFMA-busy fraction
                 from the committed rocprofv3 PMC passes (profiles/), which cannot be read from inside the process.
  hbm_kernels  — GroupNorm / LayerNorm launches of the same pass: algorithmic bytes / time vs 8 TB/s.
  cpu_baseline — the CPU path timed on the host cores on a bounded sample (one full-size forward), scaled to a
                 step; `kind` = "reference" when the reference tree is present (its own modules through
                 oracle/ref_import.py), else "port" (oracle/torch_ref.py); rank 0 at N=1.
  vae / e2e    — AutoencoderKL decode frames/s; a whole video (50-step ddim_sample_loop + 16-frame decode).

`model_tflops_per_s` / `frac_of_mfma_peak` count the REFERENCE's work per step (2 forwards x 8.665 TFLOP).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_T2V = dict(in_dim=4, dim=320, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
                num_heads=8, head_dim=64, num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                temporal_attention=True, temporal_attn_times=1, use_checkpoint=False,
                use_fps_condition=False, use_sim_mask=False)
VAE_SD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
DDIM = dict(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
            mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False)
# configs/tft2v_vcomposer_32frames_sr600_infer.yaml:38-60 of the reference
SR600_DIFF = dict(
    reverse_diffusion=dict(schedule="cosine", mean_type="v", schedule_param=dict(num_timesteps=1000, zero_terminal_snr=True)),
    forward_diffusion=dict(schedule="logsnr_cosine_interp", mean_type="v",
                           schedule_param=dict(num_timesteps=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)))
VAE_DEC_TFLOP = 1.092       # per 256x448 frame
PEAK_TFLOPS = 2500.0        # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
TOLERANCE = 1e-3            # north_star: UNet output within 1e-3 rel-L2 of the reference
GOLDEN_T2V = os.path.join(ROOT, "tests", "golden", "unet_t2v_full.pt")
# r04: two more full-size fixtures of the reference's fp32 forward (oracle/make_golden.py --only t2v_extra) — the headline
# weights at a mid-trajectory t = 501 on another input, and a second weight seed with heavy-tailed (Student-t, nu = 4)
# matrices at t = 741.  parity.unet_rel_l2 is the MAX over all three.
GOLDEN_T2V_C = os.path.join(ROOT, "tests", "golden", "unet_t2v_full_c.pt")
GOLDEN_T2V_B = os.path.join(ROOT, "tests", "golden", "unet_t2v_full_b.pt")

# name -> (model class path, ctor kwargs, latent [C,F,H,W], units per step G, UNet forward TFLOP,
#          extra conditioning builder, description)
CONFIGS = {
    "t2v": dict(cls="unet.UNetSD_T2VBase", cfg=UNET_T2V, latent=(4, 16, 32, 56), G=2, tflop=8.665,
                desc="t2v 16x448x256 latent [1,4,16,32,56], DDIM CFG step (2 UNetSD_T2VBase fwd + fused update), "
                     "guide 9, 77x1024 ctx, random-init 1411M params"),
    "i2vgen": dict(cls="unet_i2vgen.UNetSD_I2VGen", cfg=dict(UNET_T2V, concat_dim=4, upper_len=128, default_fps=8),
                   latent=(4, 16, 88, 160), G=2, tflop=88.09,
                   desc="i2vgen-xl 16x1280x704 latent [1,4,16,88,160], DDIM CFG step (2 UNetSD_I2VGen fwd + update), "
                        "77+64+4 ctx tokens, local-image stem channels"),
    "sr600": dict(cls="unet.UNetSD_SR600", cfg=dict(UNET_T2V, use_scale_shift_norm=True, inpainting=True),
                  latent=(4, 32, 90, 160), G=2, tflop=185.80,
                  desc="sr600 32x1280x720 latent [1,4,32,90,160], DPM-Solver++(2M) SDE CFG step "
                       "(GaussianDiffusion.sample: 2 UNetSD_SR600 fwd + guide_rescale + solver update), "
                       "pad-(2,1) downsample / cropped upsample / FreeU"),
    "tft2v896": dict(cls="unet_videolcm.UNetSD_TFT2V", cfg=dict(UNET_T2V, num_tokens=4), comps=["text", "image"],
                     latent=(4, 16, 64, 112), G=2, tflop=38.97,
                     desc="tft2v 16x896x512 latent [1,4,16,64,112], DDIM CFG step (2 UNetSD_TFT2V fwd + update), text+image"),
    "tft2v32f": dict(cls="unet_videolcm.UNetSD_TFT2V", cfg=dict(UNET_T2V, num_tokens=4), comps=["text", "image"],
                     latent=(4, 32, 32, 56), G=2, tflop=17.36,
                     desc="tft2v 32x448x256 latent [1,4,32,32,56], DDIM CFG step (2 UNetSD_TFT2V fwd + update), text+image"),
    "videolcm": dict(cls="unet_videolcm.UNetSD_VideoLCM", cfg=dict(UNET_T2V), comps=["text"],
                     latent=(4, 16, 32, 56), G=1, tflop=8.665,
                     desc="videolcm 16x448x256: whole videos as the engine makes them (inference_videolcm_entrance.py:"
                          "171-258) — LCMScheduler.sample_loop, 4 steps of one UNetSD_VideoLCM fwd + LCM update (no CFG), "
                          "a NEW prompt per video, then the 16-frame AutoencoderKL decode to uint8 (decoder_bs 2)"),
    # r05: the first stage of BASELINE config 5 with the reference's OWN composition list (configs/tft2v_vcomposer_infer.yaml:74)
    # — six pixel-resolution condition maps summed into the concat channels, an image token; needs precision="high" at this
    # shape (tests/golden/unet_vcomposer_full.pt: 1.04e-3 in "mixed", 8.6e-4 in "high")
    "tft2v_vcomposer": dict(cls="unet_videolcm.UNetSD_TFT2V", cfg=dict(UNET_T2V, concat_dim=8, num_tokens=4, training=False),
                            comps=["text", "mask", "depthmap", "sketch", "motion", "image", "local_image", "single_sketch"],
                            latent=(4, 32, 64, 112), G=2, tflop=78.04,
                            desc="tft2v vcomposer 32x896x512 latent [1,4,32,64,112], DDIM CFG step (2 UNetSD_TFT2V fwd + "
                                 "update), the eight-entry vcomposer composition list"),
    # BASELINE config 5: both stages back to back for ONE video (bench.py run_two_stage)
    "tft2v_sr600": dict(cls="unet_videolcm.UNetSD_TFT2V", cfg=dict(UNET_T2V, num_tokens=4), comps=["text", "image"],
                        latent=(4, 32, 64, 112), G=2, tflop=77.94,
                        desc="tft2v 32x896x512 + sr600 32x1280x720, the full 2-stage pipeline for one video "
                             "(inference_tft2v_sr600_entrance.py:274-321): 50 DDIM CFG steps at [1,4,32,64,112] + 32-frame "
                             "decode; bilinear resize to 720p, 32-frame encode, 30 DDIM-inversion forwards, 30 CFG "
                             "DPM-Solver++(2M) SDE steps at [1,4,32,90,160], 32-frame 720p decode"),
}


def randomize_(module, seed):
    """Seeded non-degenerate weights on the device (zero-init layers included, SURVEY §8c trap)."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=p.device) * (0.8 / fan_in ** 0.5))
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g, device=p.device))


def model_class(name):
    import importlib
    modname, clsname = CONFIGS[name]["cls"].split(".")
    return getattr(importlib.import_module("vgen_amd." + modname), clsname)


def calibrate_model(name, model, dev):
    """precision="high" -> "calibrated" in place (vgen_amd/calibrate.py) on the family's seeded calibration batch: noise /
    prompts / timesteps of calibrate.calibration_batch + this family's conditioning tensors with the same batch size.  Returns
    the brief report (+ wall seconds)."""
    from vgen_amd import calibrate as cal
    t0 = time.perf_counter()
    x, t, y = cal.calibration_batch(CONFIGS[name]["latent"], device=dev)
    n = x.shape[0]
    kw = conditioning(name, model, n, dev, torch.Generator(device=dev).manual_seed(424243))[0]
    kw["y"] = y
    rep = cal.calibrate_single_pass(model, x, t, **kw)
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()
    out = cal.brief_report(rep)
    out.update(seconds=round(time.perf_counter() - t0, 1), timesteps=t.tolist(),
               input="calibrate.calibration_batch seed 424242 (+ the family's conditioning, seed 424243): none of it timed or "
                     "parity-checked")
    return out


def build_model(name, dev, dtype, precision="high", state_dict=None, calibration=None):
    """The model of config `name` on `dev`, packed.  state_dict: fp32 CPU parameters to load (the golden fixture's
    weights for t2v); None = device-side seeded random init.  precision="calibrated": packed two-term and calibrated here
    (calibrate_model; the report is left on model.bench_calibration) — or, with calibration=<file>, loaded from a file another
    rank's pass saved."""
    import types
    c = CONFIGS[name]
    cls = model_class(name)
    kw = dict(c["cfg"])
    want_cal = precision == "calibrated"
    if want_cal and calibration is None:
        precision = "high"
    elif want_cal:
        kw["calibration"] = calibration
    if "comps" in c:
        kw["config"] = types.SimpleNamespace(video_compositions=c["comps"], resolution=[c["latent"][3] * 8, c["latent"][2] * 8])
    if state_dict is not None:
        with torch.device("meta"):
            model = cls(**kw, compute_dtype=dtype, precision=precision)
        model = model.to_empty(device="cpu").eval()
        model.load_state_dict(state_dict, strict=True, assign=True)
        model = model.to(dev)
    else:
        with torch.device(dev):
            model = cls(**kw, compute_dtype=dtype, precision=precision)
        model.eval()
        randomize_(model, 0)
    model.pack()
    if want_cal and calibration is None:
        model.bench_calibration = calibrate_model(name, model, dev)
    elif want_cal:
        model.bench_calibration = {"loaded_from": calibration, **{k: v for k, v in (model._calibration_report or {}).items()
                                                                  if k != "layers"}}
    return model


def build_vae(dev, dtype, precision, seed=1):
    """The SD AutoencoderKL on seeded weights in the VAE mode that goes with a UNet precision: "fast" -> to-nearest weights,
    "calibrated" -> packed two-term and calibrated (calibrate.calibrate_vae: 8 seeded latents / frames of the benchmark's
    256 x 448 shape through one decode and one encode pass), everything else -> two-term weights ("high")."""
    from vgen_amd.vae import AutoencoderKL
    vp = {"fast": "fast", "calibrated": "high"}.get(precision, "high")
    with torch.device(dev):
        vae = AutoencoderKL(ddconfig=VAE_SD, embed_dim=4, compute_dtype=dtype, precision=vp)
    vae.eval()
    randomize_(vae, seed)
    if precision == "calibrated":
        from vgen_amd.calibrate import brief_report, calibrate_vae
        t0 = time.perf_counter()
        gen = torch.Generator("cpu").manual_seed(424244)
        z = (torch.randn(8, 4, 32, 56, generator=gen) / 0.18215 * 0.2).to(dev)
        x = (torch.rand(8, 3, 256, 448, generator=gen) * 2 - 1).to(dev)
        rep = brief_report(calibrate_vae(vae, z, x))
        rep["seconds"] = round(time.perf_counter() - t0, 1)
        vae.bench_calibration = rep
    return vae


def drop_masters(model, dev):
    """fp32 masters are not needed for sampling once the operands are packed (the variants' condition stems run on
    theirs, so only the t2v trunk drops them)."""
    for p in model.parameters():
        p.data = torch.empty(0, device=dev)
    torch.cuda.empty_cache()


def conditioning(name, model, P, dev, gen):
    """[cond kwargs, uncond kwargs] (or one set) with the tensors the engines pass for this model family."""
    c = CONFIGS[name]
    C, F, H, W = c["latent"]
    y_c = torch.randn(P, 77, 1024, generator=gen, device=dev)
    y_u = torch.randn(P, 77, 1024, generator=gen, device=dev)
    kc, ku = dict(y=y_c), dict(y=y_u)
    if name == "i2vgen":
        img = torch.randn(P, 1024, generator=gen, device=dev)
        li = torch.randn(P, 4, 1, H, W, generator=gen, device=dev)
        fps = torch.full((P,), 8, dtype=torch.long, device=dev)
        kc.update(image=img.unsqueeze(1), local_image=li, fps=fps)
        ku.update(image=torch.zeros_like(img).unsqueeze(1), local_image=li, fps=fps)
    if name in ("tft2v896", "tft2v32f", "tft2v_vcomposer"):
        img = torch.randn(P, 1, 1024, generator=gen, device=dev)
        kc.update(image=img)
        ku.update(image=torch.zeros_like(img))
    if name == "tft2v_vcomposer":
        # the six spatial conditions at pixel resolution (the engine passes the same maps to both CFG branches,
        # inference_tft2v_sr600_entrance.py / unet_tf2tv.py:538-777)
        mk = lambda ch: torch.randn(P, ch, F, H * 8, W * 8, generator=gen, device=dev).half().float()
        maps = dict(depth=mk(1), sketch=mk(1), single_sketch=mk(1), motion=mk(2), local_image=mk(3), masked=mk(4))
        kc.update(maps)
        ku.update(maps)
    return [kc, ku] if c["G"] == 2 else [kc]


def golden_parity(model, gold, dev):
    """rel-L2 of `model`'s forward on the golden fixture's input vs the reference's recorded fp32 output."""
    gen = torch.Generator("cpu").manual_seed(gold["input_seed"])
    x = torch.randn(1, 4, 16, 32, 56, generator=gen)
    y = torch.randn(1, 77, 1024, generator=gen)
    with torch.no_grad():
        out = model(x.to(dev), gold["t"].to(dev), y=y.to(dev)).float().cpu()
    ref = gold["out"].float()
    return float((out - ref).norm() / ref.norm())


class StepTimer:
    """Times `steps` calls of the public per-step sampler API (after 2 setup calls and `warmup` calls)."""

    def __init__(self, diff, model, xt0, mkw, guide, dev, P):
        self.diff, self.model, self.xt0, self.mkw, self.guide, self.dev, self.P = diff, model, xt0, mkw, guide, dev, P
        self.steps_all = (1 + torch.arange(0, 1000, 20)).clamp(0, 999).flip(0).tolist()
        self.t_bufs = {s: torch.full((P,), s, dtype=torch.long, device=dev) for s in self.steps_all}

    def t_of(self, i):
        return self.t_bufs[self.steps_all[i % len(self.steps_all)]]

    def step(self, xt, i):
        return self.diff.ddim_sample(xt, self.t_of(i), self.model, self.mkw, guide_scale=self.guide, ddim_timesteps=50,
                                     eta=0.0)[0]

    def run(self, steps, warmup, world=1):
        import torch.distributed as dist
        xt = self.xt0
        for i in range(2):                                  # untimed setup: eager warm-up pass + graph capture
            xt = self.step(xt, i)
        xt = self.xt0
        for i in range(warmup):
            xt = self.step(xt, i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            xt = self.step(xt, warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_s], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_s = float(tt.item())
        return dt_s, xt


def roofline_pass(args, model, timer, xt0, kw, G, guide, precision):
    """Roofline of the dominant kernel class for `model`: ONE instrumented eager pass of the same step (every launch bracketed
    by HIP events on the launch stream), the GroupNorm / LayerNorm launches of that pass against the HBM peak.  Returns
    {"roofline": ..., "hbm_kernels": ...} measured on the model that is passed in."""
    from vgen_amd import ops
    from vgen_amd.diffusion import DiffusionDDIM
    res = {}
    d0 = DiffusionDDIM(**DDIM)
    d0.rng_parity = False
    d0.sessions = None                              # step-by-step launches: every kernel bracketed by HIP events
    xs = xt0[:1].clone()
    kw1 = [{k: (v[:1] if torch.is_tensor(v) else v) for k, v in d.items()} for d in kw]
    mk1 = kw1 if G == 2 else kw1[0]
    t1 = timer.t_of(0)[:1]
    d0.ddim_sample(xs, t1, model, mk1, guide_scale=guide, ddim_timesteps=50, eta=0.0)
    ops.KERNEL_PROFILE = []
    torch.cuda.synchronize()
    torch.cuda._sleep(int(4e8))     # let the host run ahead so event pairs bracket GPU time only
    d0.ddim_sample(xs, t1, model, mk1, guide_scale=guide, ddim_timesteps=50, eta=0.0)
    torch.cuda.synchronize()
    allrecs = ops.KERNEL_PROFILE
    ops.KERNEL_PROFILE = None
    recs = [r for r in allrecs if r[0] == "tapgemm"]
    ms = [r[1].elapsed_time(r[2]) for r in recs]
    fl = [r[3] for r in recs]
    other = {}
    for r in allrecs:
        if r[0] == "tapgemm":
            continue
        a = other.setdefault((r[0],) + tuple(r[4]), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r[1].elapsed_time(r[2])
        a[2] += r[3]
    if args.dump_shapes:
        tag = f"{args.config}_{args.dtype}_{precision}"
        orows = sorted(([list(k), v[0], round(v[1], 4), round(v[2] / (v[1] * 1e-3) / 1e9, 1)] for k, v in other.items()),
                       key=lambda r: -r[2])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"other_shapes_{tag}.json"), "w") as f:
            json.dump({"cols": ["(op,shape...)", "launches", "ms", "GB/s or GFLOP/s"], "rows": orows}, f, indent=0)
        agg = {}
        for r, m in zip(recs, ms):
            a = agg.setdefault(r[4], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += m
            a[2] += r[3]
        rows = sorted(([list(k), v[0], round(v[1], 4), round(v[2] / (v[1] * 1e-3) / 1e12, 1)] for k, v in agg.items()),
                      key=lambda r: -r[2])
        with open(os.path.join(ROOT, "gpurun_out", f"tapgemm_shapes_{tag}.json"), "w") as f:
            json.dump({"cols": ["(mode,M,N,K,epi,out)", "launches", "ms", "algorithmic TFLOP/s"], "rows": rows}, f, indent=0)
    # What an event pair adds to the kernel it brackets: in an eager launch sequence each `record / launch / record` group
    # costs the command processor a fixed time per packet that a kernel's own duration (rocprofv3's Start -> End) does not
    # contain and that the captured step does not pay (r06 evidence: the sum of rocprofv3 kernel durations per step, 29.69 ms,
    # IS the graph step's 29.74 ms).  Measured here on empty pairs (record, record: no kernel between), in the same run ahead
    # of the same sleeping stream, and subtracted once per launch; `avg_launch_us_events` keeps the raw figure.
    try:
        torch.cuda.synchronize()
        torch.cuda._sleep(int(1e8))
        pairs = []
        for _ in range(200):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            b_.record()
            pairs.append((a_, b_))
        torch.cuda.synchronize()
        gaps = sorted(a_.elapsed_time(b_) for a_, b_ in pairs)
        pair_ms = gaps[len(gaps) // 2]
    except RuntimeError:                    # no device events (tools/dryrun_bench_cpu.py): nothing to subtract
        pair_ms = 0.0
    raw_ms = sum(ms)
    ms = [max(m - pair_ms, 1e-4) for m in ms]
    tot_ms, tot_fl = sum(ms), sum(fl)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    ndw = sum(1 for r in recs if str(r[4][5]).endswith("+dw"))
    res["roofline"] = {"kernel": "tap-GEMM class: tapgemm_kernel (streaming shapes) + panel_kernel (W-panel-resident, K = 320)",
                       "bound": "mfma", "achieved": round(ach, 2),
                       "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS, 4),
                       "traffic": None, "launches_per_step": len(recs), "dual_w_launches": ndw,
                       "avg_launch_us": round(1e3 * tot_ms / max(len(recs), 1), 2),
                       "avg_launch_us_events": round(1e3 * raw_ms / max(len(recs), 1), 2),
                       "event_pair_overhead_us": round(1e3 * pair_ms, 2),
                       "avg_gflop_per_launch": round(tot_fl / max(len(recs), 1) / 1e9, 3),
                       "tapgemm_ms_per_step": round(tot_ms, 3),
                       # algorithmic FLOP of the products of one step (the CFG pair shares the layers ahead of the first
                       # cross-attention, so this is a few % below the reference's 2 x forward count); a dual-W launch
                       # executes 2x the MFMA work for its product: executed_over_algorithmic says how much
                       "tapgemm_tflop_per_step": round(tot_fl / 1e12, 3),
                       "algorithmic_bytes_per_launch": round(sum(r[5][0] for r in recs) / max(len(recs), 1)),
                       # issued MFMA work / the products' own 2 M N K: the W_lo passes of the dual-W launches AND the
                       # doubled K columns of two-term activation segments (r03 booked the latter as algorithmic)
                       "executed_over_algorithmic": round(sum(r[5][1] for r in recs) / max(tot_fl, 1.0), 3),
                       "measured_in_this_run": True,
                       # r05: what actually caps these kernels below the MFMA roofline — the CU's vector-memory path
                       # (tools/probes/vmem_probe.hip on this chip; not measured in this run)
                       "cu_vmem_ceiling_committed": {
                           "source": "profiles/r05d_vmem_probe.txt, r05f_vmem_probe_quad.txt, r05h_vmem_probe_pieces.txt",
                           "B_per_clk_per_cu": {"mfma_operand_layout_straight_from_rows": 17.5,
                                                "quad_contiguous_pieces_to_registers_or_lds_dma": "41-59",
                                                "stores_any_pattern": "10-12"},
                           "note": "a streaming 256 x 160 x 64 K-step stages 52 KiB per CU: >= 1.1 k cycles of this path "
                                   "beside 1.28 k cycles of MFMAs; stamped K-step 2.0 k cycles (DESIGN 3.1)"}}
    # HBM-side bytes per launch cannot be read from inside the process: they come from committed rocprofv3 PMC
    # passes of this command (tools/collect_evidence.sh -> profiles/) and are labelled as such
    for tname in ("r06_tapgemm_traffic.json", "r05n_tapgemm_traffic.json", "r05_tapgemm_traffic.json", "r04_tapgemm_traffic.json", "r03_tapgemm_traffic.json", "r02_tapgemm_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("precision", "fast") != precision or tj.get("dtype", "fp16") != args.dtype:
                continue
            res["roofline"]["traffic"] = round(tj["hbm_bytes_per_launch"])
            res["roofline"]["committed"] = {
                "source": f"profiles/{tname} (rocprofv3 --pmc passes of this command on an earlier box; NOT measured in this run)",
                "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)",
                "hbm_bytes_per_launch": round(tj["hbm_bytes_per_launch"]),
                "algorithmic_bytes_per_launch": tj.get("algorithmic_bytes_per_launch"),
                "mfma_busy_frac": tj.get("mfma_busy_frac")}
            break
    # the HBM-bound kernel classes of the same pass, against chip peak (north_star: "HBM GB/s ... against chip peak")
    hk = {}
    for k, v in other.items():
        if k[0] in ("groupnorm", "layernorm"):
            a = hk.setdefault(k[0], [0, 0.0, 0.0])
            a[0] += v[0]
            a[1] += v[1]
            a[2] += v[2]
    res["hbm_kernels"] = {k: {"launches": v[0], "ms_per_step": round(v[1], 3),
                              "achieved_GBs": round(v[2] / (v[1] * 1e-3) / 1e9, 1),
                              "frac_of_8TBs": round(v[2] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 3)}
                          for k, v in hk.items() if v[1] > 0}
    return res


FIXTURE_NAMES = ("t2v_full (the timed model: Gaussian weights seed 0, t=981)",
                 "t2v_full_c (the timed model, t=501, other input)",
                 "t2v_full_b (same architecture + mode, Student-t(4) weights seed 1, t=741)")


def parity_block(fx, dtype, precision):
    """The `parity` object of a mode from its per-fixture rel-L2 values (max over the fixtures against the tolerance)."""
    err = max(fx.values())
    return {"unet_rel_l2": err, "tolerance": TOLERANCE, "within_tolerance": bool(err <= TOLERANCE),
            "dtype": dtype, "precision": precision, "measured_in_this_run": True,
            "fixtures": {k: round(v, 7) for k, v in fx.items()},
            "golden": "tests/golden/unet_t2v_full{,_c,_b}.pt: the reference's fp32 UNetSD_T2VBase forward on the "
                      "same seeded weights and inputs (oracle/make_golden.py); unet_rel_l2 = max over the fixtures"}


def _finish(res, world):
    """Leave together, then print the ONE JSON line LAST: RCCL writes its version banner to stdout through C stdio, which —
    redirected to a file or a pipe — is flushed at exit, i.e. after a line printed from Python (seen in r04: the banner
    followed the JSON and a last-line parser read 'Librccl path : ...')."""
    import ctypes
    import torch.distributed as dist
    if world > 1:
        dist.barrier()                      # rank 0 ran the untimed extras (roofline pass, VAE); leave together
    if dist.is_initialized():
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                       # noqa: BLE001
        pass
    sys.stdout.flush()
    if res is not None:
        print(json.dumps(res), flush=True)


def run_videolcm(args, dev, model, world, rank):
    """BASELINE config 4 as stated: whole videos through the 4-step `LCMScheduler.sample_loop` + the 16-frame decode
    ("decode-bound, exercises VAE kernels").  Every video is a NEW prompt (fresh context tensor, as a new caption through
    the text tower would be): the scheduler's sampling session is re-bound to it (vgen_amd/session.py) — one K/V GEMM, no
    re-capture.  --steps K = denoise steps timed (rounded up to whole videos of 4); value = denoise steps / s of the
    whole-video wall time (decode included), `video` carries the split."""
    from vgen_amd.lcm import LCMScheduler
    from vgen_amd.vae import AutoencoderKL
    C, F, H, W = CONFIGS["videolcm"]["latent"]
    sched = LCMScheduler(prediction_type="v_prediction", beta_schedule="scaled_linear", clip_sample=False,
                         timestep_spacing="linspace", rescale_betas_zero_snr=True)
    sched.set_timesteps(4, device=dev)
    vae = build_vae(dev, args.dtype, args.precision)
    g = torch.Generator(device=dev).manual_seed(8888 + rank)

    def one_video():
        y = torch.randn(1, 77, 1024, generator=g, device=dev)               # a new prompt
        noise = torch.randn(1, C, F, H, W, generator=g, device=dev)
        t0 = time.perf_counter()
        lat = sched.sample_loop(noise, model, [dict(y=y)], guidance_scale=None, generator=g)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vid = vae.decode_video(lat * 0.2, scale_factor=0.18215, decoder_bs=2)
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1, lat, vid

    nvid = max(1, (args.steps + 3) // 4)
    for _ in range(2 + max(0, (args.warmup + 3) // 4)):                     # session build + capture, re-bind, warm-up
        one_video()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop_s = dec_s = 0.0
    for _ in range(nvid):
        a, b, lat, vid = one_video()
        loop_s, dec_s = loop_s + a, dec_s + b
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    steps = 4 * nvid
    return {
        "metric": "denoise_steps_per_sec", "value": round(steps / dt_s, 4), "unit": "steps/s", "n_gpus": world,
        "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt_s / steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": CONFIGS["videolcm"]["desc"], "name": "videolcm", "precision": args.precision,
                   "api": "LCMScheduler.sample_loop (4 steps) + AutoencoderKL.decode_video per video",
                   "parallelism": "single GPU", "weights": "seeded synthetic (device-side random init)"},
        "finite": bool(torch.isfinite(lat).all()), "video_shape": list(vid.shape),
        "video": {"videos": nvid, "seconds_per_video": round(dt_s / nvid, 4), "videos_per_sec": round(nvid / dt_s, 3),
                  "frames_per_sec": round(F * nvid / dt_s, 2), "sample_loop_s_per_video": round(loop_s / nvid, 4),
                  "decode_s_per_video": round(dec_s / nvid, 4), "decode_share": round(dec_s / max(loop_s + dec_s, 1e-9), 3),
                  "session_rebinds": sched.sessions.rebinds},
        "model_tflops_per_s": round(CONFIGS["videolcm"]["tflop"] * steps / dt_s, 2),
    }


def run_two_stage(args, dev, world, rank):
    """BASELINE config 5, one video end to end on one GPU: stage 1 = UNetSD_TFT2V, 50 DDIM CFG steps + decode; the decoded
    frames stand in for the .mp4 the reference writes and re-reads (video writer / cv2 reader: out of scope) and go through
    the engine's own glue (bilinear resize to 720 x 1280, inference_tft2v_sr600_entrance.py:118) into stage 2 = 32-frame
    encode, DDIM inversion (30 forwards, reverse_steps 700), `sample(solver='dpmpp_2m_sde', steps=30)` CFG 9 / rescale 0.3,
    32-frame 720p decode (decoder_bs 4)."""
    import torch.nn.functional as Fn
    from vgen_amd.diffusion import DiffusionDDIM
    from vgen_amd.diffusion_gauss import DiffusionDDIMSR
    from vgen_amd.vae import AutoencoderKL
    steps1 = steps2 = 2 if args.steps < 10 else None        # --steps < 10: a 2-step smoke run of every stage
    n1, n2 = steps1 or 50, steps2 or 30
    g = torch.Generator(device=dev).manual_seed(8888 + rank)
    vae = build_vae(dev, args.dtype, args.precision)
    times = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        times[name] = round(time.perf_counter() - t0, 4)
        return time.perf_counter()

    # ---- stage 1 -----------------------------------------------------------------------------------------------
    s1 = "tft2v_vcomposer" if args.stage1 == "vcomposer" else "tft2v896"
    p1 = args.stage1_precision or args.precision
    m1 = build_model(s1, dev, args.dtype, p1)
    C, F, H, W = CONFIGS["tft2v_sr600"]["latent"]
    if args.stage1 == "vcomposer":
        kw1 = conditioning("tft2v_vcomposer", m1, 1, dev, g)
    else:
        kw1 = conditioning("tft2v896", m1, 1, dev, g)
    d1 = DiffusionDDIM(**DDIM)
    d1.rng_parity = False
    noise = torch.randn(1, C, F, H, W, generator=g, device=dev)
    torch.cuda.synchronize()
    t_all = t0 = time.perf_counter()
    lat = d1.ddim_sample_loop(noise, m1, kw1, guide_scale=9.0, ddim_timesteps=n1, eta=0.0)
    t0 = lap("stage1_ddim_loop_s", t0)
    frames = vae.decode_video(lat * 0.2, scale_factor=0.18215, decoder_bs=2, to_uint8=False)       # [1, 3, 32, 512, 896] fp32
    t0 = lap("stage1_decode_s", t0)
    fin1 = bool(torch.isfinite(lat).all())
    del m1, d1
    gc.collect()
    torch.cuda.empty_cache()
    # ---- stage 2 -----------------------------------------------------------------------------------------------
    t_build = time.perf_counter()
    m2 = build_model("sr600", dev, args.dtype, args.precision)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build                 # weight synthesis, not pipeline time: subtracted below
    t0 = time.perf_counter()
    vid = frames[0].permute(1, 0, 2, 3).clamp(-1, 1)                                   # [32, 3, 512, 896]
    vid = Fn.interpolate(vid, size=(720, 1280), mode="bilinear")                       # the engine's glue (:118)
    zs = [vae.encode_firsr_stage(vid[i:i + 2], 0.18215) for i in range(0, vid.shape[0], 2)]
    z0 = torch.cat(zs, 0).permute(1, 0, 2, 3).unsqueeze(0).contiguous()               # [1, 4, 32, 90, 160]
    t0 = lap("stage2_resize_encode_s", t0)
    srd = DiffusionDDIMSR(**SR600_DIFF)
    y_c = torch.randn(1, 77, 1024, generator=g, device=dev)
    y_u = torch.randn(1, 77, 1024, generator=g, device=dev)
    noised = srd.reverse_diffusion.ddim_reverse_sample_loop(z0, m2, {"y": y_u}, guide_scale=None, ddim_timesteps=n2,
                                                            reverse_steps=700)
    t0 = lap("stage2_inversion_s", t0)
    out = srd.forward_diffusion.sample(noise=noised, model=m2, model_kwargs=[{"y": y_c}, {"y": y_u}], guide_scale=9.0,
                                       guide_rescale=0.3, solver="dpmpp_2m_sde", steps=n2, t_max=699, t_min=0,
                                       discretization="trailing", seed=8888)
    t0 = lap("stage2_dpm_sample_s", t0)
    video = vae.decode_video(out * 0.2, scale_factor=0.18215, decoder_bs=4)
    t0 = lap("stage2_decode_s", t0)
    total = time.perf_counter() - t_all - build_s
    return {
        "metric": "frames_per_sec_end_to_end", "value": round(F / total, 4), "unit": "frames/s", "n_gpus": world,
        "steps": n1 + 2 * n2, "warmup": 0, "ms_per_step": round(1e3 * total / (n1 + 2 * n2), 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": CONFIGS["tft2v_sr600"]["desc"], "name": "tft2v_sr600", "precision": args.precision,
                   "stage1_compositions": CONFIGS[s1]["comps"], "stage1_precision": p1,
                   "parallelism": "single GPU", "weights": "seeded synthetic (device-side random init)",
                   "smoke_run": steps1 is not None,
                   "note": "one video, cold sessions (setup + graph capture inside the timed region, as a one-shot engine "
                           "run pays them); weight synthesis of the second model excluded; " +
                           ("stage 1 runs the reference yaml's whole vcomposer composition list (six pixel-resolution condition "
                            "maps) — tests/golden/unet_vcomposer_full.pt: 1.04e-3 in 'mixed', 8.6e-4 in 'high'"
                            if args.stage1 == "vcomposer" else
                            "stage 1 runs the text + image compositions — with the whole vcomposer list (--stage1 vcomposer) at "
                            "this shape precision='mixed' measures 1.04e-3: pass --stage1-precision high for <= 1e-3")},
        "seconds_per_video": round(total, 3), "stages": times,
        "finite": fin1 and bool(torch.isfinite(out).all()), "video_shape": list(video.shape),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"])
    ap.add_argument("--config", default="t2v", choices=sorted(CONFIGS),
                    help="t2v (default, BASELINE config 2: the driver's line); videolcm = whole videos through the 4-step LCM loop "
                         "+ decode (config 4); tft2v_sr600 = the two-stage pipeline for one video (config 5; --steps < 10 runs a "
                         "2-step smoke of every stage); the others time one denoise step of that shape")
    ap.add_argument("--precision", default="calibrated",
                    help="calibrated (default, r06): every weight ONE 16-bit matrix whose rounding was calibrated at pack time "
                         "(vgen_amd/calibrate.py; single-pass launches only); mixed: packed weights as W_hi + W_lo pairs (dual-W "
                         "tap-GEMM launches) in the full-resolution level (encoder + decoder level 0, K/V projection, head); high: "
                         "two-term weights everywhere; fast: weights to nearest (the reference's autocast arithmetic; 1.33e-3); "
                         "mixed:e0d01t1-style strings select levels (vgen_amd/unet.py)")
    ap.add_argument("--variants", default="fp16/mixed,fp16/high,fp16/fast,bf16/fast",
                    help="other dtype/precision modes timed + parity-checked after the headline mode (t2v, N = 1); '' = none")
    ap.add_argument("--calibration-file", default=None,
                    help="--precision calibrated: load the timed model's calibrated weights from this file if it exists, else "
                         "calibrate and save them there (calibrate.save_calibrated) — profiling passes of one model pay the "
                         "pack-time pass once")
    ap.add_argument("--stage1", default="text_image", choices=["text_image", "vcomposer"],
                    help="--config tft2v_sr600: composition list of the first stage (vcomposer = the reference yaml's eight "
                         "entries with six pixel-resolution condition maps)")
    ap.add_argument("--stage1-precision", default=None, help="--config tft2v_sr600: precision of the first-stage UNet "
                                                             "(default: --precision); the vcomposer list needs 'high' for <= 1e-3")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-scaling-model", action="store_true")
    ap.add_argument("--partition", action="store_true",
                    help="use the multi-GPU code path (UnitPartition: session over the local units + eager "
                         "gather/update) even at --gpus 1")
    ap.add_argument("--graph-collective", action="store_true",
                    help="partitioned runs (--gpus > 1 / --partition): the whole step — local forward, RCCL all-gather, update "
                         "from the gathered buffer — as ONE hipGraph per step (vgen_amd/parallel.py ddim_step; r04, validated on "
                         "one device only: off by default)")
    ap.add_argument("--vae-size", default="256x448", help="HxW of the decoded frame")
    ap.add_argument("--dump-shapes", action="store_true", help="write per-shape kernel timings to gpurun_out/")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL).  gloo + VGEN_BENCH_ONE_DEVICE=1 runs all "
                         "ranks on cuda:0: a functional test of the multi-rank code path on a 1-GPU box, not a measurement")
    args = ap.parse_args()
    if args.no_graph:
        os.environ["VGEN_GRAPH"] = "0"
    if os.environ.get("VGEN_BENCH_WATCHDOG"):
        # diagnosis aid: dump every thread's stack and exit if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VGEN_BENCH_WATCHDOG"]), exit=True)

    import torch.distributed as dist
    from vgen_amd import ops
    from vgen_amd.diffusion import DiffusionDDIM
    from vgen_amd.parallel import UnitPartition
    from vgen_amd.synth import seeded_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    one_dev = os.environ.get("VGEN_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda", 0 if one_dev else local)
    if os.environ.get("VGEN_BENCH_DEVICE"):
        # host-logic dry run of THIS script (tools/dryrun_bench_cpu.py, test tooling): that tool pre-installs its own op
        # backend and stubs the torch.cuda calls; bench.py itself never selects anything but the HIP backend (asserted below)
        dev = torch.device(os.environ["VGEN_BENCH_DEVICE"])
    torch.cuda.set_device(dev)
    if world > 1:
        # 30 minutes instead of the default 10: the first collective is the hand-off of the calibrated weights — ranks 1 .. N-1
        # wait in it while rank 0 builds, calibrates (~85 s alone on the host) and saves, with N model builds contending for
        # the same host cores
        import datetime
        pg_timeout = datetime.timedelta(minutes=30)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
    elif args.partition and os.environ.get("VGEN_FORCE_COLLECTIVE") == "1":
        # one rank, collectives forced: the partition path's RCCL all-gather really executes on a 1-GPU box (a functional /
        # overhead measurement of the multi-GPU step path, not a scaling number)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    ops.set_backend(None)
    assert ops.backend().name == "hip"

    if args.config == "tft2v_sr600":
        res = run_two_stage(args, dev, world, rank)
        _finish(res if rank == 0 else None, world)
        return
    cfg = CONFIGS[args.config]
    C, F, H, W = cfg["latent"]
    G = cfg["G"]
    gold = sd = None
    if args.config == "t2v":
        # the golden fixture's weights: shapes + seed -> the tensors the reference's fp32 forward was recorded on
        gold = torch.load(GOLDEN_T2V, map_location="cpu", weights_only=False)
        sd = seeded_state_dict(gold["shapes"], seed=gold["seed"])
    cal_file = None
    if world > 1 and args.precision == "calibrated":
        # the persistent route (calibrate.save_calibrated / precision="calibrated", calibration=<file>): ONE rank pays the
        # pack-time pass, the others load its result — the same bits either way (the pass is deterministic)
        import tempfile
        cal_file = os.path.join(tempfile.gettempdir(), f"vgen_bench_{args.config}_{args.dtype}_{os.environ.get('MASTER_PORT', '0')}.cal")
        ok = torch.zeros(1, device=dev)
        if rank == 0:
            try:
                model = build_model(args.config, dev, args.dtype, "calibrated", state_dict=sd)
                from vgen_amd.calibrate import save_calibrated
                save_calibrated(model, cal_file)
                ok += 1
            except OSError as exc:                       # no room for the file: every rank calibrates for itself
                print(f"bench.py: could not save {cal_file} ({exc}); every rank calibrates its own copy", file=sys.stderr)
                model = None
        dist.all_reduce(ok)                              # also the barrier the other ranks wait at
        if float(ok.item()) < 1:
            cal_file = None
        if rank != 0 or model is None:
            model = build_model(args.config, dev, args.dtype, "calibrated", state_dict=sd, calibration=cal_file)
        dist.barrier()
        if rank == 0 and cal_file and os.path.exists(cal_file):
            os.remove(cal_file)
    elif args.precision == "calibrated" and args.calibration_file:
        if os.path.exists(args.calibration_file):
            model = build_model(args.config, dev, args.dtype, "calibrated", state_dict=sd, calibration=args.calibration_file)
        else:
            model = build_model(args.config, dev, args.dtype, "calibrated", state_dict=sd)
            from vgen_amd.calibrate import save_calibrated
            save_calibrated(model, args.calibration_file)
    else:
        model = build_model(args.config, dev, args.dtype, args.precision, state_dict=sd)
    if args.config == "t2v":
        drop_masters(model, dev)
    if args.config == "videolcm":
        res = run_videolcm(args, dev, model, world, rank)
        _finish(res if rank == 0 else None, world)
        return

    diff = DiffusionDDIM(**DDIM)
    diff.rng_parity = False
    P = world                                           # prompts in flight (weak scaling)
    g = torch.Generator(device=dev).manual_seed(8888)
    xt0 = torch.randn(P, C, F, H, W, generator=g, device=dev)
    kw = conditioning(args.config, model, P, dev, g)
    guide = 9.0 if G == 2 else None
    mkw = kw if G == 2 else kw[0]
    part = UnitPartition(graph_collective=args.graph_collective or None) if (world > 1 or args.partition) else None
    diff.partition = part

    inversion = None
    if args.config == "sr600":
        # the SR600 stage's own samplers (inference_tft2v_sr600_entrance.py:284-308): DDIM inversion = single forwards
        # without CFG at t = 0, 23, ...; then sample(solver='dpmpp_2m_sde') CFG steps with guide_rescale 0.3
        from vgen_amd.diffusion_gauss import DiffusionDDIMSR
        srd = DiffusionDDIMSR(**SR600_DIFF)
        srd.reverse_diffusion.partition = srd.forward_diffusion.partition = part
        zero_kw = {"y": kw[1]["y"]}
        K = args.steps
        srd.reverse_diffusion.ddim_reverse_sample_loop(xt0, model, zero_kw, guide_scale=None, ddim_timesteps=2,
                                                       reverse_steps=46)                 # setup: session + capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        noised = srd.reverse_diffusion.ddim_reverse_sample_loop(xt0, model, zero_kw, guide_scale=None,
                                                                ddim_timesteps=K, reverse_steps=23 * K)
        torch.cuda.synchronize()
        inv_s = time.perf_counter() - t0
        inversion = {"forwards": K, "seconds": round(inv_s, 4), "forwards_per_sec": round(K / inv_s, 4),
                     "api": "GaussianDiffusion.ddim_reverse_sample_loop (1 UNet fwd per step, no CFG)"}
        smp = dict(model=model, model_kwargs=kw, guide_scale=9.0, guide_rescale=0.3, solver="dpmpp_2m_sde",
                   t_max=699, t_min=0, discretization="trailing", seed=8888)
        srd.forward_diffusion.sample(noise=noised, steps=2, **smp)                        # setup
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xt = srd.forward_diffusion.sample(noise=noised, steps=K, **smp)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_s], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_s = float(tt.item())
        timer = StepTimer(diff, model, xt0, mkw, guide, dev, P)
        api = "GaussianDiffusion.sample(solver='dpmpp_2m_sde', steps=K) — K CFG model evaluations + solver updates, whole loop timed"
        cache = srd.forward_diffusion.sessions
    else:
        timer = StepTimer(diff, model, xt0, mkw, guide, dev, P)
        dt_s, xt = timer.run(args.steps, args.warmup, world)
        api = "DiffusionDDIM.ddim_sample per step (public sampler API; cached sampling session underneath)"
        cache = part.sessions if part is not None else diff.sessions
    finite = bool(torch.isfinite(xt).all())
    xt_absmax = float(xt.float().abs().nan_to_num(nan=float("inf")).max())
    sess = None
    if part is not None:
        cache = part.sessions
    if cache is not None and cache._items:
        sess = next(iter(cache._items.values()))

    steps_per_s = P * args.steps / dt_s
    res = {
        "metric": "denoise_steps_per_sec", "value": round(steps_per_s, 4), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt_s / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": cfg["desc"], "name": args.config,
                   "prompts_in_flight": P, "units_per_step": G * P, "api": api,
                   "parallelism": "single GPU" if world == 1 else
                   f"unit partition over {world} ranks, 1 all-gather/step ({args.backend}{', all ranks on one device: functional test' if one_dev else ''})"
                   + (", collective inside the step graph" if (part is not None and part.graph_collective) else ""),
                   "hipgraph": bool(sess is not None and sess.use_graph and sess._graphs) and
                   ("whole step" if (part is None and args.config != "sr600") else
                    ("whole partitioned step (forward + all-gather + update)" if (part is not None and part.graph_collective)
                     else "units' forward")),
                   "precision": args.precision,
                   # the WHOLE rule the timed model was packed under (VERDICT r04 weak #12 / ADVICE r04): level sets, the layer
                   # kinds kept single-pass inside them, the kinds added at other levels, two-term activation operands
                   "two_term_weights": ({"levels": getattr(model, "MIXED_LEVELS", None),
                                         "single_pass_kinds_inside_those_levels": list(getattr(model, "MIXED_SINGLE_KINDS", ())),
                                         "extra_kinds_by_level": {str(k): list(v) for k, v in
                                                                  getattr(model, "MIXED_EXTRA_KINDS", {}).items()},
                                         "plus": "context K/V projection, head conv"}
                                        if getattr(model, "precision", "") == "mixed" else
                                        {"high": "all", "calibrated": "none: every packed weight is ONE 16-bit matrix whose rounding "
                                         "was calibrated at pack time (vgen_amd/calibrate.py) — single-pass launches only"}
                                        .get(getattr(model, "precision", ""), "none (the 4 -> 320 stem only)")),
                   "two_term_activations": bool(getattr(model, "_asplit", False)),
                   "weights": "seeded synthetic (vgen_amd/synth.py)" + (": the golden fixture's" if gold is not None else "")},
        "finite": finite, "latent_absmax_after_timed_steps": xt_absmax,
        "model_tflops_per_s": round(G * cfg["tflop"] * steps_per_s, 2),
        "frac_of_mfma_peak": round(G * cfg["tflop"] * steps_per_s / world / PEAK_TFLOPS, 4),
    }
    if inversion is not None:
        res["inversion"] = inversion
    if getattr(model, "bench_calibration", None) is not None:
        res["calibration"] = dict(model.bench_calibration)
        if rank == 0 and world == 1 and args.config == "t2v":
            from vgen_amd.calibrate import packed_digest
            res["calibration"]["packed_digest"] = packed_digest(model)[:16]

    # ---- parity of the model that was just timed, computed here --------------------------------------------
    if rank == 0 and gold is not None and not args.no_parity:
        fx = {FIXTURE_NAMES[0]: golden_parity(model, gold, dev)}
        if os.path.exists(GOLDEN_T2V_C):
            gc_ = torch.load(GOLDEN_T2V_C, map_location="cpu", weights_only=False)
            assert gc_["seed"] == gold["seed"] and gc_.get("recipe", "gauss") == "gauss"
            fx[FIXTURE_NAMES[1]] = golden_parity(model, gc_, dev)
        if os.path.exists(GOLDEN_T2V_B) and world == 1:
            gb_ = torch.load(GOLDEN_T2V_B, map_location="cpu", weights_only=False)
            mb = build_model("t2v", dev, args.dtype, args.precision,
                             state_dict=seeded_state_dict(gb_["shapes"], seed=gb_["seed"], recipe=gb_["recipe"]))
            fx[FIXTURE_NAMES[2]] = golden_parity(mb, gb_, dev)
            if getattr(mb, "bench_calibration", None) is not None:
                res["calibration_t2v_full_b"] = mb.bench_calibration
            del mb
            gc.collect()
            torch.cuda.empty_cache()
        res["parity"] = parity_block(fx, args.dtype, args.precision)
        err = res["parity"]["unet_rel_l2"]
        if err > TOLERANCE:
            # a headline whose in-run parity is outside the north-star's tolerance must say so loudly
            res["parity_exceeds_tolerance"] = True
            print(f"bench.py: WARNING parity {err:.3e} > tolerance {TOLERANCE:.0e} in dtype={args.dtype} "
                  f"precision={args.precision}: this line is NOT a measurement of the north-star claim", file=sys.stderr)
        ypath = os.path.join(ROOT, "tests", "golden", "autocast_yardstick.json")
        if os.path.exists(ypath):
            val = json.load(open(ypath)).get(f"unet_t2v_full/{args.dtype}")
            if val is not None:      # how far the reference's OWN autocast forward lands from its fp32 forward (committed fixture)
                res["parity"]["committed_reference_own_autocast_rel_l2"] = val

    # ---- what a 2-GPU split of ONE video can reach (north_star: "cond on GPU 0, uncond on GPU 1") ----------------
    # A unit-major rank runs ONE unit's forward + the update; the pair step above runs two units in one batch.  Their ratio
    # bounds the strong scaling of a single video on 2 GPUs before any collective cost; weak scaling (P = N prompts, what
    # --gpus N measures) keeps whole pairs per rank.  No 1 -> 8 curve has been measured on hardware (the driver's to run).
    if rank == 0 and world == 1 and part is None and args.config == "t2v" and G == 2 and not args.no_scaling_model:
        from vgen_amd.session import SessionCache
        sc = SessionCache(capacity=1)
        s1 = sc.get(model, tuple(xt0.shape), dev, list(kw), torch.long, 1000, units=[0])
        tt = timer.t_of(0)
        for _ in range(3):
            s1.eval(xt0, tt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            s1.eval(xt0, tt)
        torch.cuda.synchronize()
        unit_ms = 1e2 * (time.perf_counter() - t1)
        pair_ms = 1e3 * dt_s / args.steps
        res["scaling_model"] = {
            "precision": args.precision, "single_unit_forward_ms": round(unit_ms, 3), "pair_step_ms": round(pair_ms, 3),
            "predicted_2gpu_strong_scaling_bound_one_video": round(pair_ms / unit_ms, 3),
            "note": "measured at N = 1: one unit's forward (what each rank of the cond | uncond split runs) vs the 2-unit step; "
                    "an upper bound before the all-gather; weak scaling (--gpus N: N prompts) is not bounded by it; "
                    "no multi-GPU curve has been measured",
            "launch_line": "python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                           "--master-port 29500 bench.py --gpus N --steps 20 --warmup 5 [--graph-collective]"}
        del s1, sc
        gc.collect()
        torch.cuda.empty_cache()

    # ---- roofline of the dominant kernel (instrumented eager pass, same step) ----------------------
    if rank == 0 and not args.no_roofline:
        try:
            res.update(roofline_pass(args, model, timer, xt0, kw, G, guide, args.precision))
        except Exception as exc:                                # noqa: BLE001 — an instrumentation failure must not cost the line
            res["roofline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # ---- VAE decode frames/s + a whole video end to end --------------------------------------------------
    if rank == 0 and not args.no_vae:
        from vgen_amd.vae import AutoencoderKL
        fh, fw = (int(v) for v in args.vae_size.split("x"))
        vae = build_vae(dev, args.dtype, args.precision)
        z = torch.randn(2, 4, fh // 8, fw // 8, device=dev) / 0.18215 * 0.2
        for _ in range(2):
            vae.decode(z)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nrep = 4
        for _ in range(nrep):
            vae.decode(z)
        torch.cuda.synchronize()
        fps = 2 * nrep / (time.perf_counter() - t1)
        res["vae"] = {"decode_frames_per_sec": round(fps, 2), "frame": args.vae_size, "decoder_bs": 2,
                      "precision": vae.precision,
                      # what the decode is held to.  The north-star states 1e-3 for the UNet only; the VAE's
                      # bound is the reference's OWN arithmetic: its fp16-autocast decode lands 1.90e-3 from its fp32
                      # decode (tests/golden/autocast_yardstick.json); here 1.37e-3 in "fast", 1.03e-3 in "high" (full-size
                      # 256 x 448 golden, tests/test_gpu_model.py).  A weight-only rule cannot reach 1.1e-3 cheaply: with
                      # EVERY weight two-term the decode is still at 1.03e-3 (activation roundings), so 82 % of the weight
                      # error energy would have to go — i.e. nearly all of "high"'s cost.
                      "tolerance": {"bound": "reference fp16-autocast yardstick 1.90e-3", "rel_l2_committed": {
                          "fast": 1.37e-3, "high": 1.03e-3}, "source": "tests/golden/vae_sd_full*.pt via tests/test_gpu_model.py"}}
        if getattr(vae, "bench_calibration", None) is not None:
            res["vae"]["calibration"] = vae.bench_calibration
        if args.vae_size == "256x448":
            res["vae"]["tflops_per_s"] = round(fps * VAE_DEC_TFLOP, 2)
            # the 720p frames of BASELINE configs 3 / 5 (SR600 / I2VGen decode and the SR600 stage's encode), 2 frames a call
            z7 = torch.randn(2, 4, 90, 160, device=dev) / 0.18215 * 0.2
            x7 = torch.rand(2, 3, 720, 1280, device=dev) * 2 - 1
            for fn, arg, key in ((vae.decode, z7, "decode_720p_frames_per_sec"), (vae.encode_firsr_stage, x7, "encode_720p_frames_per_sec")):
                fn(arg)
                fn(arg)                                 # eager warm-up, then the call that captures the chunk graph
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    fn(arg)
                torch.cuda.synchronize()
                res["vae"][key] = round(4 / (time.perf_counter() - t1), 2)
        if vae.precision != "fast" and args.vae_size == "256x448":
            # the single-pass VAE next to it: the north-star states a tolerance for the UNet only, and this mode (decode 1.37e-3
            # from the reference's fp32 decode) is already closer than the reference's own autocast arithmetic (1.90e-3)
            with torch.device(dev):
                vf = AutoencoderKL(ddconfig=VAE_SD, embed_dim=4, compute_dtype=args.dtype, precision="fast")
            vf.eval()
            randomize_(vf, 1)
            for _ in range(3):
                vf.decode(z)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nrep):
                vf.decode(z)
            torch.cuda.synchronize()
            res["vae"]["decode_frames_per_sec_precision_fast"] = round(2 * nrep / (time.perf_counter() - t1), 2)
            del vf
        if args.dump_shapes:
            ops.KERNEL_PROFILE = []
            torch.cuda.synchronize()
            torch.cuda._sleep(int(2e8))
            import vgen_amd.session as _vs
            _g_on, _vs._GRAPH_ON = _vs._GRAPH_ON, False         # per-launch events need the eager launch sequence
            vae.decode(z)
            _vs._GRAPH_ON = _g_on
            torch.cuda.synchronize()
            vrec = ops.KERNEL_PROFILE
            ops.KERNEL_PROFILE = None
            agg = {}
            for r in vrec:
                a = agg.setdefault((r[0],) + tuple(r[4]), [0, 0.0, 0.0])
                a[0] += 1
                a[1] += r[1].elapsed_time(r[2])
                a[2] += r[3]
            rows = sorted(([list(map(str, k)), v[0], round(v[1], 4), round(v[2] / (v[1] * 1e-3) / 1e9, 1)] for k, v in agg.items()),
                          key=lambda r: -r[2])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"vae_shapes_{args.vae_size}_{vae.precision}.json"), "w") as f:
                json.dump({"cols": ["(op,shape...)", "launches", "ms", "GFLOP/s or GB/s"], "rows": rows,
                           "total_ms": round(sum(v[1] for v in agg.values()), 3), "frames": 2}, f, indent=0)
        if not args.no_e2e and world == 1 and part is None and args.config == "t2v":
            # the engine's whole job for one prompt (inference_text2video_entrance.py:194-217): 50-step
            # ddim_sample_loop + 16 frames decoded 2 at a time to uint8 video
            vae.decode_video(xt0 * 0.2, scale_factor=0.18215, decoder_bs=2)         # the decode chunk's graph is captured
            vae.decode_video(xt0 * 0.2, scale_factor=0.18215, decoder_bs=2)         # (untimed setup, like the step graph)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lat = diff.ddim_sample_loop(xt0, model, mkw, guide_scale=guide, ddim_timesteps=50, eta=0.0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            vid = vae.decode_video(lat * 0.2, scale_factor=0.18215, decoder_bs=2)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            res["e2e"] = {"video": "16 frames 448x256, 50 DDIM CFG steps + AutoencoderKL decode to uint8",
                          "unet_precision": args.precision,
                          "loop50_s": round(t2 - t1, 4), "loop50_steps_per_sec": round(50 / (t2 - t1), 3),
                          "decode16_s": round(t3 - t2, 4),
                          "frames_per_sec": round(F / (t3 - t1), 3), "video_shape": list(vid.shape)}
        del vae

    # ---- the other modes, same measurements (timed steps + parity), after the headline ------------------------------------
    if rank == 0 and world == 1 and part is None and args.config == "t2v" and args.variants:
        res["variants"] = {}
        del model, timer, sess
        diff.sessions.clear()
        gc.collect()
        torch.cuda.empty_cache()
        for v in [s for s in args.variants.split(",") if s]:
            vdt, vpr = v.split("/")
            if (vdt, vpr) == (args.dtype, args.precision):
                continue
            try:                                                # a variant must not be able to cost the line
                vm = build_model("t2v", dev, vdt, vpr, state_dict=sd)
                drop_masters(vm, dev)
                vd = DiffusionDDIM(**DDIM)
                vd.rng_parity = False
                vt = StepTimer(vd, vm, xt0, mkw, guide, dev, P)
                k, w = min(args.steps, 10), min(args.warmup, 2)
                vdt_s, vx = vt.run(k, w)
                ent = {"value": round(k / vdt_s, 4), "unit": "steps/s", "ms_per_step": round(1e3 * vdt_s / k, 3), "steps": k,
                       "warmup": w, "dtype": vdt, "precision": vpr, "finite": bool(torch.isfinite(vx).all()),
                       "latent_absmax_after_timed_steps": float(vx.float().abs().nan_to_num(nan=float("inf")).max())}
                if not args.no_parity:
                    e = golden_parity(vm, gold, dev)
                    ent.update(unet_rel_l2=e, within_tolerance=bool(e <= TOLERANCE))
                if getattr(vm, "bench_calibration", None) is not None:
                    ent["calibration"] = vm.bench_calibration
                del vm, vd, vt, vx
            except Exception as exc:                            # noqa: BLE001
                import traceback
                ent = {"failed": f"{type(exc).__name__}: {exc}"[:300], "traceback_tail": traceback.format_exc()[-700:]}
            res["variants"][v] = ent
            gc.collect()
            try:
                torch.cuda.empty_cache()
            except Exception:                                   # noqa: BLE001 — a device error inside a variant: the
                pass                                            # measurements taken before it still print

    # ---- CPU baseline on the host cores, bounded sample ----------------------------------------------------------
    # Sample = ONE full forward of the full-size UNetSD_T2VBase (1411 M params) on the whole 16-frame latent
    # [1,4,16,32,56] — half a CFG step, ~15-20 s on 32 threads; a step is two such forwards.  32 threads: more made
    # the fp32 torch forward slower on the 256-core host (363 s per forward with 256 threads in an earlier run).
    # With the reference tree present (the build container) the forward is the REFERENCE's own UNetSD_T2VBase through
    # oracle/ref_import.py (kind "reference"); on the GPU box, which has no /root/reference, the port (oracle/torch_ref.py).
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "t2v":
        from oracle import ref_import, torch_ref
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        gen = torch.Generator("cpu").manual_seed(8888)
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)
        yy = torch.randn(1, 77, 1024, generator=gen)
        tt = torch.tensor([981])
        if ref_import.available():
            R = ref_import.load()
            rm = R["MODEL"].build(dict(type="UNetSD_T2VBase", **gold["cfg"])).eval()
            rm.load_state_dict(sd, strict=True)
            with torch.no_grad():
                t1 = time.perf_counter()
                rm(x, tt, y=yy)
                fwd_s = time.perf_counter() - t1
            kind, what = "reference", "the reference's UNetSD_T2VBase.forward (tools/modules/unet/unet_t2v.py:210-277, fp32, via oracle/ref_import.py)"
        else:
            with torch.no_grad():
                t1 = time.perf_counter()
                torch_ref.unet_forward(sd, x, tt, yy, 320)
                fwd_s = time.perf_counter() - t1
            kind, what = "port", "oracle/torch_ref.py (fp32 restatement of the reference forward; /root/reference is not on this box)"
        # the VAE leg BASELINE.md §2 promised: one 256x448 frame (latent [1,4,32,56], 1.09 TFLOP) through the fp32
        # decoder on the same host threads — the reference's AutoencoderKL when its tree is present, else the port
        vshapes = {k: tuple(v.shape) for k, v in __import__("vgen_amd.vae", fromlist=["AutoencoderKL"]).AutoencoderKL(
            ddconfig=VAE_SD, embed_dim=4).state_dict().items()}
        vsd = seeded_state_dict(vshapes, seed=0)
        zz = torch.randn(1, 4, 32, 56, generator=gen) / 0.18215 * 0.2
        with torch.no_grad():
            if ref_import.available():
                rv = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=dict(VAE_SD, video_kernel_size=[3, 1, 1]),
                                                  embed_dim=4)).eval()
                rv.load_state_dict(vsd, strict=True)
                t1 = time.perf_counter()
                rv.decode(zz)
                vae_s = time.perf_counter() - t1
            else:
                t1 = time.perf_counter()
                torch_ref.vae_decode(vsd, zz)
                vae_s = time.perf_counter() - t1
        cpu_model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        res["cpu_baseline"] = {"value": round(1.0 / (2 * fwd_s), 5), "unit": "steps/s", "cores": cores, "kind": kind,
                               "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
                               "vae_decode_frames_per_sec": round(1.0 / vae_s, 4),
                               "vae_sample": f"one 256x448 frame through the fp32 AutoencoderKL decoder ({kind}): {vae_s:.2f} s",
                               "sample": f"{what}: one forward of the full-size UNet on the 16-frame latent "
                                         f"[1,4,16,32,56]: {fwd_s:.1f} s; a CFG step is two forwards"}
    _finish(res if rank == 0 else None, world)


if __name__ == "__main__":
    main()
