"""LCM scheduler.  PARITY UNPINNED: diffusers.LCMScheduler is not available (not vendored,
not pinned) — these tests hold the device path to the CPU restatement in oracle/torch_ref.py and to the
schedule facts the engine relies on."""
import torch

from conftest import rel_l2
from oracle import torch_ref
from oracle.make_golden import dummy_model
from vgen_amd.lcm import LCMScheduler


def _sched():
    s = LCMScheduler(prediction_type="v_prediction", beta_schedule="scaled_linear", clip_sample=False,
                     timestep_spacing="linspace", rescale_betas_zero_snr=True)       # the engine's ctor call
    s.set_timesteps(4)
    return s


def test_schedule_tables_and_timesteps():
    s = _sched()
    assert s.timesteps.tolist() == [999, 759, 499, 259] == torch_ref.lcm_timesteps(4)
    ac = torch_ref.lcm_alphas_cumprod()
    assert torch.equal(s.alphas_cumprod, ac)
    assert float(ac[-1]) == 0.0 and 0.99 < float(ac[0]) < 1.0            # zero terminal SNR
    c_skip, c_out = s.boundary_scalings(999)
    assert c_skip < 1e-8 and abs(c_out - 1.0) < 1e-8                     # far from the boundary: pure x0
    assert s.scale_model_input(ac, 5) is ac


def test_sample_loop_matches_cpu_restatement(emu_backend):
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(2, 4, 3, 6, 5, generator=g)
    kw = [dict(y=torch.randn(2, 7, 16, generator=g)), dict(y=torch.randn(2, 7, 16, generator=g))]
    step_noise = [torch.randn(noise.shape, generator=g) for _ in range(4)]
    model = lambda x, t, t_w=None, **k: dummy_model(x, t, **k)
    for guide in (None, 7.5):
        s = _sched()
        out = s.sample_loop(noise, model, kw, guidance_scale=guide, step_noise=step_noise)
        ref = torch_ref.lcm_sample_loop(torch_ref.lcm_alphas_cumprod(), torch_ref.lcm_timesteps(4), noise,
                                        lambda x, t, **k: dummy_model(x, t, **k), kw, guide, step_noise)
        assert out.shape == noise.shape and rel_l2(out, ref) < 1e-6
    # step(): dict / tuple returns, last step returns the denoised sample, eps-prediction path
    s = _sched()
    v = torch.randn(noise.shape, generator=g)
    r = s.step(v, s.timesteps[0], noise, noise=step_noise[0])
    assert set(r) == {"prev_sample", "denoised"} and not torch.equal(r["prev_sample"], r["denoised"])
    s._step_index = 3
    prev, den = s.step(v, s.timesteps[3], noise, return_dict=False)
    assert torch.equal(prev, den)
    # reset(): the scheduler lets go of whatever session it cached (model reference, graphs), and keeps working
    s.sessions._items["k"] = object()
    s.reset()
    assert not s.sessions._items and s._step_index is None


def test_scheduler_matches_hand_derived_known_answers(emu_backend):
    """a23 pinned to something OUTSIDE this repository's code: tests/golden/lcm_kat.json is computed by
    tests/golden/make_lcm_kat.py in pure-Python float64 straight from the published formulas — LCM eq. (9) boundary scalings
    c_skip / c_out (sigma_data 0.5, timestep scaling 10), the skipping-step timestep grid, the scaled-linear betas with the
    zero-terminal-SNR rescale, the v-parameterised x0 and the multistep update — with no import of torch, vgen_amd or
    oracle/.  Checked: the 2 / 4 / 8-step timestep lists, alphas_cumprod at six timesteps, the boundary scalings of the four
    steps, and a whole 4-step loop on a 2 x 2 latent with a fixed model output and fixed step noise."""
    import json
    import os
    from conftest import GOLD
    from vgen_amd.lcm import LCMScheduler
    kat = json.load(open(os.path.join(GOLD, "lcm_kat.json")))
    s = LCMScheduler(prediction_type="v_prediction", beta_schedule="scaled_linear", clip_sample=False,
                     timestep_spacing="linspace", rescale_betas_zero_snr=True)
    for n in (2, 4, 8):
        assert s.set_timesteps(n).tolist() == kat[f"timesteps_{n}"]
    for t, a in kat["alphas_cumprod"].items():
        got = float(s.alphas_cumprod[int(t)])                   # fp32 tables, like the package restated
        assert abs(got - a) <= 3e-6 * max(a, 1e-3) + 1e-9, (t, got, a)
    for t, (cs, co) in kat["boundary"].items():
        g_cs, g_co = s.boundary_scalings(int(t))
        assert abs(g_cs - cs) <= 1e-12 * max(cs, 1e-30) + 1e-18 and abs(g_co - co) <= 1e-12
    s.set_timesteps(4)
    x = torch.tensor(kat["x"], dtype=torch.float32).reshape(1, 1, 1, 2, 2)
    v = torch.tensor(kat["v"], dtype=torch.float32).reshape(1, 1, 1, 2, 2)
    z = torch.tensor(kat["z"], dtype=torch.float32).reshape(1, 1, 1, 2, 2)
    s._step_index = None
    for st in kat["steps"]:
        prev, den = s.step(v, st["t"], x, return_dict=False, noise=z)
        for got, want in ((den, st["denoised"]), (prev, st["prev_sample"])):
            want = torch.tensor(want, dtype=torch.float64).reshape(got.shape)
            assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()), (st["t"], got, want)
        x = prev
