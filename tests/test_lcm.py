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
