"""-m gpu, collected LAST (file name): full-size parity of the CALIBRATED single-pass mode (vgen_amd/calibrate.py) — the mode
the bench line is quoted in — on every full-width fixture, against the reference's recorded fp32 forward.  r06 recipe: the
model packed two-term, ONE forward on the family's calibration batch (calibrate.calibration_batch: seeded noise / prompts at
timesteps spread over the schedule — never the fixture's), the decision per launch by rule (K <= 9000 and rows >= 2 K), no
wall clock anywhere.  Every result lands in gpurun_out/parity_calibrated.json (committed as profiles/r06_parity_calibrated.json).

The driver gives `pytest -m gpu` 1200 s and a calibration is ~1.5 min of model build + host factorisations.  Measured in r06
(profiles/r06d_pytest_gpu.log): the whole suite with THREE full-size calibrations took 1039 s — 161 s of headroom on a pool
whose boxes differ by +-6 %.  So the default suite runs two: t2v (both fixtures + the six-step trajectory on ONE calibrated
model) and VideoLCM (~930 s, where r05's driver run was).  VGEN_GPU_SLOW=1 adds the other six — heavy-tailed t2v_b, TFT2V, SR600,
I2VGen x 2, vcomposer — which the builder ran this round (profiles/r06_parity_calibrated.json: all <= 1e-3); t2v_b in this mode
is also calibrated and checked inside every bench.py run (parity.fixtures on the line)."""
import json
import os

import pytest
import torch

from conftest import ROOT, gold, rel_l2
from test_gpu_model import DEV, NORTH_STAR

pytestmark = pytest.mark.gpu


def _record(key, val):
    p = os.path.join(ROOT, "gpurun_out", "parity_calibrated.json")
    os.makedirs(os.path.dirname(p), exist_ok=True)
    d = json.load(open(p)) if os.path.exists(p) else {}
    d[key] = val
    json.dump(d, open(p, "w"), indent=1)


def _brief(rep):
    from vgen_amd.calibrate import brief_report
    return brief_report(rep)


def test_calibrated_t2v_meets_the_north_star_on_both_fixtures_and_along_the_trajectory(hip_backend):
    """The bench's model (Gaussian weights seed 0): t = 981 and t = 501 fixtures <= 1e-3, then the first six full-size CFG
    DDIM steps of the reference's own loop (tests/golden/ddim_traj6_full.pt) through the public per-step sampler call."""
    import full_cases as fc
    from vgen_amd import calibrate as cal
    from vgen_amd.diffusion import DiffusionDDIM
    g = fc.load("t2v")
    m = fc.build("t2v", g, "high", DEV)
    rep = fc.calibrate("t2v", m, g, DEV)
    assert rep["two_term_left"] == 0 and rep["calibrated"] > 300 and m.precision == "calibrated", _brief(rep)
    errs = {}
    for name in ("t2v", "t2v_c"):
        f = fc.load(name)
        assert f["seed"] == g["seed"]
        errs[name], nr = fc.error(fc.forward(name, m, f, DEV), f)
        assert abs(nr - 1.0) < 5e-3
    _record("unet_t2v_full/fp16/calibrated", dict(errs, report=_brief(rep), packed_digest=cal.packed_digest(m)))
    assert max(errs.values()) <= NORTH_STAR, errs
    tr = gold("ddim_traj6_full.pt")
    gen = torch.Generator("cpu").manual_seed(tr["noise_seed"])
    xt = torch.randn(1, 4, 16, 32, 56, generator=gen).to(DEV)
    y, y_u = torch.randn(1, 77, 1024, generator=gen), torch.randn(1, 77, 1024, generator=gen)
    d = DiffusionDDIM(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
                      mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False)
    d.rng_parity = False
    kw = [dict(y=y.to(DEV)), dict(y=y_u.to(DEV))]
    drift = {}
    for i, step in enumerate(tr["steps"].tolist()):
        t = torch.full((1,), int(step), dtype=torch.long, device=DEV)
        xt, x0 = d.ddim_sample(xt, t, m, kw, guide_scale=tr["guide_scale"], ddim_timesteps=tr["ddim_timesteps"], eta=0.0)
        drift[int(step)] = dict(xt=rel_l2(xt, tr["xt"][i]), x0=rel_l2(x0, tr["x0"][i]))
    _record("ddim_traj6_full/fp16/calibrated", drift)
    # the bounds the mixed mode is held to (tests/test_gpu_model.py): x_t drift <= 4e-4 over six steps, x0 <= 6e-3
    assert all(v["xt"] <= 4e-4 and v["x0"] <= 6e-3 for v in drift.values()), drift


_SLOW = pytest.mark.skipif(os.environ.get("VGEN_GPU_SLOW") != "1",
                           reason="~1.5 min of model build + host factorisations each; the driver's GPU suite has 1200 s: "
                                  "VGEN_GPU_SLOW=1 runs them (the builder's run: profiles/r06_parity_calibrated.json)")


@pytest.mark.parametrize("name", ["videolcm", pytest.param("t2v_b", marks=_SLOW), pytest.param("tft2v", marks=_SLOW), pytest.param("sr600", marks=_SLOW),
                                  pytest.param("i2vgen", marks=_SLOW), pytest.param("i2vgen_b", marks=_SLOW),
                                  pytest.param("vcomposer", marks=_SLOW)])
def test_calibrated_single_pass_on_the_other_full_width_fixtures(hip_backend, name):
    """Heavy-tailed t2v weights, UNetSD_VideoLCM [1,4,16,32,56], UNetSD_TFT2V [1,4,16,64,112], UNetSD_SR600 [1,4,32,90,160],
    UNetSD_I2VGen [1,4,16,88,160] (two weight / input tuples), the 32-frame vcomposer trunk [1,4,32,64,112]."""
    import full_cases as fc
    from vgen_amd import calibrate as cal
    g = fc.load(name)
    m = fc.build(name, g, "high", DEV)
    rep = fc.calibrate(name, m, g, DEV)
    assert rep["two_term_left"] == 0 and rep["calibrated"] > 250, _brief(rep)
    out = fc.forward(name, m, g, DEV)
    err, nr = fc.error(out, g)
    _record(f"unet_{name}_full/fp16/calibrated", dict(rel_l2=err, norm_ratio=nr, report=_brief(rep)))
    assert err <= NORTH_STAR, err
    assert abs(nr - 1.0) < 5e-3


@_SLOW
def test_calibration_auto_full_size_through_the_public_sampler(hip_backend):
    """`precision="calibrated", calibration="auto"` at full size, driven only through DiffusionDDIM.ddim_sample (the engine's
    call): the first step calibrates the model before its session captures anything; afterwards the t = 981 fixture is inside
    the north-star tolerance and the packed digest is recorded next to the explicit pass's (same batch -> same bits)."""
    import full_cases as fc
    from vgen_amd import calibrate as cal
    from vgen_amd.diffusion import DiffusionDDIM
    from vgen_amd.unet import UNetSD_T2VBase
    from oracle import torch_ref
    g = fc.load("t2v")
    with torch.device("meta"):
        m = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="calibrated", calibration="auto")
    m = m.to_empty(device="cpu").eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True, assign=True)
    m = m.to(DEV)
    assert m.precision == "high" and m._auto_cal is True
    d = DiffusionDDIM(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
                      mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False)
    d.rng_parity = False
    gen = torch.Generator("cpu").manual_seed(5)
    xt = torch.randn(1, 4, 16, 32, 56, generator=gen).to(DEV)
    kw = [dict(y=torch.randn(1, 77, 1024, generator=gen).to(DEV)), dict(y=torch.randn(1, 77, 1024, generator=gen).to(DEV))]
    for step in (981, 961, 941):
        xt, _ = d.ddim_sample(xt, torch.full((1,), step, dtype=torch.long, device=DEV), m, kw, guide_scale=9.0,
                              ddim_timesteps=50, eta=0.0)
    assert m.precision == "calibrated" and m._auto_cal is None and bool(torch.isfinite(xt).all())
    assert not any(getattr(w, "vgen_dw", None) is not None for w in cal._packed_tensors(m))
    err, nr = fc.error(fc.forward("t2v", m, g, DEV), g)
    _record("unet_t2v_full/fp16/calibrated-auto", dict(t2v=err, packed_digest=cal.packed_digest(m),
                                                        report=_brief(m._calibration_report)))
    assert err <= NORTH_STAR and abs(nr - 1.0) < 5e-3, err
