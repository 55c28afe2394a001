"""-m gpu, collected LAST on purpose (file name): the full-size parity test of the calibrated single-pass weights
(vgen_amd/calibrate.py).  The round's GPU budget ended before this test could run on an MI355X — only its tiny-model sibling
(tests/test_gpu_model.py::test_calibrated_single_pass_tiny_on_device) and the emulator run of the same product code
(profiles/r05_emu_calibrated.txt: 8.36e-4 / 8.16e-4) exist — so under `pytest -x` it must not be able to keep any other test
from running.  The assertion is the north-star tolerance itself, unchanged."""
import os

import pytest
import torch

from conftest import gold, rel_l2
from oracle import torch_ref
from test_gpu_model import DEV, NORTH_STAR, _record

pytestmark = pytest.mark.gpu


def test_calibrated_single_pass_full_size_meets_the_north_star_tolerance(hip_backend):
    """The full-size t2v UNet with EVERY weight single-pass, roundings calibrated on another noise / prompt / timestep
    (seed 424242, t = 637), against the reference's fp32 forward on the two fixtures that share its weights (t = 981 and
    t = 501).  Emulator prediction (tools/emu_calibrated.py, profiles/r05_emu_calibrated.txt) in the assertion message; the
    first GPU run of this test is the driver's."""
    from vgen_amd.calibrate import calibrate_single_pass
    from vgen_amd.unet import UNetSD_T2VBase
    g = gold("unet_t2v_full.pt")
    with torch.device("meta"):
        m = UNetSD_T2VBase(**g["cfg"], compute_dtype="fp16", precision="high")
    m = m.to_empty(device="cpu").eval()
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=g["seed"]), strict=True, assign=True)
    m = m.to(DEV)
    cg = torch.Generator("cpu").manual_seed(424242)
    xc, yc = torch.randn(1, 4, 16, 32, 56, generator=cg).to(DEV), torch.randn(1, 77, 1024, generator=cg).to(DEV)
    rep = calibrate_single_pass(m, xc, torch.tensor([637], device=DEV), y=yc)
    assert rep["two_term_left"] == 0 and rep["calibrated"] > 300, rep
    errs = {}
    for name in ("unet_t2v_full.pt", "unet_t2v_full_c.pt"):
        f = gold(name)
        assert f["seed"] == g["seed"] and int(f["t"]) != 637
        gen = torch.Generator("cpu").manual_seed(f["input_seed"])
        x, y = torch.randn(1, 4, 16, 32, 56, generator=gen), torch.randn(1, 77, 1024, generator=gen)
        errs[name] = rel_l2(m(x.to(DEV), f["t"].to(DEV), y=y.to(DEV)), f["out"])
    _record("unet_t2v_full/fp16/calibrated", dict(errs, report={k: v for k, v in rep.items()}))
    assert max(errs.values()) <= NORTH_STAR, (errs, "emulator: see profiles/r05_emu_calibrated.txt")


_SLOW = pytest.mark.skipif(os.environ.get("VGEN_GPU_SLOW") != "1",
                           reason="1-2 minutes of host factorisations each: VGEN_GPU_SLOW=1 runs them (the default suite keeps "
                                  "to the fixtures the bench line rests on, so that its run time stays where the round measured it)")


@pytest.mark.parametrize("name,emulated", [("videolcm", 8.14e-4), pytest.param("t2v_b", 6.82e-4, marks=_SLOW),
                                           pytest.param("tft2v", 8.53e-4, marks=_SLOW)])
def test_calibrated_single_pass_on_the_other_full_width_fixtures(hip_backend, name, emulated):
    """The recipe of tools/emu_calibrated.py on the GPU: the full-width model of the fixture packed two-term, calibrated on
    noise / conditioning of seed 424242 at t = 637, then the fixture's own input against the reference's fp32 forward —
    heavy-tailed t2v weights, UNetSD_VideoLCM at [1,4,16,32,56], UNetSD_TFT2V at [1,4,16,64,112].  `emulated` = what the ABI
    emulator read for the same recipe (profiles/r05_emu_calibrated.txt); first GPU run = the driver's."""
    import full_cases as fc
    from vgen_amd.calibrate import calibrate_single_pass
    g = fc.load(name)
    m = fc.build(name, g, "high", DEV)
    x, kw = fc.inputs(name, g)
    gen = torch.Generator("cpu").manual_seed(424242)
    xc = torch.randn(x.shape, generator=gen).to(DEV)
    kwc = {k: (torch.randn(v.shape, generator=gen) if v.is_floating_point() else v).to(DEV) for k, v in kw.items()}
    rep = calibrate_single_pass(m, xc, torch.full_like(g["t"], 637).to(DEV), **kwc)
    assert rep["two_term_left"] == 0 and rep["calibrated"] > 300, rep
    out = fc.forward(name, m, g, DEV)
    err, nr = fc.error(out, g)
    _record(f"unet_{name}_full/fp16/calibrated", dict(rel_l2=err, norm_ratio=nr, emulated=emulated, report=dict(rep)))
    assert err <= NORTH_STAR, (err, emulated)
    assert abs(nr - 1.0) < 5e-3
