"""TEST TOOLING (imports oracle/): the first six full-size CFG DDIM steps (tests/golden/ddim_traj6_full.pt: the reference's
own DiffusionDDIM driving its fp32 UNet) through the PRODUCT's sampler + UNet with the ABI emulator as the op backend —
per-step drift of x_t and of the predicted x0 in a precision mode.  `mixed` has a GPU measurement to hold the emulator
against (tests/test_gpu_model.py::test_full_size_trajectory_drift_in_the_benchmarked_mode: x_t 0.9 ... 2.3e-4);
`calibrated` (vgen_amd/calibrate.py, calibrated on seed 424242 / t = 637) is a prediction.
      python tools/emu_traj6.py mixed|calibrated"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import full_cases as fc  # noqa: E402
from conftest import gold, rel_l2  # noqa: E402
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402
from vgen_amd.calibrate import brief_report, calibrate_single_pass  # noqa: E402
from vgen_amd.diffusion import DiffusionDDIM  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
    ops.set_backend(EmuBackend())
    g = fc.load("t2v")
    tr = gold("ddim_traj6_full.pt")
    m = fc.build("t2v", g, "high" if mode == "calibrated" else mode)
    if mode == "calibrated":
        cg = torch.Generator("cpu").manual_seed(424242)
        xc, yc = torch.randn(1, 4, 16, 32, 56, generator=cg), torch.randn(1, 77, 1024, generator=cg)
        t0 = time.time()
        rep = calibrate_single_pass(m, xc, torch.tensor([637]), y=yc)
        print(f"calibrate_single_pass: {time.time() - t0:.0f} s {brief_report(rep)}", flush=True)
    gen = torch.Generator("cpu").manual_seed(tr["noise_seed"])
    xt = torch.randn(1, 4, 16, 32, 56, generator=gen)
    y = torch.randn(1, 77, 1024, generator=gen)
    y_u = torch.randn(1, 77, 1024, generator=gen)
    d = DiffusionDDIM(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
                      mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False)
    d.rng_parity = False
    kw = [dict(y=y), dict(y=y_u)]
    for i, step in enumerate(tr["steps"].tolist()):
        t = torch.full((1,), int(step), dtype=torch.long)
        t0 = time.time()
        xt, x0 = d.ddim_sample(xt, t, m, kw, guide_scale=tr["guide_scale"], ddim_timesteps=tr["ddim_timesteps"], eta=0.0)
        print(f"{mode}: step {i + 1} (t = {int(step)}): emulated drift x_t {rel_l2(xt, tr['xt'][i]):.3e}  x0 "
              f"{rel_l2(x0, tr['x0'][i]):.3e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
