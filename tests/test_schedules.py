"""Host-side float64 schedule tables: bit-exact vs reference-generated fixtures (and vs the live
reference where present)."""
import pytest
import torch

from conftest import gold
from vgen_amd import schedules
from vgen_amd.diffusion import DiffusionDDIM


def test_schedules_bit_exact_vs_golden():
    s = gold("ddim.pt")["schedules"]
    assert torch.equal(schedules.beta_schedule("cosine", 1000, zero_terminal_snr=True, cosine_s=0.008), s["cosine_zts"])
    assert torch.equal(schedules.beta_schedule("linear_sd", 1000, zero_terminal_snr=True, init_beta=0.00085,
                                               last_beta=0.012), s["linear_sd_zts"])
    assert torch.equal(schedules.beta_schedule("quadratic", 1000, init_beta=None, last_beta=None), s["quadratic"])
    assert torch.equal(schedules.sigma_schedule("logsnr_cosine_interp", 1000, zero_terminal_snr=True, scale_min=2.0,
                                                scale_max=4.0, logsnr_min=-15.0, logsnr_max=15.0),
                       s["sigma_logsnr_cosine_interp_zts"])
    assert torch.equal(schedules.sigma_schedule("cosine", 1000, zero_terminal_snr=True, cosine_s=0.008),
                       s["sigma_cosine_zts"])


def test_ddim_tables_and_zero_terminal_snr():
    g = gold("ddim.pt")
    d = DiffusionDDIM(**g["cfg"])
    assert d.betas.dtype == torch.float64 and d.num_timesteps == 1000
    assert torch.equal(d.alphas_cumprod, g["tables"]["alphas_cumprod"])
    assert torch.equal(d.sqrt_recipm1_alphas_cumprod, g["tables"]["sqrt_recipm1"])
    assert d.alphas_cumprod[-1] == 0            # zero terminal SNR (SURVEY §8c probe: abar_999 = 0)
    assert abs(float(d.alphas_cumprod[981]) - 7.84e-4) < 1e-5


@pytest.mark.reference
def test_schedules_vs_live_reference():
    from oracle import ref_import
    rs = ref_import.load()["schedules"]
    for name, kw in [("cosine", dict(cosine_s=0.008)), ("cosine", dict(cosine_s=0.008, zero_terminal_snr=True)),
                     ("linear_sd", dict(init_beta=0.00085, last_beta=0.012)),
                     ("linear", dict(init_beta=0.0001, last_beta=0.02)),
                     ("quadratic", dict(init_beta=0.001, last_beta=0.02, zero_terminal_snr=True))]:
        for n in (1000, 50):
            assert torch.equal(rs.beta_schedule(name, n, **kw), schedules.beta_schedule(name, n, **kw)), (name, n)
            assert torch.equal(rs.sigma_schedule(name, n, **kw), schedules.sigma_schedule(name, n, **kw)), (name, n)
