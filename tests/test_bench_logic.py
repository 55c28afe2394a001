"""bench.py's host logic that needs no GPU: the parity object of a mode, and that `--precision calibrated` (the r06 default of
every leg) is an ordinary precision of build_model — packed two-term, calibrated on the family's seeded calibration batch,
deterministic, and loadable by another rank from the file rank 0 saved."""
import types

import torch

import bench


def _fx(a, b, c):
    return dict(zip(bench.FIXTURE_NAMES, (a, b, c)))


def test_parity_block_is_the_max_over_fixtures_against_the_tolerance():
    p = bench.parity_block(_fx(7e-4, 9.99e-4, 5e-4), "fp16", "calibrated")
    assert p["unet_rel_l2"] == 9.99e-4 and p["within_tolerance"] and p["tolerance"] == 1e-3 and len(p["fixtures"]) == 3
    assert not bench.parity_block(_fx(7e-4, 1.01e-3, 5e-4), "fp16", "mixed")["within_tolerance"]


def test_defaults_name_the_calibrated_mode_on_every_leg():
    src = open(bench.__file__).read()
    assert 'ap.add_argument("--precision", default="calibrated"' in src
    assert "select_headline" not in src and "time_budget_s" not in src


def test_build_model_calibrated_is_deterministic_and_reloads_from_a_saved_file(emu_backend, tmp_path, monkeypatch):
    """A small t2v-shaped config through bench.build_model on the CPU emulator: two builds pack the same bits; a third build
    from the file the first one saved (the N > 1 hand-off) too."""
    from vgen_amd import calibrate as cal
    cfg = dict(bench.UNET_T2V, dim=64, dim_mult=[1, 2], num_heads=1, attn_scales=[1.0, 0.5], num_res_blocks=1, y_dim=1024,
               context_dim=1024)
    monkeypatch.setitem(bench.CONFIGS, "t2v", dict(bench.CONFIGS["t2v"], cfg=cfg, latent=(4, 4, 8, 8)))
    monkeypatch.setattr(cal, "calibration_units", lambda shape: 4)
    m1 = bench.build_model("t2v", "cpu", "fp16", "calibrated")
    assert m1.precision == "calibrated" and m1.bench_calibration["two_term_left"] == 0
    assert m1.bench_calibration["calibrated"] > 10 and m1.bench_calibration["timesteps"] == [875, 625, 375, 125]
    assert "layers" not in m1.bench_calibration and len(m1.bench_calibration["layers_digest"]) == 16
    m2 = bench.build_model("t2v", "cpu", "fp16", "calibrated")
    assert cal.packed_digest(m1) == cal.packed_digest(m2)
    assert m1.bench_calibration["layers_digest"] == m2.bench_calibration["layers_digest"]
    path = str(tmp_path / "rank0.cal")
    cal.save_calibrated(m1, path)
    m3 = bench.build_model("t2v", "cpu", "fp16", "calibrated", calibration=path)
    assert m3.precision == "calibrated" and cal.packed_digest(m3) == cal.packed_digest(m1)
    assert m3.bench_calibration["loaded_from"] == path and m3.bench_calibration["two_term_left"] == 0
    x = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    y = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([500])
    assert torch.equal(m1(x, t, y=y), m3(x, t, y=y))
