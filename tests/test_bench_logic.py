"""bench.py's headline selection (select_headline / parity_block) on synthetic result dicts: the calibrated single-pass
model may replace the --precision mode on the line ONLY with in-run numbers that pass every check, the replaced mode stays
under `variants`, and any doubt leaves the line as the --precision mode measured it."""
import copy
import types

import pytest

import bench


def _args(**kw):
    d = dict(dtype="fp16", precision="mixed", steps=20, warmup=5, headline="auto")
    d.update(kw)
    return types.SimpleNamespace(**d)


def _fx(a, b, c):
    return dict(zip(bench.FIXTURE_NAMES, (a, b, c)))


def _line(value=29.6, errs=(7.7e-4, 7.7e-4, 6.5e-4)):
    return {"metric": "denoise_steps_per_sec", "value": value, "ms_per_step": round(1e3 / value, 3), "steps": 20, "warmup": 5,
            "dtype": "fp16", "finite": True, "latent_absmax_after_timed_steps": 4.0,
            "model_tflops_per_s": round(17.33 * value, 2), "frac_of_mfma_peak": round(17.33 * value / 2500, 4),
            "config": {"precision": "mixed", "two_term_weights": {"levels": {"enc": [0], "dec": [0]}}},
            "parity": bench.parity_block(_fx(*errs), "fp16", "mixed"),
            "roofline": {"frac": 0.226, "traffic": 171000000}, "hbm_kernels": {"groupnorm": {"ms_per_step": 3.7}},
            "e2e": {"unet_precision": "mixed"}, "variants": {}}


def _cand(value=33.1, errs=(8.4e-4, 8.2e-4, 6.8e-4), steps=20, warmup=5, left=0, finite=True):
    par = bench.parity_block(_fx(*errs), "fp16", "calibrated")
    return {"value": value, "unit": "steps/s", "ms_per_step": round(1e3 / value, 3), "steps": steps, "warmup": warmup,
            "dtype": "fp16", "precision": "calibrated", "finite": finite, "latent_absmax_after_timed_steps": 4.1,
            "calibration": {"two_term_left": left, "seconds": 60.0}, "calibration_t2v_full_b": {"two_term_left": 0},
            "note": "n", "roofline": {"frac": 0.25, "traffic": None}, "hbm_kernels": {"groupnorm": {"ms_per_step": 3.6}},
            "parity": par, "unet_rel_l2": par["unet_rel_l2"], "within_tolerance": par["within_tolerance"]}


def test_parity_block_is_the_max_over_fixtures_against_the_tolerance():
    p = bench.parity_block(_fx(7e-4, 9.99e-4, 5e-4), "fp16", "mixed")
    assert p["unet_rel_l2"] == 9.99e-4 and p["within_tolerance"] and p["tolerance"] == 1e-3 and len(p["fixtures"]) == 3
    assert not bench.parity_block(_fx(7e-4, 1.01e-3, 5e-4), "fp16", "mixed")["within_tolerance"]


def test_calibrated_mode_is_promoted_only_with_every_check_green():
    res = _line()
    res["variants"]["fp16/calibrated"] = _cand()
    before = copy.deepcopy(res)
    bench.select_headline(res, _args(), 17.33)
    sel = res["headline_selection"]
    assert sel["selected"] == "fp16/calibrated" and all(sel["checks"].values())
    assert res["value"] == 33.1 and res["config"]["precision"] == "calibrated" and res["parity"]["precision"] == "calibrated"
    assert res["roofline"]["frac"] == 0.25 and res["roofline"]["traffic"] is None      # the calibrated model's own pass
    assert res["model_tflops_per_s"] == round(17.33 * 33.1, 2)
    assert abs(res["ms_per_step"] * res["value"] - 1e3) < 1.0
    # the replaced mode is still on the line, whole
    old = res["variants"]["fp16/mixed"]
    for k in ("value", "ms_per_step", "parity", "roofline", "hbm_kernels"):
        assert old[k] == before[k]
    assert old["steps"] == 20 and old["warmup"] == 5 and old["two_term_weights"] == before["config"]["two_term_weights"]
    assert res["variants"]["fp16/calibrated"] == {"promoted_to_headline": True}
    assert res["config"]["calibration"]["calibration"]["two_term_left"] == 0
    assert res["e2e"]["unet_precision"] == "mixed"                                     # taken on the other model: says so


@pytest.mark.parametrize("cand,why", [
    (_cand(value=29.0), "faster_or_base_outside_tolerance"),
    (_cand(errs=(8.4e-4, 1.02e-3, 6.8e-4)), "within_tolerance"),
    (_cand(steps=10, warmup=2), "same_steps_and_warmup"),
    (_cand(left=3), "no_two_term_weight_left"),
    (_cand(finite=False), "finite"),
])
def test_any_failed_check_leaves_the_line_untouched(cand, why):
    res = _line()
    res["variants"]["fp16/calibrated"] = cand
    before = copy.deepcopy(res)
    bench.select_headline(res, _args(), 17.33)
    sel = res.pop("headline_selection")
    assert sel["selected"] == "fp16/mixed" and why in sel["why"] and not sel["checks"][why]
    assert res == before


def test_a_missing_fixture_or_failed_or_fixed_headline_is_never_promoted():
    res = _line()
    c = _cand()
    c["parity"] = bench.parity_block(dict(list(_fx(8e-4, 8e-4, 8e-4).items())[:2]), "fp16", "calibrated")
    res["variants"]["fp16/calibrated"] = c
    bench.select_headline(res, _args(), 17.33)
    assert res["headline_selection"]["selected"] == "fp16/mixed" and res["value"] == 29.6
    res = _line()
    res["variants"]["fp16/calibrated"] = {"failed": "RuntimeError: x"}
    bench.select_headline(res, _args(), 17.33)
    assert res["headline_selection"]["selected"] == "fp16/mixed" and "failed" in res["headline_selection"]["why"]
    res = _line()
    res["variants"]["fp16/calibrated"] = _cand()
    bench.select_headline(res, _args(headline="fixed"), 17.33)
    assert res["headline_selection"]["selected"] == "fp16/mixed" and res["value"] == 29.6
    res = _line()
    bench.select_headline(res, _args(), 17.33)
    assert res["headline_selection"]["why"] == "calibrated mode not run"


def test_a_base_mode_outside_the_tolerance_yields_to_a_slower_calibrated_mode_inside_it():
    res = _line(errs=(1.05e-3, 7.7e-4, 6.5e-4))
    res["parity_exceeds_tolerance"] = True
    res["variants"]["fp16/calibrated"] = _cand(value=28.0)
    bench.select_headline(res, _args(), 17.33)
    assert res["headline_selection"]["selected"] == "fp16/calibrated" and res["value"] == 28.0
    assert "parity_exceeds_tolerance" not in res and not res["variants"]["fp16/mixed"]["within_tolerance"]
