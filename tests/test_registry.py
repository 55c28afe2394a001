"""The drop-in seam: yaml `type` strings resolve to the native classes through the reference's
registry API (utils/registry.py:24-153)."""
import warnings

import pytest

import vgen_amd
from vgen_amd.registry import Registry, build_from_config


def test_native_registries_resolve_reference_names():
    regs = vgen_amd.install({"MODEL": Registry("MODEL"), "AUTO_ENCODER": Registry("AUTO_ENCODER"),
                             "DIFFUSION": Registry("DIFFUSION")})
    d = regs["DIFFUSION"].build(dict(type="DiffusionDDIM", schedule="cosine",
                                     schedule_param=dict(num_timesteps=100, cosine_s=0.008), mean_type="v",
                                     var_type="fixed_small", noise_strength=0.1))
    assert d.num_timesteps == 100 and type(d).__module__ == "vgen_amd.diffusion"
    assert regs["MODEL"].get("UNetSD_T2VBase").__module__ == "vgen_amd.unet"
    assert regs["MODEL"].get("UNetSD_SR600").__module__ == "vgen_amd.unet"
    assert regs["MODEL"].get("UNetSD_I2VGen").__module__ == "vgen_amd.unet_i2vgen"
    assert regs["MODEL"].get("UNetSD_VideoLCM").__module__ == "vgen_amd.unet_videolcm"
    assert regs["MODEL"].get("UNetSD_TFT2V").__module__ == "vgen_amd.unet_videolcm"
    assert regs["DIFFUSION"].get("DiffusionDDIMSR").__module__ == "vgen_amd.diffusion_gauss"
    assert regs["AUTO_ENCODER"].get("AutoencoderKL").__module__ == "vgen_amd.vae"


def test_builder_error_contract():
    r = Registry("MODEL")
    with pytest.raises(TypeError):
        build_from_config([], r)
    with pytest.raises(KeyError):
        r.build(dict(nope=1))
    with pytest.raises(KeyError):
        r.build(dict(type="Missing"))

    @r.register_class()
    class Boom:
        def __init__(self, a):
            raise ValueError("x")

    with pytest.raises(Exception, match="Failed to init class"):
        r.build(dict(type="Boom", a=1))
    with pytest.warns(UserWarning):                      # duplicate name: warn + replace
        r.register_class("Boom")(type("Boom2", (), {}))
    assert r.get("Boom").__name__ == "Boom2"
    assert r.build(dict(type="Boom"), ).__class__.__name__ == "Boom2"


def test_extra_kwargs_and_unknown_cfg_keys_are_accepted():
    regs = vgen_amd.install({"MODEL": Registry("MODEL"), "AUTO_ENCODER": Registry("AUTO_ENCODER"),
                             "DIFFUSION": Registry("DIFFUSION")})
    import torch
    with torch.device("meta"):
        m = regs["MODEL"].build(dict(type="UNetSD_T2VBase", in_dim=4, dim=64, y_dim=1024, context_dim=1024,
                                     out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64, num_res_blocks=1,
                                     upper_len=128, default_fps=8, misc_dropout=0.4, use_checkpoint=True),
                                zero_y=None)
    assert m.out_dim == 4
    # a yaml without a `precision` key resolves to the mode that meets the north-star's 1e-3; the keyword
    # travels through the registry like any other constructor argument
    assert m.precision == "mixed" and m._asplit and m.MIXED_LEVELS["enc"] == (0,) and m.MIXED_LEVELS["dec"] == (0,)
    with torch.device("meta"):
        mf = regs["MODEL"].build(dict(type="UNetSD_T2VBase", in_dim=4, dim=64, y_dim=1024, context_dim=1024, out_dim=4,
                                      dim_mult=[1, 2], num_heads=2, head_dim=64, num_res_blocks=1, precision="fast"))
        mh = regs["MODEL"].build(dict(type="UNetSD_SR600", in_dim=4, dim=64, y_dim=1024, context_dim=1024, out_dim=4,
                                      dim_mult=[1, 2], num_heads=2, head_dim=64, num_res_blocks=1, precision="high"))
    assert mf.precision == "fast" and not mf._asplit and mh.precision == "high"


@pytest.mark.reference
def test_install_overrides_the_reference_registries_in_place():
    from oracle import ref_import
    R = ref_import.load()
    names = (("MODEL", "UNetSD_T2VBase"), ("MODEL", "UNetSD_SR600"), ("MODEL", "UNetSD_I2VGen"), ("MODEL", "UNetSD_VideoLCM"), ("MODEL", "UNetSD_TFT2V"),
             ("AUTO_ENCODER", "AutoencoderKL"), ("DIFFUSION", "DiffusionDDIM"), ("DIFFUSION", "DiffusionDDIMSR"))
    orig = {(k, n): R[k].get(n) for k, n in names}
    try:
        regs = vgen_amd.install()
        assert regs["MODEL"] is R["MODEL"]
        assert R["MODEL"].get("UNetSD_T2VBase").__module__ == "vgen_amd.unet"
        d = R["DIFFUSION"].build(dict(type="DiffusionDDIM", schedule="cosine",
                                      schedule_param=dict(num_timesteps=1000, cosine_s=0.008,
                                                          zero_terminal_snr=True), mean_type="v",
                                      var_type="fixed_small"))
        assert type(d).__module__ == "vgen_amd.diffusion"
    finally:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for (k, n), cls in orig.items():
                R[k].register_class(n)(cls)
