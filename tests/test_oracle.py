"""CPU: the oracle (oracle/torch_ref.py) is pinned (a) against the reference's own modules run
here (marker `reference`, build container only) and (b) against the committed reference-generated
fixtures (travels to the GPU box)."""
import pytest
import torch

from conftest import gold, rel_l2
from oracle import ref_import, torch_ref


# ---- (b) fixtures ---------------------------------------------------------------------------------
def test_oracle_unet_tiny_vs_golden():
    g = gold("unet_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    out = torch_ref.unet_forward(sd, g["x"], g["t"], g["y"], g["cfg"]["dim"])
    assert rel_l2(out, g["out"]) < 2e-5


def test_oracle_vae_tiny_vs_golden():
    g = gold("vae_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    assert rel_l2(torch_ref.vae_decode(sd, g["z"]), g["dec"]) < 2e-5
    mom = torch_ref.vae_encode_moments(sd, g["img"])
    assert rel_l2(mom, g["moments"]) < 2e-5
    torch.manual_seed(g["sample_seed"])
    noise = torch.randn(g["z_sample"].shape)
    assert rel_l2(torch_ref.gaussian_sample(mom, noise, 0.18215), g["z_sample"]) < 2e-5


def test_oracle_ddim_vs_golden_bit_exact():
    from oracle.make_golden import dummy_model
    from vgen_amd.schedules import beta_schedule
    g = gold("ddim.pt")
    betas = g["schedules"]["cosine_zts"]
    out = torch_ref.ddim_sample_loop(betas, g["noise"].clone(), dummy_model, g["kw"], 9.0, 50, "v", 0.0)
    assert torch.equal(out, g["out50"])
    tabs = torch_ref.ddim_tables(betas)
    assert torch.equal(tabs["ac"], g["tables"]["alphas_cumprod"])
    assert torch.equal(tabs["sqrt_recipm1"], g["tables"]["sqrt_recipm1"])
    # timestep list of the 50-step DDIM run: 981, 961, ..., 21, 1
    ts = torch_ref.ddim_timesteps(1000, 50)
    assert ts[0] == 981 and ts[-1] == 1 and len(ts) == 50 and int(ts[1]) == 961


def test_synth_weights_are_nondegenerate_and_reproducible():
    g = gold("unet_tiny.pt")
    a = torch_ref.synth_state_dict(g["shapes"], seed=1)
    b = torch_ref.synth_state_dict(g["shapes"], seed=1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    # the layers the reference zero-initialises must be non-zero (SURVEY §8c trap)
    for k in ("out.2.weight", "middle_block.0.out_layers.3.weight", "middle_block.1.proj_out.weight",
              "middle_block.0.temopral_conv.conv4.3.weight"):
        assert a[k].abs().max() > 0


# ---- (a) live reference ---------------------------------------------------------------------------
@pytest.mark.reference
def test_oracle_unet_vs_live_reference():
    from oracle.make_golden import UNET_TINY
    R = ref_import.load()
    ref = R["MODEL"].build(dict(type="UNetSD_T2VBase", **UNET_TINY)).eval()
    sd = torch_ref.synth_state_dict(torch_ref.shapes_of(ref), seed=7)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 3, 8, 8, generator=g)        # odd frame count, other resolution
    y = torch.randn(1, 5, 1024, generator=g)           # short context
    t = torch.tensor([37])
    with torch.no_grad():
        assert rel_l2(torch_ref.unet_forward(sd, x, t, y, UNET_TINY["dim"]), ref(x, t, y=y)) < 2e-5


@pytest.mark.reference
def test_oracle_blocks_vs_live_reference():
    R = ref_import.load()
    U = R["util"]
    g = torch.Generator().manual_seed(0)

    def synth(m, seed):
        sd = torch_ref.synth_state_dict(torch_ref.shapes_of(m), seed=seed)
        m.load_state_dict(sd, strict=True)
        return sd

    with torch.no_grad():
        rb = U.ResBlock(128, 256, 0.1, out_channels=64, use_scale_shift_norm=False).eval()
        sd = synth(rb, 1)
        x, e = torch.randn(6, 128, 5, 7, generator=g), torch.randn(6, 256, generator=g)
        assert rel_l2(torch_ref.resblock({"rb." + k: v for k, v in sd.items()}, "rb", x, e, 2), rb(x, e, 2)) < 2e-5
        st = U.SpatialTransformer(128, 2, 64, depth=1, context_dim=96, use_linear=True).eval()
        sd = synth(st, 2)
        x, c = torch.randn(3, 128, 4, 5, generator=g), torch.randn(3, 9, 96, generator=g)
        assert rel_l2(torch_ref.spatial_transformer({"m." + k: v for k, v in sd.items()}, "m", x, c), st(x, c)) < 2e-5
        tt = U.TemporalTransformer(64, 2, 64, depth=1, context_dim=96).eval()      # inner 128 != 64
        sd = synth(tt, 3)
        x = torch.randn(2, 64, 6, 3, 4, generator=g)
        assert rel_l2(torch_ref.temporal_transformer({"m." + k: v for k, v in sd.items()}, "m", x), tt(x)) < 2e-5


@pytest.mark.reference
def test_oracle_vae_and_sampler_vs_live_reference():
    from oracle.make_golden import DDIM_T2V, VAE_TINY, dummy_model
    R = ref_import.load()
    vae = R["AUTO_ENCODER"].build(dict(type="AutoencoderKL", ddconfig=VAE_TINY, embed_dim=4)).eval()
    sd = torch_ref.synth_state_dict(torch_ref.shapes_of(vae), seed=4)
    vae.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        z = torch.randn(1, 4, 6, 10, generator=g)
        assert rel_l2(torch_ref.vae_decode(sd, z), vae.decode(z)) < 2e-5
        img = torch.randn(1, 3, 48, 80, generator=g)
        assert rel_l2(torch_ref.vae_encode_moments(sd, img), vae.encode(img).parameters) < 2e-5
    diff = R["DIFFUSION"].build(dict(type="DiffusionDDIM", **DDIM_T2V))
    noise = torch.randn(1, 4, 2, 4, 4, generator=g)
    kw = [dict(y=torch.randn(1, 3, 8, generator=g)), dict(y=torch.randn(1, 3, 8, generator=g))]
    a = diff.ddim_sample_loop(noise.clone(), dummy_model, kw, guide_scale=7.5, ddim_timesteps=25, eta=0.0)
    b = torch_ref.ddim_sample_loop(diff.betas, noise.clone(), dummy_model, kw, 7.5, 25, "v", 0.0)
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------
# the ABI emulator (the reference of the kernel-level GPU tests) against torch's own operators
def test_abi_emulator_vs_plain_torch_operators():
    """oracle/abi_emulator.py restates every kernel on the ABI's row layouts; here each family is recomputed with
    F.conv2d / F.conv1d / F.linear / F.group_norm / F.layer_norm / F.scaled_dot_product_attention on NCHW tensors
    (tests/torch_ops_ref.py) for every kernel-test case: tap order, asymmetric / stride-2 padding, the folded nearest
    upsample + crop, the (3,1,1) temporal taps, packed GEGLU columns, the attention stride regimes."""
    import kernel_cases as kc
    import torch_ops_ref as tr
    from oracle.abi_emulator import EmuBackend
    emu = EmuBackend()
    dt = torch.float16
    for name, spec in kc.tapgemm_cases(dt).items():
        ref = tr.ref_tapgemm(spec)
        out = emu.tapgemm(kc._clone_spec(spec, "cpu")).float()
        tol = 1e-3 if spec.out_dtype != torch.float32 else 2e-5       # 16-bit outputs: one rounding of the result
        if spec.split_out:                                            # two-term rows: hi + lo is the fp32 value again
            out, tol = out[:, : spec.N] + out[:, spec.N:], 2e-5
        assert rel_l2(out, ref) < tol, (name, rel_l2(out, ref))
    g = torch.Generator().manual_seed(3)
    for nb, S, C1, C2, silu in [(2, 96, 64, 0, True), (3, 40, 64, 32, False), (1, 128, 320, 0, True)]:
        x1 = torch.randn(nb * S, C1, generator=g) * 1.7 + 0.6
        x2 = torch.randn(nb * S, C2, generator=g) if C2 else None
        ga, be_ = 1 + 0.2 * torch.randn(C1 + C2, generator=g), 0.3 * torch.randn(C1 + C2, generator=g)
        y, _ = emu.groupnorm(x1, x2, nb, S, 32, 1e-5, ga, be_, silu, False, torch.float32)
        assert rel_l2(y, tr.ref_groupnorm(x1, x2, nb, S, 32, 1e-5, ga, be_, silu)) < 2e-6
    x = torch.randn(70, 320, generator=g) * 2 + 0.5
    ga, be_ = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    assert rel_l2(emu.layernorm(x, ga, be_, 1e-5, torch.float32), tr.ref_layernorm(x, ga, be_, 1e-5)) < 2e-6
    for name, spec in kc.attn_cases(dt).items():
        spec = kc._clone_spec(spec, "cpu")
        spec.out = spec.out.float()
        emu.attention(spec)
        ref, got = tr.ref_attention(spec)
        assert rel_l2(got, ref) < 2e-6, (name, rel_l2(got, ref))


def test_clip_text_oracle_vs_transformers_clip():
    """f4's third-party dependency (open_clip) is not installed here, so the text tower's restatement cannot be pinned on
    the package itself.  `transformers` IS installed and ships an independent implementation of the same architecture
    (CLIPTextModel: pre-LN blocks, causal mask, learned positions, exact GELU when configured so): the oracle run on
    the SAME weights, re-keyed to open_clip's names, must reproduce its penultimate-layer hidden state after the final
    LayerNorm (the reference's `layer='penultimate'`, tools/modules/clip_embedder.py:55-64) and the projected
    end-of-text feature (:163-165) to fp32 rounding."""
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from oracle import torch_ref
    torch.manual_seed(0)
    L, heads, layers, d, vocab = 16, 4, 3, 64, 128
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=d, intermediate_size=4 * d, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=L, hidden_act="gelu", layer_norm_eps=1e-5,
                         projection_dim=32, eos_token_id=vocab - 1, pad_token_id=0, bos_token_id=1)
    hf = CLIPTextModelWithProjection(cfg).eval().float()
    with torch.no_grad():                                   # the default init is nearly an identity network: liven it up
        for p_ in hf.parameters():
            p_.copy_(torch.randn_like(p_) * (0.3 if p_.dim() > 1 else 0.1))
        for n_, p_ in hf.named_parameters():
            if "layer_norm" in n_ and n_.endswith("weight"):
                p_.add_(1.0)
    tm = hf.text_model
    sd = {"model.token_embedding.weight": tm.embeddings.token_embedding.weight,
          "model.positional_embedding": tm.embeddings.position_embedding.weight,
          "model.ln_final.weight": tm.final_layer_norm.weight, "model.ln_final.bias": tm.final_layer_norm.bias,
          "model.text_projection": hf.text_projection.weight.t()}
    for i, lyr in enumerate(tm.encoder.layers):
        p = f"model.transformer.resblocks.{i}."
        a = lyr.self_attn
        sd[p + "attn.in_proj_weight"] = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight])
        sd[p + "attn.in_proj_bias"] = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias])
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = a.out_proj.weight, a.out_proj.bias
        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = lyr.layer_norm1.weight, lyr.layer_norm1.bias
        sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = lyr.layer_norm2.weight, lyr.layer_norm2.bias
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = lyr.mlp.fc1.weight, lyr.mlp.fc1.bias
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = lyr.mlp.fc2.weight, lyr.mlp.fc2.bias
    sd = {k: v.detach() for k, v in sd.items()}
    tokens = torch.randint(2, vocab - 1, (3, L))
    tokens[:, 0] = 1
    for b, e in enumerate((5, 11, L - 1)):                  # end-of-text = the highest id, as open_clip's argmax assumes
        tokens[b, e] = vocab - 1
        tokens[b, e + 1:] = 0
    with torch.no_grad():
        out = hf(input_ids=tokens, output_hidden_states=True)
        want_pen = tm.final_layer_norm(out.hidden_states[-2])           # penultimate block's output through ln_final
        want_last = out.last_hidden_state                               # all blocks + ln_final
        x_pen, _ = torch_ref.clip_text_forward(sd, tokens, heads, layer_idx=1)
        x_last, xt = torch_ref.clip_text_forward(sd, tokens, heads, layer_idx=0)
    assert rel_l2(x_pen, want_pen) < 2e-6
    assert rel_l2(x_last, want_last) < 2e-6
    assert rel_l2(xt, out.text_embeds) < 2e-6
