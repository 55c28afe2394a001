"""OpenCLIP text tower (vgen_amd/clip_text.py): host logic on the ABI emulator (CPU) and the HIP path
(-m gpu) against the oracle restatement (oracle/torch_ref.py::clip_text_forward — parity unpinned against the
un-vendored open_clip package, see its header)."""
import pytest
import torch

from conftest import rel_l2
from oracle import torch_ref

TINY = dict(vocab=300, ctx=20, width=128, layers=3, heads=2, embed_dim=64)


def _tower(cfg, dtname, layer, seed=5):
    from vgen_amd.clip_text import FrozenOpenCLIPEmbedder
    m = FrozenOpenCLIPEmbedder(text_cfg=cfg, layer=layer, compute_dtype=dtname)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = torch_ref.synth_state_dict(shapes, seed=seed)
    sd["model.logit_scale"] = torch.tensor(2.6592)
    m.load_state_dict(sd, strict=True)
    return m, sd


def _tokens(cfg, B, seed=1):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(1, cfg["vocab"] - 2, (B, cfg["ctx"]), generator=g)
    for b in range(B):                                   # an EOT (largest id) somewhere, zeros after it
        e = 3 + 5 * b
        t[b, e] = cfg["vocab"] - 1
        t[b, e + 1:] = 0
    return t


def test_state_dict_follows_open_clip_text_branch():
    from vgen_amd.clip_text import FrozenOpenCLIPEmbedder
    with torch.device("meta"):
        m = FrozenOpenCLIPEmbedder()                       # ViT-H-14 text tower
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert keys["model.token_embedding.weight"] == (49408, 1024) and keys["model.positional_embedding"] == (77, 1024)
    assert keys["model.transformer.resblocks.23.attn.in_proj_weight"] == (3072, 1024)
    assert keys["model.transformer.resblocks.0.mlp.c_fc.weight"] == (4096, 1024)
    assert keys["model.text_projection"] == (1024, 1024) and keys["model.ln_final.bias"] == (1024,)
    assert sum(torch.Size(s).numel() for s in keys.values()) == 354032641      # open_clip ViT-H-14 text branch


@pytest.mark.parametrize("layer", ["penultimate", "last"])
def test_text_tower_host_logic_vs_oracle(emu_backend, layer):
    m, sd = _tower(TINY, "fp16", layer)
    tok = _tokens(TINY, 3)
    x = m.encode_with_transformer(tok)
    ref, ref_t = torch_ref.clip_text_forward(sd, tok, TINY["heads"], layer_idx=m.layer_idx)
    assert x.shape == ref.shape and x.dtype == torch.float32
    assert rel_l2(x, ref) < 3e-3
    xt, x2 = m.encode_text_and_tokens(tok)
    assert torch.equal(x2, x) and rel_l2(xt, ref_t) < 3e-3
    # causal: changing a later token must not change earlier positions
    tok2 = tok.clone()
    tok2[:, 12] = 7
    assert torch.equal(m.encode_with_transformer(tok2)[:, :12], x[:, :12])


@pytest.mark.gpu
@pytest.mark.parametrize("dtname,tol", [("fp16", 2.5e-3), ("bf16", 2e-2)])
def test_text_tower_tiny_on_device(hip_backend, dtname, tol):
    m, sd = _tower(TINY, dtname, "penultimate")
    m = m.to("cuda:0")
    tok = _tokens(TINY, 3)
    x = m.encode_with_transformer(tok.to("cuda:0"))
    ref, ref_t = torch_ref.clip_text_forward(sd, tok, TINY["heads"], layer_idx=1)
    assert rel_l2(x, ref) < tol
    xt, _ = m.encode_text_and_tokens(tok.to("cuda:0"))
    assert rel_l2(xt, ref_t) < tol
    assert torch.equal(x, m.encode_with_transformer(tok.to("cuda:0")))


@pytest.mark.gpu
def test_text_tower_vit_h_on_device(hip_backend):
    """Full ViT-H/14 text tower (354 M parameters, 77 tokens, penultimate layer as the engines use it) vs the oracle
    run on the host CPU; the UNet consumes the result as its context y."""
    from vgen_amd.clip_text import ARCHS
    import json, os
    from conftest import ROOT
    cfg = ARCHS["ViT-H-14"]
    m, sd = _tower(cfg, "fp16", "penultimate", seed=9)
    m = m.to("cuda:0")
    tok = _tokens(cfg, 2)
    x = m.encode_with_transformer(tok.to("cuda:0"))
    ref, _ = torch_ref.clip_text_forward(sd, tok, cfg["heads"], layer_idx=1)
    err = rel_l2(x, ref)
    p = os.path.join(ROOT, "gpurun_out", "parity.json")
    d = json.load(open(p)) if os.path.exists(p) else {}
    d["clip_text_vit_h/fp16"] = err
    json.dump(d, open(p, "w"), indent=1)
    assert x.shape == (2, 77, 1024) and err < 2.5e-3, err
