"""FrozenOpenCLIPEmbedder — the OpenCLIP ViT-H/14 TEXT tower on the C ABI.

Reference: tools/modules/clip_embedder.py:12-77 (`FrozenOpenCLIPEmbedder`: token_embedding + positional_embedding,
the transformer's ResidualAttentionBlocks up to the penultimate one, ln_final) and :145-227 (`...TextVisualEmbedder`,
the same text path plus `xt = x[eot] @ text_projection`).  The transformer itself is the un-vendored third-party
`open_clip` (no version pinned in the reference's requirements): restated here from its published architecture —

    block(x) = x + out_proj(MHA(ln_1(x), causal mask)) ;  x + c_proj(gelu(c_fc(ln_2(x))))
    ViT-H-14 text: width 1024, 16 heads of 64, 24 layers, context 77, vocab 49408, exact (erf) GELU, eps 1e-5

— so parity is UNPINNED against open_clip itself; the oracle (oracle/torch_ref.py::clip_text_forward) builds the same
block from `torch.nn.MultiheadAttention`, the module open_clip uses, and is pinned on `transformers`' independent
CLIP text model run on the same weights (tests/test_oracle.py::test_clip_text_oracle_vs_transformers_clip).  Parameter names follow open_clip's `CLIP` text
branch under the reference's `model.` prefix, so the text keys of a stock checkpoint load (`strict=False` skips the
`model.visual.*` keys; the image tower is out of scope for this path).

Tokenisation (open_clip.tokenize: BPE over its bundled vocabulary) is not part of the hot path: `forward(text)` uses
it when open_clip is importable; the engines can equally pass token ids to `encode_with_transformer`.

Execution: activations are rows [B*77, 1024]; fp32 residual stream, LayerNorm / tap-GEMM / causal flash attention
kernels of libvgen_hip.so, 16-bit GEMM operands, fp32 accumulation — the same kernels as the UNet's transformer
blocks (head_dim 64), plus a GELU cast pass and the token-embedding gather.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import Attn, TapGemm

HEAD_DIM = 64


class _MLPP(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.c_fc = nn.Linear(d, hidden)
        self.gelu = nn.Identity()               # nn.GELU in open_clip; no parameters
        self.c_proj = nn.Linear(hidden, d)


class _BlockP(nn.Module):
    """open_clip ResidualAttentionBlock parameter container (ln_1, attn.in_proj_*, attn.out_proj, ln_2, mlp)."""

    def __init__(self, d, heads, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _MLPP(d, int(d * mlp_ratio))


class _TransformerP(nn.Module):
    def __init__(self, d, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_BlockP(d, heads) for _ in range(layers)])


class _TextModelP(nn.Module):
    def __init__(self, vocab, ctx, width, layers, heads, embed_dim):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.empty(ctx, width))
        self.transformer = _TransformerP(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=width ** -0.5)


ARCHS = {"ViT-H-14": dict(vocab=49408, ctx=77, width=1024, layers=24, heads=16, embed_dim=1024)}


class FrozenOpenCLIPEmbedder(nn.Module):
    LAYERS = ["last", "penultimate"]

    def __init__(self, pretrained=None, arch="ViT-H-14", device="cuda", max_length=77, freeze=True, layer="last",
                 compute_dtype=None, text_cfg=None, **kwargs):
        super().__init__()
        assert layer in self.LAYERS
        cfg = dict(text_cfg) if text_cfg is not None else ARCHS.get(arch)
        if cfg is None:
            raise NotImplementedError(f"FrozenOpenCLIPEmbedder: unknown arch {arch!r} (text tower of ViT-H-14 is built)")
        if cfg["width"] != cfg["heads"] * HEAD_DIM:
            raise NotImplementedError("the attention kernels are built for head_dim 64")
        self.cfg = cfg
        self.model = _TextModelP(**cfg)
        self.device = device
        self.max_length = max_length
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        self.compute_dtype = ops.sixteen(compute_dtype)
        self._packed = None
        if pretrained is not None:
            sd = torch.load(pretrained, map_location="cpu")
            sd = sd.get("state_dict", sd)
            own = self.state_dict()
            self.load_state_dict({k: v for k, v in (("model." + k if not k.startswith("model.") else k, v)
                                                    for k, v in sd.items()) if k in own}, strict=True)
        if freeze:
            self.freeze()

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def pack(self):
        dt = self.compute_dtype
        f32 = lambda t: t.detach().float().contiguous()
        w16 = lambda t: t.detach().to(dt).contiguous()
        m = self.model
        P = {"tok": f32(m.token_embedding.weight), "pos": f32(m.positional_embedding),
             "lnf": (f32(m.ln_final.weight), f32(m.ln_final.bias)),
             "proj": f32(m.text_projection.t()), "blocks": []}
        for b in m.transformer.resblocks:
            P["blocks"].append(dict(
                ln1=(f32(b.ln_1.weight), f32(b.ln_1.bias)), ln2=(f32(b.ln_2.weight), f32(b.ln_2.bias)),
                qkv=(w16(b.attn.in_proj_weight), f32(b.attn.in_proj_bias)),          # rows q | k | v
                o=(w16(b.attn.out_proj.weight), f32(b.attn.out_proj.bias)),
                fc=(w16(b.mlp.c_fc.weight), f32(b.mlp.c_fc.bias)),
                pr=(w16(b.mlp.c_proj.weight), f32(b.mlp.c_proj.bias))))
        self._packed = P
        return P

    # -- the tower -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _tower(self, tokens):
        """tokens int64 [B, L] -> rows [B*L, width] fp32 after ln_final (clip_embedder.py:154-161)."""
        be, dt = ops.backend(), self.compute_dtype
        if self._packed is None:
            self.pack()
        P = self._packed
        B, Lk = tokens.shape
        d, heads = self.cfg["width"], self.cfg["heads"]
        assert Lk == P["pos"].shape[0], "context length is fixed by positional_embedding"
        M = B * Lk
        x = be.embed_tokens(tokens.to(P["tok"].device).long().contiguous(), P["tok"], P["pos"])
        nblocks = len(P["blocks"]) - self.layer_idx                      # text_transformer_forward: :55-64
        for blk in P["blocks"][:nblocks]:
            n = be.layernorm(x, *blk["ln1"], 1e-5, dt)
            qkv = be.tapgemm(TapGemm(A=n, W=blk["qkv"][0], M=M, N=3 * d, C1=d, bias=blk["qkv"][1], out_dtype=dt))
            o = torch.empty((M, d), dtype=dt, device=x.device)
            ld = 3 * d
            be.attention(Attn(q=qkv, k=qkv[:, d:], v=qkv[:, 2 * d:], out=o, heads=heads, nq=Lk, nk=Lk, nbatch=B, inner=1,
                              q_s=(ld, Lk * ld, 0), k_s=(ld, Lk * ld, 0), v_s=(ld, Lk * ld, 0), o_s=(d, Lk * d, 0),
                              scale=HEAD_DIM ** -0.5, causal=True))
            x = be.tapgemm(TapGemm(A=o, W=blk["o"][0], M=M, N=d, C1=d, bias=blk["o"][1], residual=x))
            n = be.layernorm(x, *blk["ln2"], 1e-5, dt)
            h = be.tapgemm(TapGemm(A=n, W=blk["fc"][0], M=M, N=blk["fc"][0].shape[0], C1=d, bias=blk["fc"][1]))
            h = be.act_cast(h, 2, dt)                                         # nn.GELU, exact (erf) form
            x = be.tapgemm(TapGemm(A=h, W=blk["pr"][0], M=M, N=d, C1=h.shape[1], bias=blk["pr"][1], residual=x))
        return be.layernorm(x, *P["lnf"], 1e-5, torch.float32)

    @torch.no_grad()
    def encode_with_transformer(self, text):
        B, Lk = text.shape
        return self._tower(text).view(B, Lk, -1)

    @torch.no_grad()
    def encode_text_and_tokens(self, text):
        """(xt, x) of FrozenOpenCLIPTextVisualEmbedder.encode_with_transformer (:190-198): the EOT-token feature
        through text_projection, and the per-token features."""
        B, Lk = text.shape
        x = self._tower(text).view(B, Lk, -1)
        eot = x[torch.arange(B, device=x.device), text.to(x.device).argmax(dim=-1)].contiguous()
        xt = ops.backend().linear_f32(eot, self._packed["proj"], None)
        return xt, x

    def forward(self, text):
        if torch.is_tensor(text):
            return self.encode_with_transformer(text)
        try:
            import open_clip
        except ImportError as e:                               # the BPE tokenizer lives in the un-vendored package
            raise RuntimeError("FrozenOpenCLIPEmbedder.forward(str) needs open_clip.tokenize; pass token ids "
                               "[B, 77] (int64) to run the tower without it") from e
        return self.encode_with_transformer(open_clip.tokenize(text))

    def encode(self, text):
        return self(text)
