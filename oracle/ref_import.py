"""TEST INFRASTRUCTURE — loads the reference's own modules (read-only tree at /root/reference)
on CPU so the restatement in oracle/torch_ref.py and the golden fixtures can be pinned against
the real thing.  Only usable in the build container (the GPU box has no /root/reference);
nothing under vgen_amd/ imports this file.

Recipe = SURVEY.md Appendix C: four stub modules for un-vendored third-party deps
(xformers.ops.memory_efficient_attention restated as softmax(q k^T / sqrt(d)) v via
F.scaled_dot_product_attention — xformers==0.0.13 is pinned in the reference's requirements.txt
but is absent here), bare package objects so `tools/__init__.py` (cv2, open_clip, ...) is skipped,
and the leaf files loaded by path.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = os.environ.get("VGEN_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "tools", "modules"))


_loaded = {}


def load():
    """Returns dict(MODEL=..., AUTO_ENCODER=..., DIFFUSION=..., util=<module>, ...)."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    import torch.nn.functional as F

    if REF not in sys.path:
        sys.path.insert(0, REF)
    xf, xo = types.ModuleType("xformers"), types.ModuleType("xformers.ops")
    def _mea(q, k, v, attn_bias=None, op=None):
        # [B*heads, M, 64] sequences are independent: above ~2 GB of fp32 scores evaluate them in batch chunks (the
        # 14 080-token self-attention of the 704p / 720p latents would need 63 GB at once on the CPU) — same numbers
        import torch as _t
        per = q.shape[1] * k.shape[1] * 4
        n = max(1, int(2e9 // max(per, 1)))
        if q.dim() != 3 or q.shape[0] <= n:
            return F.scaled_dot_product_attention(q, k, v)
        return _t.cat([F.scaled_dot_product_attention(q[i:i + n], k[i:i + n], v[i:i + n]) for i in range(0, q.shape[0], n)], 0)

    xo.memory_efficient_attention = _mea
    xo.LowerTriangularMask = type("LowerTriangularMask", (), {})
    xf.ops = xo
    sys.modules.update({"xformers": xf, "xformers.ops": xo, "open_clip": types.ModuleType("open_clip")})
    r = types.ModuleType("rotary_embedding_torch")
    r.RotaryEmbedding = object
    sys.modules["rotary_embedding_torch"] = r
    fc = types.ModuleType("fairscale.nn.checkpoint")
    fc.checkpoint_wrapper = lambda m, *a, **k: m
    sys.modules.update({"fairscale": types.ModuleType("fairscale"),
                        "fairscale.nn": types.ModuleType("fairscale.nn"),
                        "fairscale.nn.checkpoint": fc})
    for pkg in ("tools", "tools.modules", "tools.modules.unet", "tools.modules.diffusions"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg.replace(".", "/"))]
        sys.modules[pkg] = m

    def _load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name.replace(".", "/") + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    # torchsde (==0.2.6 in the reference's requirements.txt) is not installed: a deterministic stand-in
    # for BrownianTree so that sample_dpmpp_2m_sde can run; increments are N(0, |tb - ta|) drawn from a
    # generator keyed on (entropy, ta, tb).  Parity of the true torchsde noise is unpinned.
    import torch as _torch
    ts = types.ModuleType("torchsde")

    class BrownianTree:
        def __init__(self, t0, w0, t1, entropy=None, **kw):
            self.shape, self.entropy = w0.shape, int(entropy or 0)

        def __call__(self, ta, tb):
            key = (self.entropy * 1000003 + int(float(ta) * 1e6) * 7919 + int(float(tb) * 1e6)) % (2 ** 62)
            g = _torch.Generator("cpu").manual_seed(key)
            return _torch.randn(self.shape, generator=g) * (float(tb) - float(ta)) ** 0.5

    ts.BrownianTree = BrownianTree
    sys.modules["torchsde"] = ts
    mods = {}
    for n in ("tools.modules.unet.util", "tools.modules.unet.unet_t2v", "tools.modules.unet.unet_sr600",
              "tools.modules.unet.unet_i2vgen", "tools.modules.unet.unet_videolcm",
              "tools.modules.unet.unet_tf2tv",
              "tools.modules.autoencoder",
              "tools.modules.diffusions.schedules", "tools.modules.diffusions.losses",
              "tools.modules.diffusions.diffusion_ddim", "tools.modules.diffusions.diffusion_gauss"):
        mods[n.rsplit(".", 1)[-1]] = _load(n)
    from utils.registry_class import AUTO_ENCODER, DIFFUSION, MODEL
    _loaded.update(mods)
    _loaded.update(MODEL=MODEL, AUTO_ENCODER=AUTO_ENCODER, DIFFUSION=DIFFUSION)
    return _loaded
