"""Per-block parity against the reference (tests/golden/unet_blocks_tiny.pt, oracle/make_golden.py make_blocks): every
ResBlock (+ temporal convs), SpatialTransformer, TemporalTransformer, Downsample and Upsample of the tiny
UNetSD_T2VBase is run ALONE on the reference's own input for that block and compared with the reference's output."""
import torch

from conftest import gold, rel_l2
from oracle import torch_ref


def run_blocks(dtname, dev):
    """-> {block name: (kind, rel-L2 of the block's output vs the reference's)}"""
    from vgen_amd import ops
    from vgen_amd.unet import UNetSD_T2VBase, _ResBlockP, _SpatialTransformerP, _TemporalTransformerP, _DownP, _UpP
    g0, g = gold("unet_tiny.pt"), gold("unet_blocks_tiny.pt")
    m = UNetSD_T2VBase(**g0["cfg"], compute_dtype=dtname, precision="fast").eval()
    m.load_state_dict(torch_ref.synth_state_dict(g0["shapes"], seed=g0["seed"]), strict=True)
    m = m.to(dev)
    m.pack()
    be, dt, P = ops.backend(), m.compute_dtype, m._packed
    B, F = g["B"], g["F"]
    emb_all = m._embed(g["t"].to(dev), None, B, dev)
    kv_all = m._context_kv(g["y"].to(dev), dev)
    Lctx = g["y"].shape[1]
    rows = lambda v: v.permute(0, 2, 3, 1).reshape(-1, v.shape[1]).contiguous().to(dev)   # [(b f) h w, C] fp32
    res = {}
    for i, r in enumerate(g["recs"]):
        mod = m.get_submodule(r["name"])
        H, W = r["H"], r["W"]
        src = r["src"]
        x1 = rows(r["x"] if src is None else g["outs"][src[0]])
        x2 = rows(g["outs"][src[1]]) if src is not None and len(src) == 2 else None
        if isinstance(mod, _ResBlockP):
            o = m._resblock(mod, x1, x2, emb_all, B, F, H, W)
        elif isinstance(mod, _SpatialTransformerP):
            o = m._spatial_tx(mod, x1, kv_all, B, F, H, W, Lctx)
        elif isinstance(mod, _TemporalTransformerP):
            o = m._temporal_tx(mod, x1, B, F, H, W)
        elif isinstance(mod, _DownP):
            o, H, W = m._conv3x3(be.act_cast(x1, 0, dt), P[mod._pname], B * F, H, W, x1.shape[1], stride=2, pad=mod.pad)
        elif isinstance(mod, _UpP):
            o, H, W = m._conv3x3(be.act_cast(x1, 0, dt), P[mod._pname], B * F, H, W, x1.shape[1], ups=1, crop=mod.crop)
        else:
            raise TypeError(type(mod))
        want = g["outs"][i]
        assert (want.shape[2], want.shape[3]) == (H, W) and o.shape == (want.shape[0] * H * W, want.shape[1]), r["name"]
        res[r["name"]] = (r["kind"], rel_l2(o.float().cpu(), rows(want).cpu()))
    return res


def run_vae_blocks(dtname, dev):
    """Every ResnetBlock / AttnBlock / Upsample / Downsample of the reference's tiny AutoencoderKL (encoder and decoder)
    alone on the reference's input (tests/golden/vae_blocks_tiny.pt) -> {name: (kind, rel-L2)}"""
    from vgen_amd import ops
    from vgen_amd.vae import AutoencoderKL, _ResnetBlockP, _AttnBlockP, _ResampleP
    g0, g = gold("vae_tiny.pt"), gold("vae_blocks_tiny.pt")
    v = AutoencoderKL(ddconfig=g0["ddconfig"], embed_dim=4, compute_dtype=dtname).eval()
    v.load_state_dict(torch_ref.synth_state_dict(g0["shapes"], seed=g0["seed"]), strict=True)
    v = v.to(dev)
    v.pack()
    be, dt, P = ops.backend(), v.compute_dtype, v._packed
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().to(dev)
    res = {}
    for i, r in enumerate(g["recs"]):
        xin = r["x"] if r["src"] is None else g["outs"][r["src"]]
        n, _, H, W = xin.shape
        mod = v.get_submodule(r["name"])
        x = rows(xin)
        if isinstance(mod, _ResnetBlockP):
            o = v._resnet(mod, x, n, H, W)
        elif isinstance(mod, _AttnBlockP):
            o = v._attn(mod, x, n, H, W)
        elif isinstance(mod, _ResampleP):
            a = be.act_cast(x, 0, dt)
            if r["kind"] == "Upsample":
                o, H, W = v._conv(a, P[mod._pname], n, H, W, x.shape[1], ups=1)
            else:
                o, H, W = v._conv(a, P[mod._pname], n, H, W, x.shape[1], stride=2, pad=0)
        else:
            raise TypeError(type(mod))
        want = g["outs"][i]
        assert tuple(want.shape[2:]) == (H, W), (r["name"], want.shape, H, W)
        res[r["name"]] = (r["kind"], rel_l2(o.float().cpu(), rows(want).cpu()))
    return res
