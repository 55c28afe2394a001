"""AutoencoderKL — MI355X-native drop-in for the reference's per-frame SD KL-VAE.

Interface parity (reference: tools/modules/autoencoder.py:30-103 AutoencoderKL, :484-578
Encoder, :581-686 Decoder, :212-225 DiagonalGaussianDistribution): same registry name and
constructor (`ddconfig, embed_dim, pretrained, ...`), same state_dict keys (stock SD checkpoints
load through `init_from_ckpt` with strict=True), `decode(z[n,4,h,w]) -> [n,3,8h,8w]`,
`encode_firsr_stage(x, scale_factor) -> [n,4,h/8,w/8]`, `encode(x) -> posterior`.

Execution: channels-last rows [n*H*W, C]; every conv is the tap-GEMM HIP kernel (nearest-2x
upsample and the encoder's asymmetric (0,1,0,1) padding are folded into the conv's gather),
GroupNorm+swish is the fused GN kernel, the single-head 512-channel mid attention is
scores-GEMM -> row softmax kernel -> PV-GEMM.
"""
from __future__ import annotations


import collections
import logging

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .ops import TapGemm
from .unet import _f32, _w16_cat, pack_conv3x3, pack_linear, pack_small_conv3x3, split_weights


def _norm(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class _ResnetBlockP(nn.Module):
    # reference: ResnetBlock (temb_channels = 0), autoencoder.py:272-335
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = _norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = _norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)


class _AttnBlockP(nn.Module):
    # reference: AttnBlock, autoencoder.py:391-442
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.norm = _norm(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)


class _ResampleP(nn.Module):
    def __init__(self, c, stride_pad):
        super().__init__()
        s, p = stride_pad
        self.conv = nn.Conv2d(c, c, 3, stride=s, padding=p)


class _MidP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block_1 = _ResnetBlockP(c, c)
        self.attn_1 = _AttnBlockP(c)
        self.block_2 = _ResnetBlockP(c, c)


class _LevelP(nn.Module):
    pass


class _EncoderP(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels,
                 double_z=True, **ignore):
        super().__init__()
        if list(attn_resolutions):
            raise NotImplementedError("attn_resolutions must be empty (SD VAE)")
        assert resamp_with_conv
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i in range(self.num_resolutions):
            lvl = _LevelP()
            bi, bo = ch * in_ch_mult[i], ch * ch_mult[i]
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                lvl.block.append(_ResnetBlockP(bi, bo))
                bi = bo
            if i != self.num_resolutions - 1:
                lvl.downsample = _ResampleP(bi, (2, 0))
            self.down.append(lvl)
        self.mid = _MidP(bi)
        self.norm_out = _norm(bi)
        self.conv_out = nn.Conv2d(bi, 2 * z_channels if double_z else z_channels, 3, padding=1)


class _DecoderP(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels,
                 give_pre_end=False, tanh_out=False, **ignore):
        super().__init__()
        if list(attn_resolutions):
            raise NotImplementedError("attn_resolutions must be empty (SD VAE)")
        if give_pre_end or tanh_out:
            raise NotImplementedError
        assert resamp_with_conv
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        bi = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, bi, 3, padding=1)
        self.mid = _MidP(bi)
        self.up = nn.ModuleList()
        for i in reversed(range(self.num_resolutions)):
            lvl = _LevelP()
            bo = ch * ch_mult[i]
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                lvl.block.append(_ResnetBlockP(bi, bo))
                bi = bo
            if i != 0:
                lvl.upsample = _ResampleP(bi, (1, 1))
            self.up.insert(0, lvl)
        self.norm_out = _norm(bi)
        self.conv_out = nn.Conv2d(bi, out_ch, 3, padding=1)


class DiagonalGaussianDistribution(object):
    """reference: autoencoder.py:212-246 (sampling members)."""

    def __init__(self, moments_rows, nimg, zc, H, W, deterministic=False):
        self._m, self.nimg, self.zc, self.H, self.W = moments_rows, nimg, zc, H, W
        self.deterministic = deterministic

    @property
    def parameters(self):
        return self._m.view(self.nimg, self.H, self.W, 2 * self.zc).permute(0, 3, 1, 2)

    def sample(self, scale=1.0):
        dev = self._m.device
        shape = (self.nimg, self.zc, self.H, self.W)
        noise = torch.randn(shape).to(device=dev) if not self.deterministic else torch.zeros(shape, device=dev)
        return ops.backend().gaussian_sample(self._m, noise.float().contiguous(), self.nimg, self.zc,
                                             self.H * self.W, scale)

    def mode(self):
        return self.parameters[:, : self.zc].contiguous()


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim, pretrained=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, ema_decay=None, learn_logvar=False,
                 use_vid_decoder=False, compute_dtype=None, precision=None, **kwargs):
        super().__init__()
        self.learn_logvar = learn_logvar
        self.image_key = image_key
        self.encoder = _EncoderP(**ddconfig)
        self.decoder = _DecoderP(**ddconfig)
        assert ddconfig["double_z"]
        if embed_dim != ddconfig["z_channels"]:
            # quant_conv / post_quant_conv and DiagonalGaussianDistribution are laid out for 2*z_channels moments
            # (every config of the reference uses embed_dim == z_channels == 4, tools/modules/config.py:118-135)
            raise NotImplementedError(f"AutoencoderKL: embed_dim ({embed_dim}) != z_channels ({ddconfig['z_channels']})")
        self.zc = ddconfig["z_channels"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.compute_dtype = ops.sixteen(compute_dtype)
        # "high": weights as hi + lo operand pairs (vgen_amd/unet.py); "calibrated" (r06): ONE 16-bit matrix per
        # layer whose rounding a calibration pass chose (vgen_amd/calibrate.py::calibrate_vae on a "high" model, or
        # `precision="calibrated", calibration=<file save_calibrated wrote>`): "high"'s accuracy at "fast"'s launches
        self.precision = precision or "fast"
        self.calibration = kwargs.pop("calibration", None)
        if (self.precision == "calibrated") != bool(self.calibration):
            raise ValueError("AutoencoderKL: precision='calibrated' and calibration=<file> go together (to calibrate a model, "
                             "build it with precision='high' and call vgen_amd.calibrate.calibrate_vae)")
        assert self.precision in ("fast", "high", "calibrated")
        self._epoch = 0
        self._packed = None
        self._attn_qb = None          # query-block override of the mid attention (tests force several blocks)
        self._graphs = collections.OrderedDict()   # (kind, input shape, device) -> captured launch sequence of a chunk (LRU)
        self._graph_pool = None       # ONE memory pool for all of this model's chunk graphs (as UnitSession does)
        if pretrained is not None:
            self.init_from_ckpt(pretrained, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        sd_new = collections.OrderedDict()
        for k in list(sd.keys()):
            if k.find("first_stage_model") >= 0:
                sd_new[k.split("first_stage_model.")[-1]] = sd[k]
        self.load_state_dict(sd_new, strict=True)
        logging.info(f"Restored from {path}")

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        self._packed = None
        self._epoch = getattr(self, "_epoch", 0) + 1
        if getattr(self, "precision", None) == "calibrated" and not getattr(self, "calibration", None):
            self.precision = "high"          # an in-place calibration went with the packed operands (as in the UNets)
        self.clear_graphs()

    def clear_graphs(self):
        """Drop every captured chunk graph together with the memory pool they share (a chunk's peak activations stay
        reserved while its graph lives: several GB for 720p chunks).  The engines call this between the stages of a
        pipeline; weight loads / device moves do it themselves."""
        self._graphs = collections.OrderedDict()
        self._graph_pool = None

    # captured graphs are per-process device objects: a copy / pickle of the model starts without them (ADVICE r04:
    # CUDAGraph is neither picklable nor deep-copyable)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_graphs"] = collections.OrderedDict()
        st["_graph_pool"] = None
        return st

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # -- r04: a decode / encode chunk is ONE hipGraph replay ------------------------------------------------------
    _GRAPH_SHAPES = 4             # captured input shapes kept per model (each holds a chunk's peak activations)

    def _graphed(self, kind, rows_fn, x):
        """rows_fn(x) -> (rows, n, H, W) through a captured graph keyed on (kind, x.shape): the engines decode a video as
        F / decoder_bs identical chunks (inference_text2video_entrance.py:208-217) — ~110 launches each, many of them a few
        microseconds long at the low-resolution levels.  First call of a shape runs eagerly (warms the allocator), the
        second captures, later ones copy the input into the graph's static buffer and replay.  The returned rows live in
        the graph's memory pool: valid until the next call of the same shape (every caller consumes them at once).
        VGEN_GRAPH=0, CPU tensors (the tests' ABI emulator) and capture failures take the eager path."""
        from .session import _GRAPH_ON
        if not _GRAPH_ON or x.device.type != "cuda":
            return rows_fn(x)
        key = (kind, tuple(x.shape), str(x.device), self._attn_qb)
        st = self._graphs.get(key)
        if st is None:
            while len(self._graphs) >= self._GRAPH_SHAPES:
                self._graphs.popitem(last=False)        # least recently USED shape (hits move to the end below)
            self._graphs[key] = {"calls": 1}
            return rows_fn(x)
        self._graphs.move_to_end(key)
        if "graph" not in st:
            if st.get("failed"):
                return rows_fn(x)
            xin = x.float().contiguous().clone()
            torch.cuda.synchronize(x.device)
            g = torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            try:
                with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                    out = rows_fn(xin)
            except Exception as ex:             # noqa: BLE001 — capture is an optimisation: stay correct, say so
                import warnings
                warnings.warn(f"AutoencoderKL: hipGraph capture of {kind} failed ({type(ex).__name__}: {ex}); running eagerly")
                torch.cuda.synchronize(x.device)
                st["failed"] = True
                return rows_fn(x)
            st.update(graph=g, xin=xin, out=out)
        else:
            st["xin"].copy_(x)
        st["graph"].replay()
        return st["out"]

    # -- packing ---------------------------------------------------------------------------------
    @torch.no_grad()
    def pack(self):
        with split_weights(self.precision == "high"):
            P = self._pack()
        if self.precision == "calibrated":
            from .calibrate import load_calibrated
            load_calibrated(self, self.calibration)
        return P

    def _pack(self):
        dt = self.compute_dtype
        P = {}
        for name, m in self.named_modules():
            m._pname = name
            if isinstance(m, _ResnetBlockP):
                d = {"gn1": (_f32(m.norm1.weight), _f32(m.norm1.bias)),
                     "conv1": (pack_conv3x3(m.conv1.weight, dt), _f32(m.conv1.bias)),
                     "gn2": (_f32(m.norm2.weight), _f32(m.norm2.bias))}
                b2 = _f32(m.conv2.bias)
                if m.cin != m.cout:
                    w2 = _w16_cat([pack_conv3x3(m.conv2.weight), pack_linear(m.nin_shortcut.weight)], dt)
                    b2 = (b2 + _f32(m.nin_shortcut.bias)).contiguous()
                else:
                    w2 = pack_conv3x3(m.conv2.weight, dt)
                d["conv2"] = (w2, b2)
                P[name] = d
            elif isinstance(m, _AttnBlockP):
                P[name] = {
                    "gn": (_f32(m.norm.weight), _f32(m.norm.bias)),
                    "qk": (pack_linear(torch.cat([m.q.weight, m.k.weight], 0), dt),
                           torch.cat([_f32(m.q.bias), _f32(m.k.bias)]).contiguous()),
                    "v": pack_linear(m.v.weight, dt), "vb": _f32(m.v.bias),
                    "o": (pack_linear(m.proj_out.weight, dt), _f32(m.proj_out.bias))}
            elif isinstance(m, _ResampleP):
                P[name] = (pack_conv3x3(m.conv.weight, dt), _f32(m.conv.bias))
        for side in ("encoder", "decoder"):
            net = getattr(self, side)
            ci = net.conv_in
            kpad = ((27 * ci.in_channels + 63) // 64) * 64          # split stem: fp32-accurate input / weights
            P[side + ".conv_in"] = (pack_small_conv3x3(ci.weight, kpad, dt, split=True), _f32(ci.bias), kpad)
            P[side + ".norm_out"] = (_f32(net.norm_out.weight), _f32(net.norm_out.bias))
            P[side + ".conv_out"] = (pack_conv3x3(net.conv_out.weight, dt), _f32(net.conv_out.bias))
        for nm in ("quant_conv", "post_quant_conv"):
            c = getattr(self, nm)
            P[nm] = (_f32(c.weight.reshape(c.out_channels, c.in_channels)), _f32(c.bias))
        dev = self.quant_conv.weight.device
        P["eye"] = {n: torch.eye(n, dtype=torch.float32, device=dev).contiguous()
                    for n in {self.decoder.conv_out.out_channels}}
        self._packed = P
        return P

    # -- blocks ------------------------------------------------------------------------------------
    def _conv(self, A, Wb, nimg, Hi, Wi, C1, stride=1, ups=0, pad=1, **kw):
        W, b = Wb
        if stride == 1:
            Ho, Wo = Hi << ups, Wi << ups
        else:                                   # F.pad(0,1,0,1) + stride-2 valid conv (autoencoder.py:476-478)
            Ho, Wo = (Hi + 1 - 3) // 2 + 1, (Wi + 1 - 3) // 2 + 1
        # r04: like the UNet's convs, a conv whose fp32 output feeds a GroupNorm leaves per-64-row-slab column statistics
        # behind (vgen_tapgemm_args.colstats), so that norm skips its statistics pass: 10 -> 6 B / element on the 117 MB
        # tensors of the 256 x 448 level.  Only where the launch is never split along K and the frame is whole slabs.
        M = nimg * Ho * Wo
        kw["colstats"] = bool(kw.get("colstats", False)) and M >= ops.COLSTATS_MIN_ROWS and (Ho * Wo) % ops.CS_ROWS == 0 \
            and W.shape[0] % 4 == 0
        g = TapGemm(A=A, W=W, M=M, N=W.shape[0], C1=C1, mode=L.TAP_CONV3X3, taps=9,
                    Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, stride=stride, pad_t=pad, pad_l=pad, ups=ups, bias=b, **kw)
        return ops.backend().tapgemm(g), Ho, Wo

    def _resnet(self, m: _ResnetBlockP, x, n, H, W):
        be, dt = ops.backend(), self.compute_dtype
        P = self._packed[m._pname]
        skip = m.cin != m.cout
        a1, raw = be.groupnorm(x, None, n, H * W, 32, 1e-6, *P["gn1"], True, skip, dt)
        h, _, _ = self._conv(a1, P["conv1"], n, H, W, m.cin, colstats=True)
        a2, _ = be.groupnorm(h, None, n, H * W, 32, 1e-6, *P["gn2"], True, False, dt)
        if skip:
            h, _, _ = self._conv(a2, P["conv2"], n, H, W, m.cout, A2=raw, C2=m.cin, colstats=True)
        else:
            h, _, _ = self._conv(a2, P["conv2"], n, H, W, m.cout, residual=x, colstats=True)
        return h

    def _attn(self, m: _AttnBlockP, x, n, H, W):
        """reference: AttnBlock.forward (autoencoder.py:418-442): softmax(q k^T / sqrt(c)) v, one head."""
        be, dt = ops.backend(), self.compute_dtype
        P = self._packed[m._pname]
        c, hw = m.c, H * W
        M = n * hw
        a, _ = be.groupnorm(x, None, n, hw, 32, 1e-6, *P["gn"], False, False, dt)
        Wqk, bqk = P["qk"]
        qk = be.tapgemm(TapGemm(A=a, W=Wqk, M=M, N=2 * c, C1=c, bias=bqk, out_dtype=dt))
        hwp = ((hw + 63) // 64) * 64
        o = torch.empty((M, c), dtype=dt, device=x.device)
        scale = float(int(c) ** (-0.5))
        # scores are formed per block of QB queries: S [QB, hw] fp32 stays <= 64 MiB (a whole 90x160 latent frame
        # of the 720p configs would need 829 MB for hw x hw; 32x56 frames fit in one block)
        QB = min(hw, self._attn_qb or max(256, ((16 << 20) // hwp) // 256 * 256))
        S = torch.empty((QB, hw), dtype=torch.float32, device=x.device)
        Pm = torch.zeros((QB, hwp), dtype=dt, device=x.device)          # pad columns stay zero
        vt = torch.zeros((c, hwp), dtype=dt, device=x.device)
        for i in range(n):
            rows = slice(i * hw, (i + 1) * hw)
            # V^T[c, p] = sum_ci Wv[c, ci] a[p, ci]  (bias folded after PV: softmax rows sum to 1)
            be.tapgemm(TapGemm(A=P["v"], W=a[rows], M=c, N=hw, C1=c, out_dtype=dt, out=vt))
            for q0 in range(0, hw, QB):
                qb = min(QB, hw - q0)
                qr = slice(i * hw + q0, i * hw + q0 + qb)
                be.tapgemm(TapGemm(A=qk[qr, :c], W=qk[rows, c:], M=qb, N=hw, C1=c, out=S[:qb]))
                be.softmax_rows(S[:qb], hw, scale, dt, out=Pm[:qb])
                be.tapgemm(TapGemm(A=Pm[:qb], W=vt, M=qb, N=c, C1=hwp, bias=P["vb"], out_dtype=dt, out=o[qr]))
        Wo, bo = P["o"]
        return be.tapgemm(TapGemm(A=o, W=Wo, M=M, N=c, C1=c, bias=bo, residual=x))

    def _mid(self, mid: _MidP, h, n, H, W):
        h = self._resnet(mid.block_1, h, n, H, W)
        h = self._attn(mid.attn_1, h, n, H, W)
        return self._resnet(mid.block_2, h, n, H, W)

    def _stem(self, side, src, n, Cin, H, W, strides):
        be, dt = ops.backend(), self.compute_dtype
        Wc, bc, kpad = self._packed[side + ".conv_in"]
        col = be.im2col3x3_small(src, n, 1, Cin, H, W, strides, kpad, dt, split=True)
        return be.tapgemm(TapGemm(A=col, W=Wc, M=n * H * W, N=Wc.shape[0], C1=kpad, bias=bc))

    # -- public API ----------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z, **kwargs):
        o, n, H, W = self._graphed("decode", self._decode_rows, z)
        be = ops.backend()
        oc = o.shape[1]
        out = torch.empty((n, oc, H, W), dtype=torch.float32, device=z.device)
        be.pointwise_small(o, n, 1, oc, H, W, (H * W * oc, 0, 1, W * oc, oc), self._packed["eye"][oc], None, oc,
                           out, (oc * H * W, 0, H * W, W, 1))
        return out

    @torch.no_grad()
    def decode_to_uint8(self, z, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        """decode + the engines' post-processing (utils/video_op.py:181-188) in one pass: the decoder's rows
        are already (frame, y, x, channel)-ordered, so the byte image [n, H, W, 3] is written straight from
        them — no NCHW tensor, no fp32 D2H of 22 MB per video."""
        o, n, H, W = self._graphed("decode", self._decode_rows, z)
        key = (tuple(mean), tuple(std), o.device)
        if getattr(self, "_u8_consts", (None,))[0] != key:
            self._u8_consts = (key, torch.tensor(mean, dtype=torch.float32, device=o.device),
                               torch.tensor(std, dtype=torch.float32, device=o.device))
        return ops.backend().frames_u8(o, self._u8_consts[1], self._u8_consts[2]).view(n, H, W, o.shape[1])

    @torch.no_grad()
    def decode_video(self, latents, scale_factor=0.18215, decoder_bs=2, to_uint8=True, group=None, shard=None, **u8):
        """Engine glue a21 (inference_text2video_entrance.py:208-217): latents [B, 4, F, h, w] -> frames, i.e.
        `1/scale_factor * x`, '(b f) c h w' chunks of decoder_bs through decode, back to per-video order.
        Returns uint8 [B, F, H, W, 3] (to_uint8) or fp32 [B, 3, F, H, W] like the reference.

        Frame sharding ("the VAE shards by frame ... one final all-gather"): the AutoencoderKL is a per-frame
        2-D network, so with an initialised process group (`shard` defaults to that; `group` selects it) every rank
        decodes a contiguous block of ceil(B F / W) frames and ONE all_gather_into_tensor (RCCL over xGMI with the nccl
        backend; 16 x 448 x 256 x 3 uint8 = 5.5 MB) hands every rank the whole video.  The latents must be the same
        on all ranks — they are after the unit partition's redundant update (vgen_amd/parallel.py)."""
        import torch.distributed as dist
        B, C, F, h, w = latents.shape
        z = (latents.float() * (1.0 / scale_factor)).permute(0, 2, 1, 3, 4).reshape(B * F, C, h, w)
        n = B * F
        inited = dist.is_available() and dist.is_initialized()
        W_ = dist.get_world_size(group) if inited else 1
        if shard is None:
            shard = W_ > 1
        shard = bool(shard) and inited         # shard=True with a 1-rank group still runs the collective (RCCL smoke test)
        rank = dist.get_rank(group) if shard else 0
        per = (n + W_ - 1) // W_ if shard else n                 # frames per rank (the last rank's block may be short)
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        outs = []
        for i in range(lo, hi, decoder_bs):
            zc = z[i:min(i + decoder_bs, hi)]
            outs.append(self.decode_to_uint8(zc, **u8) if to_uint8 else self.decode(zc))
        if not shard:
            o = torch.cat(outs, 0)
        else:
            H, Wd = 8 * h, 8 * w
            fshape = (H, Wd, 3) if to_uint8 else (3, H, Wd)
            mine = torch.zeros((per,) + fshape, dtype=torch.uint8 if to_uint8 else torch.float32, device=latents.device)
            if outs:
                mine[: hi - lo].copy_(torch.cat(outs, 0))
            allf = mine.new_empty((W_ * per,) + fshape)
            dist.all_gather_into_tensor(allf, mine, group=group)   # the one collective of the decode
            o = allf[:n]
        if to_uint8:
            return o.view(B, F, *o.shape[1:])
        return o.view(B, F, *o.shape[1:]).permute(0, 2, 1, 3, 4)

    def to_host_async(self, frames, slot=0):
        """Hand decoded frames to the host-side writer without stalling the GPU queue (f3; consumer:
        utils/video_op.py:181-213, which today receives `video.cpu()` — a synchronous 22 MB fp32 copy per video,
        inference_text2video_entrance.py:225): `frames` (the uint8 [B, F, H, W, 3] of decode_video: 5.5 MB) is copied
        into a reused PINNED host buffer on a side stream that first waits for the compute stream; returns (host tensor,
        event).  The next prompt's sampling is enqueued immediately; the writer calls `event.synchronize()` before it
        reads.  `slot` selects one of the reusable pinned buffers (two prompts in flight: slots 0 / 1)."""
        if not frames.is_cuda:
            return frames, None
        key = (slot, tuple(frames.shape), frames.dtype)
        bufs = self.__dict__.setdefault("_pinned", {})
        host = bufs.get(key)
        if host is None:
            host = torch.empty(frames.shape, dtype=frames.dtype, pin_memory=True)
            bufs[key] = host
        side = self.__dict__.get("_copy_stream")
        if side is None:
            side = torch.cuda.Stream(device=frames.device)
            self.__dict__["_copy_stream"] = side
        side.wait_stream(torch.cuda.current_stream(frames.device))
        with torch.cuda.stream(side):
            host.copy_(frames, non_blocking=True)
            frames.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(side)
        return host, ev

    @torch.no_grad()
    def _decode_rows(self, z):
        """-> (decoder output rows [n*H*W, out_ch] fp32, n, H, W)"""
        be, dt = ops.backend(), self.compute_dtype
        if self._packed is None:
            self.pack()
        P = self._packed
        n, zc, H, W = z.shape
        z = z.float().contiguous()
        dec = self.decoder
        # post_quant_conv 1x1 (autoencoder.py:100-101): NCHW -> rows [n*H*W, zc]
        zq = torch.empty((n * H * W, zc), dtype=torch.float32, device=z.device)
        Wp, bp = P["post_quant_conv"]
        nchw = (zc * H * W, 0, H * W, W, 1)
        rows = lambda c: (H * W * c, 0, 1, W * c, c)
        be.pointwise_small(z, n, 1, zc, H, W, nchw, Wp, bp, zc, zq, rows(zc))
        h = self._stem("decoder", zq, n, zc, H, W, rows(zc))
        h = self._mid(dec.mid, h, n, H, W)
        for i in reversed(range(dec.num_resolutions)):
            lvl = dec.up[i]
            for blk in lvl.block:
                h = self._resnet(blk, h, n, H, W)
            if i != 0:
                a = be.act_cast(h, 0, dt)
                h, H, W = self._conv(a, P[lvl.upsample._pname], n, H, W, h.shape[1], ups=1, colstats=True)
        a, _ = be.groupnorm(h, None, n, H * W, 32, 1e-6, *P["decoder.norm_out"], True, False, dt)
        o, _, _ = self._conv(a, P["decoder.conv_out"], n, H, W, h.shape[1])
        return o, n, H, W

    @torch.no_grad()
    def _encode_rows(self, x):
        be, dt = ops.backend(), self.compute_dtype
        if self._packed is None:
            self.pack()
        P = self._packed
        n, cin, H, W = x.shape
        x = x.float().contiguous()
        enc = self.encoder
        h = self._stem("encoder", x, n, cin, H, W, (cin * H * W, 0, H * W, W, 1))
        for i in range(enc.num_resolutions):
            lvl = enc.down[i]
            for blk in lvl.block:
                h = self._resnet(blk, h, n, H, W)
            if i != enc.num_resolutions - 1:
                a = be.act_cast(h, 0, dt)
                h, H, W = self._conv(a, P[lvl.downsample._pname], n, H, W, h.shape[1], stride=2, pad=0, colstats=True)
        h = self._mid(enc.mid, h, n, H, W)
        a, _ = be.groupnorm(h, None, n, H * W, 32, 1e-6, *P["encoder.norm_out"], True, False, dt)
        m, _, _ = self._conv(a, P["encoder.conv_out"], n, H, W, h.shape[1])
        # quant_conv 1x1 on the moments, rows -> rows
        zc2 = m.shape[1]
        Wq, bq = P["quant_conv"]
        mom = torch.empty_like(m)
        r = (H * W * zc2, 0, 1, W * zc2, zc2)
        be.pointwise_small(m, n, 1, zc2, H, W, r, Wq, bq, zc2, mom, r)
        return mom, n, H, W

    def encode(self, x):
        mom, n, H, W = self._graphed("encode", self._encode_rows, x)
        return DiagonalGaussianDistribution(mom.clone(), n, self.zc, H, W)     # the posterior outlives the next chunk

    def encode_firsr_stage(self, x, scale_factor=1.0):
        return self.encode(x).sample(scale_factor)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
