"""UNetSD_VideoLCM drop-in for the text-to-video LCM configuration (reference:
tools/modules/unet/unet_videolcm.py:186-784 with `video_compositions: ['text']`,
configs/videolcm_t2v_infer.yaml:46-67).

With only the text (and optionally the global-image) composition the reference builds no spatial condition
stem: `pre_image` is an empty Sequential (:409), the `concat` buffer stays zero (:598) and the context is the
text tokens (:714-741) plus, with 'image', num_tokens tokens from `pre_image_condition` (:284-288, 743-745).
The forward is then the t2v trunk behind a stem conv with `in_dim + concat_dim` input channels of which the
last `concat_dim` see zeros.  The spatial compositions (depthmap / motion / canny / mask / sketch / single_sketch /
local_image, :294-372, 598-699) each add a `concat_dim`-channel map into that buffer; their stems depend only on
the conditioning maps, so they are evaluated once per conditioning tensor on the C ABI's stem kernels and cached (prompt
constants ahead of the hot path, like UNetSD_I2VGen's).  'histogram' adds a different context token per FRAME
(:375-380, 747-755): the trunk then projects K/V per (prompt, frame) — 16 x 78 rows instead of 77 — and the
cross-attention kernel addresses them frame-major through its strides.  `UNetSD_TFT2V` (unet_tf2tv.py) is the same class without the unused t_w argument.
The LCM sampler passes float timesteps (inference_videolcm_entrance.py:239); `vgen_timestep_embedding` takes
fp32 t, so those work unchanged.
"""
from __future__ import annotations

import torch

import torch.nn as nn

from . import ops
from .unet import UNetSD_T2VBase, _f32, pack_linear
from .unet_i2vgen import _FrameTransformer, conv_stack, frame_transformer, mlp_f32

# spatial compositions: composition name -> (forward kwarg, stem attribute, transformer attribute, input channels)
# (unet_videolcm.py:294-372 / unet_tf2tv.py likewise; every stem is Conv3x3 - SiLU - AdaptiveAvgPool(res/2) -
#  Conv3x3 s2 - SiLU - Conv3x3 s2 -> concat_dim channels at latent resolution, then one Transformer_v2 over frames)
_SPATIAL = {
    "depthmap": ("depth", "depth_embedding", "depth_embedding_after", 1),
    "motion": ("motion", "motion_embedding", "motion_embedding_after", 2),
    "canny": ("canny", "canny_embedding", "canny_embedding_after", 1),
    "mask": ("masked", "masked_embedding", "mask_embedding_after", 4),
    "sketch": ("sketch", "sketch_embedding", "sketch_embedding_after", 1),
    "single_sketch": ("single_sketch", "single_sketch_embedding", "single_sketch_embedding_after", 1),
    "local_image": ("local_image", "local_image_embedding", "local_image_embedding_after", 3),
}
# order in which the reference accumulates the maps into `concat` (unet_videolcm.py:599-699): the fp32 sum is
# order-dependent, so the same order keeps three or more active compositions bit-comparable with the oracle
_ACCUMULATION_ORDER = ("depthmap", "local_image", "motion", "canny", "sketch", "single_sketch", "mask")
_UNSUPPORTED = ()
_SUPPORTED_COMPOSITIONS = ("text", "image", "histogram") + tuple(_SPATIAL)


def _compositions(config):
    comps = getattr(config, "video_compositions", None)
    if comps is None and isinstance(config, dict):
        comps = config.get("video_compositions")
    return list(comps or ["text"])


class _ComposerTrunk(UNetSD_T2VBase):
    """Shared body of UNetSD_VideoLCM / UNetSD_TFT2V for the compositions whose stems are built natively:
    'text' (context = y) and 'image' (num_tokens extra context tokens from `pre_image_condition`, a
    Linear-SiLU-Linear on the global image embedding: unet_videolcm.py:284-288, 743-745 — two tap-GEMMs)."""

    @staticmethod
    def _extra_stem_channels(kwargs):
        return kwargs["_composer_concat"]

    @staticmethod
    def _default_precision(config, kwargs):
        """precision=None: "mixed" like every trunk — unless the composition list carries SPATIAL condition stems.  The
        full-width vcomposer fixture (32 frames 896 x 512, tests/golden/unet_vcomposer_full.pt) measures 1.01e-3 from the
        reference's fp32 forward in "mixed" and 8.6e-4 in "high": the summed condition maps change the activation
        statistics the 16-bit operands see, and level-0 two-term weights alone no longer buy the north-star's 1e-3
        (DESIGN §4.1).  An explicit keyword always wins."""
        if kwargs.get("precision") is None and any(c in _SPATIAL for c in _compositions(config)):
            kwargs["precision"] = "high"
        return kwargs

    def _init_composer(self, config, concat_dim, num_tokens, black_image_feature, inpainting, adapter_layers,
                       hist_dim=156):
        self.cfg = config
        self.hist_dim = hist_dim
        self.concat_dim = concat_dim
        self.num_tokens = num_tokens
        self.video_compositions = _compositions(config)
        self.black_image_feature = black_image_feature
        res = getattr(config, "resolution", None) or (config.get("resolution") if isinstance(config, dict) else None)
        self.resolution = res
        if "image" in self.video_compositions:
            cd = self.context_dim
            self.pre_image_condition = nn.Sequential(nn.Linear(cd, cd), nn.SiLU(), nn.Linear(cd, cd * num_tokens))
        # spatial condition stems: prompt constants (they depend on the conditioning maps only), evaluated once per
        # conditioning tensor on the stem kernels (vgen_conv3x3_small / vgen_adaptive_avgpool2d / vgen_frame_transformer)
        # and cached — like UNetSD_I2VGen's; the nn.Modules hold the parameters, registered in the reference's
        # order and under its names so stock checkpoints load strict
        c4 = concat_dim * 4
        for name in ("depthmap", "motion", "canny", "mask", "sketch", "single_sketch", "local_image"):
            if name not in self.video_compositions:
                continue
            kwarg, stem, after, cin = _SPATIAL[name]
            if res is None:
                raise ValueError(f"{type(self).__name__}: config.resolution is needed for the '{name}' stem")
            if name == "mask" and not inpainting:
                setattr(self, stem, None)
            else:
                setattr(self, stem, nn.Sequential(
                    nn.Conv2d(cin, c4, 3, padding=1), nn.SiLU(), nn.AdaptiveAvgPool2d((res[1] // 2, res[0] // 2)),
                    nn.Conv2d(c4, c4, 3, stride=2, padding=1), nn.SiLU(), nn.Conv2d(c4, concat_dim, 3, stride=2, padding=1)))
            setattr(self, after, _FrameTransformer(heads=2, dim=concat_dim, dim_head=concat_dim, mlp_dim=concat_dim,
                                                   depth=adapter_layers))
        if "histogram" in self.video_compositions:
            # one extra context token PER FRAME (:375-380, 748-755): Linear - SiLU - Linear on [B, F, hist_dim]
            self.hist_context_embedding = nn.Sequential(nn.Linear(self.hist_dim, self.embed_dim), nn.SiLU(),
                                                        nn.Linear(self.embed_dim, self.context_dim))
        self._zeros = None
        self._pic = None
        self._stem_cache = {}

    @staticmethod
    def _check(config, name):
        extra = [c for c in _compositions(config) if c not in _SUPPORTED_COMPOSITIONS]
        if extra:
            raise NotImplementedError(f"{name}: condition stems {extra} are not built natively yet; supported "
                                      f"video_compositions: {list(_SUPPORTED_COMPOSITIONS)} (SURVEY §8 f2)")
        if getattr(config, "use_text_clip_vip_model", False):
            raise NotImplementedError(f"{name}: use_text_clip_vip_model")

    def pack(self, device=None):
        super().pack(device)
        self._pic = None
        self._stem_cache = {}

    def invalidate(self):
        """Parameters (may) have changed or moved: drop the packed operands AND every cached stem output."""
        super().invalidate()
        self._pic = None
        self._zeros = None
        self._stem_cache = {}

    def _image_tokens(self, image, B):
        be, dt = ops.backend(), self.compute_dtype
        if self._pic is None:
            l0, l2 = self.pre_image_condition[0], self.pre_image_condition[2]
            self._pic = ((pack_linear(l0.weight, dt), _f32(l0.bias)), (pack_linear(l2.weight, dt), _f32(l2.bias)))
        x = image.to(dtype=torch.float32).reshape(B, self.context_dim).contiguous()
        h = self._linear(be.act_cast(x, 0, dt), self._pic[0], B)
        o = self._linear(be.act_cast(h, 1, dt), self._pic[1], B)
        return o.view(B, self.num_tokens, self.context_dim)

    @torch.no_grad()
    def _spatial_stem(self, name, cond, B):
        """One composition's contribution to the concat buffer (:600-699): [B, c, F, Hc, Wc] -> [B, concat_dim, F, h, w].
        Cached per conditioning tensor object (+ version), holding a reference (see UNetSD_I2VGen.condition_stems)."""
        key = (name, id(cond), cond._version)
        hit = self._stem_cache.get(key)
        if hit is not None and hit[0] is cond:
            return hit[1]
        kwarg, stem, after, cin = _SPATIAL[name]
        Bc, c, F, Hc, Wc = cond.shape
        z = conv_stack(cond.float().permute(0, 2, 1, 3, 4).reshape(Bc * F, c, Hc, Wc).contiguous(), getattr(self, stem))
        out = frame_transformer(getattr(self, after), z, Bc, F)                    # [B, concat_dim, F, h, w]
        if len(self._stem_cache) >= 32:
            self._stem_cache.clear()
        self._stem_cache[key] = (cond, out)
        return out

    @torch.no_grad()
    def _prepare(self, shape, device, y=None, image=None, t_w=None, fps=None, video_mask=None, focus_present_mask=None,
                 prob_focus_present=0., mask_last_frame_num=0, **conds):
        """Everything ahead of the trunk for one set of conditions -> (concat [B, concat_dim, F, H, W],
        ctx [B or B*F, L, D], ctx_per_frame)."""
        B, C, F, H, W = shape
        histogram = conds.pop("histogram", None)
        if self._packed is None:
            self.pack()
        concat = None
        for name in _ACCUMULATION_ORDER:                                # the reference's order of accumulation
            kwarg, stem, after, cin = _SPATIAL[name]
            cond = conds.get(kwarg)
            if cond is None:
                continue
            if name not in self.video_compositions:
                raise ValueError(f"condition '{kwarg}' given but '{name}' is not in video_compositions")
            c = self._spatial_stem(name, cond, B)
            concat = c if concat is None else ops.backend().lincomb4(concat, c, None, None, 1.0, 1.0, 0.0, 0.0)
        unknown = [k for k, v in conds.items() if v is not None and k not in [a[0] for a in _SPATIAL.values()]]
        if unknown:
            raise TypeError(f"{type(self).__name__}.forward: unexpected arguments {unknown}")
        if concat is None:
            zshape = (B, self.concat_dim, F, H, W)
            if self._zeros is None or self._zeros.shape != zshape or self._zeros.device != device:
                self._zeros = torch.zeros(zshape, dtype=torch.float32, device=device)
            concat = self._zeros
        ctx = y if y is not None else self.zero_y.repeat(B, 1, 1)          # full zero_y here (:740)
        ctx = ctx.float()
        if image is not None:
            if "image" not in self.video_compositions:
                raise ValueError("image condition given but 'image' is not in video_compositions")
            ctx = torch.cat([ctx, self._image_tokens(image, B).to(ctx.device)], 1)
        per_frame = False
        if histogram is not None:
            if "histogram" not in self.video_compositions:
                raise ValueError("histogram condition given but 'histogram' is not in video_compositions")
            key = ("histogram", id(histogram), histogram._version)
            hit = self._stem_cache.get(key)
            if hit is not None and hit[0] is histogram:
                hc = hit[1]
            else:
                hc = mlp_f32(histogram.reshape(B * F, -1), self.hist_context_embedding).view(B, F, 1, self.context_dim)
                self._stem_cache[key] = (histogram, hc)
            # the shared tokens repeated per frame, then this frame's histogram token (:747-755)
            ctx = torch.cat([ctx.unsqueeze(1).expand(B, F, ctx.shape[1], ctx.shape[2]), hc.to(ctx.device)], 2)
            ctx = ctx.reshape(B * F, ctx.shape[2], ctx.shape[3])
            per_frame = True
        return concat.to(device), ctx, per_frame

    @torch.no_grad()
    def forward(self, x, t, **kw):
        self._maybe_auto_calibrate(tuple(x.shape), x.device, kw, t.dtype)
        concat, ctx, per_frame = self._prepare(tuple(x.shape), x.device, **kw)
        return self._trunk(torch.cat([x.float(), concat], 1), t, ctx, kw.get("fps"), ctx_per_frame=per_frame)

    def _prepare_units(self, shape, device, kwargs_list):
        """CFG units of one latent batch (see UNetSD_T2VBase._prepare_units): stems / context are prepared per
        unit on the caller's own conditioning tensors (cached), then stacked along the batch."""
        prep = [self._prepare(tuple(shape), device, **kw) for kw in kwargs_list]
        same = len({(p[1].shape[1], p[2]) for p in prep}) == 1 and \
            len({kw.get("fps") is None for kw in kwargs_list}) == 1
        if not same:
            return None
        fps = None
        if self.use_fps_condition and kwargs_list[0].get("fps") is not None:
            fps = torch.cat([kw["fps"].reshape(-1).to(device) for kw in kwargs_list], 0)
        # per-frame contexts are frame-major per prompt: cat along dim 0 is right
        return dict(extra=torch.cat([p[0] for p in prep], 0), ctx=torch.cat([p[1] for p in prep], 0),
                    per_frame=prep[0][2], fps=fps)


def _trunk_kwargs(loc):
    keys = ("config", "in_dim", "dim", "y_dim", "context_dim", "hist_dim", "out_dim", "num_tokens", "dim_mult", "num_heads",
            "head_dim", "num_res_blocks", "attn_scales", "use_scale_shift_norm", "dropout", "temporal_attn_times",
            "temporal_attention", "use_checkpoint", "use_image_dataset", "use_sim_mask", "training", "inpainting",
            "use_fps_condition", "p_all_zero", "p_all_keep", "zero_y", "adapter_transformer_layers", "compute_dtype")
    return {k: loc[k] for k in keys}


class UNetSD_VideoLCM(_ComposerTrunk):
    def __init__(self, config=None, in_dim=4, dim=512, y_dim=512, context_dim=1024, hist_dim=156, concat_dim=8,
                 out_dim=6, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64, num_res_blocks=3,
                 attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1, temporal_attn_times=1,
                 temporal_attention=True, use_checkpoint=False, use_image_dataset=False, use_fps_condition=False,
                 use_sim_mask=False, misc_dropout=0.5, training=True, inpainting=True, p_all_zero=0.1,
                 p_all_keep=0.1, zero_y=None, black_image_feature=None, adapter_transformer_layers=1, num_tokens=4,
                 use_lcm=True, compute_dtype=None, **kwargs):
        self._check(config, "UNetSD_VideoLCM")
        kwargs = self._default_precision(config, kwargs)
        super().__init__(**_trunk_kwargs(locals()), _composer_concat=concat_dim, **kwargs)
        self._init_composer(config, concat_dim, num_tokens, black_image_feature, inpainting, adapter_transformer_layers,
                            hist_dim)


class UNetSD_TFT2V(_ComposerTrunk):
    """reference: tools/modules/unet/unet_tf2tv.py:189-777 — UNetSD_VideoLCM without the (unused) t_w input;
    configs/tft2v_t2v_infer.yaml uses video_compositions ['text', 'image']."""

    def __init__(self, config=None, in_dim=4, dim=512, y_dim=512, context_dim=1024, hist_dim=156, concat_dim=8,
                 out_dim=6, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64, num_res_blocks=3,
                 attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1, temporal_attn_times=1,
                 temporal_attention=True, use_checkpoint=False, use_image_dataset=False, use_fps_condition=False,
                 use_sim_mask=False, misc_dropout=0.5, training=True, inpainting=True, p_all_zero=0.1,
                 p_all_keep=0.1, zero_y=None, black_image_feature=None, adapter_transformer_layers=1, num_tokens=4,
                 compute_dtype=None, **kwargs):
        self._check(config, "UNetSD_TFT2V")
        kwargs = self._default_precision(config, kwargs)
        super().__init__(**_trunk_kwargs(locals()), _composer_concat=concat_dim, **kwargs)
        self._init_composer(config, concat_dim, num_tokens, black_image_feature, inpainting, adapter_transformer_layers,
                            hist_dim)
