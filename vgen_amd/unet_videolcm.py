"""UNetSD_VideoLCM drop-in for the text-to-video LCM configuration (reference:
tools/modules/unet/unet_videolcm.py:186-784 with `video_compositions: ['text']`,
configs/videolcm_t2v_infer.yaml:46-67).

With only the text composition the reference builds no condition stem at all: `pre_image` is an empty
Sequential (:409), the `concat` buffer stays zero (:598) and the context is the text tokens (:714-741), so
the forward is the t2v trunk behind a stem conv with `in_dim + concat_dim` input channels of which the last
`concat_dim` see zeros.  Other compositions (depth / sketch / motion / image / histogram ... stems,
:294-372, 598-699) are not built here: asking for them raises NotImplementedError ("next").
The LCM sampler passes float timesteps (inference_videolcm_entrance.py:239); `vgen_timestep_embedding` takes
fp32 t, so those work unchanged.
"""
from __future__ import annotations

import torch

from .unet import UNetSD_T2VBase

_UNSUPPORTED = ("depth", "image", "motion", "local_image", "single_sketch", "masked", "canny", "sketch", "histogram")


class UNetSD_VideoLCM(UNetSD_T2VBase):
    @staticmethod
    def _extra_stem_channels(kwargs):
        return kwargs["_lcm_concat"]

    def __init__(self, config=None, in_dim=4, dim=512, y_dim=512, context_dim=1024, hist_dim=156, concat_dim=8,
                 out_dim=6, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64, num_res_blocks=3,
                 attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1, temporal_attn_times=1,
                 temporal_attention=True, use_checkpoint=False, use_image_dataset=False, use_fps_condition=False,
                 use_sim_mask=False, misc_dropout=0.5, training=True, inpainting=True, p_all_zero=0.1,
                 p_all_keep=0.1, zero_y=None, black_image_feature=None, adapter_transformer_layers=1, num_tokens=4,
                 use_lcm=True, compute_dtype=None, **kwargs):
        comps = list(getattr(config, "video_compositions", None) or
                     (config.get("video_compositions") if isinstance(config, dict) else None) or ["text"])
        extra = [c for c in comps if c != "text"]
        if extra:
            raise NotImplementedError(f"UNetSD_VideoLCM: condition stems {extra} are not built natively yet; "
                                      "only video_compositions == ['text'] (configs/videolcm_t2v_infer.yaml)")
        if getattr(config, "use_text_clip_vip_model", False):
            raise NotImplementedError("UNetSD_VideoLCM: use_text_clip_vip_model")
        super().__init__(config=config, in_dim=in_dim, dim=dim, y_dim=y_dim, context_dim=context_dim,
                         hist_dim=hist_dim, out_dim=out_dim, num_tokens=num_tokens, dim_mult=dim_mult,
                         num_heads=num_heads, head_dim=head_dim, num_res_blocks=num_res_blocks,
                         attn_scales=attn_scales, use_scale_shift_norm=use_scale_shift_norm, dropout=dropout,
                         temporal_attn_times=temporal_attn_times, temporal_attention=temporal_attention,
                         use_checkpoint=use_checkpoint, use_image_dataset=use_image_dataset,
                         use_sim_mask=use_sim_mask, training=training, inpainting=inpainting,
                         use_fps_condition=use_fps_condition, p_all_zero=p_all_zero, p_all_keep=p_all_keep,
                         zero_y=zero_y, adapter_transformer_layers=adapter_transformer_layers,
                         compute_dtype=compute_dtype, _lcm_concat=concat_dim, **kwargs)
        self.cfg = config
        self.concat_dim = concat_dim
        self.video_compositions = comps
        self.black_image_feature = black_image_feature
        self._zeros = None

    @torch.no_grad()
    def forward(self, x, t, t_w=None, y=None, fps=None, video_mask=None, focus_present_mask=None,
                prob_focus_present=0., mask_last_frame_num=0, **conds):
        given = [k for k in _UNSUPPORTED if conds.get(k) is not None]
        if given:
            raise NotImplementedError(f"UNetSD_VideoLCM: conditions {given} need the native stems (SURVEY §8 f2)")
        B, C, F, H, W = x.shape
        shape = (B, self.concat_dim, F, H, W)
        if self._zeros is None or self._zeros.shape != shape or self._zeros.device != x.device:
            self._zeros = torch.zeros(shape, dtype=torch.float32, device=x.device)
        ctx = y if y is not None else self.zero_y.repeat(B, 1, 1)          # full zero_y here (:740)
        return self._trunk(torch.cat([x.float(), self._zeros], 1), t, ctx, fps)

    def forward_units(self, x, t, kwargs_list):
        G = len(kwargs_list)
        if any(kw.get("y") is None for kw in kwargs_list) or any(k not in ("y", "fps", "t_w") for kw in kwargs_list for k in kw):
            return tuple(self.forward(x, t, **kw) for kw in kwargs_list)
        y = torch.cat([kw["y"] for kw in kwargs_list], 0)
        fps = None
        if all(kw.get("fps") is not None for kw in kwargs_list):
            fps = torch.cat([kw["fps"].reshape(-1) for kw in kwargs_list], 0)
        out = self.forward(x.repeat(G, 1, 1, 1, 1), t.repeat(G), y=y, fps=fps)
        return tuple(out.chunk(G, 0))
