"""ctypes binding of libvgen_hip.so (include/vgen_hip.h).

The product path has NO fallback: if the shared object is missing or a call fails, a
VgenHipError is raised.  (tests inject oracle/abi_emulator.py through vgen_amd.ops.set_backend
to exercise the host logic on CPU; nothing in this package imports the oracle.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VGEN_HIP_LIB: load another build of the library instead (same-box A/B of two kernel versions; tuning switch)
LIB_PATH = os.environ.get("VGEN_HIP_LIB") or os.path.join(HERE, "libvgen_hip.so")

VGEN_BF16, VGEN_F16, VGEN_F32 = 0, 1, 2
TAP_LINEAR, TAP_CONV3X3, TAP_TEMPORAL3 = 0, 1, 2
EPI_NONE, EPI_GEGLU = 0, 1
ABI_VERSION = 5


class VgenHipError(RuntimeError):
    pass


class TapGemmArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int32), ("dtype", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64), ("C1", C.c_int32), ("taps", C.c_int32),
        ("mode", C.c_int32),
        ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("ups", C.c_int32),
        ("F", C.c_int32), ("S", C.c_int64),
        ("A2", C.c_void_p), ("lda2", C.c_int64), ("C2", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("rowbias_ld", C.c_int64),
        ("rows_per_rb", C.c_int64), ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_dtype", C.c_int32),
        ("epilogue", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("crop_t", C.c_int32),
        ("colstats", C.c_void_p), ("dualw", C.c_int32), ("split_out", C.c_int32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
        ("dtype", C.c_int32), ("heads", C.c_int32), ("nq", C.c_int32), ("nk", C.c_int32),
        ("nbatch", C.c_int64), ("inner", C.c_int64),
        ("q_rs", C.c_int64), ("q_bo", C.c_int64), ("q_bi", C.c_int64),
        ("k_rs", C.c_int64), ("k_bo", C.c_int64), ("k_bi", C.c_int64),
        ("v_rs", C.c_int64), ("v_bo", C.c_int64), ("v_bi", C.c_int64),
        ("o_rs", C.c_int64), ("o_bo", C.c_int64), ("o_bi", C.c_int64),
        ("scale", C.c_float), ("causal", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol include/vgen_hip.h declares
_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "vgen_version": (C.c_int, []),
    "vgen_last_error": (C.c_char_p, []),
    "vgen_groupnorm_ws_bytes": (_sz, [_i64, _i64]),
    "vgen_groupnorm": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _i32,
                                 _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "vgen_groupnorm_cs": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _i32,
                                    _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "vgen_layernorm": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _i32, _vp]),
    "vgen_tapgemm_ws_bytes": (_sz, [C.POINTER(TapGemmArgs)]),
    "vgen_tapgemm": (C.c_int, [C.POINTER(TapGemmArgs), _vp]),
    "vgen_tapgemm_query_plan": (C.c_int, [C.POINTER(TapGemmArgs), _vp]),
    "vgen_tapgemm_set_plans": (C.c_int, [_vp, _i32]),
    "vgen_attention": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "vgen_softmax_rows": (C.c_int, [_vp, _i64, _i32, _i64, _f32, _vp, _i64, _i32, _vp]),
    "vgen_act_cast": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "vgen_cast_split": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i32, _vp]),
    "vgen_timestep_embedding": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "vgen_conv3x3_small": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "vgen_adaptive_avgpool2d": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vgen_frame_transformer": (C.c_int, [_vp, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _i32, _f32, _i32, _vp]),
    "vgen_embed_tokens": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vgen_linear_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "vgen_im2col3x3_small": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64,
                                       _i64, _vp, _i32, _i32, _i32, _vp]),
    "vgen_pointwise_small": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64,
                                       _i64, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i64,
                                       _vp]),
    "vgen_cfg_ddim_step_units": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _i64, _i64, _vp,
                                           _vp, _vp, _i32, _i64, _i64, _vp]),
    "vgen_cfg_ddim_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _i64, _i64, _vp,
                                     _vp, _vp]),
    "vgen_gaussian_sample": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _f32, _vp, _vp]),
    "vgen_lowfreq_filter": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "vgen_scale_channels": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _f32, _vp]),
    "vgen_frames_u8": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "vgen_cfg_stats_ws_bytes": (_sz, [_i64]),
    "vgen_cfg_stats": (C.c_int, [_vp, _vp, _f32, _i32, _i64, _i64, _vp, _vp, _sz, _vp]),
    "vgen_gauss_x0": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    "vgen_lincomb4": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _i64, _vp]),
    "vgen_repeat_rows": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "vgen_gather_rows_f32": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "vgen_dpmpp2m_sde_step": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _i64, _vp]),
}

_lib = None


def load(path: str | None = None):
    """Load libvgen_hip.so; raises VgenHipError (never falls back) when unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch owns the device memory and the streams this library enqueues on, so both must run on ONE HIP
    # runtime: import torch first.  Its wheel bundles its own libamdhip64; loaded after /opt/rocm's copy (which a
    # bare CDLL of libvgen_hip.so would pull in) the process ends up with two runtimes and every launch from this
    # library fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise VgenHipError(
            f"{p} not found: build it with `python -m vgen_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the hot path.")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # missing libamdhip64 etc.
        raise VgenHipError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VgenHipError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.vgen_version()
    if v != ABI_VERSION:
        raise VgenHipError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = _lib.vgen_last_error().decode("utf-8", "replace") if _lib is not None else "?"
        raise VgenHipError(f"{what} failed (rc={rc}): {msg}")
