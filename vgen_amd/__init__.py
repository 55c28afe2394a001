"""vgen_amd — MI355X-native (gfx950) sampling path for VGen-style video diffusion.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of
include/vgen_hip.h), `lib`/`ops` (ctypes binding), and the host-side mirrors of the reference's
operator interface for this path: `UNetSD_T2VBase`, `AutoencoderKL`, `DiffusionDDIM`, the
registry seam (`install`) and the unit partition for multi-GPU sampling.
"""
from .registry import AUTO_ENCODER, DIFFUSION, MODEL, Registry, install  # noqa: F401

__all__ = ["MODEL", "AUTO_ENCODER", "DIFFUSION", "Registry", "install"]


def __getattr__(name):
    if name == "UNetSD_T2VBase":
        from .unet import UNetSD_T2VBase
        return UNetSD_T2VBase
    if name == "AutoencoderKL":
        from .vae import AutoencoderKL
        return AutoencoderKL
    if name == "DiffusionDDIM":
        from .diffusion import DiffusionDDIM
        return DiffusionDDIM
    raise AttributeError(name)
