"""Registry seam — how `inference.py` + `configs/*.yaml` resolve to the MI355X-native modules.

The reference never imports model classes directly: engines call `MODEL.build(cfg.UNet)`,
`AUTO_ENCODER.build(cfg.auto_encoder)`, `DIFFUSION.build(cfg.Diffusion)` and the yaml's `type`
string is looked up in a registry (utils/registry.py:24-69, utils/registry_class.py:9-19).
Registering an existing name only warns and REPLACES the entry (utils/registry.py:116-119), so

    import vgen_amd; vgen_amd.install()        # after `from tools import *` in inference.py

swaps 'UNetSD_T2VBase', 'AutoencoderKL', 'DiffusionDDIM' in place.  When the reference tree is not
importable (e.g. on the benchmark box) the same API is served by the stand-alone registries below,
which implement the identical build/lookup/error behaviour.
"""
from __future__ import annotations

import copy
import inspect
import warnings


class Registry(object):
    """name -> class/function map with `build(dict(type=..., **kw), **extra)`."""

    def __init__(self, name, build_func=None, allow_types=("class", "function")):
        self.name = name
        self.allow_types = allow_types
        self.class_map = {}
        self.func_map = {}
        self.build_func = build_func or build_from_config

    def get(self, req_type):
        return self.class_map.get(req_type) or self.func_map.get(req_type)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, table, kind, obj, name, check):
        if not check(obj):
            raise TypeError(f"Registry {self.name}: expected a {kind}, got {type(obj)}")
        if kind not in self.allow_types:
            raise TypeError(f"Register {self.name} only allows type {self.allow_types}, got {kind}")
        key = name or obj.__name__
        if key in table:
            warnings.warn(f"{kind.capitalize()} {key} already registered by {table[key]}, "
                          f"will be replaced by {obj}")
        table[key] = obj
        return obj

    def register_class(self, name=None):
        return lambda cls: self._register(self.class_map, "class", cls, name, inspect.isclass)

    def register_function(self, name=None):
        return lambda fn: self._register(self.func_map, "function", fn, name, inspect.isfunction)

    def __repr__(self):
        keys = sorted(list(self.class_map) + list(self.func_map))
        return f"{self.__class__.__name__} [{self.name}]: " + ", ".join(keys)


def build_from_config(cfg, registry, **kwargs):
    """Same contract as the reference builder: TypeError for a non-dict / non-Registry,
    KeyError for a missing `type` or an unknown name, and constructor failures re-raised as
    Exception("Failed to init class ...")."""
    if not isinstance(cfg, dict):
        raise TypeError(f"config must be type dict, got {type(cfg)}")
    if "type" not in cfg:
        raise KeyError(f"config must contain key type, got {cfg}")
    if not hasattr(registry, "class_map"):
        raise TypeError(f"registry must be type Registry, got {type(registry)}")
    cfg = copy.deepcopy(cfg)
    req = cfg.pop("type")
    entry = req
    if isinstance(req, str):
        entry = registry.get(req)
        if entry is None:
            raise KeyError(f"{req} not found in {registry.name} registry")
    if kwargs is not None:
        cfg.update(kwargs)
    if inspect.isclass(entry):
        try:
            return entry(**cfg)
        except Exception as e:
            raise Exception(f"Failed to init class {entry}, with {e}")
    if inspect.isfunction(entry):
        try:
            return entry(**cfg)
        except Exception as e:
            raise Exception(f"Failed to invoke function {entry}, with {e}")
    raise TypeError(f"type must be str or class, got {type(entry)}")


MODEL = Registry("MODEL")
AUTO_ENCODER = Registry("AUTO_ENCODER")
EMBEDDER = Registry("EMBEDDER")
DIFFUSION = Registry("DIFFUSION")


def _native_classes():
    from .diffusion import DiffusionDDIM
    from .diffusion_gauss import DiffusionDDIMSR
    from .unet import UNetSD_SR600, UNetSD_T2VBase
    from .unet_i2vgen import UNetSD_I2VGen
    from .unet_videolcm import UNetSD_TFT2V, UNetSD_VideoLCM
    from .vae import AutoencoderKL
    from .clip_text import FrozenOpenCLIPEmbedder
    return {"MODEL": [UNetSD_T2VBase, UNetSD_SR600, UNetSD_I2VGen, UNetSD_VideoLCM, UNetSD_TFT2V], "AUTO_ENCODER": [AutoencoderKL],
            "DIFFUSION": [DiffusionDDIM, DiffusionDDIMSR], "EMBEDDER": [FrozenOpenCLIPEmbedder]}


def install(registries=None, quiet=True):
    """Register the native classes under the reference's names.

    `registries`: dict with MODEL / AUTO_ENCODER / DIFFUSION registry objects; default = the
    reference's `utils.registry_class` when importable, else this module's own registries.
    Returns the dict of registries that now resolve to vgen_amd.
    """
    if registries is None:
        try:
            from utils import registry_class as rc      # reference tree on sys.path
            registries = {"MODEL": rc.MODEL, "AUTO_ENCODER": rc.AUTO_ENCODER, "DIFFUSION": rc.DIFFUSION,
                          "EMBEDDER": rc.EMBEDDER}
        except Exception:
            registries = {"MODEL": MODEL, "AUTO_ENCODER": AUTO_ENCODER, "DIFFUSION": DIFFUSION, "EMBEDDER": EMBEDDER}
    with warnings.catch_warnings():
        if quiet:
            warnings.simplefilter("ignore")
        for key, classes in _native_classes().items():
            if key not in registries:             # caller passed a subset (e.g. only MODEL)
                continue
            for cls in classes:
                registries[key].register_class()(cls)
    return registries
