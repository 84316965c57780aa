"""MI355X-native HideAndSeek multi-UAV pursuit-evasion environment step.

One hot path (the vectorised environment step of thu-uav/Multi-UAV-pursuit-evasion's
HideAndSeek task) as hand-written HIP kernels for gfx950 behind a C ABI (include/hns.h) and a
Python class that mirrors the reference's torchrl ``EnvBase`` surface.  Import name: ``hns_amd``.
"""
from . import abi, config  # noqa: F401
from .config import load_cfg, make_cfg, resolve_hns_cfg  # noqa: F401

__all__ = ["abi", "config", "load_cfg", "make_cfg", "resolve_hns_cfg"]
