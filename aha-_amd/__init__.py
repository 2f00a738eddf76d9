"""aha_amd -- MI355X-native per-frame streaming inference for the Aha! hot path.

Directory name is ``aha-_amd``; import it as ``aha_amd`` (see the shim ``aha_amd.py`` at
the repository root).  The compute path is the C-ABI library ``libaha_amd.so`` (hand-written
HIP for gfx950, declared in ``include/aha_amd.h``); this package is the Python host that
mirrors the reference's driver / model / cache interfaces on top of it.
"""
from .config import LiveConfig, LMConfig, VisionConfig, preset  # noqa: F401

__all__ = ["LiveConfig", "LMConfig", "VisionConfig", "preset"]
