"""dot_ring_amd — MI355X-native (gfx950) Ring-VRF hot path behind dot-ring's Python API.

The arithmetic lives in libdotring_hip.so (hand-written HIP, see dot_ring_amd/csrc and include/dotring_hip.h);
this package is the host-side mirror of the reference interface.  There is no CPU fallback.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
