"""VRF base class: `Scheme[CurveVariant]` specialisation (dot_ring/vrf/vrf.py:10-48)."""
from __future__ import annotations

from ..curve import CurveVariant


class VRF:
    cv = None

    _specialised: dict = {}

    def __class_getitem__(cls, curve_variant):
        if not isinstance(curve_variant, CurveVariant):
            return cls
        # one class object per (scheme, curve): dataclass equality compares classes, and per-class memos stay alive
        key = (cls, curve_variant.name)
        hit = VRF._specialised.get(key)
        if hit is None or hit.cv is not curve_variant:
            hit = type(f"{cls.__name__}[{curve_variant.name}]", (cls,), {"cv": curve_variant})
            VRF._specialised[key] = hit
        return hit

    @classmethod
    def prove(cls, *args, **kwargs):
        raise NotImplementedError(f"{cls.__name__} does not implement prove")

    def verify(self, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} does not implement verify")

    def encode(self) -> bytes:
        raise NotImplementedError(f"{self.__class__.__name__} does not implement encode")

    @classmethod
    def decode(cls, data: bytes):
        raise NotImplementedError(f"{cls.__name__} does not implement from_bytes")

    @classmethod
    def batch_verify(cls, *args, **kwargs):
        raise NotImplementedError(f"{cls.__name__} does not implement batch_verify")
