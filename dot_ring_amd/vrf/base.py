"""Common base of the four schemes: `Scheme[curve_variant]` yields the scheme bound to one suite
(reference behaviour: dot_ring/vrf/vrf.py:10-48).  Bound classes are created once per (scheme, suite) and reused, so
dataclass equality between proofs works and per-class memos survive."""
from __future__ import annotations

from ..curve import CurveVariant

_BOUND: dict = {}


def _unsupported(owner, what: str):
    name = owner.__name__ if isinstance(owner, type) else type(owner).__name__
    return NotImplementedError(f"{name} does not implement {what}")


class VRF:
    cv = None

    def __class_getitem__(cls, variant):
        if not isinstance(variant, CurveVariant):
            return cls
        bound = _BOUND.get((cls, variant.name))
        if bound is None or bound.cv is not variant:
            bound = _BOUND[(cls, variant.name)] = type(f"{cls.__name__}[{variant.name}]", (cls,), {"cv": variant})
        return bound

    # the interface every scheme fills in
    @classmethod
    def prove(cls, *args, **kwargs):
        raise _unsupported(cls, "prove")

    def verify(self, *args, **kwargs):
        raise _unsupported(self, "verify")

    def encode(self) -> bytes:
        raise _unsupported(self, "encode")

    @classmethod
    def decode(cls, data: bytes):
        raise _unsupported(cls, "from_bytes")

    @classmethod
    def batch_verify(cls, *args, **kwargs):
        raise _unsupported(cls, "batch_verify")
