"""Scalar / point codecs (dot_ring/vrf/codec.py:9-51)."""
from __future__ import annotations

from ..curve import valid_points


def scalar_len(cv) -> int:
    return (cv.curve.params.subgroup_order.bit_length() + 7) // 8


def point_len(cv) -> int:
    return cv.curve.params.encoding.point_len


def enc_scalar(cv, value: int) -> bytes:
    return int(value % cv.curve.params.subgroup_order).to_bytes(scalar_len(cv), "little")


def dec_scalar(cv, value: bytes) -> int:
    if len(value) != scalar_len(cv):
        raise ValueError(f"scalar must be exactly {scalar_len(cv)} bytes")
    scalar = int.from_bytes(value, "little")
    if scalar >= cv.curve.params.subgroup_order:
        raise ValueError("scalar is not canonical")
    return scalar


def dec_scalar_mod(cv, value: bytes) -> int:
    return int.from_bytes(value, "little") % cv.curve.params.subgroup_order


def enc_point(point) -> bytes:
    return point.point_to_string()


def dec_points(cv, values) -> list:
    """Decode + subgroup-validate several points with two kernel launches in total; raises like dec_point."""
    pts = []
    for value in values:
        if len(value) != point_len(cv):
            raise ValueError(f"point must be exactly {point_len(cv)} bytes")
        pts.append(cv.point_type.string_to_point(value))
    if not all(valid_points(pts)):
        raise ValueError("point is not a valid nonidentity subgroup point")
    return pts


def dec_point(cv, value: bytes):
    return dec_points(cv, [value])[0]


def enc_64(value: int) -> bytes:
    if not 0 <= value < 1 << 64:
        raise ValueError("value does not fit in uint64")
    return value.to_bytes(8, "little")
