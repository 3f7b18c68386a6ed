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
    """Decode + subgroup-validate several points in ONE kernel launch (dr_bsn_decode_points); raises like dec_point."""
    from .. import runtime

    values = [bytes(v) for v in values]
    for value in values:
        if len(value) != point_len(cv):
            raise ValueError(f"point must be exactly {point_len(cv)} bytes")
    if not values:
        return []
    raw, ok = runtime.context().bsn_decode_points(b"".join(values))
    if not all(ok):
        raise ValueError("point is not a valid nonidentity subgroup point")
    frm, mk = int.from_bytes, cv.point_type._trusted
    return [mk(frm(raw[i : i + 32], "little"), frm(raw[i + 32 : i + 64], "little")) for i in range(0, len(raw), 64)]


def dec_point(cv, value: bytes):
    return dec_points(cv, [value])[0]


def enc_64(value: int) -> bytes:
    if not 0 <= value < 1 << 64:
        raise ValueError("value does not fit in uint64")
    return value.to_bytes(8, "little")
