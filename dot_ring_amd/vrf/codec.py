"""Byte codecs of the VRF layer: little-endian scalars of the group order's width, compressed points
(reference behaviour: dot_ring/vrf/codec.py:9-51 — same function names, lengths and error texts, because the
reference's tests match on them).  Point decoding is a batch operation here: decompression and the subgroup check of
any number of points are one kernel launch (dr_bsn_decode_points)."""
from __future__ import annotations

_U64 = 1 << 64


def _order(cv) -> int:
    return cv.curve.params.subgroup_order


def scalar_len(cv) -> int:
    return -(-_order(cv).bit_length() // 8)


def point_len(cv) -> int:
    return cv.curve.params.encoding.point_len


def enc_64(value: int) -> bytes:
    if value < 0 or value >= _U64:
        raise ValueError("value does not fit in uint64")
    return value.to_bytes(8, "little")


def enc_scalar(cv, value: int) -> bytes:
    return (int(value) % _order(cv)).to_bytes(scalar_len(cv), "little")


def dec_scalar_mod(cv, value: bytes) -> int:
    """Any length, reduced mod the group order (nonces, challenges, seeds)."""
    return int.from_bytes(value, "little") % _order(cv)


def dec_scalar(cv, value: bytes) -> int:
    """Exactly scalar_len bytes and canonical (below the group order)."""
    width = scalar_len(cv)
    if len(value) != width:
        raise ValueError(f"scalar must be exactly {width} bytes")
    k = int.from_bytes(value, "little")
    if k >= _order(cv):
        raise ValueError("scalar is not canonical")
    return k


def enc_point(point) -> bytes:
    return point.point_to_string()


def dec_points(cv, values) -> list:
    """dec_point for several encodings at once; raises ValueError if ANY of them is malformed, the identity or outside
    the prime-order subgroup."""
    from .. import runtime

    width = point_len(cv)
    blobs = []
    for v in values:
        v = bytes(v)
        if len(v) != width:
            raise ValueError(f"point must be exactly {width} bytes")
        blobs.append(v)
    if not blobs:
        return []
    xy, flags = runtime.context().bsn_decode_points(b"".join(blobs), cv.curve.params.curve_id)
    if 0 in flags:
        # the reference reports the first bad point: "Invalid point encoding" when it does not decompress
        # (point.py:176-205), otherwise the subgroup message of dec_point
        cv.point_type.string_to_point(blobs[bytes(flags).index(0)])
        raise ValueError("point is not a valid nonidentity subgroup point")
    make, le = cv.point_type._trusted, int.from_bytes
    return [make(le(xy[o : o + 32], "little"), le(xy[o + 32 : o + 64], "little")) for o in range(0, 64 * len(blobs), 64)]


def dec_point(cv, value: bytes):
    (point,) = dec_points(cv, (value,))
    return point
