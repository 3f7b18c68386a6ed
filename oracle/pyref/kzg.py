"""ORACLE (test infrastructure only) — BLS12-381 G1 codecs, SRS file reader and KZG commit/open.

Restates the behaviour at these reference call sites (the G1 arithmetic itself is in the un-vendored
blst fork, see oracle/c/oracle.c header):
  SRS file layout / loader        dot_ring/ring_proof/pcs/srs.py:42-90
  commit (all-zero -> infinity)   dot_ring/ring_proof/pcs/kzg.py:152-175
  open = synthetic division + commit   dot_ring/ring_proof/pcs/kzg.py:178-192, pcs/utils.py:27-35
  compress / serialize / decompress    dot_ring/ring_proof/pcs/kzg.py:129-144 (zcash encoding produced by blst)
A G1 point is an affine int pair (x, y) or None for infinity.
"""
from __future__ import annotations

import os

from .. import coracle

FP = coracle.FP_P
FR = coracle.FR_P
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
DEFAULT_SRS = os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "..", "dot_ring_amd", "data", "bls12-381-srs-2-11-uncompressed-zcash.bin"
)


class SRS:
    """g1: list of affine int pairs; g1_raw_le: packed LE bytes for the C oracle; g2_raw: the two 192-byte G2 records."""

    def __init__(self, path: str = DEFAULT_SRS):
        with open(path, "rb") as f:
            blob = f.read()
        g1_count = int.from_bytes(blob[:8], "little")
        self.g1 = []
        for i in range(g1_count):
            rec = blob[8 + 96 * i : 8 + 96 * (i + 1)]
            if len(rec) != 96:
                raise ValueError(f"Unexpected end-of-file when reading G1 point {i}.")
            self.g1.append((int.from_bytes(rec[:48], "big"), int.from_bytes(rec[48:], "big")))
        off = 8 + 96 * g1_count
        g2_count = int.from_bytes(blob[off : off + 8], "little")
        if g2_count < 2:
            raise ValueError("SRS file must contain at least two G2 points")
        self.g2_raw = [blob[off + 8 + 192 * i : off + 8 + 192 * (i + 1)] for i in range(2)]
        self.g1_raw_le = coracle.g1_pack(self.g1)

    @classmethod
    def from_tau(cls, tau: int, count: int) -> "SRS":
        """Synthetic known-tau SRS [tau^i]G1 (for sizes the shipped file does not cover; no G2 part)."""
        self = cls.__new__(cls)
        self.g1 = []
        t = 1
        for _ in range(count):
            self.g1.append(coracle.g1_mul(G1_GEN, t))
            t = t * tau % FR
        self.g2_raw = [b"", b""]
        self.g1_raw_le = coracle.g1_pack(self.g1)
        return self


_default = None


def default_srs() -> SRS:
    global _default
    if _default is None:
        _default = SRS()
    return _default


# ------------------------------------------------------------------ encodings (zcash / blst)
def serialize(pt) -> bytes:
    """96 bytes: BE x || BE y; infinity = 0x40 then zeros."""
    if pt is None:
        return b"\x40" + bytes(95)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def compress(pt) -> bytes:
    """48 bytes BE x with flags: 0x80 compressed, 0x40 infinity, 0x20 y is the larger root."""
    if pt is None:
        return b"\xc0" + bytes(47)
    x, y = pt
    raw = bytearray(x.to_bytes(48, "big"))
    raw[0] |= 0x80
    if y > FP - y:
        raw[0] |= 0x20
    return bytes(raw)


def decompress(data: bytes):
    if len(data) != 48:
        raise ValueError(f"invalid BLS12-381 G1 length: expected 48, got {len(data)}")
    flags = data[0] >> 5
    if not flags & 4:
        raise ValueError("invalid BLS12-381 G1 encoding")
    x = int.from_bytes(data, "big") & ((1 << 381) - 1)
    if flags & 2:
        if x != 0 or flags & 1:
            raise ValueError("invalid BLS12-381 G1 encoding")
        return None
    if x >= FP:
        raise ValueError("invalid BLS12-381 G1 encoding")
    y = coracle.g1_recover_y(x, bool(flags & 1))
    if y is None:
        raise ValueError("invalid BLS12-381 G1 encoding")
    return x, y


def deserialize(data: bytes):
    if len(data) != 96:
        raise ValueError("invalid BLS12-381 G1 length")
    if data[0] & 0x40:
        return None
    pt = (int.from_bytes(data[:48], "big"), int.from_bytes(data[48:], "big"))
    if not coracle.g1_on_curve(pt):
        raise ValueError("invalid BLS12-381 G1 encoding")
    return pt


# ------------------------------------------------------------------ KZG
def commit(coeffs, srs: SRS | None = None):
    srs = srs or default_srs()
    n = len(coeffs)
    if n > len(srs.g1):
        raise ValueError("polynomial degree exceeds SRS size")
    if not any(c % FR for c in coeffs):
        return None
    ks = b"".join((c % FR).to_bytes(32, "little") for c in coeffs)
    return coracle.g1_unpack1(coracle.g1_msm_raw(srs.g1_raw_le[: 96 * n], ks, n))


def synthetic_div(poly, x: int):
    """pcs/utils.py:27 — quotient by (X - x) and the value f(x)."""
    q = [0] * (len(poly) - 1)
    rem = poly[-1] % FR
    for i in range(len(poly) - 2, -1, -1):
        q[i] = rem
        rem = (rem * x + poly[i]) % FR
    return q, rem


def open_at(coeffs, x: int, srs: SRS | None = None):
    q, y = synthetic_div(coeffs, x)
    return commit(q, srs), y


def msm(points, scalars):
    return coracle.g1_msm(points, [s % FR for s in scalars])
