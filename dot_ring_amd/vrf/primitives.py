"""Transcript, nonce, challenge, delinearisation (dot_ring/vrf/primitives.py:26-174, domain.py:6-17)."""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum

from .codec import dec_scalar_mod, enc_64, enc_point, enc_scalar

SECURITY_PARAMETER = 128
CHALLENGE_LEN = SECURITY_PARAMETER // 8


class DomSep(IntEnum):
    TINY_VRF = 0x00
    THIN_VRF = 0x01
    PEDERSEN_VRF = 0x02
    NONCE_EXPAND = 0x10
    NONCE = 0x11
    PEDERSEN_BLINDING = 0x12
    POINT_TO_HASH = 0x20
    DELINEARIZE = 0x30
    CHALLENGE = 0x40
    BATCH_VERIFY = 0x50
    HASH_TO_CURVE = 0x60


@dataclass(frozen=True)
class VrfIo:
    input: object
    output: object

    def encode(self) -> bytes:
        return enc_point(self.input) + enc_point(self.output)


def squeeze_transcript_bytes(hash_fn, absorbed: bytes, size: int) -> bytes:
    if hash_fn().name in ("shake_128", "shake_256"):
        return hash_fn(absorbed).digest(size)
    seed = hash_fn(absorbed).digest()
    blocks = -(-size // len(seed))
    return b"".join(hash_fn(seed + i.to_bytes(8, "little")).digest() for i in range(blocks))[:size]


class VrfTranscript:
    """Append-only; squeezes are consecutive slices of one counter-mode stream; no absorb after squeeze."""

    def __init__(self, label: bytes, hash_fn):
        self._hash_fn = hash_fn
        self._absorbed = bytearray(label)
        self._sealed = False
        self._offset = 0

    def copy(self) -> "VrfTranscript":
        other = VrfTranscript(bytes(self._absorbed), self._hash_fn)
        other._sealed, other._offset = self._sealed, self._offset
        return other

    def absorb(self, data: bytes) -> None:
        if self._sealed:
            raise ValueError("cannot absorb after squeeze")
        self._absorbed += data

    def squeeze(self, size: int) -> bytes:
        self._sealed = True
        start = self._offset
        self._offset += size
        return squeeze_transcript_bytes(self._hash_fn, bytes(self._absorbed), self._offset)[start:]


def new_transcript(cv) -> VrfTranscript:
    return VrfTranscript(cv.curve.params.suite_id, cv.curve.params.hash_fn)


def nonce(cv, secret_scalar: int, transcript: VrfTranscript | None = None) -> int:
    t = transcript.copy() if transcript is not None else new_transcript(cv)
    t_exp = t.copy()
    t_exp.absorb(bytes([DomSep.NONCE_EXPAND]))
    t_exp.absorb(enc_scalar(cv, secret_scalar))
    t.absorb(bytes([DomSep.NONCE]))
    t.absorb(t_exp.squeeze(64))
    k = dec_scalar_mod(cv, t.squeeze((cv.curve.params.subgroup_order.bit_length() + SECURITY_PARAMETER + 7) // 8))
    if k == 0:
        raise ValueError("nonce scalar is zero")
    return k


def challenge(cv, points, transcript: VrfTranscript | None = None) -> int:
    t = transcript.copy() if transcript is not None else new_transcript(cv)
    t.absorb(bytes([DomSep.CHALLENGE]))
    for point in points:
        t.absorb(enc_point(point))
    return dec_scalar_mod(cv, t.squeeze(CHALLENGE_LEN))


def point_to_hash(cv, point, size: int = 32) -> bytes:
    t = new_transcript(cv)
    t.absorb(bytes([DomSep.POINT_TO_HASH]))
    t.absorb(enc_point(point))
    return t.squeeze(size)


def vrf_transcript_scalars(cv, scheme, ios, ad: bytes):
    t = new_transcript(cv)
    t.absorb(bytes([scheme]))
    t.absorb(enc_64(len(ios)))
    for io in ios:
        t.absorb(io.encode())
    t.absorb(enc_64(len(ad)))
    t.absorb(ad)
    zs = []
    if ios:
        d = t.copy()
        d.absorb(bytes([DomSep.DELINEARIZE]))
        zs = [1] + [dec_scalar_mod(cv, d.squeeze(CHALLENGE_LEN)) for _ in range(len(ios) - 1)]
    return t, zs


def vrf_transcript(cv, scheme, ios, ad: bytes):
    """Returns (transcript, merged VrfIo); the two delinearised sums are one grouped-MSM launch."""
    t, zs = vrf_transcript_scalars(cv, scheme, ios, ad)
    if not ios:
        zero = cv.point_type.identity()
        return t, VrfIo(zero, zero)
    if len(ios) == 1:
        return t, ios[0]
    from ..curve import msm_groups

    m = len(ios)
    merged = msm_groups([io.input for io in ios] + [io.output for io in ios], zs + zs, m)
    return t, VrfIo(merged[0], merged[1])


def secret_from_seed_scalar(cv, seed: bytes) -> int:
    if len(seed) != 32:
        raise ValueError("seed must be exactly 32 bytes")
    base_secret = dec_scalar_mod(cv, seed)
    for counter in range(256):
        t = new_transcript(cv)
        t.absorb(seed)
        if counter:
            t.absorb(bytes([counter]))
        secret = nonce(cv, base_secret, t)
        if secret != 0:
            return secret
    raise RuntimeError("failed to derive non-zero secret scalar")
