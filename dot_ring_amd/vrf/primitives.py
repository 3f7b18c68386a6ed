"""Host-side sigma-protocol plumbing for the single-proof API: domain separators, the append-only VRF transcript, nonce
and challenge derivation, delinearisation of several (input, output) pairs.  Behaviour follows
dot_ring/vrf/primitives.py:26-174 and vrf/domain.py:6-17 (the KATs pin every byte); the batch entry points run the same
logic natively (csrc/hostproto.hpp)."""
from __future__ import annotations

from enum import IntEnum
from typing import NamedTuple

from .codec import dec_scalar_mod, enc_64, enc_point, enc_scalar

SECURITY_PARAMETER = 128
CHALLENGE_LEN = SECURITY_PARAMETER // 8
_XOF_NAMES = frozenset(("shake_128", "shake_256"))


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


class VrfIo(NamedTuple):
    """One (input point, output point) pair of a VRF statement."""
    input: object
    output: object

    def encode(self) -> bytes:
        return enc_point(self.input) + enc_point(self.output)


def squeeze_transcript_bytes(hash_fn, absorbed: bytes, size: int) -> bytes:
    """The first `size` bytes of the stream an absorbed byte string defines: the XOF output for SHAKE suites, else
    H(seed || LE64(0)) || H(seed || LE64(1)) || ... with seed = H(absorbed)."""
    if hash_fn().name in _XOF_NAMES:
        return hash_fn(absorbed).digest(size)
    seed = hash_fn(absorbed).digest()
    stream, counter = bytearray(), 0
    while len(stream) < size:
        stream += hash_fn(seed + counter.to_bytes(8, "little")).digest()
        counter += 1
    return bytes(stream[:size])


class VrfTranscript:
    """Append-only byte log; successive squeezes return consecutive slices of the one stream the log defines, and the
    log is frozen by the first squeeze."""
    __slots__ = ("_hash_fn", "_log", "_taken")

    def __init__(self, label: bytes, hash_fn):
        self._hash_fn, self._log, self._taken = hash_fn, bytearray(label), None

    def copy(self) -> "VrfTranscript":
        twin = VrfTranscript(self._log, self._hash_fn)
        twin._taken = self._taken
        return twin

    def absorb(self, data: bytes) -> None:
        if self._taken is not None:
            raise ValueError("cannot absorb after squeeze")
        self._log += data

    def squeeze(self, size: int) -> bytes:
        start = self._taken or 0
        self._taken = start + size
        return squeeze_transcript_bytes(self._hash_fn, bytes(self._log), self._taken)[start:]


def new_transcript(cv) -> VrfTranscript:
    params = cv.curve.params
    return VrfTranscript(params.suite_id, params.hash_fn)


def _fork(cv, transcript, *chunks) -> VrfTranscript:
    t = new_transcript(cv) if transcript is None else transcript.copy()
    for chunk in chunks:
        t.absorb(chunk)
    return t


def nonce(cv, secret_scalar: int, transcript: VrfTranscript | None = None) -> int:
    """Deterministic nonce bound to the transcript: expand the secret under NONCE_EXPAND, absorb 64 expanded bytes under
    NONCE, reduce (order bits + 128) squeezed bits mod the order."""
    expanded = _fork(cv, transcript, bytes([DomSep.NONCE_EXPAND]), enc_scalar(cv, secret_scalar)).squeeze(64)
    width = (cv.curve.params.subgroup_order.bit_length() + SECURITY_PARAMETER + 7) // 8
    k = dec_scalar_mod(cv, _fork(cv, transcript, bytes([DomSep.NONCE]), expanded).squeeze(width))
    if k == 0:
        raise ValueError("nonce scalar is zero")
    return k


def challenge(cv, points, transcript: VrfTranscript | None = None) -> int:
    t = _fork(cv, transcript, bytes([DomSep.CHALLENGE]), *(enc_point(p) for p in points))
    return dec_scalar_mod(cv, t.squeeze(CHALLENGE_LEN))


def point_to_hash(cv, point, size: int = 32) -> bytes:
    return _fork(cv, None, bytes([DomSep.POINT_TO_HASH]), enc_point(point)).squeeze(size)


def vrf_transcript_scalars(cv, scheme, ios, ad: bytes):
    """Transcript of a statement (scheme tag, the io pairs, the additional data) and the delinearisation weights
    z_0 = 1, z_1.. = 128-bit squeezes under DELINEARIZE — host hashing only, no kernel launch."""
    ios = list(ios)
    t = _fork(cv, None, bytes([scheme]), enc_64(len(ios)), *(io.encode() for io in ios), enc_64(len(ad)), ad)
    if not ios:
        return t, []
    d = _fork(cv, t, bytes([DomSep.DELINEARIZE]))
    return t, [1] + [dec_scalar_mod(cv, d.squeeze(CHALLENGE_LEN)) for _ in ios[1:]]


def vrf_transcript(cv, scheme, ios, ad: bytes):
    """(transcript, merged VrfIo): sum_i z_i * input_i and sum_i z_i * output_i as one grouped-MSM launch."""
    ios = list(ios)
    t, zs = vrf_transcript_scalars(cv, scheme, ios, ad)
    if len(ios) == 1:
        return t, ios[0]
    if not ios:
        origin = cv.point_type.identity()
        return t, VrfIo(origin, origin)
    from ..curve import msm_groups

    merged_in, merged_out = msm_groups([io.input for io in ios] + [io.output for io in ios], zs + zs, len(ios))
    return t, VrfIo(merged_in, merged_out)


def secret_from_seed_scalar(cv, seed: bytes) -> int:
    """Secret scalar of a 32-byte seed: nonce of (seed mod n) over the transcript suite_id || seed [|| counter]; the
    counter only matters if a derived scalar is zero."""
    if len(seed) != 32:
        raise ValueError("seed must be exactly 32 bytes")
    base = dec_scalar_mod(cv, seed)
    for counter in range(256):
        t = _fork(cv, None, seed, bytes([counter]) if counter else b"")
        secret = nonce(cv, base, t)
        if secret:
            return secret
    raise RuntimeError("failed to derive non-zero secret scalar")
