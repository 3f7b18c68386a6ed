"""Fiat-Shamir transcript of the ring proof and its phases
(dot_ring/ring_proof/transcript/transcript.py:21-136, phases.py:18-128).  Stays on the host (hashlib)."""
from __future__ import annotations

import hashlib
import math
import struct

_FOOTER = b"\x00\x00\x00\x09"


def _framed(label: bytes) -> bytes:
    return label + struct.pack(">I", len(label))


class FiatShamirTranscript:
    def __init__(self, modulus: int, initial: bytes):
        self.modulus = modulus
        self._shake = hashlib.shake_128()
        self._challenge_bytes = math.ceil((modulus.bit_length() + 128) / 8)
        self._shake.update(_framed(initial))

    def copy(self) -> "FiatShamirTranscript":
        clone = FiatShamirTranscript.__new__(FiatShamirTranscript)
        clone.modulus, clone._challenge_bytes = self.modulus, self._challenge_bytes
        clone._shake = self._shake.copy()
        return clone

    def label(self, lbl: bytes) -> None:
        self._shake.update(_framed(lbl))

    def absorb_labeled(self, label: bytes, data: bytes) -> None:
        if len(data) >= 1 << 31:
            raise ValueError("transcript items of 2 GiB or more are not supported")
        self._shake.update(_framed(label) + data + struct.pack(">I", len(data)))

    def challenges(self, label: bytes, n: int) -> list:
        if n <= 0:
            return []
        prefix = _framed(label) + b"challenge"
        out = []
        self._shake.update(prefix)
        for i in range(n):
            out.append(int.from_bytes(self._shake.digest(self._challenge_bytes), "big") % self.modulus)
            self._shake.update(_FOOTER if i == n - 1 else _FOOTER + prefix)
        return out

    def challenge(self, label: bytes) -> int:
        return self.challenges(label, 1)[0]


def _le32(v) -> bytes:
    return int(v).to_bytes(32, "little")


def serialize_instance(point) -> bytes:
    if point.x is None or point.y is None:
        raise ValueError("Cannot serialize identity point")
    return _le32(point.x) + _le32(point.y)


def phase1_alphas_after_vk(t: FiatShamirTranscript, result_point, witness_commitments: bytes):
    t.absorb_labeled(b"instance", serialize_instance(result_point))
    t.absorb_labeled(b"committed_cols", bytes(witness_commitments))
    return t, t.challenges(b"constraints_aggregation", 7)


def phase2_eval_point(t: FiatShamirTranscript, quotient_commitment: bytes):
    t.absorb_labeled(b"quotient", bytes(quotient_commitment))
    return t, t.challenge(b"evaluation_point")


def phase3_nu_vector(t: FiatShamirTranscript, evals, lin_eval) -> list:
    t.absorb_labeled(b"register_evaluations", b"".join(_le32(e) for e in evals))
    t.absorb_labeled(b"shifted_linearization_evaluation", _le32(lin_eval))
    return t.challenges(b"kzg_aggregation", 8)


def derive_challenges_after_vk(t, result_point, witness_commitments: bytes, quotient_commitment: bytes, evals, lin_eval):
    t = t.copy()
    t, alphas = phase1_alphas_after_vk(t, result_point, witness_commitments)
    t, zeta = phase2_eval_point(t, quotient_commitment)
    return t, alphas, zeta, phase3_nu_vector(t, evals, lin_eval)


def serialize_verifier_key(g1_record: bytes, g2_records, commitment_records) -> bytes:
    """G1[0] || G2[0] || G2[1] (file byte order) || the three fixed-column commitments (root.py:54-72)."""
    return bytes(g1_record) + b"".join(g2_records) + b"".join(commitment_records)
