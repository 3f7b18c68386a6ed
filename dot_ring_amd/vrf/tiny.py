"""Tiny VRF (dot_ring/vrf/ietf/tiny.py:26-88): proof = O || c (16 bytes) || s = 80 bytes; the verifier recomputes
R = s*M - c*(pk + z*O) and compares challenges.  Shared machinery in ietf.py."""
from __future__ import annotations

from dataclasses import dataclass

from .codec import dec_point, dec_scalar, dec_scalar_mod, enc_point, enc_scalar, point_len, scalar_len
from .ietf import IetfVRF
from .primitives import CHALLENGE_LEN, DomSep, challenge


@dataclass
class TinyVRF(IetfVRF):
    output_point: object
    c: int
    s: int

    SCHEME = DomSep.TINY_VRF
    THIN = False

    @classmethod
    def _from_parts(cls, output_point, r_point, c, s):
        return cls(output_point, c, s)

    def encode(self) -> bytes:
        return enc_point(self.output_point) + self.c.to_bytes(CHALLENGE_LEN, "little") + enc_scalar(self.cv, self.s)

    @classmethod
    def decode(cls, proof_bytes: bytes) -> "TinyVRF":
        cv = cls.cv
        pl = point_len(cv)
        want = pl + CHALLENGE_LEN + scalar_len(cv)
        if len(proof_bytes) != want:
            raise ValueError(f"invalid Tiny VRF proof length: expected {want}, got {len(proof_bytes)}")
        try:
            gamma = dec_point(cv, proof_bytes[:pl])
        except ValueError as exc:
            raise ValueError("Invalid output point") from exc
        return cls(gamma, dec_scalar_mod(cv, proof_bytes[pl : pl + CHALLENGE_LEN]), dec_scalar(cv, proof_bytes[pl + CHALLENGE_LEN :]))

    def verify(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes = b"") -> bool:
        if self._small_host_serves():
            return self._verify_small(public_key, input, additional_data, salt)
        transcript, merged = self._verifier_view(public_key, input, additional_data, salt)
        r = self.cv.point_type.msm([merged.input, merged.output], [self.s, -self.c])
        return self.c == challenge(self.cv, [r], transcript)
