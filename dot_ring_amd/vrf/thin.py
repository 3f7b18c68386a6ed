"""Thin VRF (dot_ring/vrf/ietf/thin.py:38-152): proof = O || R || s = 96 bytes; the verifier checks
s*M - c*(pk + z*O) = R with c recomputed from R, which also allows ONE 5B-point MSM for a batch.  Shared machinery in
ietf.py."""
from __future__ import annotations

from dataclasses import dataclass

from .codec import dec_points, dec_scalar, dec_scalar_mod, enc_point, enc_scalar, point_len, scalar_len
from .ietf import IetfVRF
from .primitives import CHALLENGE_LEN, DomSep, challenge, squeeze_transcript_bytes, vrf_transcript_scalars


@dataclass
class ThinVRF(IetfVRF):
    output_point: object
    r: object
    s: int

    SCHEME = DomSep.THIN_VRF
    THIN = True

    @classmethod
    def _from_parts(cls, output_point, r_point, c, s):
        return cls(output_point, r_point, s)

    def encode(self) -> bytes:
        return enc_point(self.output_point) + enc_point(self.r) + enc_scalar(self.cv, self.s)

    @classmethod
    def decode(cls, proof_bytes: bytes) -> "ThinVRF":
        cv = cls.cv
        pl = point_len(cv)
        want = 2 * pl + scalar_len(cv)
        if len(proof_bytes) != want:
            raise ValueError(f"invalid Thin VRF proof length: expected {want}, got {len(proof_bytes)}")
        gamma, r = dec_points(cv, (proof_bytes[:pl], proof_bytes[pl : 2 * pl]))          # one launch for both points
        return cls(gamma, r, dec_scalar(cv, proof_bytes[2 * pl :]))

    def verify(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes = b"") -> bool:
        if self._small_host_serves():
            return self._verify_small(public_key, input, additional_data, salt)
        transcript, merged = self._verifier_view(public_key, input, additional_data, salt)
        c = challenge(self.cv, [self.r], transcript)
        return self.cv.point_type.msm([merged.input, merged.output], [self.s, -c]) == self.r

    @classmethod
    def batch_verify(cls, proofs, public_keys, inputs, additional_data, salts=None) -> bool:
        """Random linear combination of all B relations (weights squeezed from the transcript of every (c, s)): one
        5B-point MSM that must vanish; public keys are decoded and inputs hashed to the curve in one launch each."""
        cv = cls.cv
        count = len(proofs)
        salts = [b""] * count if salts is None else salts
        try:
            if not (len(public_keys) == len(inputs) == len(additional_data) == len(salts) == count):
                return False
            if count == 0:
                return True
            input_points = cv.point_type.encode_to_curve_batch(list(inputs), list(salts))
            pk_points = dec_points(cv, list(public_keys))
            rows = []
            for proof, pk, ipt, ad in zip(proofs, pk_points, input_points, additional_data):
                ios = cls._statement(pk, ipt, proof.output_point)
                transcript, zs = vrf_transcript_scalars(cv, cls.SCHEME, ios, ad)
                rows.append((challenge(cv, [proof.r], transcript), ios, zs, proof))
        except (AttributeError, TypeError, ValueError):
            return False
        sp = cv.curve.params
        log = bytes(sp.suite_id) + bytes([DomSep.BATCH_VERIFY]) + b"".join(enc_scalar(cv, c) + enc_scalar(cv, pr.s) for c, _, _, pr in rows)
        stream = squeeze_transcript_bytes(sp.hash_fn, log, CHALLENGE_LEN * count)
        points, scalars = [], []
        for i, (c, ios, zs, pr) in enumerate(rows):
            w = dec_scalar_mod(cv, stream[CHALLENGE_LEN * i : CHALLENGE_LEN * (i + 1)])
            for io, z in zip(ios, zs):
                points += [io.input, io.output]
                scalars += [w * pr.s * z, -(w * c * z)]
            points.append(pr.r)
            scalars.append(-w)
        return cv.point_type.msm(points, scalars).is_identity()
