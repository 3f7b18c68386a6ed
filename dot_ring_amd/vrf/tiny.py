"""Tiny VRF (dot_ring/vrf/ietf/tiny.py:26-88).  Envelope: gamma || c(16) || s."""
from __future__ import annotations

import os
from dataclasses import dataclass

from .. import _native, runtime
from ..curve import msm_groups, scalar_mul_batch
from .base import VRF
from .codec import dec_point, dec_scalar, dec_scalar_mod, enc_point, enc_scalar, point_len, scalar_len
from .primitives import CHALLENGE_LEN, DomSep, VrfIo, challenge, nonce, point_to_hash, vrf_transcript, vrf_transcript_scalars


@dataclass
class TinyVRF(VRF):
    output_point: object
    c: int
    s: int

    @classmethod
    def decode(cls, proof_bytes: bytes) -> "TinyVRF":
        pl, sl = point_len(cls.cv), scalar_len(cls.cv)
        expected = pl + CHALLENGE_LEN + sl
        if len(proof_bytes) != expected:
            raise ValueError(f"invalid Tiny VRF proof length: expected {expected}, got {len(proof_bytes)}")
        try:
            output_point = dec_point(cls.cv, proof_bytes[:pl])
        except ValueError as exc:
            raise ValueError("Invalid output point") from exc
        c = dec_scalar_mod(cls.cv, proof_bytes[pl : pl + CHALLENGE_LEN])
        s = dec_scalar(cls.cv, proof_bytes[pl + CHALLENGE_LEN :])
        return cls(output_point, c, s)

    def encode(self) -> bytes:
        return enc_point(self.output_point) + self.c.to_bytes(CHALLENGE_LEN, "little") + enc_scalar(self.cv, self.s)

    @classmethod
    def prove_batch(cls, alphas, secret_keys, additional_data, salts=None) -> list:
        """Additive API: element i equals prove(alphas[i], secret_keys[i], additional_data[i])."""
        cv = cls.cv
        count = len(alphas)
        if count and os.environ.get("DOTRING_NATIVE_HOST", "1") != "0":
            # one dr_ietf_prove_batch call: transcripts on the library's worker threads, four kernel launches in all
            if not (len(secret_keys) == len(additional_data) == count) or (salts is not None and len(salts) != count):
                raise ValueError("batch arguments must have equal lengths")
            sp = cv.curve.params
            order = sp.subgroup_order
            le = lambda v: int(v).to_bytes(32, "little")
            gen = sp.generator
            bb = sp.auxiliary_points.blinding_base or gen
            suite = _native.vrf_suite(sp.suite_id, sp.xof, le(gen[0]) + le(gen[1]), le(bb[0]) + le(bb[1]))
            sks = b"".join(bytes(sk) if len(sk) == 32 else le(int.from_bytes(sk, "little") % order) for sk in secret_keys)
            out, frm, mk, plen = [], int.from_bytes, cv.point_type._trusted, 80
            ctx = runtime.context()
            for lo in range(0, count, 65536):
                hi = min(count, lo + 65536)
                blob, aux = ctx.ietf_prove_batch(suite, False, [bytes(a) for a in alphas[lo:hi]], [bytes(a) for a in additional_data[lo:hi]],
                                                 salts[lo:hi] if salts else None, sks[32 * lo : 32 * hi])
                for k in range(hi - lo):
                    raw, a = blob[plen * k : plen * k + plen], aux[128 * k : 128 * k + 128]
                    o = mk(frm(a[0:32], "little"), frm(a[32:64], "little"))
                    out.append(cls(o, frm(raw[32:48], "little"), frm(raw[48:80], "little")))
            return out
        salts = salts or [b""] * count
        gen = cv.point_type.generator_point()
        xs = [dec_scalar_mod(cv, sk) for sk in secret_keys]
        inputs = cv.point_type.encode_to_curve_batch(alphas, salts)
        firsts = scalar_mul_batch([gen] * count + inputs, xs + xs)            # pk_i, O_i
        pks, outs = firsts[:count], firsts[count:]
        # transcripts + delinearisation scalars on the host, then ONE grouped launch for the merged inputs of all proofs
        # (vrf_transcript would launch once per proof): merged.input_i = 1*G + z_i*I_i
        transcripts, pts, zs_all = [], [], []
        for i in range(count):
            t, zs = vrf_transcript_scalars(cv, DomSep.TINY_VRF, [VrfIo(gen, pks[i]), VrfIo(inputs[i], outs[i])], additional_data[i])
            transcripts.append(t)
            pts += [gen, inputs[i]]
            zs_all += zs
        merged_in = msm_groups(pts, zs_all, 2)
        ks = [nonce(cv, x, t) for x, t in zip(xs, transcripts)]
        rs = scalar_mul_batch(merged_in, ks)
        order = cv.curve.params.subgroup_order
        proofs = []
        for i in range(count):
            c = challenge(cv, [rs[i]], transcripts[i])
            proofs.append(cls(outs[i], c, (ks[i] + c * xs[i]) % order))
        return proofs

    @classmethod
    def prove(cls, alpha: bytes, secret_key: bytes, additional_data: bytes, salt: bytes = b"") -> "TinyVRF":
        return cls.prove_batch([alpha], [secret_key], [additional_data], [salt])[0]

    def verify(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes = b"") -> bool:
        cv = self.cv
        input_point = cv.point_type.encode_to_curve(input, salt)
        try:
            public_key_pt = dec_point(cv, public_key)
        except ValueError as exc:
            raise ValueError("Invalid public key") from exc
        ios = [VrfIo(cv.point_type.generator_point(), public_key_pt), VrfIo(input_point, self.output_point)]
        transcript, merged = vrf_transcript(cv, DomSep.TINY_VRF, ios, additional_data)
        r = cv.point_type.msm([merged.input, merged.output], [self.s, -self.c])
        return self.c == challenge(cv, [r], transcript)

    @classmethod
    def proof_to_hash(cls, gamma, mul_cofactor: bool = False) -> bytes:
        if mul_cofactor:
            gamma = gamma.double().double()
        return point_to_hash(cls.cv, gamma)
