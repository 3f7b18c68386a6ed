"""Thin VRF (dot_ring/vrf/ietf/thin.py:38-152).  Envelope: gamma || R || s.  Same seams as Tiny VRF (SURVEY §2:
"rides for free once boundary A is native"): scalar multiplications and the batch-verify MSM run on the GPU."""
from __future__ import annotations

import os
from dataclasses import dataclass

from .. import _native, runtime
from ..curve import msm_groups, scalar_mul_batch
from .base import VRF
from .codec import dec_point, dec_points, dec_scalar, dec_scalar_mod, enc_point, enc_scalar, point_len, scalar_len
from .primitives import (CHALLENGE_LEN, DomSep, VrfIo, challenge, nonce, point_to_hash, squeeze_transcript_bytes, vrf_transcript,
                         vrf_transcript_scalars)


@dataclass
class ThinVRF(VRF):
    output_point: object
    r: object
    s: int

    @classmethod
    def decode(cls, proof_bytes: bytes) -> "ThinVRF":
        pl, sl = point_len(cls.cv), scalar_len(cls.cv)
        expected = 2 * pl + sl
        if len(proof_bytes) != expected:
            raise ValueError(f"invalid Thin VRF proof length: expected {expected}, got {len(proof_bytes)}")
        out, r = dec_points(cls.cv, [proof_bytes[:pl], proof_bytes[pl : 2 * pl]])
        return cls(out, r, dec_scalar(cls.cv, proof_bytes[2 * pl :]))

    def encode(self) -> bytes:
        return enc_point(self.output_point) + enc_point(self.r) + enc_scalar(self.cv, self.s)

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
            out, frm, mk, plen = [], int.from_bytes, cv.point_type._trusted, 96
            ctx = runtime.context()
            for lo in range(0, count, 65536):
                hi = min(count, lo + 65536)
                blob, aux = ctx.ietf_prove_batch(suite, True, [bytes(a) for a in alphas[lo:hi]], [bytes(a) for a in additional_data[lo:hi]],
                                                 salts[lo:hi] if salts else None, sks[32 * lo : 32 * hi])
                for k in range(hi - lo):
                    raw, a = blob[plen * k : plen * k + plen], aux[128 * k : 128 * k + 128]
                    o = mk(frm(a[0:32], "little"), frm(a[32:64], "little"))
                    r = mk(frm(a[64:96], "little"), frm(a[96:128], "little"))
                    out.append(cls(o, r, frm(raw[64:96], "little")))
            return out
        gen = cv.point_type.generator_point()
        xs = [dec_scalar_mod(cv, sk) for sk in secret_keys]
        inputs = cv.point_type.encode_to_curve_batch(alphas, salts)
        firsts = scalar_mul_batch([gen] * count + inputs, xs + xs)
        pks, outs = firsts[:count], firsts[count:]
        # transcripts + delinearisation scalars on the host, then ONE grouped launch for the merged inputs of all proofs
        # (vrf_transcript would launch once per proof): merged.input_i = 1*G + z_i*I_i
        transcripts, pts, zs_all = [], [], []
        for i in range(count):
            t, zs = vrf_transcript_scalars(cv, DomSep.THIN_VRF, [VrfIo(gen, pks[i]), VrfIo(inputs[i], outs[i])], additional_data[i])
            transcripts.append(t)
            pts += [gen, inputs[i]]
            zs_all += zs
        merged_in = msm_groups(pts, zs_all, 2)
        ks = [nonce(cv, x, t) for x, t in zip(xs, transcripts)]
        rs = scalar_mul_batch(merged_in, ks)
        order = cv.curve.params.subgroup_order
        return [cls(outs[i], rs[i], (ks[i] + challenge(cv, [rs[i]], transcripts[i]) * xs[i]) % order) for i in range(count)]

    @classmethod
    def prove(cls, alpha: bytes, secret_key: bytes, additional_data: bytes, salt: bytes = b"") -> "ThinVRF":
        return cls.prove_batch([alpha], [secret_key], [additional_data], [salt])[0]

    def verify(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes = b"") -> bool:
        cv = self.cv
        input_point = cv.point_type.encode_to_curve(input, salt)
        try:
            pk = dec_point(cv, public_key)
        except ValueError as exc:
            raise ValueError("Invalid public key") from exc
        transcript, merged = vrf_transcript(cv, DomSep.THIN_VRF, [VrfIo(cv.point_type.generator_point(), pk), VrfIo(input_point, self.output_point)],
                                            additional_data)
        c = challenge(cv, [self.r], transcript)
        return cv.point_type.msm([merged.input, merged.output], [self.s, -c]) == self.r

    @classmethod
    def proof_to_hash(cls, gamma, mul_cofactor: bool = False) -> bytes:
        if mul_cofactor:
            gamma = gamma.double().double()
        return point_to_hash(cls.cv, gamma)

    @classmethod
    def batch_verify(cls, proofs, public_keys, inputs, additional_data, salts=None) -> bool:
        """thin.py:108 — one 5B-point MSM on the GPU."""
        cv = cls.cv
        if salts is None:
            salts = [b""] * len(proofs)
        items = []
        try:
            if not (len(proofs) == len(public_keys) == len(inputs) == len(additional_data) == len(salts)):
                raise ValueError("batch arguments must have equal lengths")
            input_points = cv.point_type.encode_to_curve_batch(list(inputs), list(salts))
            pk_points = dec_points(cv, list(public_keys)) if public_keys else []
            gen = cv.point_type.generator_point()
            for proof, pk, ipt, ad in zip(proofs, pk_points, input_points, additional_data):
                ios = [VrfIo(gen, pk), VrfIo(ipt, proof.output_point)]
                transcript, zs = vrf_transcript_scalars(cv, DomSep.THIN_VRF, ios, ad)
                items.append((challenge(cv, [proof.r], transcript), ios, zs, proof.r, proof.s))
        except (AttributeError, TypeError, ValueError):
            return False
        if not items:
            return True
        absorbed = bytearray(cv.curve.params.suite_id)
        absorbed.append(DomSep.BATCH_VERIFY)
        for c, _, _, _, s in items:
            absorbed += enc_scalar(cv, c) + enc_scalar(cv, s)
        raw = squeeze_transcript_bytes(cv.curve.params.hash_fn, bytes(absorbed), CHALLENGE_LEN * len(items))
        points, scalars = [], []
        for index, (c, ios, zs, r, s) in enumerate(items):
            coeff = dec_scalar_mod(cv, raw[CHALLENGE_LEN * index : CHALLENGE_LEN * (index + 1)])
            for io, z in zip(ios, zs):
                points += [io.input, io.output]
                scalars += [coeff * s * z, -(coeff * c * z)]
            points.append(r)
            scalars.append(-coeff)
        return cv.point_type.msm(points, scalars).is_identity()
