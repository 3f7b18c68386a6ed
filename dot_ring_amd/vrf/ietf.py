"""What TinyVRF and ThinVRF share (dot_ring/vrf/ietf/tiny.py:26-88, thin.py:38-152): the statement
(G, pk), (I, O) with I = encode_to_curve(alpha), O = x*I; the transcript over both pairs; the delinearised input
M = G + z*I; the nonce k; R = k*M; the 128-bit challenge c over R; the response s = k + c*x.  Tiny publishes
(O, c, s), Thin publishes (O, R, s).  prove_batch is ONE native call (dr_ietf_prove_batch: transcripts on the
library's worker threads, four kernel launches in all); DOTRING_NATIVE_HOST=0 keeps the Python orchestration over
the same kernels."""
from __future__ import annotations

import os

from .. import _native, runtime
from ..curve import msm_groups, scalar_mul_batch
from .base import VRF
from .codec import dec_point, dec_scalar_mod
from .primitives import VrfIo, challenge, nonce, point_to_hash, vrf_transcript, vrf_transcript_scalars


class IetfVRF(VRF):
    """Mixin with the scheme-independent parts; subclasses set SCHEME (DomSep), THIN and _from_parts()."""
    SCHEME = None
    THIN = False

    @classmethod
    def _from_parts(cls, output_point, r_point, c: int, s: int):
        raise NotImplementedError

    @classmethod
    def _statement(cls, public_key_point, input_point, output_point):
        return [VrfIo(cls.cv.point_type.generator_point(), public_key_point), VrfIo(input_point, output_point)]

    # ---- proving
    @classmethod
    def prove_batch(cls, alphas, secret_keys, additional_data, salts=None) -> list:
        """Additive API: element i equals prove(alphas[i], secret_keys[i], additional_data[i])."""
        count = len(alphas)
        if not (len(secret_keys) == len(additional_data) == count) or (salts is not None and len(salts) != count):
            raise ValueError("batch arguments must have equal lengths")
        if count == 0:
            return []
        if os.environ.get("DOTRING_NATIVE_HOST", "1") == "0":
            return cls._prove_batch_python(alphas, secret_keys, additional_data, salts)
        cv = cls.cv
        sp = cv.curve.params
        le = lambda v: int(v).to_bytes(32, "little")
        gen = sp.generator
        bb = sp.auxiliary_points.blinding_base or gen                      # unused by these two schemes
        suite = _native.vrf_suite(sp.suite_id, sp.xof, le(gen[0]) + le(gen[1]), le(bb[0]) + le(bb[1]), sp.curve_id)
        sks = b"".join(bytes(sk) if len(sk) == 32 else le(int.from_bytes(sk, "little") % sp.subgroup_order) for sk in secret_keys)
        plen = 96 if cls.THIN else 80
        ctx, make, frm, out = runtime.context(), cv.point_type._trusted, int.from_bytes, []
        for lo in range(0, count, 65536):
            hi = min(count, lo + 65536)
            blob, aux = ctx.ietf_prove_batch(suite, cls.THIN, [bytes(a) for a in alphas[lo:hi]], [bytes(a) for a in additional_data[lo:hi]],
                                             salts[lo:hi] if salts else None, sks[32 * lo : 32 * hi])
            for k in range(hi - lo):
                raw, xy = blob[plen * k : plen * (k + 1)], aux[128 * k : 128 * (k + 1)]
                o = make(frm(xy[0:32], "little"), frm(xy[32:64], "little"))
                r = make(frm(xy[64:96], "little"), frm(xy[96:128], "little"))
                c = 0 if cls.THIN else frm(raw[32:48], "little")
                out.append(cls._from_parts(o, r, c, frm(raw[plen - 32 :], "little")))
        return out

    @classmethod
    def _prove_batch_python(cls, alphas, secret_keys, additional_data, salts) -> list:
        cv = cls.cv
        count = len(alphas)
        gen = cv.point_type.generator_point()
        xs = [dec_scalar_mod(cv, sk) for sk in secret_keys]
        inputs = cv.point_type.encode_to_curve_batch(alphas, salts or [b""] * count)
        both = scalar_mul_batch([gen] * count + inputs, xs + xs)               # pk_i = x_i*G, O_i = x_i*I_i: one launch
        pks, outs = both[:count], both[count:]
        # transcripts and delinearisation weights on the host, then ONE grouped launch for all merged inputs M_i = G + z_i*I_i
        transcripts, terms, weights = [], [], []
        for i in range(count):
            t, zs = vrf_transcript_scalars(cv, cls.SCHEME, cls._statement(pks[i], inputs[i], outs[i]), additional_data[i])
            transcripts.append(t)
            terms += [gen, inputs[i]]
            weights += zs
        merged = msm_groups(terms, weights, 2)
        ks = [nonce(cv, x, t) for x, t in zip(xs, transcripts)]
        rs = scalar_mul_batch(merged, ks)
        order = cv.curve.params.subgroup_order
        proofs = []
        for i in range(count):
            c = challenge(cv, [rs[i]], transcripts[i])
            proofs.append(cls._from_parts(outs[i], rs[i], c, (ks[i] + c * xs[i]) % order))
        return proofs

    @classmethod
    def prove(cls, alpha: bytes, secret_key: bytes, additional_data: bytes, salt: bytes = b""):
        return cls.prove_batch([alpha], [secret_key], [additional_data], [salt])[0]

    # ---- verifying (single proof; the relation is checked by the subclass)
    @classmethod
    def _suite_struct(cls):
        sp = cls.cv.curve.params
        le = lambda v: int(v).to_bytes(32, "little")
        gen = sp.generator
        bb = sp.auxiliary_points.blinding_base or gen
        return _native.vrf_suite(sp.suite_id, sp.xof, le(gen[0]) + le(gen[1]), le(bb[0]) + le(bb[1]), sp.curve_id)

    @classmethod
    def _small_host_serves(cls) -> bool:
        """one proof of an Elligator suite of Bandersnatch: the library checks it on a host core (dr_ietf_verify_batch, ~0.6 ms)
        instead of three kernel launch chains.  DOTRING_SMALL_HOST_MAX=0 / DOTRING_NATIVE_HOST=0 keep the kernels."""
        sp = cls.cv.curve.params
        return (sp.curve_id == _native.CURVE_BANDERSNATCH and sp.e2c != "tai" and os.environ.get("DOTRING_NATIVE_HOST", "1") != "0"
                and os.environ.get("DOTRING_SMALL_HOST_MAX", "64") != "0")

    def _verify_small(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes) -> bool:
        if len(public_key) != 32:
            raise ValueError("Invalid public key")
        verdict = runtime.context().ietf_verify_batch(self._suite_struct(), self.THIN, self.encode(), bytes(public_key), [bytes(input)],
                                                      [bytes(additional_data)], [bytes(salt)])[0]
        if verdict == 2:
            raise ValueError("Invalid public key")
        return verdict == 1

    def _verifier_view(self, public_key: bytes, input: bytes, additional_data: bytes, salt: bytes):
        """(transcript, merged io) of the statement this proof is about."""
        cv = self.cv
        try:
            pk = dec_point(cv, public_key)
        except ValueError as exc:
            raise ValueError("Invalid public key") from exc
        ios = self._statement(pk, cv.point_type.encode_to_curve(input, salt), self.output_point)
        return vrf_transcript(cv, self.SCHEME, ios, additional_data)

    @classmethod
    def proof_to_hash(cls, gamma, mul_cofactor: bool = False) -> bytes:
        if mul_cofactor:
            # gamma * cofactor (tiny.py:88, pedersen/vrf.py:167): 4 = two doublings on Bandersnatch, 8 = three on JubJub
            for _ in range(cls.cv.curve.params.cofactor.bit_length() - 1):
                gamma = gamma.double()
        return point_to_hash(cls.cv, gamma)
