"""ORACLE (test infrastructure only) — Tiny and Pedersen VRF-AD over Bandersnatch, plain Python.

Restates:
  transcript / nonce / challenge / delinearisation   dot_ring/vrf/primitives.py:26-174
  domain separators                                   dot_ring/vrf/domain.py:6-17
  Tiny VRF prove / verify / codec                     dot_ring/vrf/ietf/tiny.py:35-88
  Pedersen VRF prove / verify / codec                 dot_ring/vrf/pedersen/vrf.py:44-169
Proof objects here are plain dicts / tuples; only the byte encodings are compared with the KATs.
"""
from __future__ import annotations

from . import bandersnatch as bsn

TINY, THIN, PEDERSEN = 0x00, 0x01, 0x02
NONCE_EXPAND, NONCE, PEDERSEN_BLINDING = 0x10, 0x11, 0x12
POINT_TO_HASH, DELINEARIZE, CHALLENGE, BATCH_VERIFY, HASH_TO_CURVE = 0x20, 0x30, 0x40, 0x50, 0x60
CHALLENGE_LEN = 16


def _squeeze_stream(suite: bsn.Suite, absorbed: bytes, size: int) -> bytes:
    # primitives.py:165 — XOF: digest(size); hash: seed=H(absorbed), blocks H(seed || LE64(ctr))
    h = suite.hash_fn()
    if suite.xof:
        return h(absorbed).digest(size)
    seed = h(absorbed).digest()
    out = b""
    ctr = 0
    while len(out) < size:
        out += h(seed + ctr.to_bytes(8, "little")).digest()
        ctr += 1
    return out[:size]


class Transcript:
    """primitives.py:26 — append-only; squeezes are consecutive slices of one stream."""

    def __init__(self, suite: bsn.Suite, absorbed: bytes | None = None):
        self.suite = suite
        self.absorbed = bytearray(suite.suite_id if absorbed is None else absorbed)
        self.squeezed = False
        self.offset = 0

    def fork(self) -> "Transcript":
        t = Transcript(self.suite, bytes(self.absorbed))
        t.squeezed, t.offset = self.squeezed, self.offset
        return t

    def absorb(self, data: bytes) -> None:
        if self.squeezed:
            raise ValueError("cannot absorb after squeeze")
        self.absorbed += data

    def squeeze(self, size: int) -> bytes:
        self.squeezed = True
        out = _squeeze_stream(self.suite, bytes(self.absorbed), self.offset + size)[self.offset :]
        self.offset += size
        return out


def nonce(suite, secret: int, transcript: Transcript | None = None) -> int:
    t = transcript.fork() if transcript is not None else Transcript(suite)
    t_exp = t.fork()
    t_exp.absorb(bytes([NONCE_EXPAND]))
    t_exp.absorb(bsn.enc_scalar(secret))
    secret_hash = t_exp.squeeze(64)
    t.absorb(bytes([NONCE]))
    t.absorb(secret_hash)
    k = bsn.dec_scalar_mod(t.squeeze((bsn.N.bit_length() + 128 + 7) // 8))
    if k == 0:
        raise ValueError("nonce scalar is zero")
    return k


def challenge(suite, points, transcript: Transcript | None = None) -> int:
    t = transcript.fork() if transcript is not None else Transcript(suite)
    t.absorb(bytes([CHALLENGE]))
    for pt in points:
        t.absorb(bsn.enc_point(pt))
    return bsn.dec_scalar_mod(t.squeeze(CHALLENGE_LEN))


def point_to_hash(suite, pt, size: int = 32) -> bytes:
    t = Transcript(suite)
    t.absorb(bytes([POINT_TO_HASH]))
    t.absorb(bsn.enc_point(pt))
    return t.squeeze(size)


def vrf_transcript(suite, scheme: int, ios, ad: bytes):
    """primitives.py:99 — returns (transcript, merged (input, output))."""
    t = Transcript(suite)
    t.absorb(bytes([scheme]))
    t.absorb(len(ios).to_bytes(8, "little"))
    for inp, out in ios:
        t.absorb(bsn.enc_point(inp) + bsn.enc_point(out))
    t.absorb(len(ad).to_bytes(8, "little"))
    t.absorb(ad)
    if not ios:
        return t, (bsn.IDENTITY, bsn.IDENTITY)
    if len(ios) == 1:
        return t, ios[0]
    d = t.fork()
    d.absorb(bytes([DELINEARIZE]))
    zs = [1] + [bsn.dec_scalar_mod(d.squeeze(CHALLENGE_LEN)) for _ in range(len(ios) - 1)]
    return t, (bsn.msm([io[0] for io in ios], zs), bsn.msm([io[1] for io in ios], zs))


def secret_from_seed(suite, seed: bytes):
    """curve.py:391 + primitives.py:147 -> (public_key_bytes, secret_key_bytes)."""
    if len(seed) != 32:
        raise ValueError("seed must be exactly 32 bytes")
    base = bsn.dec_scalar_mod(seed)
    counter = 0
    while True:
        t = Transcript(suite)
        t.absorb(seed)
        if counter:
            t.absorb(bytes([counter]))
        secret = nonce(suite, base, t)
        if secret != 0:
            break
        counter += 1
    sk = bsn.enc_scalar(secret)
    return bsn.public_key_from_secret(sk), sk


# ------------------------------------------------------------------ Tiny VRF
def tiny_prove(suite, alpha: bytes, sk: bytes, ad: bytes, salt: bytes = b"") -> bytes:
    x = bsn.dec_scalar_mod(sk)
    pk = bsn.mul(bsn.G, x)
    inp = bsn.encode_to_curve(suite, alpha, salt)
    out = bsn.mul(inp, x)
    t, merged = vrf_transcript(suite, TINY, [(bsn.G, pk), (inp, out)], ad)
    k = nonce(suite, x, t)
    r = bsn.mul(merged[0], k)
    c = challenge(suite, [r], t)
    s = (k + c * x) % bsn.N
    return bsn.enc_point(out) + c.to_bytes(CHALLENGE_LEN, "little") + bsn.enc_scalar(s)


def tiny_verify(suite, proof: bytes, pk: bytes, alpha: bytes, ad: bytes, salt: bytes = b"") -> bool:
    if len(proof) != 80:
        raise ValueError("invalid Tiny VRF proof length")
    out = bsn.dec_point(proof[:32])
    c = bsn.dec_scalar_mod(proof[32:48])
    s = bsn.dec_scalar(proof[48:])
    inp = bsn.encode_to_curve(suite, alpha, salt)
    pk_pt = bsn.dec_point(pk)
    t, merged = vrf_transcript(suite, TINY, [(bsn.G, pk_pt), (inp, out)], ad)
    r = bsn.msm([merged[0], merged[1]], [s, -c])
    return c == challenge(suite, [r], t)


# ------------------------------------------------------------------ Pedersen VRF
def pedersen_prove(suite, alpha: bytes, sk: bytes, ad: bytes, salt: bytes = b""):
    """Returns (proof_bytes(192), blinding_factor)."""
    x = bsn.dec_scalar_mod(sk)
    pk = bsn.mul(bsn.G, x)
    inp = bsn.encode_to_curve(suite, alpha, salt)
    out = bsn.mul(inp, x)
    t, merged = vrf_transcript(suite, PEDERSEN, [(inp, out)], ad)
    tb = t.fork()
    tb.absorb(bytes([PEDERSEN_BLINDING]))
    b = nonce(suite, x, tb)
    bb = suite.blinding_base
    blinded = bsn.add(pk, bsn.mul(bb, b))
    t.absorb(bsn.enc_point(blinded))
    k = nonce(suite, x, t)
    kb = nonce(suite, b, t)
    r = bsn.msm([bsn.G, bb], [k, kb])
    ok = bsn.mul(merged[0], k)
    c = challenge(suite, [r, ok], t)
    s = (k + c * x) % bsn.N
    sb = (kb + c * b) % bsn.N
    proof = b"".join(bsn.enc_point(q) for q in (out, blinded, r, ok)) + bsn.enc_scalar(s) + bsn.enc_scalar(sb)
    return proof, b


def pedersen_decode(proof: bytes):
    if len(proof) != 192:
        raise ValueError("invalid Pedersen VRF proof length")
    pts = [bsn.dec_point(proof[32 * i : 32 * i + 32]) for i in range(4)]
    return (*pts, bsn.dec_scalar(proof[128:160]), bsn.dec_scalar(proof[160:192]))


def pedersen_verify(suite, proof: bytes, alpha: bytes, ad: bytes, salt: bytes = b"") -> bool:
    out, blinded, r, ok, s, sb = pedersen_decode(proof)
    inp = bsn.encode_to_curve(suite, alpha, salt)
    t, merged = vrf_transcript(suite, PEDERSEN, [(inp, out)], ad)
    t.absorb(bsn.enc_point(blinded))
    c = challenge(suite, [r, ok], t)
    if bsn.msm([merged[0], merged[1]], [s, -c]) != ok:
        return False
    return bsn.msm([bsn.G, suite.blinding_base, blinded], [s, sb, -c]) == r
