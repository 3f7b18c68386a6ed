"""ORACLE (test infrastructure only) — Bandersnatch (and JubJub) curve, codecs and hash-to-curve in plain Python ints.

Restates, for the two Bandersnatch suites and the JubJub suite (same base field; `with using(JUBJUB):` swaps the
module's curve constants — dot_ring/curve/specs/jubjub.py:17-56 — for the duration of a block):
  curve constants / suites          dot_ring/curve/specs/bandersnatch.py:57-144
  affine twisted-Edwards law        dot_ring/curve/twisted_edwards/te_affine_point.py:69-167
  compressed point codec            dot_ring/curve/point.py:150-214, te_affine_point.py:297-316
  subgroup validation (dec_point)   dot_ring/vrf/codec.py:39-45, dot_ring/curve/curve.py:56-67
  hash_to_field / expand_message    dot_ring/curve/curve.py:110-237
  Elligator2 map + Montgomery->TE   dot_ring/curve/twisted_edwards/te_curve.py:48-95, te_affine_point.py:212-295
  try-and-increment (JubJub)        dot_ring/curve/point.py:252-296
Scalar multiplications go through the C oracle (oracle/c/oracle.c), which tests check against the
affine law below.
"""
from __future__ import annotations

import hashlib
from contextlib import contextmanager
from dataclasses import dataclass

from .. import coracle

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
N = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1
COFACTOR = 4
A = -5 % P
D = 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7
G = (
    18886178867200960497001835917649091219057080094937609519140440539760939937304,
    19188667384257783945677642223292697773471335439753913231509108946878080696678,
)
IDENTITY = (0, 1)


@dataclass(frozen=True)
class Suite:
    name: str
    suite_id: bytes
    xof: bool                      # True: SHAKE128 transcript + expand_message_xof
    blinding_base: tuple
    accumulator_base: tuple
    padding_point: tuple
    curve: str = "bandersnatch"    # which constant set `using` installs
    e2c: str = "ell2"              # "ell2" (Elligator 2, RO) or "tai" (try and increment)

    @property
    def dst(self) -> bytes:
        return self.suite_id + b"\x60"

    def hash_fn(self):
        return hashlib.shake_128 if self.xof else hashlib.sha512


SHA512 = Suite(
    "Bandersnatch",
    b"Bandersnatch-SHA512-ELL2-v1",
    False,
    (23335687741101763108036518445642207119627658113885888016488710494487028845889,
     5552214580375038693022409684979828600325210968745774080859660443337357929963),
    (14056632001415368875257708737821299882600475929746323097150942355715730684350,
     10322661992765989500407719465917595459409463902187386706652408883505670839210),
    (26913883415342152801331916189968962157924271221160514298872262294143390094043,
     30874728313203001508631936119690348239461579770372782660098261717479009115354),
)
SHAKE128 = Suite(
    "Bandersnatch_SHAKE128",
    b"Bandersnatch-SHAKE128-ELL2-v1",
    True,
    (6153734995852631824944342602386415873379775188383988340041079006556670120775,
     27204351599954061630605768787803524395123895650061061132592995395630473050754),
    (27631238720955528589004064829276283990465032040945349648037876197995278250917,
     37605358688136619817560700742505556266961225274493904038881144193539047100140),
    (1834402953989431481748983728202937234471322740714585873803966488035889514523,
     52100941849053769665273763352270294131006971127418863694682093199651869272752),
)
JUBJUB = Suite(
    "JubJub",
    b"JubJub-SHA512-TAI-v1",
    False,
    (38206460563694846719174258613922853630278999941532690543235578292520143148532,
     34254498978062207918041301829525626783549813531091321004550549786528984401675),
    (48142684311216766702182564801462043940571084233680216669499475549492432046964,
     34380560660182334518990118617091967209302636551264477863958902286043397647879),
    (17348704025397475127937572481155408456556065464328870407269802701696798733683,
     24318278422173803457621119807961883607097742387673491974779969503617097905596),
    curve="jubjub",
    e2c="tai",
)

_CURVES = {
    "bandersnatch": dict(N=N, COFACTOR=COFACTOR, A=A, D=D, G=G),
    "jubjub": dict(                                            # specs/jubjub.py:17-29
        N=0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7,
        COFACTOR=8,
        A=-1 % P,
        D=19257038036680949359750312669786877991949435402254120286184196891950884077233,
        G=(8076246640662884909881801758704306714034609987455869804520522091855516602923,
           13262374693698910701929044844600465831413122818447359594527400194675274060458),
    ),
}
_ACTIVE = "bandersnatch"


@contextmanager
def using(suite: Suite):
    """Install the curve constants of `suite` as this module's N / COFACTOR / A / D / G for the duration of the block."""
    global _ACTIVE
    before = _ACTIVE
    globals().update(_CURVES[suite.curve])
    _ACTIVE = suite.curve
    try:
        yield suite
    finally:
        globals().update(_CURVES[before])
        _ACTIVE = before


# ------------------------------------------------------------------ group law (affine)
def on_curve(pt) -> bool:
    x, y = pt
    return (A * x * x + y * y) % P == (1 + D * x * x % P * y * y) % P


def add(p1, p2):
    """x3 = (x1y2+x2y1)/(1+d x1x2y1y2), y3 = (y1y2 - a x1x2)/(1 - d x1x2y1y2)."""
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 % P * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + x2 * y1) * pow(1 + t, -1, P) % P
    y3 = (y1 * y2 - A * x1 * x2) * pow(1 - t, -1, P) % P
    return x3, y3


def neg(pt):
    return (-pt[0]) % P, pt[1]


def sub(p1, p2):
    return add(p1, neg(p2))


def double(pt):
    return add(pt, pt)


def mul_py(pt, k: int):
    """Plain double-and-add on the affine law (slow; the ground truth the C oracle is checked against)."""
    acc = IDENTITY
    while k > 0:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def mul(pt, k: int):
    k %= N
    if k == 0 or pt == IDENTITY:
        return IDENTITY
    if _ACTIVE != "bandersnatch":          # the C oracle's TE kernels are the reference's: a = -5 only
        return mul_py(pt, k)
    return coracle.te_mul(pt, k, glv=False)


def msm(points, scalars):
    acc = IDENTITY
    for pt, k in zip(points, scalars, strict=True):
        acc = add(acc, mul(pt, k))
    return acc


# ------------------------------------------------------------------ field helpers
def sqrt(v: int):
    """A square root of v in Fr, or None (Tonelli-Shanks in the C oracle; squared back here)."""
    v %= P
    if v == 0:
        return 0
    r = coracle.fr_sqrt(v)
    if r is None:
        return None
    assert r * r % P == v
    return r


def is_square(v: int) -> bool:
    v %= P
    return v == 0 or pow(v, (P - 1) // 2, P) == 1


# ------------------------------------------------------------------ codecs
def enc_point(pt) -> bytes:
    x, y = pt
    raw = bytearray(y.to_bytes(32, "little"))
    if x > (-x) % P:
        raw[31] |= 0x80
    return bytes(raw)


def decompress(data: bytes):
    """point.py:176 string_to_point — raises ValueError on a bad encoding; no subgroup check."""
    if len(data) != 32:
        raise ValueError("point must be exactly 32 bytes")
    sign = data[31] >> 7
    raw = bytearray(data)
    raw[31] &= 0x7F
    y = int.from_bytes(raw, "little")
    if y >= P:
        raise ValueError("Invalid point encoding")
    den = (A - D * y * y) % P
    if den == 0:
        raise ValueError("Invalid point encoding")
    x2 = (1 - y * y) * pow(den, -1, P) % P
    x = sqrt(x2)
    if x is None:
        raise ValueError("Invalid point encoding")
    lo, hi = sorted((x, (-x) % P))
    pt = (hi if sign else lo, y)
    if pt != IDENTITY and not on_curve(pt):
        raise ValueError("Point is not on the curve")
    return pt


def in_prime_subgroup(pt) -> bool:
    """curve.py:56 valid_point — non-identity, on curve, [h]P != O and [h^-1 mod n][h]P == P."""
    if pt == IDENTITY or not on_curve(pt):
        return False
    cleared = mul_py(pt, COFACTOR)
    if cleared == IDENTITY:
        return False
    return mul(cleared, pow(COFACTOR, -1, N)) == pt


def dec_point(data: bytes):
    pt = decompress(data)
    if not in_prime_subgroup(pt):
        raise ValueError("point is not a valid nonidentity subgroup point")
    return pt


def enc_scalar(k: int) -> bytes:
    return (k % N).to_bytes(32, "little")


def dec_scalar(data: bytes) -> int:
    if len(data) != 32:
        raise ValueError("scalar must be exactly 32 bytes")
    k = int.from_bytes(data, "little")
    if k >= N:
        raise ValueError("scalar is not canonical")
    return k


def dec_scalar_mod(data: bytes) -> int:
    return int.from_bytes(data, "little") % N


# ------------------------------------------------------------------ hash to curve (Elligator2, RO variant)
def _expand_xmd(suite: Suite, msg: bytes, length: int) -> bytes:
    # curve.py:145 — note Z_pad is expand_len = 48 zero bytes (not SHA-512's 128-byte block)
    dst_prime = suite.dst + bytes([len(suite.dst)])
    msg_prime = bytes(48) + msg + length.to_bytes(2, "big") + b"\x00" + dst_prime
    b0 = hashlib.sha512(msg_prime).digest()
    blocks = [hashlib.sha512(b0 + b"\x01" + dst_prime).digest()]
    ell = -(-length // 64)
    for i in range(2, ell + 1):
        mixed = bytes(a ^ b for a, b in zip(b0, blocks[-1]))
        blocks.append(hashlib.sha512(mixed + bytes([i]) + dst_prime).digest())
    return b"".join(blocks)[:length]


def _expand_xof(suite: Suite, msg: bytes, length: int) -> bytes:
    # curve.py:201
    dst_prime = suite.dst + bytes([len(suite.dst)])
    return hashlib.shake_128(msg + length.to_bytes(2, "big") + dst_prime).digest(length)


def hash_to_field(suite: Suite, msg: bytes, count: int):
    length = count * 48
    raw = (_expand_xof if suite.xof else _expand_xmd)(suite, msg, length)
    return [int.from_bytes(raw[48 * i : 48 * i + 48], "big") % P for i in range(count)]


_MONT_DEN_INV = pow((A - D) % P, -1, P)
MONT_A = 2 * (A + D) * _MONT_DEN_INV % P      # bandersnatch.py:39 elligator2_map_from_edwards
MONT_B = 4 * _MONT_DEN_INV % P
ELL2_Z = 5


def map_to_curve_ell2(u: int):
    """te_curve.py:48 — Elligator 2 onto the Montgomery model, scaled by B."""
    a_over_b = MONT_A * pow(MONT_B, -1, P) % P
    inv_b2 = pow(MONT_B * MONT_B % P, -1, P)
    tv1 = ELL2_Z * u * u % P
    if tv1 == P - 1:
        tv1 = 0
    x1 = -a_over_b * pow(tv1 + 1, -1, P) % P
    gx1 = ((x1 + a_over_b) * x1 + inv_b2) * x1 % P
    x2 = (-x1 - a_over_b) % P
    gx2 = tv1 * gx1 % P
    e2 = is_square(gx1)
    x, y2 = (x1, gx1) if e2 else (x2, gx2)
    y = sqrt(y2)
    if y is None:
        raise ValueError("No square root exists")
    if e2 ^ (y % 2 == 1):
        y = -y % P
    return x * MONT_B % P, y * MONT_B % P


def from_mont(s: int, t: int):
    """te_affine_point.py:268 — Montgomery (s,t) -> twisted Edwards (v,w)."""
    tv1 = (s + 1) % P
    tv2 = tv1 * t % P
    tv2 = pow(tv2, -1, P) if tv2 else 0
    v = tv2 * tv1 % P * s % P
    w = tv2 * t % P * (s - 1) % P
    if tv2 == 0:
        w = 1
    pt = (v, w)
    if pt != IDENTITY and not on_curve(pt):
        raise ValueError("Point is not on the curve")
    return pt


def _te_add_ref(p1, p2):
    """te_affine_point.py:69 __add__ semantics (identity / doubling shortcuts, doubling special cases)."""
    if p1 == IDENTITY:
        return p2
    if p2 == IDENTITY:
        return p1
    if p1 == p2:
        return _te_double_ref(p1)
    return add(p1, p2)


def _te_double_ref(pt):
    # te_affine_point.py:138
    x1, y1 = pt
    if y1 == 0:
        return IDENTITY
    dx = (A * x1 * x1 + y1 * y1) % P
    dy = (2 - A * x1 * x1 - y1 * y1) % P
    if dx == 0 or dy == 0:
        return IDENTITY
    return 2 * x1 * y1 * pow(dx, -1, P) % P, (y1 * y1 - A * x1 * x1) * pow(dy, -1, P) % P


def _squeeze(suite: Suite, absorbed: bytes, size: int) -> bytes:
    # vrf/primitives.py:165 (restated again in vrf.py, which imports this module)
    if suite.xof:
        return hashlib.shake_128(absorbed).digest(size)
    seed = hashlib.sha512(absorbed).digest()
    out, ctr = b"", 0
    while len(out) < size:
        out += hashlib.sha512(seed + ctr.to_bytes(8, "little")).digest()
        ctr += 1
    return out[:size]


def encode_to_curve_tai(suite: Suite, alpha: bytes, salt: bytes = b""):
    """point.py:252 — candidate_c = the first 32 squeezed bytes of suite_id || 0x60 || LE64(len data) || data || c, read
    as a compressed point (255 value bits + the sign bit); the first c whose candidate decompresses, times the
    cofactor (skipped if that is the identity)."""
    data = salt + alpha
    prefix = suite.suite_id + b"\x60" + len(data).to_bytes(8, "little") + data
    for counter in range(256):
        cand = bytearray(_squeeze(suite, prefix + bytes([counter]), 32))
        sign = cand[-1] & 0x80
        cand[-1] &= 0x7F          # shave = 256 - 255 bits
        cand[-1] |= sign
        try:
            pt = decompress(bytes(cand))
        except ValueError:
            continue
        pt = mul_py(pt, COFACTOR)
        if pt != IDENTITY:
            return pt
    raise ValueError("hash_to_curve_tai failed")


def encode_to_curve(suite: Suite, alpha: bytes, salt: bytes = b""):
    """te_affine_point.py:212 _e2c_ell2_ro: two field elements, two maps, add, clear cofactor (2 doublings)."""
    if suite.e2c == "tai":
        return encode_to_curve_tai(suite, alpha, salt)
    u0, u1 = hash_to_field(suite, salt + alpha, 2)
    q0 = from_mont(*map_to_curve_ell2(u0))
    q1 = from_mont(*map_to_curve_ell2(u1))
    r = _te_add_ref(q0, q1)
    for _ in range(2):
        r = _te_double_ref(r)
    return r


def public_key_from_secret(sk: bytes) -> bytes:
    """curve.py:384 — note: the raw little-endian integer, reduced only inside the scalar mul."""
    return enc_point(mul(G, int.from_bytes(sk, "little")))
