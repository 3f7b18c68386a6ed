"""Curve layer of the API mirror: the Bandersnatch suites (and JubJub), points, codecs, hash-to-curve.

Mirrors (names, argument meaning, error behaviour) the parts of the reference the Ring-VRF path touches:
  dot_ring/curve/specs/bandersnatch.py:57-306   suites, BandersnatchPoint.__mul__/msm, CurveVariant objects
  dot_ring/curve/specs/jubjub.py:17-66          JubJub: same field, a = -1, cofactor 8, try-and-increment
  dot_ring/curve/point.py:150-214               compressed codec
  dot_ring/curve/twisted_edwards/*              affine law, Elligator2 encode_to_curve
  dot_ring/curve/curve.py:56-67,110-237,384-401 valid_point, hash_to_field, key derivation
Scalar multiplications and MSMs run on the GPU (seam A of include/dotring_hip.h); single affine additions,
hashing and the Elligator map stay host-side big-int code exactly as they are in the reference.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Callable

from . import _native, runtime

FIELD_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
SUBGROUP_ORDER = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1
_P, _N = FIELD_MODULUS, SUBGROUP_ORDER
_A = -5
_D = 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7


@dataclass(frozen=True)
class AuxiliaryPoints:
    blinding_base: tuple
    accumulator_base: tuple
    padding_point: tuple


@dataclass(frozen=True)
class Encoding:
    endian: str = "little"
    point_len: int = 32
    challenge_len: int = 16
    uncompressed: bool = False


@dataclass(frozen=True)
class SuiteParams:
    """The fields of the reference's BandersnatchParams that callers read (bandersnatch.py:46-106)."""
    suite_id: bytes
    hash_fn: Callable
    auxiliary_points: AuxiliaryPoints
    xof: bool
    field_modulus: int = FIELD_MODULUS
    subgroup_order: int = SUBGROUP_ORDER
    cofactor: int = 4
    a: int = _A
    d: int = _D
    generator: tuple = (
        18886178867200960497001835917649091219057080094937609519140440539760939937304,
        19188667384257783945677642223292697773471335439753913231509108946878080696678,
    )
    encoding: Encoding = Encoding()
    curve_id: int = _native.CURVE_BANDERSNATCH      # DR_CURVE_* of include/dotring_hip.h
    e2c: str = "ell2"                               # "ell2" (Elligator 2, RO) or "tai" (try and increment)

    @property
    def h2c_dst(self) -> bytes:
        return self.suite_id + b"\x60"


class BandersnatchCurve:
    def __init__(self, params: SuiteParams):
        self.params = params

    # -- curve.py:110-237
    def hash_to_field(self, msg: bytes, count: int) -> list[int]:
        if count < 0:
            raise ValueError("Count must be non-negative")
        if msg is None:
            raise ValueError("Message cannot be None")
        length = 48 * count
        dst_prime = self.params.h2c_dst + bytes([len(self.params.h2c_dst)])
        if self.params.xof:
            raw = hashlib.shake_128(msg + length.to_bytes(2, "big") + dst_prime).digest(length)
        else:
            # expand_message_xmd with SHA-512; Z_pad is 48 zero bytes in this suite (bandersnatch.py:85, curve.py:170)
            b0 = hashlib.sha512(bytes(48) + msg + length.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
            blocks = [hashlib.sha512(b0 + b"\x01" + dst_prime).digest()]
            for i in range(2, -(-length // 64) + 1):
                blocks.append(hashlib.sha512(bytes(x ^ y for x, y in zip(b0, blocks[-1])) + bytes([i]) + dst_prime).digest())
            raw = b"".join(blocks)[:length]
        return [int.from_bytes(raw[48 * i : 48 * i + 48], "big") % _P for i in range(count)]

    def mod_sqrt(self, val: int) -> int:
        return _native.fr_sqrt(val % _P)        # raises ValueError("No square root exists")

    def is_square(self, val: int) -> bool:
        val %= _P
        return val == 0 or pow(val, (_P - 1) // 2, _P) == 1

    def valid_point(self, point: "BandersnatchPoint") -> bool:
        """Non-identity member of the prime-order subgroup (curve.py:56)."""
        return bool(valid_points([point])[0])


class BandersnatchPoint:
    """Affine twisted Edwards point; `curve` and the coefficient shortcuts are bound per suite by the subclasses below
    (the class keeps its Bandersnatch name: that is what the reference's callers import)."""
    curve: BandersnatchCurve
    _A, _D, _N, _H, _CV = _A, _D, _N, 4, _native.CURVE_BANDERSNATCH
    __slots__ = ("x", "y")

    def __init__(self, x: int, y: int):
        self.x, self.y = x, y
        if (x, y) != (0, 1):
            if not (0 <= x < _P and 0 <= y < _P):
                raise ValueError("Invalid point coordinates")
            if not self._on_curve(x, y):
                raise ValueError("Point is not on the curve")

    @classmethod
    def _on_curve(cls, x: int, y: int) -> bool:
        return (cls._A * x * x + y * y) % _P == (1 + cls._D * x * x % _P * y * y) % _P

    @classmethod
    def _trusted(cls, x: int, y: int):
        """Construct from coordinates that are already known to be on the curve (kernel outputs): skips the
        range / on-curve checks of __init__."""
        pt = object.__new__(cls)
        pt.x, pt.y = x, y
        return pt

    # -- basics
    def __eq__(self, other):
        return isinstance(other, BandersnatchPoint) and self.x == other.x and self.y == other.y

    def __hash__(self):
        return (self.x + self.y) % self._N

    def __repr__(self):
        return f"{type(self).__name__}({self.x}, {self.y})"

    @classmethod
    def identity(cls):
        return cls(0, 1)

    @classmethod
    def generator_point(cls):
        return cls(*cls.curve.params.generator)

    def is_identity(self) -> bool:
        return self.x == 0 and self.y == 1

    def is_on_curve(self) -> bool:
        return self._on_curve(self.x, self.y)

    # -- group law: single additions are host big-int code, as in te_affine_point.py:69-167
    def __add__(self, other):
        if not isinstance(other, BandersnatchPoint):
            raise TypeError("Can only add TEAffinePoints")
        if self.is_identity():
            return other
        if other.is_identity():
            return self
        if self == other:
            return self.double()
        x1, y1, x2, y2 = self.x, self.y, other.x, other.y
        a, d = self._A, self._D
        t = d * x1 % _P * x2 % _P * y1 % _P * y2 % _P
        return type(self)((x1 * y2 + x2 * y1) * pow(1 + t, -1, _P) % _P, (y1 * y2 - a * x1 * x2) * pow(1 - t, -1, _P) % _P)

    def double(self):
        x1, y1, a = self.x, self.y, self._A
        if y1 == 0:
            return self.identity()
        dx, dy = (a * x1 * x1 + y1 * y1) % _P, (2 - a * x1 * x1 - y1 * y1) % _P
        if dx == 0 or dy == 0:
            return self.identity()
        return type(self)(2 * x1 * y1 * pow(dx, -1, _P) % _P, (y1 * y1 - a * x1 * x1) * pow(dy, -1, _P) % _P)

    def __neg__(self):
        return type(self)(-self.x % _P, self.y)

    def __sub__(self, other):
        return self + (-other)

    # -- scalar multiplication / MSM on the GPU
    def __mul__(self, scalar: int):
        return scalar_mul_batch([self], [scalar])[0]

    __rmul__ = __mul__

    @classmethod
    def msm(cls, points, scalars):
        if len(points) != len(scalars):
            raise ValueError("Points and scalars must have same length")
        if not points:
            return cls.identity()
        raw = runtime.context().bsn_msm(pack_points(points), pack_scalars(scalars, cls._N), cls._CV)
        return cls(int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little"))

    # -- codec (point.py:150-214)
    def point_to_string(self) -> bytes:
        raw = bytearray(self.y.to_bytes(32, "little"))
        if self.x > -self.x % _P:
            raw[31] |= 0x80
        return bytes(raw)

    @classmethod
    def string_to_point(cls, octet_string: bytes):
        if not octet_string:
            raise ValueError("Empty octet string")
        sign = (octet_string[-1] >> 7) & 1
        raw = bytearray(octet_string)
        raw[-1] &= 0x7F
        y = int.from_bytes(raw, "little")
        if y >= _P:
            raise ValueError("Invalid point encoding")
        den = (cls._A - cls._D * y * y) % _P
        if den == 0:
            raise ValueError("Invalid point encoding")
        try:
            x = cls.curve.mod_sqrt((1 - y * y) * pow(den, -1, _P) % _P)
        except ValueError:
            raise ValueError("Invalid point encoding") from None
        lo, hi = sorted((x, -x % _P))
        return cls(hi if sign else lo, y)

    # -- hash to curve (te_affine_point.py:212-295, te_curve.py:48-95)
    @classmethod
    def _suite_struct(cls):
        """dr_vrf_suite of this point type's suite (cached on the class)."""
        st = cls.__dict__.get("_suite_cache")
        if st is None:
            sp = cls.curve.params
            le = lambda pt: pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")  # noqa: E731
            st = _native.vrf_suite(sp.suite_id, sp.xof, le(sp.generator), le(sp.auxiliary_points.blinding_base), sp.curve_id)
            cls._suite_cache = st
        return st

    @classmethod
    def encode_to_curve(cls, alpha_string: bytes, salt: bytes = b""):
        if cls.curve.params.e2c == "tai":          # point.py:252-296, through the batch entry point (hashing native, sqrt on the GPU)
            return cls.encode_to_curve_batch([alpha_string], [salt])[0]
        u0, u1 = cls.curve.hash_to_field(salt + alpha_string, 2)
        r = cls.map_to_curve(u0) + cls.map_to_curve(u1)
        return r.double().double()

    @classmethod
    def hash_to_field_pairs(cls, alpha_strings, salts=None) -> bytes:
        """Host half of encode_to_curve for many inputs: two field elements per input, packed little-endian."""
        salts = salts or [b""] * len(alpha_strings)
        return b"".join(u.to_bytes(32, "little") for a, s in zip(alpha_strings, salts) for u in cls.curve.hash_to_field(s + a, 2))

    @classmethod
    def encode_to_curve_from_field(cls, us: bytes):
        """Device half: Elligator2 maps, addition and cofactor clearing for packed (u0, u1) pairs."""
        if not us:
            return []
        return unpack_points(cls, runtime.context().bsn_encode_to_curve_batch(us))

    @classmethod
    def encode_to_curve_batch(cls, alpha_strings, salts=None):
        """encode_to_curve for many inputs: hash_to_field on the host, Elligator2 + cofactor clearing on the GPU; for a
        try-and-increment suite the candidates are hashed natively and decompressed + cofactor-cleared on the GPU."""
        if cls.curve.params.e2c == "tai":
            if not alpha_strings:
                return []
            return unpack_points(cls, runtime.context().encode_to_curve_batch(cls._suite_struct(), list(alpha_strings), salts))
        return cls.encode_to_curve_from_field(cls.hash_to_field_pairs(alpha_strings, salts))

    @classmethod
    def map_to_curve(cls, u: int):
        inv_den = pow((_A - _D) % _P, -1, _P)
        mont_a, mont_b = 2 * (_A + _D) * inv_den % _P, 4 * inv_den % _P
        a_over_b = mont_a * pow(mont_b, -1, _P) % _P
        inv_b2 = pow(mont_b * mont_b % _P, -1, _P)
        tv1 = 5 * u * u % _P
        if tv1 == _P - 1:
            tv1 = 0
        x1 = -a_over_b * pow(tv1 + 1, -1, _P) % _P
        gx1 = ((x1 + a_over_b) * x1 + inv_b2) * x1 % _P
        e2 = cls.curve.is_square(gx1)
        x, y2 = (x1, gx1) if e2 else ((-x1 - a_over_b) % _P, tv1 * gx1 % _P)
        y = cls.curve.mod_sqrt(y2)
        if e2 ^ (y % 2 == 1):
            y = -y % _P
        s, t = x * mont_b % _P, y * mont_b % _P
        # Montgomery (s,t) -> twisted Edwards (v,w)
        tv1 = (s + 1) % _P
        tv2 = tv1 * t % _P
        tv2 = pow(tv2, -1, _P) if tv2 else 0
        v, w = tv2 * tv1 % _P * s % _P, tv2 * t % _P * (s - 1) % _P
        return cls(v, 1 if tv2 == 0 else w)


# ------------------------------------------------------------------ batched helpers over the C ABI
def pack_points(points) -> bytes:
    return b"".join(p.x.to_bytes(32, "little") + p.y.to_bytes(32, "little") for p in points)


def pack_scalars(scalars, order: int = _N) -> bytes:
    return b"".join((int(s) % order).to_bytes(32, "little") for s in scalars)


def unpack_points(cls, raw: bytes):
    """Kernel outputs are group elements by construction: no per-point curve check."""
    frm, mk = int.from_bytes, cls._trusted
    return [mk(frm(raw[i : i + 32], "little"), frm(raw[i + 32 : i + 64], "little")) for i in range(0, len(raw), 64)]


def scalar_mul_batch(points, scalars):
    """[k_i * P_i] in one kernel launch."""
    if len(points) != len(scalars):
        raise ValueError("Points and scalars must have same length")
    if not points:
        return []
    cls = type(points[0])
    first = points[0]
    if all(p is first for p in points) and (first.x, first.y) in _fixed_bases(cls):
        # k_i * G (key derivation, curve.py:384) or k_i * B: the constant's fixed-base window table — 64 table additions
        # over four lanes instead of ~250 dependent doublings (dr_te_fixed_base_msm_groups)
        raw = runtime.context().te_fixed_base_msm_groups(pack_points([first]), pack_scalars(scalars, cls._N), cls._CV)
        return unpack_points(cls, raw)
    raw = runtime.context().bsn_scalar_mul_batch(pack_points(points), pack_scalars(scalars, cls._N), cls._CV)
    return unpack_points(cls, raw)


def _fixed_bases(cls):
    """the suite's constant points that get a fixed-base table: generator and Pedersen blinding base"""
    params = cls.curve.params
    bb = params.auxiliary_points.blinding_base
    return (tuple(params.generator),) + ((tuple(bb),) if bb else ())


def msm_groups(points, scalars, m: int):
    """[sum_{j<m} k_{g*m+j} * P_{g*m+j}] for consecutive groups of m terms, one launch."""
    if not points:
        return []
    cls = type(points[0])
    raw = runtime.context().bsn_msm_groups(pack_points(points), pack_scalars(scalars, cls._N), m, cls._CV)
    return unpack_points(cls, raw)


def valid_points(points) -> list[bool]:
    """curve.py:56 for a whole batch: [h]P != O and [h^-1 mod n][h]P == P (h the cofactor), one launch for all points."""
    live = [i for i, p in enumerate(points) if not p.is_identity() and p.is_on_curve()]
    out = [False] * len(points)
    if not live:
        return out
    h, order = type(points[live[0]])._H, type(points[live[0]])._N
    cleared = scalar_mul_batch_raw([points[i] for i in live], [h] * len(live))
    back = scalar_mul_batch([c for c in cleared], [pow(h, -1, order)] * len(live))
    for i, c, b in zip(live, cleared, back):
        out[i] = (not c.is_identity()) and b == points[i]
    return out


def scalar_mul_batch_raw(points, small_scalars):
    """Scalar multiplication WITHOUT reduction mod n for points that may lie outside the prime-order subgroup:
    the kernel reduces scalars mod n, which is only sound on the subgroup, so small cofactor multiples are done
    with host doublings (4P = two doublings, 8P = three, as te_affine_point.py:235 clear_cofactor)."""
    out = []
    for p, k in zip(points, small_scalars):
        if k != type(p)._H:
            raise ValueError("only the cofactor multiple is supported here")
        for _ in range(k.bit_length() - 1):
            p = p.double()
        out.append(p)
    return out


# ------------------------------------------------------------------ suites / curve variants
def _suite(name: str, suite_id: bytes, xof: bool, bb, ab, pp, **curve_consts):
    params = SuiteParams(suite_id=suite_id, hash_fn=hashlib.shake_128 if xof else hashlib.sha512,
                         auxiliary_points=AuxiliaryPoints(bb, ab, pp), xof=xof, **curve_consts)
    curve = BandersnatchCurve(params)
    point_type = type(f"{name}Point", (BandersnatchPoint,), {
        "curve": curve, "__slots__": (), "_A": params.a, "_D": params.d, "_N": params.subgroup_order, "_H": params.cofactor,
        "_CV": params.curve_id})
    return CurveVariant(name, curve, point_type)


class CurveVariant:
    """curve.py:353 — name, curve, point_type + key derivation."""

    def __init__(self, name, curve, point_type):
        self.name, self.curve, self.point_type = name, curve, point_type

    def point(self, x, y=None):
        if isinstance(x, BandersnatchPoint):
            return x
        if y is None:
            x, y = x
        return self.point_type(x, y)

    def public_key_from_secret(self, secret_key: bytes) -> bytes:
        if not isinstance(secret_key, (bytes, bytearray)):
            raise TypeError("secret_key must be bytes")
        return (self.point_type.generator_point() * int.from_bytes(secret_key, "little")).point_to_string()

    def secret_from_seed(self, seed: bytes):
        if not isinstance(seed, (bytes, bytearray)):
            raise TypeError("seed must be bytes")
        from .vrf.codec import enc_scalar
        from .vrf.primitives import secret_from_seed_scalar

        secret_key = enc_scalar(self, secret_from_seed_scalar(self, bytes(seed)))
        return self.public_key_from_secret(secret_key), secret_key


Bandersnatch = _suite(
    "Bandersnatch", b"Bandersnatch-SHA512-ELL2-v1", False,
    (23335687741101763108036518445642207119627658113885888016488710494487028845889,
     5552214580375038693022409684979828600325210968745774080859660443337357929963),
    (14056632001415368875257708737821299882600475929746323097150942355715730684350,
     10322661992765989500407719465917595459409463902187386706652408883505670839210),
    (26913883415342152801331916189968962157924271221160514298872262294143390094043,
     30874728313203001508631936119690348239461579770372782660098261717479009115354),
)
Bandersnatch_SHAKE128 = _suite(
    "Bandersnatch_SHAKE128", b"Bandersnatch-SHAKE128-ELL2-v1", True,
    (6153734995852631824944342602386415873379775188383988340041079006556670120775,
     27204351599954061630605768787803524395123895650061061132592995395630473050754),
    (27631238720955528589004064829276283990465032040945349648037876197995278250917,
     37605358688136619817560700742505556266961225274493904038881144193539047100140),
    (1834402953989431481748983728202937234471322740714585873803966488035889514523,
     52100941849053769665273763352270294131006971127418863694682093199651869272752),
)
# dot_ring/curve/specs/jubjub.py:17-66 — the other twisted Edwards curve over the BLS12-381 scalar field
JubJub = _suite(
    "JubJub", b"JubJub-SHA512-TAI-v1", False,
    (38206460563694846719174258613922853630278999941532690543235578292520143148532,
     34254498978062207918041301829525626783549813531091321004550549786528984401675),
    (48142684311216766702182564801462043940571084233680216669499475549492432046964,
     34380560660182334518990118617091967209302636551264477863958902286043397647879),
    (17348704025397475127937572481155408456556065464328870407269802701696798733683,
     24318278422173803457621119807961883607097742387673491974779969503617097905596),
    subgroup_order=0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7,
    cofactor=8,
    a=-1,
    d=19257038036680949359750312669786877991949435402254120286184196891950884077233,
    generator=(8076246640662884909881801758704306714034609987455869804520522091855516602923,
               13262374693698910701929044844600465831413122818447359594527400194675274060458),
    curve_id=_native.CURVE_JUBJUB,
    e2c="tai",
)
