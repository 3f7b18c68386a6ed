"""KZG over BLS12-381 with the G1 work on the GPU (seam B).

Mirrors dot_ring/ring_proof/pcs/{kzg,srs,utils,opening,protocol}.py: the class below satisfies the reference's PCS
protocol (pcs/protocol.py:10-40) and is what RingProofParams.pcs selects.  A commitment is an opaque value to the
layers above — here the 96-byte affine record x||y (big-endian) or None for the point at infinity.
Pairings stay on the host (dr_pairing_check), G1 MSMs / commits run through dr_g1_msm*.
"""
from __future__ import annotations

import hashlib
import os
import secrets
from dataclasses import dataclass
from functools import lru_cache
from typing import Any, NamedTuple

from .. import _native, runtime

SCALAR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


class PcsVerification(NamedTuple):
    commitment: Any
    proof: Any
    point: int
    value: int


class LinearPcsVerification(NamedTuple):
    commitment_terms: tuple
    proof: Any
    point: int
    value: int


@dataclass(frozen=True)
class Opening:
    proof: Any
    y: int


def synthetic_div_with_eval(poly, x):
    """Quotient by (X - x) and f(x) in one Horner pass (pcs/utils.py:27)."""
    q = [0] * (len(poly) - 1)
    rem = poly[-1]
    for i in range(len(poly) - 2, -1, -1):
        q[i] = rem
        rem = (rem * x + poly[i]) % SCALAR_MODULUS
    return q, rem


def _candidate_srs_files(min_g1_count: int):
    paths = []
    env = os.environ.get("DOT_RING_BLS12_381_SRS")
    if env:
        paths.append(env)
    paths.append(os.path.join(_DATA, "bls12-381-srs-2-11-uncompressed-zcash.bin"))
    paths.append(os.path.join(_DATA, "bls12-381-srs-2-16-uncompressed-zcash.bin"))
    usable = []
    for path in paths:
        if not os.path.exists(path):
            continue
        with open(path, "rb") as f:
            header = f.read(8)
        if len(header) == 8 and int.from_bytes(header, "little") >= min_g1_count:
            usable.append(path)
    return usable


def _precompute_tables(dev) -> None:
    """Fixed-base tables of an SRS in HBM: 12-bit windows feeding the bucket pipeline — for an SRS of up to ~16 k points a row per bit,
    over which batches of hundreds of MSMs recode their scalars in non-adjacent form (dr_srs_table_info).  DOTRING_SRS_WINDOW overrides
    the width (0 = no table at all)."""
    explicit = os.environ.get("DOTRING_SRS_WINDOW")
    bits = int(explicit) if explicit is not None else 12
    if bits:
        dev.precompute(bits)


class SRS:
    """G1 powers (resident in HBM once used) + the two G2 points (pcs/srs.py:42-148)."""

    def __init__(self, g1_raw: bytes, g2_raw: list):
        self.g1_raw = g1_raw                                   # count * 96 bytes, BE x||y
        self.g2_raw = g2_raw                                   # two 192-byte records in file byte order
        self.count = len(g1_raw) // 96
        self._devices = {}          # id(ctx) -> (ctx, device handle): one upload per context/thread

    @property
    def g1(self):
        return self.g1_points

    @property
    def g1_points(self):
        return _PointList(self.g1_raw)

    @property
    def g2_points(self):
        out = []
        for rec in self.g2_raw:
            x0, x1, y0, y1 = (int.from_bytes(rec[48 * i : 48 * i + 48], "big") for i in range(4))
            out.append(((x1, x0), (y1, y0)))
        return out

    def device(self) -> _native.Srs:
        """The SRS in HBM, with its fixed-base window table (DOTRING_SRS_WINDOW bits per window, default 12; 0 = none)."""
        ctx = runtime.context()
        hit = self._devices.get(id(ctx))
        if hit is None or hit[0] is not ctx:
            dev = ctx.srs_load(self.g1_raw)
            _precompute_tables(dev)
            hit = (ctx, dev)
            self._devices[id(ctx)] = hit
        return hit[1]

    @classmethod
    def from_loaded(cls, max_deg: int) -> "SRS":
        need = max_deg + 1
        files = _candidate_srs_files(need)
        if not files:
            raise ValueError(f"no BLS12-381 SRS file with at least {need} G1 points is available")
        with open(files[0], "rb") as f:
            blob = f.read()
        g1_count = int.from_bytes(blob[:8], "little")
        take = min(need, g1_count)
        g1_raw = blob[8 : 8 + 96 * take]
        if len(g1_raw) != 96 * take:
            raise ValueError("Unexpected end-of-file when reading G1 points.")
        off = 8 + 96 * g1_count
        g2_count_raw = blob[off : off + 8]
        if len(g2_count_raw) < 8:
            raise ValueError("File too short to contain G2 vector length header.")
        if int.from_bytes(g2_count_raw, "little") < 2:
            raise ValueError("SRS file must contain at least two G2 points")
        g2_raw = [blob[off + 8 + 192 * i : off + 8 + 192 * (i + 1)] for i in range(2)]
        if any(len(r) != 192 for r in g2_raw):
            raise ValueError("Unexpected end-of-file when reading G2 points.")
        return cls(g1_raw, g2_raw)

    @classmethod
    def synthetic(cls, tau: int, count: int) -> "SRS":
        """Known-tau SRS [tau^i]G1, [1, tau]G2 for domains the shipped file does not cover (domain 4096 needs 12289
        points, the file holds 6145: SURVEY R5).  tau is public, so this is for tests and benchmarks only; G1 powers
        are generated on the GPU, tau*G2 on the host.  Generators = point 0 of the shipped SRS."""
        base = cls.default()
        tau %= SCALAR_MODULUS
        ctx = runtime.context()
        dev = ctx.srs_powers(base.g1_raw[:96], tau, count)
        self = cls(dev.download(0, count), [base.g2_raw[0], _native.g2_mul(base.g2_raw[0], tau)])
        _precompute_tables(dev)
        self._devices[id(ctx)] = (ctx, dev)
        return self

    @staticmethod
    @lru_cache(maxsize=2)
    def default(max_deg: int = 6144) -> "SRS":
        return SRS.from_loaded(max_deg)


class _PointList:
    """Lazy list of (x, y) int pairs over the raw SRS bytes."""

    def __init__(self, raw: bytes):
        self._raw = raw

    def __len__(self):
        return len(self._raw) // 96

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        rec = self._raw[96 * i : 96 * i + 96]
        if len(rec) != 96:
            raise IndexError(i)
        return int.from_bytes(rec[:48], "big"), int.from_bytes(rec[48:], "big")


def _to_record(point) -> bytes | None:
    """Accept a 96-byte record, None, or an (x, y) int pair."""
    if point is None or isinstance(point, (bytes, bytearray)):
        return None if point is None else bytes(point)
    x, y = point
    return int(x).to_bytes(48, "big") + int(y).to_bytes(48, "big")


def _random_nonzero_coefficients(count: int, order: int):
    """kzg.py:84 — first coefficient 1, the rest from a secrets-seeded SHAKE256 stream, rejection-sampled."""
    if count <= 0:
        return []
    coeffs = [1]
    byte_len = (order.bit_length() + 7) // 8
    limit = (1 << (8 * byte_len)) - ((1 << (8 * byte_len)) % order)
    seed, counter = secrets.token_bytes(32), 0
    while len(coeffs) < count:
        raw = hashlib.shake_256(seed + counter.to_bytes(8, "little")).digest(byte_len * (count - len(coeffs)) * 2)
        counter += 1
        for off in range(0, len(raw), byte_len):
            cand = int.from_bytes(raw[off : off + byte_len], "big")
            if cand < limit and cand % order:
                coeffs.append(cand % order)
                if len(coeffs) == count:
                    break
    return coeffs


class KZG:
    commitment_size = 48
    scalar_modulus = SCALAR_MODULUS
    srs = None            # set lazily (the reference loads at import; here loading is deferred to first use)

    @classmethod
    def with_srs(cls, srs: SRS) -> type:
        """A PCS class bound to its own SRS (pass as RingProofParams(pcs=...)); KZG itself keeps the shipped one."""
        return type("KZG", (cls,), {"srs": srs})

    @classmethod
    def _srs(cls) -> SRS:
        if cls.srs is None:
            cls.srs = SRS.default()
        return cls.srs

    @classmethod
    def ensure_srs_size(cls, max_degree: int) -> None:
        if max_degree >= cls._srs().count:
            cls.srs = SRS.default(max_degree)

    # ---- encodings
    @staticmethod
    def normalize_g1(point):
        rec = _to_record(point)
        if rec is None:
            raise ValueError("point at infinity has no affine coordinates")
        return int.from_bytes(rec[:48], "big"), int.from_bytes(rec[48:], "big")

    @classmethod
    def compress_g1(cls, point) -> bytes:
        return _native.g1_compress(_to_record(point))

    @classmethod
    def serialize_g1_uncompressed(cls, point) -> bytes:
        rec = _to_record(point)
        return rec if rec is not None else b"\x40" + bytes(95)

    @classmethod
    def decompress_g1(cls, data: bytes):
        if len(data) != cls.commitment_size:
            raise ValueError(f"invalid BLS12-381 G1 length: expected {cls.commitment_size}, got {len(data)}")
        return _native.g1_decompress(bytes(data))

    # ---- G1 work on the GPU
    @classmethod
    def msm_g1(cls, points, scalars):
        recs = b"".join(bytes(96) if (r := _to_record(p)) is None else r for p in points)
        ks = b"".join((int(s) % SCALAR_MODULUS).to_bytes(32, "little") for s in scalars)
        return runtime.context().g1_msm_points(recs, ks)

    @classmethod
    def commit(cls, coeffs):
        if len(coeffs) > 0:
            cls.ensure_srs_size(len(coeffs) - 1)
        srs = cls._srs()
        if len(coeffs) > srs.count:
            raise ValueError("polynomial degree exceeds SRS size")
        if not any(coeffs):
            return None
        ks = b"".join((int(c) % SCALAR_MODULUS).to_bytes(32, "little") for c in coeffs)
        return runtime.context().g1_msm(srs.device(), ks)

    @classmethod
    def commit_batch(cls, polys):
        """Commit several polynomials of EQUAL length in one batched MSM (same bases, one launch chain)."""
        if not polys:
            return []
        n = len(polys[0])
        if any(len(p) != n for p in polys):
            raise ValueError("commit_batch needs polynomials of equal length")
        if n == 0:
            return [None] * len(polys)
        cls.ensure_srs_size(n - 1)
        srs = cls._srs()
        if n > srs.count:
            raise ValueError("polynomial degree exceeds SRS size")
        ks = b"".join((int(c) % SCALAR_MODULUS).to_bytes(32, "little") for p in polys for c in p)
        return runtime.context().g1_msm_batch(srs.device(), ks, n)

    @classmethod
    def open(cls, coeffs, x) -> Opening:
        q, y = synthetic_div_with_eval(coeffs, x)
        return Opening(cls.commit(q), y)

    # ---- verification: fold on the GPU, two Miller loops + final exponentiation on the host
    @classmethod
    def _pairing_equal(cls, lhs, rhs) -> bool:
        """e(lhs, [1]G2) == e(rhs, [tau]G2)"""
        g2 = cls._srs().g2_raw
        return _native.pairing_check([(lhs, g2[0]), (_native.g1_neg(rhs), g2[1])])

    @classmethod
    def verify(cls, commitment, proof, point, value) -> bool:
        # e(C - [v]G1 + [z]proof, G2) == e(proof, tau G2)
        g1 = cls._srs().g1_raw[:96]
        lhs = cls.msm_g1([commitment, g1, proof], [1, -value, point])
        return cls._pairing_equal(lhs, _to_record(proof))

    @classmethod
    def batch_verify(cls, verifications) -> bool:
        if not verifications:
            return True
        if len(verifications) == 1:
            return cls.verify(*verifications[0])
        linear = [LinearPcsVerification(((v[0], 1),), v[1], v[2], v[3]) for v in verifications]
        return cls.batch_verify_linear_preconverted(linear)

    @classmethod
    def batch_verify_linear_preconverted(cls, verifications) -> bool:
        """kzg.py:304 — random linear combination of all claims, two MSMs, one pairing equation."""
        if not verifications:
            return True
        order = SCALAR_MODULUS
        coeffs = _random_nonzero_coefficients(len(verifications), order)
        lhs, rhs = {}, {}

        def add(table, point, scalar):
            scalar %= order
            if scalar == 0:
                return
            key = _to_record(point)
            table[key] = (table.get(key, 0) + scalar) % order

        sum_v = 0
        for coeff, ver in zip(coeffs, verifications):
            for commitment, scalar in ver.commitment_terms:
                add(lhs, commitment, coeff * scalar)
            sum_v = (sum_v + coeff * ver.value) % order
            add(lhs, ver.proof, coeff * ver.point)
            add(rhs, ver.proof, coeff)
        add(lhs, cls._srs().g1_raw[:96], -sum_v)
        lhs_items = [(k, s) for k, s in lhs.items() if s and k is not None]
        rhs_items = [(k, s) for k, s in rhs.items() if s and k is not None]
        lhs_point = cls.msm_g1([k for k, _ in lhs_items], [s for _, s in lhs_items])
        rhs_point = cls.msm_g1([k for k, _ in rhs_items], [s for _, s in rhs_items])
        return cls._pairing_equal(lhs_point, rhs_point)
