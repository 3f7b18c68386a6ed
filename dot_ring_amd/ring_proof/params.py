"""RingProofParams (dot_ring/ring_proof/params.py:119-287) — same fields, defaults, validation and errors."""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import lru_cache

from ..curve import Bandersnatch, CurveVariant
from .pcs import KZG

ROOT_OF_UNITY_2048 = 49307615728544765012166121802278658070711169839041683575071795236746050763237
DEFAULT_DOMAIN_SIZE = 512
DEFAULT_MAX_RING_SIZE = 255
ZK_ROWS = 3
MAX_PIOP_DOMAIN_SIZE = 4096


def _is_power_of_two(n: int) -> bool:
    return n > 0 and n & (n - 1) == 0


def _sqrt_mod_prime(n: int, prime: int) -> int:
    """The square root the reference's Tonelli-Shanks returns (params.py:63-105) — WHICH of the two roots matters: three square roots of
    the shipped 2048-th root of unity fix omega for N = 4096, and with it the row order of the domain and every proof byte.
    Closed form of that loop: with prime - 1 = q 2^s, c = z^q for the smallest non-residue z and n^q = c^e (e even for a square), it
    returns n^((q+1)/2) c^((2^s - e)/2) — the root whose 2-power part has its logarithm below 2^(s-1); for e = 0 plainly n^((q+1)/2).
    e is read bit by bit in the group of order 2^s (s = 32 for the BLS12-381 scalar field)."""
    n %= prime
    if n == 0:
        return 0
    s = ((prime - 1) & -(prime - 1)).bit_length() - 1
    q = (prime - 1) >> s
    if s == 1:
        return pow(n, (prime + 1) // 4, prime)
    z = next(v for v in range(2, prime) if pow(v, (prime - 1) // 2, prime) == prime - 1)
    c, rest, e = pow(z, q, prime), pow(n, q, prime), 0
    c_inv = pow(c, -1, prime)
    for k in range(s):                        # bit k of e: (n^q c^-(bits below k))^(2^(s-1-k)) is -1 exactly when it is set
        if pow(rest, 1 << (s - 1 - k), prime) != 1:
            e |= 1 << k
            rest = rest * pow(c_inv, 1 << k, prime) % prime
    if e & 1:
        raise ValueError("No square root exists for provided value")
    root = pow(n, (q + 1) // 2, prime)
    return root * pow(c, ((1 << s) - e) // 2, prime) % prime if e else root


@lru_cache(maxsize=8)
def _extend_root_to_size(base_root: int, base_size: int, target_size: int, prime: int):
    root, size = base_root, base_size
    while size < target_size:
        root, size = _sqrt_mod_prime(root, prime), size * 2
    return root, size


@lru_cache(maxsize=32)
def _domain_for_size(size: int, prime: int, base_root: int, base_size: int):
    omega = pow(base_root, base_size // size, prime)
    out, cur = [], 1
    for _ in range(size):
        out.append(cur)
        cur = cur * omega % prime
    return tuple(out)


@dataclass
class RingProofParams:
    """Shape of the ring proof: PIOP domain N, the 4N evaluation domain, ring capacity, roots of unity, PCS and suite.
    Field names, defaults, derived properties and every error text follow the reference (params.py:119-287) — its
    tests construct this class directly and match on the messages; the checks themselves are organised as a list of
    (violated?, message) pairs evaluated in the reference's order."""
    domain_size: int = DEFAULT_DOMAIN_SIZE
    max_ring_size: int = DEFAULT_MAX_RING_SIZE
    padding_rows: int = 4
    radix_domain_size: int | None = None
    base_root: int = ROOT_OF_UNITY_2048
    base_root_size: int = 2048
    pcs: type = field(default=KZG, compare=False, hash=False, repr=False)
    test_vectors: bool = False
    cv: CurveVariant = field(default_factory=lambda: Bandersnatch, compare=False, hash=False)

    def __post_init__(self) -> None:
        aux = self.cv.curve.params.auxiliary_points
        missing = [n for n in ("blinding_base", "accumulator_base", "padding_point") if getattr(aux, n) is None]
        if missing:
            raise ValueError(f"{self.cv.name} ring proofs require auxiliary point {missing[0]}")
        n = self.domain_size
        radix = self.radix_domain_size = 4 * n if self.radix_domain_size is None else self.radix_domain_size
        p = self.prime

        def primitive_root_ok() -> bool:
            return pow(self.base_root, self.base_root_size, p) == 1 and pow(self.base_root, self.base_root_size // 2, p) != 1

        for violated, message in (
            (lambda: not _is_power_of_two(n), f"domain_size must be a power of two, got {n}"),
            (lambda: not _is_power_of_two(radix), f"radix_domain_size must be a power of two, got {radix}"),
            (lambda: radix % n != 0, f"domain_size {n} must divide radix_domain_size {radix}"),
            (lambda: n > MAX_PIOP_DOMAIN_SIZE, f"domain_size {n} exceeds supported SRS domain size {MAX_PIOP_DOMAIN_SIZE}"),
            (lambda: radix <= self.base_root_size and self.base_root_size % radix != 0,
             f"radix_domain_size {radix} must divide base_root_size {self.base_root_size}"),
            (lambda: not primitive_root_ok(), f"{self.cv.name} ring proofs require a primitive {self.base_root_size}-th root of unity"),
        ):
            if violated():
                raise ValueError(message)
        if radix > self.base_root_size:          # 4N beyond the shipped 2048-th root: take square roots (reference schedule)
            self.base_root, self.base_root_size = _extend_root_to_size(self.base_root, self.base_root_size, radix, p)
        if self.base_root_size % radix != 0:
            raise ValueError(f"radix_domain_size {radix} must divide base_root_size {self.base_root_size}")
        if self.padding_rows < 1:
            raise ValueError("padding_rows must be >= 1 to preserve accumulator structure")
        if self.padding_rows >= n:
            raise ValueError("padding_rows must be less than domain_size")
        if self.padding_rows != ZK_ROWS + 1:
            raise ValueError(f"padding_rows must be {ZK_ROWS + 1} to match the {ZK_ROWS} hidden rows")
        capacity = n - self.row_overhead
        if capacity <= 0:
            raise ValueError(
                "domain_size is too small for the scalar bit decomposition: "
                f"domain_size={n}, scalar_bits={self.scalar_bits}, padding_rows={self.padding_rows}")
        if self.max_ring_size == DEFAULT_MAX_RING_SIZE and capacity != DEFAULT_MAX_RING_SIZE:
            self.max_ring_size = capacity
        elif self.max_ring_size > capacity:
            raise ValueError(f"max_ring_size {self.max_ring_size} exceeds supported size {capacity}")

    # ---- derived quantities
    @property
    def prime(self) -> int:
        return self.cv.curve.params.field_modulus

    @property
    def scalar_bits(self) -> int:
        return self.cv.curve.params.subgroup_order.bit_length()

    @property
    def row_overhead(self) -> int:
        return self.scalar_bits + self.padding_rows

    def _root_of_order(self, size: int) -> int:
        return pow(self.base_root, self.base_root_size // size, self.prime)

    @property
    def omega(self) -> int:
        return self._root_of_order(self.domain_size)

    @property
    def radix_omega(self) -> int:
        return self._root_of_order(self.radix_domain_size)

    @property
    def domain(self) -> list:
        return list(_domain_for_size(self.domain_size, self.prime, self.base_root, self.base_root_size))

    @property
    def radix_domain(self) -> list:
        return list(_domain_for_size(self.radix_domain_size, self.prime, self.base_root, self.base_root_size))

    @property
    def radix_shift(self) -> int:
        return self.radix_domain_size // self.domain_size

    @property
    def last_index(self) -> int:
        return self.domain_size - self.padding_rows

    @property
    def max_effective_ring_size(self) -> int:
        return self.domain_size - self.row_overhead

    @property
    def required_srs_degree(self) -> int:
        return max(self.domain_size - 1, self.radix_domain_size - self.domain_size)

    @classmethod
    def from_ring_size(cls, ring_size: int, padding_rows: int = 4, base_root: int = ROOT_OF_UNITY_2048,
                       base_root_size: int = 2048, test_vectors: bool = False, cv: CurveVariant = Bandersnatch, pcs: type = KZG):
        """Smallest power-of-two domain with room for `ring_size` keys, the scalar bits and the padding rows."""
        if ring_size <= 0:
            raise ValueError(f"ring_size must be positive, got {ring_size}")
        overhead = cv.curve.params.subgroup_order.bit_length() + padding_rows
        domain_size = 1 << max(0, (ring_size + overhead - 1).bit_length())
        return cls(domain_size=domain_size, max_ring_size=domain_size - overhead, padding_rows=padding_rows,
                   base_root=base_root, base_root_size=base_root_size, test_vectors=test_vectors, cv=cv, pcs=pcs)
