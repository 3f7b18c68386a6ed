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
    """Tonelli-Shanks with the reference's schedule (params.py:63) — the root it returns fixes omega for N = 4096."""
    if n == 0:
        return 0
    if prime % 4 == 3:
        return pow(n, (prime + 1) // 4, prime)
    if pow(n, (prime - 1) // 2, prime) != 1:
        raise ValueError("No square root exists for provided value")
    q, s = prime - 1, 0
    while q % 2 == 0:
        s, q = s + 1, q // 2
    z = 2
    while pow(z, (prime - 1) // 2, prime) != prime - 1:
        z += 1
    m, c, x, t = s, pow(z, q, prime), pow(n, (q + 1) // 2, prime), pow(n, q, prime)
    while t != 1:
        i, probe = 1, t * t % prime
        while i < m and probe != 1:
            probe = probe * probe % prime
            i += 1
        b = pow(c, 1 << (m - i - 1), prime)
        x, t, c, m = x * b % prime, t * b * b % prime, b * b % prime, i
    return x


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
        for name in ("blinding_base", "accumulator_base", "padding_point"):
            if getattr(aux, name) is None:
                raise ValueError(f"{self.cv.name} ring proofs require auxiliary point {name}")
        if self.radix_domain_size is None:
            self.radix_domain_size = self.domain_size * 4
        radix = self.radix_domain_size
        if not _is_power_of_two(self.domain_size):
            raise ValueError(f"domain_size must be a power of two, got {self.domain_size}")
        if not _is_power_of_two(radix):
            raise ValueError(f"radix_domain_size must be a power of two, got {radix}")
        if radix % self.domain_size != 0:
            raise ValueError(f"domain_size {self.domain_size} must divide radix_domain_size {radix}")
        if self.domain_size > MAX_PIOP_DOMAIN_SIZE:
            raise ValueError(f"domain_size {self.domain_size} exceeds supported SRS domain size {MAX_PIOP_DOMAIN_SIZE}")
        if self.base_root_size % radix != 0 and radix <= self.base_root_size:
            raise ValueError(f"radix_domain_size {radix} must divide base_root_size {self.base_root_size}")
        if pow(self.base_root, self.base_root_size, self.prime) != 1 or pow(self.base_root, self.base_root_size // 2, self.prime) == 1:
            raise ValueError(f"{self.cv.name} ring proofs require a primitive {self.base_root_size}-th root of unity")
        if radix > self.base_root_size:
            self.base_root, self.base_root_size = _extend_root_to_size(self.base_root, self.base_root_size, radix, self.prime)
        if self.base_root_size % radix != 0:
            raise ValueError(f"radix_domain_size {radix} must divide base_root_size {self.base_root_size}")
        if self.padding_rows < 1:
            raise ValueError("padding_rows must be >= 1 to preserve accumulator structure")
        if self.padding_rows >= self.domain_size:
            raise ValueError("padding_rows must be less than domain_size")
        if self.padding_rows != ZK_ROWS + 1:
            raise ValueError(f"padding_rows must be {ZK_ROWS + 1} to match the {ZK_ROWS} hidden rows")
        max_supported = self.domain_size - self.row_overhead
        if max_supported <= 0:
            raise ValueError(
                "domain_size is too small for the scalar bit decomposition: "
                f"domain_size={self.domain_size}, scalar_bits={self.scalar_bits}, padding_rows={self.padding_rows}")
        if self.max_ring_size == DEFAULT_MAX_RING_SIZE and max_supported != DEFAULT_MAX_RING_SIZE:
            self.max_ring_size = max_supported
        elif self.max_ring_size > max_supported:
            raise ValueError(f"max_ring_size {self.max_ring_size} exceeds supported size {max_supported}")

    @property
    def prime(self) -> int:
        return self.cv.curve.params.field_modulus

    @property
    def scalar_bits(self) -> int:
        return self.cv.curve.params.subgroup_order.bit_length()

    @property
    def row_overhead(self) -> int:
        return self.scalar_bits + self.padding_rows

    @property
    def omega(self) -> int:
        return pow(self.base_root, self.base_root_size // self.domain_size, self.prime)

    @property
    def domain(self) -> list:
        return list(_domain_for_size(self.domain_size, self.prime, self.base_root, self.base_root_size))

    @property
    def radix_omega(self) -> int:
        return pow(self.base_root, self.base_root_size // self.radix_domain_size, self.prime)

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
        if ring_size <= 0:
            raise ValueError(f"ring_size must be positive, got {ring_size}")
        overhead = cv.curve.params.subgroup_order.bit_length() + padding_rows
        need = ring_size + overhead
        domain_size = 1
        while domain_size < need:
            domain_size *= 2
        return cls(domain_size=domain_size, max_ring_size=domain_size - overhead, padding_rows=padding_rows,
                   base_root=base_root, base_root_size=base_root_size, test_vectors=test_vectors, cv=cv, pcs=pcs)
