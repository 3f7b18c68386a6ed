"""Column container (dot_ring/ring_proof/columns/columns.py:21-66)."""
from __future__ import annotations

import secrets
from dataclasses import dataclass

from .params import DEFAULT_DOMAIN_SIZE, ZK_ROWS
from .pcs import KZG
from .poly import inverse_fft


@dataclass
class Column:
    name: str
    evals: list
    coeffs: list | None = None
    _commitment: object = None
    size: int = DEFAULT_DOMAIN_SIZE
    _has_commitment: bool = False

    def __post_init__(self):
        if self._commitment is not None:
            self._has_commitment = True

    def pad(self, prime: int, hidden: bool = False, test_vectors: bool = False) -> None:
        """Zero-pad to the column size; hidden columns get ZK_ROWS random rows unless test_vectors (columns.py:29)."""
        if hidden and not test_vectors:
            capacity = self.size - ZK_ROWS
            if len(self.evals) > capacity:
                raise ValueError(f"{self.name} evals length {len(self.evals)} exceeds capacity {capacity} (size={self.size}, ZK_ROWS={ZK_ROWS})")
            self.evals += [0] * (capacity - len(self.evals))
            self.evals += [secrets.randbelow(prime) for _ in range(ZK_ROWS)]
        else:
            if len(self.evals) > self.size:
                raise ValueError(f"{self.name} evals length {len(self.evals)} exceeds column size {self.size}")
            self.evals += [0] * (self.size - len(self.evals))

    def interpolate(self, domain_omega: int, prime: int, hidden: bool = False, test_vectors: bool = False) -> None:
        if self.coeffs is None:
            self.pad(prime, hidden, test_vectors)
            self.coeffs = inverse_fft(self.evals, domain_omega, prime)

    def commit(self, pcs=KZG) -> None:
        if self.coeffs is None:
            raise ValueError("call interpolate() first")
        if not self._has_commitment:
            self.set_commitment(pcs.commit(self.coeffs))

    def set_commitment(self, commitment) -> None:
        self._commitment = commitment
        self._has_commitment = True

    @property
    def commitment(self):
        if not self._has_commitment:
            raise ValueError(f"{self.name} commitment is not set")
        return self._commitment
