"""Polynomial helpers of the ring prover: NTTs on the GPU (seam C), coefficient-space ops on the host.

Mirrors dot_ring/ring_proof/polynomial/fft.py:87-144 (inverse_fft / evaluate_poly_fft) and ops.py:51-224.
"""
from __future__ import annotations

from .. import runtime


def _pack(values, prime) -> bytes:
    return b"".join((int(v) % prime).to_bytes(32, "little") for v in values)


def _unpack(raw: bytes):
    return [int.from_bytes(raw[i : i + 32], "little") for i in range(0, len(raw), 32)]


def ntt_batch(vectors, omega: int, prime: int, scale: int | None = None):
    """Transform several equal-length vectors in one launch chain."""
    if not vectors:
        return []
    n = len(vectors[0])
    if n < 2 or n & (n - 1) or any(len(v) != n for v in vectors):
        raise ValueError(f"native NTT plan size must be a power of two >= 2, got {n}")
    raw = runtime.context().ntt(b"".join(_pack(v, prime) for v in vectors), n.bit_length() - 1, omega, scale)
    flat = _unpack(raw)
    return [flat[i * n : (i + 1) * n] for i in range(len(vectors))]


def inverse_fft(values, omega: int, prime: int):
    n = len(values)
    if n == 1:
        return [values[0] % prime]
    return ntt_batch([values], pow(omega, -1, prime), prime, pow(n, -1, prime))[0]


def inverse_fft_batch(vectors, omega: int, prime: int):
    n = len(vectors[0])
    return ntt_batch(vectors, pow(omega, -1, prime), prime, pow(n, -1, prime))


def fold(poly, size: int, prime: int):
    out = [0] * size
    for i, c in enumerate(poly):
        out[i % size] = (out[i % size] + c) % prime
    return out


def evaluate_poly_fft(poly, domain_size: int, omega: int, prime: int):
    return ntt_batch([fold(poly, domain_size, prime)], omega, prime)[0]


def evaluate_polys_fft(polys, domain_size: int, omega: int, prime: int):
    return ntt_batch([fold(p, domain_size, prime) for p in polys], omega, prime)


def poly_evaluate_single(poly, x: int, prime: int) -> int:
    acc = 0
    x %= prime
    for c in reversed(poly):
        acc = (acc * x + c) % prime
    return acc


def poly_mul_small(a, b, prime: int):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % prime
    return out


def poly_divide_by_vanishing(poly, domain_size: int, prime: int):
    """Quotient by X^N - 1 for a multiple of it: q_j = sum_{i>=1} p_{j+iN} (ops.py:207; reduced mod p here,
    which leaves every commitment unchanged because the SRS points have order p)."""
    if domain_size <= 0:
        raise ValueError("domain_size must be positive")
    if len(poly) < domain_size:
        return [0]
    q = [sum(poly[j + i * domain_size] for i in range(1, (len(poly) - j - 1) // domain_size + 1)) % prime
         for j in range(len(poly) - domain_size)]
    while q and q[-1] == 0:
        q.pop()
    return q
