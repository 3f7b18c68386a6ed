"""`BlsScalarNTTPlan` with the constructor and methods of dot_ring/ring_proof/polynomial/ntt.pyx:29-163, over `dr_ntt`.

The reference builds a plan from per-stage twiddle tables and a bit-reversal permutation (polynomial/fft.py:14-55:
`twiddles[s][j] = omega^(j * n / 2^(s+1))`, `rev[i]` = bit-reversed i) and calls `transform(list)` /
`transform_scaled(list, scale)` in place on Python lists (fft.py:70, 84).  The GPU kernel derives its own tables from the
n-th root of unity, which is the second twiddle of the last stage; the plan checks that the tables it was given are the
ones that root generates, so a caller with another permutation or twiddle set gets an error instead of a different transform.
"""
from __future__ import annotations

from .. import runtime

_P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class BlsScalarNTTPlan:
    def __init__(self, twiddles: list, rev: list):
        self.n = len(rev)
        self.stages = len(twiddles)
        if self.n < 2 or self.n & (self.n - 1):
            raise ValueError(f"native NTT plan size must be a power of two >= 2, got {self.n}")
        expected = self.n.bit_length() - 1
        if self.stages != expected:
            raise ValueError(f"native NTT plan expected {expected} twiddle stages, got {self.stages}")
        for val in rev:
            if val < 0 or val >= self.n:
                raise ValueError(f"bit-reverse index {val} is outside plan size {self.n}")
        m = 2
        for stage in twiddles:
            if len(stage) != m >> 1:
                raise ValueError(f"native NTT plan stage for m={m} expected {m >> 1} twiddles, got {len(stage)}")
            m <<= 1
        # omega = the primitive n-th root the tables were generated from (n = 2: the single stage is [1] and omega = -1)
        self.omega = int(twiddles[-1][1]) % _P if self.n > 2 else _P - 1
        bits = expected
        if any(int(rev[i]) != int(f"{i:0{bits}b}"[::-1], 2) for i in range(self.n)):
            raise ValueError("native NTT plan: rev is not the bit-reversal permutation")
        for s, stage in enumerate(twiddles):
            step = pow(self.omega, self.n >> (s + 1), _P)
            w = 1
            for j, t in enumerate(stage):
                if int(t) % _P != w:
                    raise ValueError(f"native NTT plan: twiddles of stage {s} are not powers of one n-th root of unity")
                w = w * step % _P
        if pow(self.omega, self.n, _P) != 1 or pow(self.omega, self.n >> 1, _P) == 1:
            raise ValueError("native NTT plan: twiddles do not come from a primitive n-th root of unity")

    def transform(self, coeffs: list) -> None:
        """ntt.pyx:104 — in-place NTT of a Python list of integers."""
        if len(coeffs) <= 1:
            return
        self._run(coeffs, None)

    def transform_scaled(self, coeffs: list, scale) -> None:
        """ntt.pyx:110 — in-place NTT, every output multiplied by `scale`."""
        if len(coeffs) <= 1:
            return
        self._run(coeffs, scale)

    def _run(self, coeffs: list, scale) -> None:
        if len(coeffs) != self.n:
            raise ValueError(f"coefficient length {len(coeffs)} does not match native NTT plan size {self.n}")
        data = b"".join((int(c) % _P).to_bytes(32, "little") for c in coeffs)
        out = runtime.context().ntt(data, self.n.bit_length() - 1, self.omega, None if scale is None else int(scale) % _P)
        coeffs[:] = [int.from_bytes(out[32 * i : 32 * i + 32], "little") for i in range(self.n)]
