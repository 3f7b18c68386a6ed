#!/usr/bin/env python3
"""ORACLE tooling (test infrastructure): NTT golden vectors from the REFERENCE'S OWN kernel.

Runs `BlsScalarNTTPlan.transform / transform_scaled` of /root/reference/dot_ring/ring_proof/polynomial/ntt.pyx — compiled from where it
lies into oracle/_ref/pyx by oracle/build_ref_ntt.sh, over the reference's bls12_381_scalar.c — on seeded inputs, with plans built the
way the reference builds them (bit-reversal list and per-stage twiddle lists, ring_proof/polynomial/fft.py:14-55), and writes
tests/golden/ntt/ntt_cases.json.  The reference cannot travel to the GPU box; this file of inputs (as seeds) and expected outputs
(whole vectors up to 32 points, SHA-256 of the little-endian output bytes plus the first and last elements beyond) does.

    bash oracle/build_ref_ntt.sh && python3 oracle/gen_ntt_fixtures.py

Cases: n = 2 ... 16384; forward (omega_n, no scale), inverse (omega_n^-1, scale n^-1: fft.py:84-105), and an arbitrary scale; roots
omega_n = ROOT_OF_UNITY_2048^(2048 / n) up to 2048 (ring_proof/params.py:12) and 7^((p - 1) / n) beyond (7 generates Fr^*).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_ref", "pyx"))

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ROOT_OF_UNITY_2048 = 49307615728544765012166121802278658070711169839041683575071795236746050763237


def seeded_inputs(n: int, tag: str) -> list[int]:
    """n field elements from SHAKE256("ntt-fixture" || tag): 48 bytes each, little-endian, mod p; every 7th one an edge value."""
    stream = hashlib.shake_256(b"ntt-fixture" + tag.encode()).digest(48 * n)
    vals = [int.from_bytes(stream[48 * i : 48 * i + 48], "little") % P for i in range(n)]
    edges = (0, 1, P - 1, P - 2, 2, (P - 1) // 2, (P + 1) // 2)
    for i in range(0, n, 7):
        vals[i] = edges[(i // 7) % len(edges)]
    return vals


def bit_reverse(n: int) -> list[int]:
    bits = n.bit_length() - 1
    return [int(format(i, f"0{bits}b")[::-1], 2) if bits else 0 for i in range(n)]


def stage_twiddles(n: int, omega: int) -> list[list[int]]:
    out, m = [], 2
    while m <= n:
        step = pow(omega, n // m, P)
        out.append([pow(step, j, P) for j in range(m // 2)])
        m *= 2
    return out


def root_of_unity(n: int) -> int:
    w = pow(ROOT_OF_UNITY_2048, 2048 // n, P) if n <= 2048 else pow(7, (P - 1) // n, P)
    assert pow(w, n, P) == 1 and pow(w, n // 2, P) == P - 1
    return w


def digest(vals) -> str:
    return hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in vals)).hexdigest()


def main() -> int:
    from dot_ring.ring_proof.polynomial.ntt import BlsScalarNTTPlan      # the reference's kernel (oracle/_ref/pyx)

    cases = []
    for log2n in range(1, 15):
        n = 1 << log2n
        w = root_of_unity(n)
        for kind in ("forward", "inverse", "scaled"):
            omega = w if kind != "inverse" else pow(w, -1, P)
            scale = None if kind == "forward" else pow(n, -1, P) if kind == "inverse" else (0x1234567 + 977 * log2n) * pow(3, 200 + log2n, P) % P
            tag = f"{log2n}-{kind}"
            vals = seeded_inputs(n, tag)
            plan = BlsScalarNTTPlan(stage_twiddles(n, omega), bit_reverse(n))
            out = list(vals)
            if scale is None:
                plan.transform(out)
            else:
                plan.transform_scaled(out, scale)
            rec = {"log2n": log2n, "kind": kind, "omega": hex(omega), "scale": None if scale is None else hex(scale), "input_tag": tag,
                   "input_sha256": digest(vals), "output_sha256": digest(out), "output_head": [hex(v) for v in out[:4]],
                   "output_tail": [hex(v) for v in out[-4:]]}
            if n <= 32:
                rec["input"] = [hex(v) for v in vals]
                rec["output"] = [hex(v) for v in out]
            cases.append(rec)
    dst = os.path.join(ROOT, "tests", "golden", "ntt")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "ntt_cases.json"), "w") as f:
        json.dump({"source": "BlsScalarNTTPlan of /root/reference/dot_ring/ring_proof/polynomial/ntt.pyx (built by oracle/build_ref_ntt.sh), "
                             "driven by oracle/gen_ntt_fixtures.py", "modulus": hex(P),
                   "input_rule": "SHAKE256(b'ntt-fixture' + input_tag) -> 48-byte little-endian chunks mod p; element 7k replaced by the "
                                 "edge value (0, 1, p-1, p-2, 2, (p-1)/2, (p+1)/2)[k mod 7]",
                   "cases": cases}, f, indent=1)
    print(f"wrote {len(cases)} cases to tests/golden/ntt/ntt_cases.json")
    return 0


if __name__ == "__main__":
    sys.exit(main())
