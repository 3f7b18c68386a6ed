"""The Bernstein-Yang inversion of dot_ring_amd/csrc/divstep28.hip.h, compiled for the host with g++ (the routine is plain
integer C++ shared by host and device builds), against big-integer arithmetic: random values, the edge values of the field,
values built to need many division steps, and the stated output bounds (|out| < 21 p, limbs within the lazy-operand range
of the Montgomery product that follows it on the device)."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FQ_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR_P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FIELDS = {"fq": (FQ_P, 14, 28, 40), "fr": (FR_P, 9, 29, 26)}


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("divstep") / "divstep_host_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "divstep_host_check.cpp"), "-o", str(exe)], check=True)
    return str(exe)


def _run(checker, xs, field="fq"):
    out = subprocess.run([checker, field], input="".join(f"{x:x}\n" for x in xs), capture_output=True, text=True, check=True).stdout
    rows = [list(map(int, line.split())) for line in out.strip().splitlines()]
    assert len(rows) == len(xs)
    return rows


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_divstep_inversion_against_big_integers(checker, field):
    p, n, bits, max_batches = FIELDS[field]
    rng = random.Random(381)
    xs = [rng.randrange(1, p) for _ in range(4000)]
    xs += [1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << (p.bit_length() - 2), (1 << (p.bit_length() - 2)) - 1, (1 << bits) - 1,
           1 << bits, (1 << (bits * (n - 1))) + 1, p >> 1, p - (1 << 200)]
    xs += [pow(2, k, p) for k in range(0, 2 * p.bit_length(), 19)]             # powers of two: long runs of even steps
    xs += [(p - pow(2, k, p)) % p for k in range(1, p.bit_length(), 23)]
    xs += [pow(3, -k, p) for k in range(1, 40)]
    worst = 0
    for x, row in zip(xs, _run(checker, xs, field)):
        limbs, batches = row[:n], row[n]
        value = sum(l << (bits * i) for i, l in enumerate(limbs))
        assert value * x % p == 1, hex(x)
        assert abs(value) < (max_batches // 2 + 1) * p
        assert all(abs(l) < (1 << bits) for l in limbs[: n - 1]) and abs(limbs[n - 1]) < (1 << 27)
        worst = max(worst, batches)
    assert worst <= max_batches


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_divstep_zero_maps_to_zero(checker, field):
    n = FIELDS[field][1]
    (row,) = _run(checker, [0], field)
    assert row[:n] == [0] * n and row[n] == 0
