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


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("divstep") / "divstep_host_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "divstep_host_check.cpp"), "-o", str(exe)], check=True)
    return str(exe)


def _run(checker, xs):
    out = subprocess.run([checker], input="".join(f"{x:x}\n" for x in xs), capture_output=True, text=True, check=True).stdout
    rows = [list(map(int, line.split())) for line in out.strip().splitlines()]
    assert len(rows) == len(xs)
    return rows


def test_divstep_inversion_against_big_integers(checker):
    rng = random.Random(381)
    xs = [rng.randrange(1, FQ_P) for _ in range(4000)]
    xs += [1, 2, 3, FQ_P - 1, FQ_P - 2, (FQ_P - 1) // 2, (FQ_P + 1) // 2, 1 << 380, (1 << 380) - 1, (1 << 28) - 1, 1 << 28,
           (1 << 364) + 1, FQ_P >> 1, FQ_P - (1 << 200)]
    xs += [pow(2, k, FQ_P) for k in range(0, 760, 19)]                       # powers of two: long runs of even steps
    xs += [(FQ_P - pow(2, k, FQ_P)) % FQ_P for k in range(1, 380, 23)]
    xs += [pow(3, -k, FQ_P) for k in range(1, 40)]
    worst = 0
    for x, row in zip(xs, _run(checker, xs)):
        limbs, batches = row[:14], row[14]
        value = sum(l << (28 * i) for i, l in enumerate(limbs))
        assert value * x % FQ_P == 1, hex(x)
        assert abs(value) < 21 * FQ_P
        assert all(abs(l) < (1 << 28) for l in limbs[:13]) and abs(limbs[13]) < (1 << 23)
        worst = max(worst, batches)
    assert worst <= 40


def test_divstep_zero_maps_to_zero(checker):
    (row,) = _run(checker, [0])
    assert row[:14] == [0] * 14 and row[14] == 0
