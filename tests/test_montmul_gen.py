"""The generated gfx950 Montgomery sequences over unsaturated limbs (dot_ring_amd/csrc/gen_montmul28.py) are executed
here by a small interpreter of the emitted asm TEXT — instruction by instruction, with the register widths and
signedness of the hardware instructions — and compared with big-integer arithmetic.  No GPU needed: this is the check
that the text the assembler sees computes (a b + m p) / R for every operand shape the kernels use."""
from __future__ import annotations

import os
import random
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dot_ring_amd", "csrc")

FQ_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _s32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >> 31 else v


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _parse(header_text: str, name: str):
    m = re.search(r"DR_DEV void " + name + r"\((.*?)\) \{\n(?:\s+int32_t [^\n]*\n)?\s+asm\(\"(.*?)\"\n\s+: (.*?)\n\s+: (.*?)\n\s+: (.*?)\);", header_text, re.S)
    assert m, name
    body, outs, ins = m.group(2), m.group(3), m.group(4)
    lines = body.split("\\n\\t")
    out_ops = re.findall(r'"=&v"\(([^)]+)\)', outs)
    in_ops = re.findall(r'"([vs])"\(((?:[^()]|\([^()]*\))+)\)', ins)
    return lines, out_ops, in_ops


def _run(lines, out_ops, in_ops, values):
    """values: operand expression -> 32-bit value for every input.  Returns {output expression: signed 32-bit value}."""
    regs = {}
    for idx, (_, expr) in enumerate(in_ops):
        regs[f"%{len(out_ops) + idx}"] = values[expr] & 0xFFFFFFFF
    written = set()

    def rd(op):
        if op.startswith("0x"):
            return int(op, 16)
        if op.isdigit():
            return int(op)
        if op == "v[2:3]":
            return (regs["v3"] << 32) | regs["v2"]
        assert op in regs, f"read of unwritten register {op}"
        return regs[op]

    def wr(op, v):
        if op == "v[2:3]":
            regs["v2"], regs["v3"] = v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF
        else:
            regs[op] = v & 0xFFFFFFFF
            written.add(op)

    for line in lines:
        mnem, rest = line.split(None, 1)
        ops = [o.strip() for o in rest.split(",")]
        if mnem == "v_mad_i64_i32":
            # D.i64 = S0.i32 * S1.i32 + S2.i64 (VOP3, carry-out to the named SGPR pair ignored here)
            dst, vcc, a, b, c = ops[0], ops[1], ops[2], ops[3], ",".join(ops[4:])
            assert vcc == "vcc"
            full = _s32(rd(a)) * _s32(rd(b)) + _s64(rd(c))
            assert -(1 << 63) <= full < (1 << 63), "64-bit accumulator overflow"
            wr(dst, full & ((1 << 64) - 1))
        elif mnem == "v_mul_lo_u32":
            wr(ops[0], (rd(ops[1]) * rd(ops[2])) & 0xFFFFFFFF)
        elif mnem == "v_and_b32":
            wr(ops[0], rd(ops[1]) & rd(ops[2]))
        elif mnem == "v_ashrrev_i64":
            wr(ops[0], (_s64(rd(",".join(ops[2:]))) >> rd(ops[1])) & ((1 << 64) - 1))
        elif mnem == "v_lshlrev_b32":
            wr(ops[0], (rd(ops[2]) << rd(ops[1])) & 0xFFFFFFFF)
        elif mnem == "v_mov_b32":
            wr(ops[0], rd(ops[1]))
        else:
            raise AssertionError(f"unknown instruction {mnem}")
    return {expr: _s32(regs[f"%{i}"]) for i, expr in enumerate(out_ops)}


@pytest.fixture(scope="module")
def header(tmp_path_factory):
    out = tmp_path_factory.mktemp("gen") / "montmul28_gen.hip.h"
    subprocess.run([sys.executable, os.path.join(CSRC, "gen_montmul28.py"), str(out)], check=True)
    text = out.read_text()
    committed = os.path.join(CSRC, "montmul28_gen.hip.h")
    if os.path.exists(committed):
        assert open(committed).read() == text, "montmul28_gen.hip.h is stale: rerun gen_montmul28.py"
    return text


def _limbs(v, n, bits):
    return [(v >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def _lazy(rng, n, bits, limb_bound, p, value_bound_p):
    """a lazy element: signed limbs with |limb| < limb_bound, |value| < value_bound_p * p"""
    while True:
        l = [rng.randrange(-limb_bound + 1, limb_bound) for _ in range(n - 1)]
        top = rng.randrange(-(value_bound_p * p >> (bits * (n - 1))), (value_bound_p * p >> (bits * (n - 1))) + 1)
        l.append(top)
        v = sum(x << (bits * i) for i, x in enumerate(l))
        if abs(v) < value_bound_p * p:
            return l, v


CASES = [("montmul14x28_asm", "montsqr14x28_asm", 14, 28, FQ_P)]


@pytest.mark.parametrize("mul_name,sqr_name,n,bits,p", CASES)
def test_generated_montgomery_sequences(header, mul_name, sqr_name, n, bits, p):
    rng = random.Random(2802)
    R = 1 << (bits * n)
    n0 = (-pow(p, -1, 1 << bits)) % (1 << bits)
    consts = {f"(int32_t)FP::P[{i}]": x for i, x in enumerate(_limbs(p, n, bits))}
    consts["FP::N0"] = n0
    mul = _parse(header, mul_name)
    sqr = _parse(header, sqr_name)
    assert sum(1 for l in mul[0] if l.startswith("v_mad")) == 2 * n * n
    assert sum(1 for l in sqr[0] if l.startswith("v_mad")) == n * n + n * (n + 1) // 2

    def check(res, a_val, b_val, tag):
        limbs = [res[f"r[{i}]"] for i in range(n)]
        assert all(0 <= x < (1 << bits) for x in limbs[:-1]), tag
        got = sum(x << (bits * i) for i, x in enumerate(limbs))
        assert (got * R - a_val * b_val) % p == 0, tag
        assert -p // 2 < got < p + p // 2, tag                      # "normal": (-p/2, 1.5 p)

    shapes = [((1 << bits), (1 << bits), 2, 2),          # two products
              ((1 << 30), (1 << bits), 31, 31),          # lazy x normal at the stated bounds
              ((1 << 29), (1 << 29), 31, 31),
              ((1 << bits), (1 << 30), 8, 31)]
    for la, lb, va, vb in shapes:
        for _ in range(12):
            a, av = _lazy(rng, n, bits, la, p, va)
            b, bv = _lazy(rng, n, bits, lb, p, vb)
            vals = dict(consts)
            vals.update({f"a[{i}]": a[i] for i in range(n)})
            vals.update({f"b[{i}]": b[i] for i in range(n)})
            check(_run(*mul, vals), av, bv, (la, lb))
    # extreme limbs: every limb at +-(bound - 1)
    for sa in (1, -1):
        for sb in (1, -1):
            a = [sa * ((1 << 30) - 1)] * (n - 1) + [sa * 3]
            b = [sb * ((1 << bits) - 1)] * (n - 1) + [sb * 3]
            vals = dict(consts)
            vals.update({f"a[{i}]": a[i] for i in range(n)})
            vals.update({f"b[{i}]": b[i] for i in range(n)})
            res = _run(*mul, vals)
            av = sum(x << (bits * i) for i, x in enumerate(a))
            bv = sum(x << (bits * i) for i, x in enumerate(b))
            got = sum(res[f"r[{i}]"] << (bits * i) for i in range(n))
            assert (got * R - av * bv) % p == 0
    # squaring: |limb| <= 2^29
    for _ in range(24):
        a, av = _lazy(rng, n, bits, 1 << 29, p, 31)
        vals = dict(consts)
        vals.update({f"a[{i}]": a[i] for i in range(n)})
        check(_run(*sqr, vals), av, av, "sqr")
    for s in (1, -1):
        a = [s * (1 << 29)] * (n - 1) + [s * 3]
        vals = dict(consts)
        vals.update({f"a[{i}]": a[i] for i in range(n)})
        res = _run(*sqr, vals)
        av = sum(x << (bits * i) for i, x in enumerate(a))
        got = sum(res[f"r[{i}]"] << (bits * i) for i in range(n))
        assert (got * R - av * av) % p == 0
    # fused a b + c d (one reduction): a with limbs up to 2^29, the others below 2^28, both signs
    mul2 = _parse(header, "montmul2_14x28_asm")
    assert sum(1 for l in mul2[0] if l.startswith("v_mad")) == 3 * n * n
    for trial in range(24):
        a, av = _lazy(rng, n, bits, 1 << 29, p, 8)
        b, bv = _lazy(rng, n, bits, 1 << bits, p, 8)
        c, cv_ = _lazy(rng, n, bits, 1 << bits, p, 4)
        d_, dv = _lazy(rng, n, bits, 1 << bits, p, 2)
        if trial == 0:
            a = [(1 << 29) - 1] * (n - 1) + [3]; av = sum(x << (bits * i) for i, x in enumerate(a))
            b = [(1 << bits) - 1] * (n - 1) + [3]; bv = sum(x << (bits * i) for i, x in enumerate(b))
            c = [-((1 << bits) - 1)] * (n - 1) + [-3]; cv_ = sum(x << (bits * i) for i, x in enumerate(c))
            d_ = [-((1 << bits) - 1)] * (n - 1) + [-3]; dv = sum(x << (bits * i) for i, x in enumerate(d_))
        vals = dict(consts)
        for nm, arr in (("a", a), ("b", b), ("c", c), ("d", d_)):
            vals.update({f"{nm}[{i}]": arr[i] for i in range(n)})
        res = _run(*mul2, vals)
        limbs = [res[f"r[{i}]"] for i in range(n)]
        assert all(0 <= x < (1 << bits) for x in limbs[:-1])
        got = sum(x << (bits * i) for i, x in enumerate(limbs))
        assert (got * R - (av * bv + cv_ * dv)) % p == 0
        assert -p // 2 < got < p + p // 2
    # zero, one, p - 1 in canonical limbs
    for av in (0, 1, p - 1, R % p):
        for bv in (0, 1, p - 1):
            vals = dict(consts)
            vals.update({f"a[{i}]": x for i, x in enumerate(_limbs(av, n, bits))})
            vals.update({f"b[{i}]": x for i, x in enumerate(_limbs(bv, n, bits))})
            check(_run(*mul, vals), av, bv, "canonical")
