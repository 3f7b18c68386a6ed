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
FR_P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


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
        elif mnem == "v_sub_u32":
            wr(ops[0], (rd(ops[1]) - rd(ops[2])) & 0xFFFFFFFF)
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


# per field: asm names, limbs, bits, modulus, the operand shapes the kernels use — (limb bound a, limb bound b, |a| / p, |b| / p):
# the value bounds keep |a b| / R below p / 2 ("normal" result) — and the shapes of the squaring and of the fused product
CASES = [
    dict(mul="montmul14x28_asm", sqr="montsqr14x28_asm", mul2="montmul2_14x28_asm", n=14, bits=28, p=FQ_P,
         shapes=[(1 << 28, 1 << 28, 2, 2), (1 << 30, 1 << 28, 31, 31), (1 << 29, 1 << 29, 31, 31), (1 << 28, 1 << 30, 8, 31)],
         extreme=(1 << 30, 1 << 28), sqr_shape=(1 << 29, 31), mul2_shape=((1 << 29, 8), (1 << 28, 8), (1 << 28, 4), (1 << 28, 2))),
    dict(mul="montmul9x29_asm", sqr="montsqr9x29_asm", mul2="montmul2_9x29_asm", n=9, bits=29, p=FR_P,
         shapes=[(1 << 29, 1 << 29, 2, 2), (1 << 30, 1 << 29, 6, 5), (3 << 28, 3 << 28, 5, 5), (1 << 29, 1 << 30, 9, 3), (1 << 29, 1 << 29, 14, 1)],
         extreme=(1 << 30, 1 << 29), sqr_shape=(3 << 28, 5), mul2_shape=((1 << 29, 4), (1 << 29, 4), (1 << 29, 4), (1 << 29, 4))),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["mul"])
def test_generated_montgomery_sequences(header, case):
    n, bits, p = case["n"], case["bits"], case["p"]
    rng = random.Random(2802)
    R = 1 << (bits * n)
    n0 = (-pow(p, -1, 1 << bits)) % (1 << bits)
    consts = {f"(int32_t)FP::P[{i}]": x for i, x in enumerate(_limbs(p, n, bits))}
    consts["FP::N0"] = n0
    mul = _parse(header, case["mul"])
    sqr = _parse(header, case["sqr"])
    assert sum(1 for l in mul[0] if l.startswith("v_mad")) == 2 * n * n
    assert sum(1 for l in sqr[0] if l.startswith("v_mad")) == n * n + n * (n + 1) // 2

    def check(res, prod, tag):
        limbs = [res[f"r[{i}]"] for i in range(n)]
        assert all(0 <= x < (1 << bits) for x in limbs[:-1]), tag
        got = sum(x << (bits * i) for i, x in enumerate(limbs))
        assert (got * R - prod) % p == 0, tag
        assert -p // 2 < got < p + p // 2, tag                      # "normal": (-p/2, 1.5 p)

    def value(l):
        return sum(x << (bits * i) for i, x in enumerate(l))

    for la, lb, va, vb in case["shapes"]:
        for _ in range(12):
            a, av = _lazy(rng, n, bits, la, p, va)
            b, bv = _lazy(rng, n, bits, lb, p, vb)
            vals = dict(consts)
            vals.update({f"a[{i}]": a[i] for i in range(n)})
            vals.update({f"b[{i}]": b[i] for i in range(n)})
            check(_run(*mul, vals), av * bv, (la, lb))
    # extreme limbs: every limb at +-(bound - 1) — the 64-bit column sums must hold (the interpreter asserts it); the value is
    # far outside the lazy range, so only the congruence is checked
    ea, eb = case["extreme"]
    for sa in (1, -1):
        for sb in (1, -1):
            a = [sa * (ea - 1)] * (n - 1) + [sa * 3]
            b = [sb * (eb - 1)] * (n - 1) + [sb * 3]
            vals = dict(consts)
            vals.update({f"a[{i}]": a[i] for i in range(n)})
            vals.update({f"b[{i}]": b[i] for i in range(n)})
            res = _run(*mul, vals)
            got = value([res[f"r[{i}]"] for i in range(n)])
            assert (got * R - value(a) * value(b)) % p == 0
    # squaring
    sl, sv = case["sqr_shape"]
    for _ in range(24):
        a, av = _lazy(rng, n, bits, sl, p, sv)
        vals = dict(consts)
        vals.update({f"a[{i}]": a[i] for i in range(n)})
        check(_run(*sqr, vals), av * av, "sqr")
    for s_ in (1, -1):
        a = [s_ * (sl - 1)] * (n - 1) + [s_ * 3]
        vals = dict(consts)
        vals.update({f"a[{i}]": a[i] for i in range(n)})
        res = _run(*sqr, vals)
        got = value([res[f"r[{i}]"] for i in range(n)])
        assert (got * R - value(a) ** 2) % p == 0
    # fused a b + c d (one reduction), both signs
    mul2 = _parse(header, case["mul2"])
    assert sum(1 for l in mul2[0] if l.startswith("v_mad")) == 3 * n * n
    (la, va), (lb, vb), (lc, vc), (ld, vd) = case["mul2_shape"]
    for trial in range(24):
        a, av = _lazy(rng, n, bits, la, p, va)
        b, bv = _lazy(rng, n, bits, lb, p, vb)
        c, cv_ = _lazy(rng, n, bits, lc, p, vc)
        d_, dv = _lazy(rng, n, bits, ld, p, vd)
        extreme = trial == 0
        if extreme:
            a = [la - 1] * (n - 1) + [3]; av = value(a)
            b = [lb - 1] * (n - 1) + [3]; bv = value(b)
            c = [-(lc - 1)] * (n - 1) + [-3]; cv_ = value(c)
            d_ = [-(ld - 1)] * (n - 1) + [-3]; dv = value(d_)
        vals = dict(consts)
        for nm, arr in (("a", a), ("b", b), ("c", c), ("d", d_)):
            vals.update({f"{nm}[{i}]": arr[i] for i in range(n)})
        res = _run(*mul2, vals)
        limbs = [res[f"r[{i}]"] for i in range(n)]
        assert all(0 <= x < (1 << bits) for x in limbs[:-1])
        got = value(limbs)
        assert (got * R - (av * bv + cv_ * dv)) % p == 0
        if not extreme:
            assert -p // 2 < got < p + p // 2
    # zero, one, p - 1 in canonical limbs
    for av in (0, 1, p - 1, R % p):
        for bv in (0, 1, p - 1):
            vals = dict(consts)
            vals.update({f"a[{i}]": x for i, x in enumerate(_limbs(av, n, bits))})
            vals.update({f"b[{i}]": x for i, x in enumerate(_limbs(bv, n, bits))})
            check(_run(*mul, vals), av * bv, "canonical")
