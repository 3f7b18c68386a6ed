#!/usr/bin/env python3
"""Generate montmul28_gen.hip.h: Montgomery multiplication / squaring over UNSATURATED signed limbs for gfx950, one
inline-asm statement per operation (hipcc pads every asm statement with an s_nop, and left to itself it re-associates
the column sums into trees of v_lshl_add_u64 + v_mov — measured 14.5 k instructions for the bucket walk against 5.3 k
with these sequences).

Radix 2^B (B = 28 for Fq: 14 limbs, R = 2^392; B = 29 for Fr: 9 limbs, R = 2^261).  A 64-bit accumulator holds a whole
column of partial products, so one partial product is ONE v_mad_i64_i32 — no carry instruction:

    column k <  n:  acc += sum_{i+j=k} a_i b_j + sum_{i+j=k, i<k} m_i p_j ;  m_k = (acc.lo * n0) mod 2^B ; acc += m_k p_0
    column k >= n:  acc += sum_{i+j=k} (a_i b_j + m_i p_j) ;                 r_{k-n} = acc.lo mod 2^B
    acc >>= B   (arithmetic: limbs and accumulator are signed)

m_i is last read in column i + n - 1 and r_i is first written in column i + n, so the m's live in the result registers.
The squaring variant takes the off-diagonal products once against the doubled operand (n(n+1)/2 instead of n^2).
tests/test_montmul_gen.py interprets the emitted text instruction by instruction against big-integer arithmetic.
"""
import sys

ACC = "v[2:3]"
ACC_LO = "v2"


def gen(name: str, n: int, bits: int, square: bool, neg_n0: bool = False) -> str:
    mask = (1 << bits) - 1
    # operands: outputs r[0..n-1] (double as m), [a2[0..n-1] when squaring]; inputs a[], b[] (mul only), p[] (sgpr), n0 (sgpr)
    R = lambda i: f"%{i}"
    if square:                       # a2[0..n-2]: the top limb is never the smaller index of a pair
        A2 = lambda i: f"%{n + i}"
        A = lambda i: f"%{2 * n - 1 + i}"
        B = A
        P = lambda i: f"%{3 * n - 1 + i}"
        N0 = f"%{4 * n - 1}"
    else:
        A = lambda i: f"%{n + i}"
        B = lambda i: f"%{2 * n + i}"
        P = lambda i: f"%{3 * n + i}"
        N0 = f"%{4 * n}"
    M = R
    lines = []
    started = [False]

    def mac(x, y):
        addend = ACC if started[0] else "0"
        lines.append(f"v_mad_i64_i32 {ACC}, vcc, {x}, {y}, {addend}")
        started[0] = True

    if square:
        for i in range(n - 1):
            lines.append(f"v_lshlrev_b32 {A2(i)}, 1, {A(i)}")
    for k in range(2 * n - 1):
        lo, hi = max(0, k - n + 1), min(k, n - 1)
        if square:
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    mac(A2(i), A(k - i))
            if k % 2 == 0:
                mac(A(k // 2), A(k // 2))
        else:
            for i in range(lo, hi + 1):
                mac(A(i), B(k - i))
        for i in range(lo, hi + 1):
            if k < n and i == k:
                continue
            mac(M(i), P(k - i))
        if k < n:
            if neg_n0:                  # n0 = -1 mod 2^B (p = 1 mod 2^B): m_k = -acc mod 2^B, a subtraction instead of a product
                lines.append(f"v_sub_u32 {M(k)}, 0, {ACC_LO}")
            else:
                lines.append(f"v_mul_lo_u32 {M(k)}, {ACC_LO}, {N0}")
            lines.append(f"v_and_b32 {M(k)}, {hex(mask)}, {M(k)}")
            mac(M(k), P(0))
        else:
            lines.append(f"v_and_b32 {R(k - n)}, {hex(mask)}, {ACC_LO}")
        lines.append(f"v_ashrrev_i64 {ACC}, {bits}, {ACC}")
    lines.append(f"v_mov_b32 {R(n - 1)}, {ACC_LO}")
    body = "\\n\\t".join(lines)
    outs = [f'"=&v"(r[{i}])' for i in range(n)]
    if square:
        outs += [f'"=&v"(a2_{i})' for i in range(n - 1)]
    ins = [f'"v"(a[{i}])' for i in range(n)]
    if not square:
        ins += [f'"v"(b[{i}])' for i in range(n)]
    ins += [f'"s"((int32_t)FP::P[{i}])' for i in range(n)] + ['"s"(FP::N0)']
    n_mac = sum(1 for l in lines if l.startswith("v_mad"))
    decl = ""
    if square:
        decl = "    int32_t " + ", ".join(f"a2_{i}" for i in range(n - 1)) + ";\n"
    args = f"int32_t (&r)[{n}], const int32_t (&a)[{n}]" + ("" if square else f", const int32_t (&b)[{n}]")
    return f'''
// {name}: {n} limbs of {bits} bits, {n_mac} v_mad_i64_i32, {len(lines)} instructions
template <class FP>
DR_DEV void {name}({args}) {{
{decl}    asm("{body}"
        : {", ".join(outs)}
        : {", ".join(ins)}
        : "vcc", "v2", "v3");
}}
'''


def gen2(name: str, n: int, bits: int, neg_n0: bool = False) -> str:
    """(a b + c d + m p) / R with ONE reduction: the two products share every column sum (3 n^2 multiply-adds instead of 4 n^2).
    Column bound for 14 x 28: 14 (|a_i b_j| + |c_i d_j| + m p) < 2^63 needs e.g. |a_i| < 2^29, the other limbs below 2^28;
    for 9 x 29: 9 (2 * 2^58 + 2^58) = 2^62.75 — all four operands with limbs below 2^29."""
    mask = (1 << bits) - 1
    R = lambda i: f"%{i}"
    A = lambda i: f"%{n + i}"
    B = lambda i: f"%{2 * n + i}"
    C = lambda i: f"%{3 * n + i}"
    D = lambda i: f"%{4 * n + i}"
    P = lambda i: f"%{5 * n + i}"
    N0 = f"%{6 * n}"
    M = R
    lines = []
    started = [False]

    def mac(x, y):
        addend = ACC if started[0] else "0"
        lines.append(f"v_mad_i64_i32 {ACC}, vcc, {x}, {y}, {addend}")
        started[0] = True

    for k in range(2 * n - 1):
        lo, hi = max(0, k - n + 1), min(k, n - 1)
        for i in range(lo, hi + 1):
            mac(A(i), B(k - i))
        for i in range(lo, hi + 1):
            mac(C(i), D(k - i))
        for i in range(lo, hi + 1):
            if k < n and i == k:
                continue
            mac(M(i), P(k - i))
        if k < n:
            if neg_n0:                  # n0 = -1 mod 2^B (p = 1 mod 2^B): m_k = -acc mod 2^B, a subtraction instead of a product
                lines.append(f"v_sub_u32 {M(k)}, 0, {ACC_LO}")
            else:
                lines.append(f"v_mul_lo_u32 {M(k)}, {ACC_LO}, {N0}")
            lines.append(f"v_and_b32 {M(k)}, {hex(mask)}, {M(k)}")
            mac(M(k), P(0))
        else:
            lines.append(f"v_and_b32 {R(k - n)}, {hex(mask)}, {ACC_LO}")
        lines.append(f"v_ashrrev_i64 {ACC}, {bits}, {ACC}")
    lines.append(f"v_mov_b32 {R(n - 1)}, {ACC_LO}")
    body = "\\n\\t".join(lines)
    outs = [f'"=&v"(r[{i}])' for i in range(n)]
    ins = [f'"v"({v}[{i}])' for v in "abcd" for i in range(n)]
    ins += [f'"s"((int32_t)FP::P[{i}])' for i in range(n)] + ['"s"(FP::N0)']
    n_mac = sum(1 for l in lines if l.startswith("v_mad"))
    return f'''
// {name}: a b + c d, {n} limbs of {bits} bits, {n_mac} v_mad_i64_i32, {len(lines)} instructions
template <class FP>
DR_DEV void {name}(int32_t (&r)[{n}], const int32_t (&a)[{n}], const int32_t (&b)[{n}], const int32_t (&c)[{n}], const int32_t (&d)[{n}]) {{
    asm("{body}"
        : {", ".join(outs)}
        : {", ".join(ins)}
        : "vcc", "v2", "v3");
}}
'''


def main(path):
    out = ["// GENERATED by gen_montmul28.py — do not edit.\n#pragma once\n"]
    out.append(gen("montmul14x28_asm", 14, 28, False))
    out.append(gen("montsqr14x28_asm", 14, 28, True))
    out.append(gen2("montmul2_14x28_asm", 14, 28))
    out.append(gen("montmul9x29_asm", 9, 29, False, neg_n0=True))          # Fr: p = 1 mod 2^32
    out.append(gen("montsqr9x29_asm", 9, 29, True, neg_n0=True))
    out.append(gen2("montmul2_9x29_asm", 9, 29, neg_n0=True))
    with open(path, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else __file__.replace("gen_montmul28.py", "montmul28_gen.hip.h"))
