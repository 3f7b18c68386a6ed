// BLS12-381 base field for gfx950 in an UNSATURATED radix: 14 signed limbs of 28 bits, lazy reduction.
//
// Why (round-2 measurement, tools/ubench_limbs.hip): with saturated 32-bit limbs every partial product of the Montgomery
// multiplication costs v_mad_u64_u32 + v_addc_co_u32 (the 64-bit accumulator overflows after one product: 288 + 288
// instructions for Fq, field.hip.h).  With 28-bit limbs a 64-bit accumulator absorbs a whole column — 28 products of < 2^58
// — so a partial product is ONE v_mad_i64_i32 and the carry instructions disappear: 392 multiply-adds per product
// instead of 576 multiply-add + carry instructions.  The 11 spare bits (R = 2^392 against a 381-bit p) also make
// additions and subtractions plain limb-wise v_add/v_sub (no carry chain, no conditional subtraction of p).
//
// Value of an element: sum l[i] * 2^(28 i), limbs SIGNED.  Montgomery form x -> x * 2^392 mod p.
//   "normal"  : limbs 0..12 in [0, 2^28), limb 13 small and signed — what mul() returns; value in (-p/2, 1.5 p)
//   "lazy"    : sums / differences of a few normal values: |limb| < 2^30, |value| < 32 p
// mul() accepts lazy operands as long as  14 * max|a_i| * max|b_j| + 2^60 < 2^63  (e.g. 2^30 x 2^28, 2^29 x 2^29) and
// |a|, |b| < 32 p (then |a b| / R < p / 2).  Canonical values (standard 12 x u32 words, < p) exist only in memory.
#pragma once
#include "field.hip.h"
#include "montmul28_gen.hip.h"
#include "divstep28.hip.h"

namespace dr {

constexpr int L28 = 14;
constexpr uint32_t M28 = 0x0fffffffu;

struct Fq28Params {
    static constexpr uint32_t P[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                       0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
    static constexpr uint32_t N0 = 0xffcfffdu;                     // -p^-1 mod 2^28
    static constexpr uint32_t ONE[14] = {0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                                         0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au};   // 2^392 mod p
    static constexpr uint32_t R2[14] = {0x10370edu, 0x6d1c345u, 0xe243d62u, 0xec45c53u, 0x3b1d65au, 0x093317du, 0xb4f36a0u,
                                        0x5d74088u, 0xc10ea72u, 0x865d118u, 0x7320a75u, 0xfd5cd50u, 0xcc8a759u, 0x000c8d4u};    // R^2 mod p
    static constexpr uint32_t R3[14] = {0x1f7b890u, 0x294cc4du, 0x9f3af22u, 0xb5ba56cu, 0xcb5c0ccu, 0xc0d975cu, 0xc89a8c5u,
                                        0x6c968b4u, 0x22672eau, 0x91de8c9u, 0x35652a6u, 0x84977c8u, 0x424bbb9u, 0x00141abu};    // R^3 mod p
    static constexpr uint32_t FOUR[14] = {0xd1ff2e0u, 0x6000000u, 0x00ac467u, 0x3379b48u, 0x1c84b80u, 0x0e88243u, 0x0dd9a7eu,
                                          0x683dcf8u, 0x6c26d0bu, 0x4a5eec2u, 0x457663cu, 0x04b29f1u, 0x967f3e8u, 0x0015de9u};  // 4 R mod p (curve b)
    // 2^400 mod p: mul28(x 2^384, K400) = x 2^392 — from the 32-bit-limb Montgomery form (R = 2^384) into this one
    static constexpr uint32_t K400[14] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                                          0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau};
    // 2^384 mod p: mul28(x 2^392, K384) = x 2^384 — back into the 32-bit-limb Montgomery form
    static constexpr uint32_t K384[14] = {0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                          0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u};
};

struct Fq28 {
    int32_t l[L28];
    DR_DEV static Fq28 zero() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < L28; i++) r.l[i] = 0;
        return r;
    }
    DR_DEV static Fq28 one() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < L28; i++) r.l[i] = (int32_t)Fq28Params::ONE[i];
        return r;
    }
    template <const uint32_t (&C)[14]>
    DR_DEV static Fq28 constant() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < L28; i++) r.l[i] = (int32_t)C[i];
        return r;
    }
};

DR_DEV Fq28 add(const Fq28& a, const Fq28& b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < L28; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
DR_DEV Fq28 sub(const Fq28& a, const Fq28& b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < L28; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
DR_DEV Fq28 dbl(const Fq28& a) { return add(a, a); }
DR_DEV Fq28 neg(const Fq28& a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < L28; i++) r.l[i] = -a.l[i];
    return r;
}
DR_DEV Fq28 cneg(const Fq28& a, bool negate) {       // negate ? -a : a
    const int32_t s = negate ? -1 : 0;
    Fq28 r;
#pragma unroll
    for (int i = 0; i < L28; i++) r.l[i] = (a.l[i] ^ s) - s;
    return r;
}

// carry propagation: limbs 0..12 into [0, 2^28), the rest into the signed top limb.  Value unchanged.
DR_DEV Fq28 carry(const Fq28& a) {
    Fq28 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L28 - 1; i++) {
        const int32_t t = a.l[i] + c;
        r.l[i] = t & (int32_t)M28;
        c = t >> 28;
    }
    r.l[L28 - 1] = a.l[L28 - 1] + c;
    return r;
}

// Montgomery product (a b + m p) / 2^392, column by column in one signed 64-bit accumulator.  mul() / sqr() are the
// generated asm sequences (gen_montmul28.py); the C++ statements of the same algorithm stay as mul_cxx / sqr_cxx
// (hipcc turns them into ~2.7x the instructions: it re-associates the column sums).
DR_DEV Fq28 mul(const Fq28& a, const Fq28& b) {
    Fq28 r;
    montmul14x28_asm<Fq28Params>(r.l, a.l, b.l);
    return r;
}
DR_DEV Fq28 sqr(const Fq28& a) {
    Fq28 r;
    montsqr14x28_asm<Fq28Params>(r.l, a.l);
    return r;
}
// a b + c d with one Montgomery reduction (montmul2_14x28_asm): 657 instructions against 2 x 461.  Operand bounds: the column
// sums hold 14 (|a_i b_j| + |c_i d_j| + m p) — one operand may carry limbs up to 2^29, the others stay below 2^28.
DR_DEV Fq28 mul2(const Fq28& a, const Fq28& b, const Fq28& c, const Fq28& d) {
    Fq28 r;
    montmul2_14x28_asm<Fq28Params>(r.l, a.l, b.l, c.l, d.l);
    return r;
}
DR_DEV Fq28 mul_cxx(const Fq28& a, const Fq28& b) {
    using FP = Fq28Params;
    Fq28 r;
    int32_t m[L28];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L28; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)(int32_t)FP::P[k - i];
        m[k] = (int32_t)(((uint32_t)acc * FP::N0) & M28);
        acc += (int64_t)m[k] * (int64_t)(int32_t)FP::P[0];
        acc >>= 28;                                               // exact: the low 28 bits are zero now
    }
#pragma unroll
    for (int k = L28; k < 2 * L28 - 1; k++) {
#pragma unroll
        for (int i = k - L28 + 1; i < L28; i++) {
            acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
            acc += (int64_t)m[i] * (int64_t)(int32_t)FP::P[k - i];
        }
        r.l[k - L28] = (int32_t)((uint32_t)acc & M28);
        acc >>= 28;
    }
    r.l[L28 - 1] = (int32_t)acc;
    return r;
}

// a^2: the off-diagonal products once, against the doubled operand (105 + 196 multiply-adds instead of 392).
// Needs 14 * 2 max|a_i|^2 + 2^60 < 2^63: |a_i| <= 2^29.
DR_DEV Fq28 sqr_cxx(const Fq28& a) {
    using FP = Fq28Params;
    Fq28 r;
    int32_t m[L28], a2[L28];
#pragma unroll
    for (int i = 0; i < L28; i++) a2[i] = a.l[i] + a.l[i];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L28 - 1; k++) {
        const int lo = k < L28 ? 0 : k - L28 + 1;
#pragma unroll
        for (int i = lo; 2 * i < k; i++) acc += (int64_t)a2[i] * (int64_t)a.l[k - i];
        if ((k & 1) == 0) acc += (int64_t)a.l[k / 2] * (int64_t)a.l[k / 2];
        if (k < L28) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)(int32_t)FP::P[k - i];
            m[k] = (int32_t)(((uint32_t)acc * FP::N0) & M28);
            acc += (int64_t)m[k] * (int64_t)(int32_t)FP::P[0];
        } else {
#pragma unroll
            for (int i = lo; i < L28; i++) acc += (int64_t)m[i] * (int64_t)(int32_t)FP::P[k - i];
            r.l[k - L28] = (int32_t)((uint32_t)acc & M28);
        }
        acc >>= 28;
    }
    r.l[L28 - 1] = (int32_t)acc;
    return r;
}

// ---------------------------------------------------------------- memory form: 12 x u32 words, canonical (< p)
// words -> limbs (no arithmetic: the value is reinterpreted in radix 2^28)
DR_DEV Fq28 unpack28(const uint32_t (&w)[12]) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < L28; i++) {
        const int bit = 28 * i, j = bit >> 5, sh = bit & 31;
        uint32_t v = w[j] >> sh;
        if (sh > 4 && j + 1 < 12) v |= w[j + 1] << (32 - sh);
        r.l[i] = (int32_t)(v & M28);
    }
    return r;
}

// any lazy value with |value| < 8 p -> the canonical representative in [0, p), as 12 words
DR_DEV void canon28(const Fq28& a, uint32_t (&w)[12]) {
    using FP = Fq28Params;
    // carry first (signed), then + 8p makes the value positive (8p < 2^384) and the second carry pass is unsigned
    const Fq28 c = carry(a);
    uint32_t u[L28], cy = 0;
#pragma unroll
    for (int i = 0; i < L28; i++) {
        u[i] = (uint32_t)c.l[i] + 8u * FP::P[i] + cy;              // < 2^28 + 2^31 + 2^4; the top limb stays >= 0 for |a| < 8p
        if (i < L28 - 1) { cy = u[i] >> 28; u[i] &= M28; }
    }
    Fq28 t;
#pragma unroll
    for (int i = 0; i < L28; i++) t.l[i] = (int32_t)u[i];
    // pack: limb i occupies bits [28 i, 28 i + 28); value < 16 p < 2^385: keep a 13th word
    uint32_t x[13];
#pragma unroll
    for (int j = 0; j < 13; j++) x[j] = 0;
#pragma unroll
    for (int i = 0; i < L28; i++) {
        const int bit = 28 * i, j = bit >> 5, sh = bit & 31;
        const uint32_t v = (uint32_t)t.l[i];
        x[j] |= v << sh;
        if (sh > 4 && j + 1 < 13) x[j + 1] |= v >> (32 - sh);
    }
    // subtract 8p, 4p, 2p, p where they fit
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        uint32_t d[13], borrow = 0;
#pragma unroll
        for (int j = 0; j < 13; j++) {
            // word j of (p << s), p in 32-bit words
            uint32_t pw = j < 12 ? FqParams::P[j] << s : 0u;
            if (s > 0 && j > 0) pw |= FqParams::P[j - 1] >> (32 - s);
            d[j] = subb(x[j], pw, borrow);
        }
#pragma unroll
        for (int j = 0; j < 13; j++) x[j] = borrow ? x[j] : d[j];
    }
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = x[j];
}

DR_DEV bool is_zero_mod_p(const Fq28& a) {            // exact; cold paths only
    uint32_t w[12], acc = 0;
    canon28(a, w);
#pragma unroll
    for (int j = 0; j < 12; j++) acc |= w[j];
    return acc == 0;
}
// cheap filter for "a normal value (mul output) that may be 0 mod p": its value is 0 or p, so limb 0 is 0 or p_0.
// Never misses a zero; says "maybe" for 2 of 2^28 non-zero values.
DR_DEV bool maybe_zero_normal(const Fq28& a) { return a.l[0] == 0 || a.l[0] == (int32_t)Fq28Params::P[0]; }

DR_DEV void load_words12(const uint32_t* p, uint32_t (&w)[12]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}
DR_DEV void store_words12(uint32_t* p, const uint32_t (&w)[12]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
DR_DEV Fq28 load_fq28(const uint32_t* p) {
    uint32_t w[12];
    load_words12(p, w);
    return unpack28(w);
}
DR_DEV void store_fq28(uint32_t* p, const Fq28& v) {
    uint32_t w[12];
    canon28(v, w);
    store_words12(p, w);
}

// standard form (canonical words) <-> Montgomery form
DR_DEV Fq28 to_mont28(const uint32_t (&w)[12]) { return mul(unpack28(w), Fq28::constant<Fq28Params::R2>()); }
DR_DEV void from_mont28(const Fq28& a, uint32_t (&w)[12]) {
    Fq28 one_std = Fq28::zero();
    one_std.l[0] = 1;
    canon28(mul(a, one_std), w);
}

// a^-1 (Montgomery in, Montgomery out; 0 -> 0): Bernstein-Yang division steps (divstep28.hip.h) on the canonical limbs of
// A = aR give +-A^-1 = a^-1 R^-1 as a lazy signed value (< 21 p), then one product with R^3.  (Round 1-2a: the word-wise
// binary Euclid of field.hip.h, ~137 k instructions per inversion against ~25 k.)
DR_DEV Fq28 inv(const Fq28& a) {
    uint32_t w[12];
    canon28(a, w);
    const Fq28 x = unpack28(w);
    Fq28 r;
    inv_divsteps28(Fq28Params::P, Fq28Params::N0, x.l, r.l);
    return mul(r, Fq28::constant<Fq28Params::R3>());
}

}  // namespace dr
