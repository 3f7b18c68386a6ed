// BLS12-381 scalar field (= the base field of Bandersnatch and JubJub) for gfx950 in an UNSATURATED radix: 9 signed limbs of
// 29 bits, lazy reduction — the same construction as fq28.hip.h, for the twisted Edwards kernels (seam A).
//
// With saturated 32-bit limbs a product is 128 v_mad_u64_u32 + 128 carry instructions + ... = 305 instructions
// (field.hip.h, montmul_gen.hip.h) and a squaring costs the same.  With 29-bit limbs a signed 64-bit accumulator absorbs a whole
// column (18 products below 2^58), a partial product is ONE v_mad_i64_i32: 206 instructions per product, 178 per squaring,
// 287 for a fused a b + c d; additions and subtractions are 9 plain v_add / v_sub.  p = 1 mod 2^32, so -1/p mod 2^29 = -1 and
// the Montgomery factor of a column is a negation.
//
// Value of an element: sum l[i] * 2^(29 i), limbs SIGNED.  Montgomery form x -> x * 2^261 mod p.  R / p = 70.7: far less
// room than Fq has (2^11), so the rules are tighter than fq28's:
//   "normal" : limbs 0..7 in [0, 2^29), limb 8 small and signed — what mul() / sqr() / mul2() return; value in (-p/2, 1.5 p)
//              (in (-0.04 p, 1.04 p) when the operands were normal)
//   mul(a, b): 9 max|a_i| max|b_j| + 9 * 2^58 < 2^63, i.e. max|a_i| max|b_j| <= 2^59.3 (2^30 x 2^29, or 1.6 * 2^29 twice), and
//              |a| |b| <= 35 p^2 for a normal result (beyond that the result is still correct, just wider by |a b| / (70.7 p))
//   sqr(a)   : max|a_i| <= 1.6 * 2^29;   mul2(a, b, c, d): all limbs below 2^29 in magnitude, |a b + c d| <= 35 p^2
//   carry()  : limbs back into [0, 2^29) (value unchanged) — needed after ~3 additions of normal values or before a product of
//              two sums
// Memory / LDS / shuffles between lanes keep 8 x u32 canonical words (pack / unpack): layouts are those of field.hip.h.
#pragma once
#include "field.hip.h"
#include "montmul28_gen.hip.h"
#include "divstep28.hip.h"

namespace dr {

constexpr int L29 = 9;
constexpr uint32_t M29 = 0x1fffffffu;

struct Fr29Params {
    static constexpr uint32_t P[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    static constexpr uint32_t N0 = 0x1fffffffu;                     // -p^-1 mod 2^29 = -1
    static constexpr uint32_t ONE[9] = {0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu};   // 2^261 mod p
    static constexpr uint32_t R2[9] = {0x0a71b3c0u, 0x1d32207eu, 0x1663d999u, 0x1c5abc93u, 0x03b58c44u, 0x0be37438u, 0x0829f771u, 0x1660139eu, 0x0027fd91u};    // R^2 mod p
    static constexpr uint32_t R3[9] = {0x19d7065du, 0x0020db85u, 0x16122e43u, 0x0edb1ff8u, 0x0fda6124u, 0x0517ac72u, 0x12e6a522u, 0x19d54edau, 0x0009750bu};    // R^3 mod p
    static constexpr uint32_t P8[9] = {0x00000008u, 0x1fffffc0u, 0x1cb7fdffu, 0x1a402fffu, 0x0c02a9deu, 0x00202687u, 0x0906673bu, 0x13299d7du, 0x039f6d3au};    // 8 p
    // between this Montgomery form (2^261) and field.hip.h's (2^256): mul(x 2^256, 2^266) = x 2^261; mul(x 2^261, 2^256) = x 2^256
    static constexpr uint32_t K266[9] = {0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu};
    static constexpr uint32_t K256[9] = {0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u};
    static constexpr uint32_t P3[9] = {0x00000003u, 0x1fffffe8u, 0x1ec4ff3fu, 0x11d811ffu, 0x1880ffb3u, 0x040c0e72u, 0x1f6266b6u, 0x1f2f9b0eu, 0x015bc8f5u};    // 3 p
};

struct Fs {                          // an element of Fr in signed 29-bit limbs
    int32_t l[L29];
    DR_DEV static Fs zero() {
        Fs r;
#pragma unroll
        for (int i = 0; i < L29; i++) r.l[i] = 0;
        return r;
    }
    DR_DEV static Fs one() {
        Fs r;
#pragma unroll
        for (int i = 0; i < L29; i++) r.l[i] = (int32_t)Fr29Params::ONE[i];
        return r;
    }
    template <const uint32_t (&C)[9]>
    DR_DEV static Fs constant() {
        Fs r;
#pragma unroll
        for (int i = 0; i < L29; i++) r.l[i] = (int32_t)C[i];
        return r;
    }
};

DR_DEV Fs add(const Fs& a, const Fs& b) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
DR_DEV Fs sub(const Fs& a, const Fs& b) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
DR_DEV Fs dbl(const Fs& a) { return add(a, a); }
DR_DEV Fs neg(const Fs& a) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = -a.l[i];
    return r;
}
DR_DEV Fs cneg(const Fs& a, bool negate) {
    const int32_t s = negate ? -1 : 0;
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = (a.l[i] ^ s) - s;
    return r;
}
// limbs 0..7 into [0, 2^29), the rest into the signed top limb.  Value unchanged.
DR_DEV Fs carry(const Fs& a) {
    Fs r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L29 - 1; i++) {
        const int32_t t = a.l[i] + c;
        r.l[i] = t & (int32_t)M29;
        c = t >> 29;
    }
    r.l[L29 - 1] = a.l[L29 - 1] + c;
    return r;
}
// a - 3p: centres a value known to lie in (0, 6p) around zero (limbs of a normal -> limbs in (-2^29, 2^29))
DR_DEV Fs sub_3p(const Fs& a) { return sub(a, Fs::constant<Fr29Params::P3>()); }

DR_DEV Fs mul(const Fs& a, const Fs& b) {
    Fs r;
    montmul9x29_asm<Fr29Params>(r.l, a.l, b.l);
    return r;
}
DR_DEV Fs sqr(const Fs& a) {
    Fs r;
    montsqr9x29_asm<Fr29Params>(r.l, a.l);
    return r;
}
DR_DEV Fs mul2(const Fs& a, const Fs& b, const Fs& c, const Fs& d) {      // a b + c d, one reduction
    Fs r;
    montmul2_9x29_asm<Fr29Params>(r.l, a.l, b.l, c.l, d.l);
    return r;
}

// ---------------------------------------------------------------- 8 x u32 words <-> limbs
// canonical words (< p, < 2^255) reinterpreted in radix 2^29: a normal value
DR_DEV Fs unpack29(const uint32_t (&w)[8]) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) {
        const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
        uint32_t v = w[j] >> sh;
        if (sh > 3 && j + 1 < 8) v |= w[j + 1] << (32 - sh);
        r.l[i] = (int32_t)(v & M29);
    }
    return r;
}
// any value with |value| < 8 p -> the canonical representative in [0, p), as 8 words
DR_DEV void canon29(const Fs& a, uint32_t (&w)[8]) {
    using FP = Fr29Params;
    const Fs c = carry(a);
    // + 8p (pre-carried limbs) makes the value positive; second carry pass unsigned
    uint32_t u[L29], cy = 0;
#pragma unroll
    for (int i = 0; i < L29; i++) {
        u[i] = (uint32_t)c.l[i] + FP::P8[i] + cy;                   // < 2^29 + 2^29 + 1; the top limb stays >= 0 for |a| < 8p
        if (i < L29 - 1) { cy = u[i] >> 29; u[i] &= M29; }
    }
    // pack: value < 16 p < 2^259: nine words
    uint32_t x[9];
#pragma unroll
    for (int j = 0; j < 9; j++) x[j] = 0;
#pragma unroll
    for (int i = 0; i < L29; i++) {
        const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
        x[j] |= u[i] << sh;
        if (sh > 3 && j + 1 < 9) x[j + 1] |= u[i] >> (32 - sh);
    }
    // subtract 8p, 4p, 2p, p where they fit
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        uint32_t d[9], borrow = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            uint32_t pw = j < 8 ? FrParams::P[j] << s : 0u;        // word j of (p << s)
            if (s > 0 && j > 0) pw |= FrParams::P[j - 1] >> (32 - s);
            d[j] = subb(x[j], pw, borrow);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = borrow ? x[j] : d[j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = x[j];
}

// ---------------------------------------------------------------- raw 9-word records
// Prover-internal buffers (NTT tiles, the 4N-domain evaluations between the NTT and the constraint kernel, per-proof scalars)
// keep the nine limbs as they are — 36 bytes per element, no canonicalisation on either side.  What a record holds (normal,
// carried, value bound) is part of the producing kernel's contract.
DR_DEV Fs fs_load9(const uint32_t* p) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = (int32_t)p[i];
    return r;
}
DR_DEV void fs_store9(uint32_t* p, const Fs& v) {
#pragma unroll
    for (int i = 0; i < L29; i++) p[i] = (uint32_t)v.l[i];
}
// kernel-argument copy of an Fs value
struct FsArg {
    int32_t l[L29];
};
DR_DEV Fs from_arg(const FsArg& a) {
    Fs r;
#pragma unroll
    for (int i = 0; i < L29; i++) r.l[i] = a.l[i];
    return r;
}
// Any lazy value with |value| < 30 p and limbs within int32 -> carried limbs (0..7 in [0, 2^29), signed top limb) and
// |value| < 0.51 p: q = round(value / p) from the top limb (p / 2^232 = 7597479.33; the low limbs add less than one unit of it),
// then value - q p with 64-bit limb arithmetic (|q| P[i] does not fit 32 bits).  ~75 instructions: where a growing sum leaves a
// kernel without passing through a product (the forward NTT's last stage).
DR_DEV Fs reduce_small(const Fs& a) {
    const Fs c = carry(a);
    const int32_t q = __float2int_rn((float)c.l[L29 - 1] * (1.0f / 7597479.5f));
    Fs r;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < L29 - 1; i++) {
        const int64_t t = (int64_t)c.l[i] - (int64_t)q * (int64_t)Fr29Params::P[i] + cy;
        r.l[i] = (int32_t)(t & (int64_t)M29);
        cy = t >> 29;
    }
    r.l[L29 - 1] = c.l[L29 - 1] - q * (int32_t)Fr29Params::P[L29 - 1] + (int32_t)cy;
    return r;
}

// the same for a value known to lie in (-p, 3p) — every product of normal operands, and a product plus a canonical value: + p,
// then 2p and p come off where they fit
// (two conditional subtractions instead of four: ~70 instructions against ~120)
DR_DEV void canon29_small(const Fs& a, uint32_t (&w)[8]) {
    using FP = Fr29Params;
    const Fs c = carry(a);
    uint32_t u[L29], cy = 0;
#pragma unroll
    for (int i = 0; i < L29; i++) {
        u[i] = (uint32_t)c.l[i] + FP::P[i] + cy;
        if (i < L29 - 1) { cy = u[i] >> 29; u[i] &= M29; }
    }
    uint32_t x[9];
#pragma unroll
    for (int j = 0; j < 9; j++) x[j] = 0;
#pragma unroll
    for (int i = 0; i < L29; i++) {
        const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
        x[j] |= u[i] << sh;
        if (sh > 3 && j + 1 < 9) x[j + 1] |= u[i] >> (32 - sh);
    }
#pragma unroll
    for (int s = 1; s >= 0; s--) {
        uint32_t d[9], borrow = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            uint32_t pw = j < 8 ? FrParams::P[j] << s : 0u;
            if (s > 0 && j > 0) pw |= FrParams::P[j - 1] >> (32 - s);
            d[j] = subb(x[j], pw, borrow);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = borrow ? x[j] : d[j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = x[j];
}

// The 8-word containers of field.hip.h (Fr = Fe<FrParams>) carry values between lanes, LDS and memory; which form the words
// are in is said by the function: pack / unpack keep the Montgomery form (2^261), fs_from_std / fs_to_std convert standard form.
DR_DEV Fr pack(const Fs& a) {
    Fr r;
    canon29(a, r.l);
    return r;
}
DR_DEV Fs unpack(const Fr& w) { return unpack29(w.l); }
DR_DEV Fs fs_from_std(const Fr& std_words) { return mul(unpack29(std_words.l), Fs::constant<Fr29Params::R2>()); }
DR_DEV Fr fs_to_std(const Fs& a) {
    Fs one_std = Fs::zero();
    one_std.l[0] = 1;
    return pack(mul(a, one_std));
}
// values in the 2^256 Montgomery form of field.hip.h (per-ring tables, kernel arguments, columns shared with the NTT kernels)
DR_DEV Fs from_mont256(const Fr& w) { return mul(unpack29(w.l), Fs::constant<Fr29Params::K266>()); }
DR_DEV Fr to_mont256(const Fs& a) { return pack(mul(a, Fs::constant<Fr29Params::K256>())); }
// exact tests for |value| < 4 p.  Canonicalising costs ~120 instructions, so a filter goes first: value = k p with |k| <= 4 and
// p = 1 mod 2^29, so limb 0 of a multiple of p is k mod 2^29 — anything else is non-zero (the square-root loop compares after
// every squaring; 9 of 2^29 non-zero values pass the filter and take the exact test).
DR_DEV bool is_zero(const Fs& a) {
    const uint32_t d = (uint32_t)a.l[0] & M29;
    if (d > 4u && d < M29 - 3u) return false;
    return pack(a).is_zero();
}
DR_DEV bool equal(const Fs& a, const Fs& b) { return is_zero(sub(a, b)); }

// a^-1 (Montgomery in and out; 0 -> 0): division steps on the canonical limbs of A = aR give +-A^-1 as a lazy signed value
// below 14 p (26 batches of 29 steps cover the 738 steps a 255-bit modulus can need), then one product with R^3
DR_DEV Fs inv(const Fs& a) {
    const Fs x = unpack(pack(a));
    Fs r;
    inv_divsteps<9, 29, 26>(Fr29Params::P, Fr29Params::N0, x.l, r.l);
    return mul(r, Fs::constant<Fr29Params::R3>());
}

}  // namespace dr
