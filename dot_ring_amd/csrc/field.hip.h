// Device-side Montgomery prime-field arithmetic for gfx950 (CDNA4), 32-bit limbs held in VGPRs.
//
// Two instantiations:
//   Fr  — BLS12-381 scalar field = Bandersnatch base field, 8 x u32  (replaces the reference's
//         dot_ring/curve/native_field/bls12_381_scalar.c:43-264, 4 x u64 CIOS on the CPU)
//   Fq  — BLS12-381 base field, 12 x u32 (the arithmetic the reference reaches through blst at
//         dot_ring/ring_proof/pcs/kzg.py:152-175)
//
// Design notes (MI355X):
//  * everything is fully unrolled over compile-time limb counts so limbs live in VGPRs and the modulus
//    limbs become scalar literals (SGPR / inline constants) — no constant-memory traffic;
//  * multiplication is product-scanning (Comba) Montgomery: each column sums its 32x32->64 partial
//    products into a 96-bit accumulator.  One partial product = one v_mad_u64_u32 (64-bit accumulate is
//    free in that instruction) + one carry add into the third word;
//  * values are kept fully reduced (< p) between operations, like the reference, so equality tests and
//    canonical encodings need no extra normalisation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DR_DEV __device__ __forceinline__

#include "montmul_gen.hip.h"

namespace dr {

// ---------------------------------------------------------------- field parameter packs
struct FrParams {
    static constexpr int N = 8;
    // p = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    static constexpr uint32_t P[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                      0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t N0 = 0xffffffffu;  // -p^-1 mod 2^32
    // R^2 mod p, R = 2^256
    static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                       0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
    // R mod p (Montgomery one)
    static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                        0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
};

struct FqParams {
    static constexpr int N = 12;
    // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static constexpr uint32_t P[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                       0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr uint32_t N0 = 0xfffcfffdu;
    static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                        0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
    static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                         0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
};

// ---------------------------------------------------------------- carry primitives
// clang's multiprecision builtins lower to v_add_co_u32 / v_addc_co_u32 (v_sub_co / v_subb_co) chains.
DR_DEV uint32_t addc(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned int c;
    uint32_t r = __builtin_addc(a, b, carry, &c);
    carry = c;
    return r;
}
DR_DEV uint32_t subb(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned int c;
    uint32_t r = __builtin_subc(a, b, borrow, &c);
    borrow = c;
    return r;
}

template <class FP>
struct Fe {
    static constexpr int N = FP::N;
    uint32_t l[N];

    DR_DEV static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    DR_DEV static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = FP::ONE[i];
        return r;
    }
    DR_DEV bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    DR_DEV bool operator==(const Fe& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
};

// r = a - p if a >= p (a < 2p, possibly with an extra carry-out bit `hi`)
template <class FP>
DR_DEV void cond_sub_p(Fe<FP>& a, uint32_t hi) {
    constexpr int N = FP::N;
    uint32_t d[N], borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = subb(a.l[i], FP::P[i], borrow);
    bool take = (hi != 0) | (borrow == 0);
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = take ? d[i] : a.l[i];
}

template <class FP>
DR_DEV Fe<FP> add(const Fe<FP>& a, const Fe<FP>& b) {
    constexpr int N = FP::N;
    Fe<FP> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc(a.l[i], b.l[i], carry);
    cond_sub_p(r, carry);
    return r;
}

template <class FP>
DR_DEV Fe<FP> sub(const Fe<FP>& a, const Fe<FP>& b) {
    constexpr int N = FP::N;
    Fe<FP> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = subb(a.l[i], b.l[i], borrow);
    uint32_t mask = 0u - borrow, carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc(r.l[i], FP::P[i] & mask, carry);
    return r;
}

template <class FP>
DR_DEV Fe<FP> neg(const Fe<FP>& a) {
    constexpr int N = FP::N;
    Fe<FP> r;
    uint32_t borrow = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    uint32_t mask = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = subb(FP::P[i] & mask, a.l[i], borrow);
    return r;
}

template <class FP>
DR_DEV Fe<FP> dbl(const Fe<FP>& a) { return add(a, a); }

// Montgomery product a*b*R^-1 mod p (inputs and output < p): generated single-statement asm, see gen_montmul.py
template <class FP>
DR_DEV Fe<FP> mul(const Fe<FP>& a, const Fe<FP>& b) {
    Fe<FP> r;
    uint32_t top;
    if constexpr (FP::N == 8) top = montmul8_asm<FP>(r.l, a.l, b.l);
    else top = montmul12_asm<FP>(r.l, a.l, b.l);
    cond_sub_p(r, top);
    return r;
}

template <class FP>
DR_DEV Fe<FP> sqr(const Fe<FP>& a) { return mul(a, a); }

template <class FP>
DR_DEV Fe<FP> to_mont(const Fe<FP>& a) {
    Fe<FP> r2;
#pragma unroll
    for (int i = 0; i < FP::N; i++) r2.l[i] = FP::R2[i];
    return mul(a, r2);
}
template <class FP>
DR_DEV Fe<FP> from_mont(const Fe<FP>& a) {
    Fe<FP> one = Fe<FP>::zero();
    one.l[0] = 1;
    return mul(a, one);
}

// a^(p-2) by square-and-multiply over the (compile-time) exponent: ~N*32 squarings + ~N*16 multiplications.
template <class FP>
DR_DEV Fe<FP> inv_fermat(const Fe<FP>& a) {
    constexpr int N = FP::N;
    // table a^0..a^15 would cost 16*N VGPRs; use plain square-and-multiply, MSB first, exponent from constants
    Fe<FP> r = Fe<FP>::one();
    bool started = false;
#pragma unroll 1
    for (int i = N - 1; i >= 0; i--) {
        // exponent limb of p-2 (only limb 0 differs from p; p[0] >= 2 for both fields' low limbs? Fr: p[0]=1 -> borrow)
        uint32_t e = FP::P[i];
        if (FP::P[0] >= 2) {
            if (i == 0) e -= 2;
        } else {
            // p[0] == 1: p-2 = (p - 1) - 1 -> limb0 = 0xffffffff, borrow through zero limbs above
            if (i == 0) e = 0xffffffffu;
            else if (i == 1) e = FP::P[1] - 1;   // Fr: p[1] = 0xffffffff, no further borrow
        }
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            if (started) r = sqr(r);
            if ((e >> bit) & 1) {
                r = started ? mul(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

// Inversion by a branch-free binary extended Euclid on the stored (Montgomery) value A = aR:
//   invariants  x1*A = u,  x2*A = v  (mod p);  start u = A, v = p, x1 = 1, x2 = 0;
//   step: if u is odd and u < v swap (u, x1) <-> (v, x2); if u is odd subtract: u -= v, x1 -= x2; halve u and x1.
// Each step removes a bit from u or v, so 2*bits steps end with u = 0, v = 1, x2 = A^-1 = a^-1 R^-1; one Montgomery
// product with R^3 gives a^-1 R.  ~16*N instructions per step: Fr 61 k, Fq 137 k instructions against 116 k / 370 k
// for the Fermat power — every affine conversion in the latency-bound kernels is an inversion.  0 -> 0.
template <class FP>
DR_DEV void inv_words(const uint32_t (&a)[FP::N], uint32_t (&out)[FP::N]) {      // out = a^-1 mod p on plain words (a < p)
    constexpr int N = FP::N;
    uint32_t u[N], v[N], x1[N], x2[N];
#pragma unroll
    for (int i = 0; i < N; i++) { u[i] = a[i]; v[i] = FP::P[i]; x1[i] = i == 0 ? 1u : 0u; x2[i] = 0u; }
    int bits = 32 * N;
    while (bits > 0 && !((FP::P[(bits - 1) >> 5] >> ((bits - 1) & 31)) & 1u)) bits--;      // bit length of p (compile-time)
#pragma unroll 1
    for (int it = 0; it < 2 * bits; it++) {
        const uint32_t odd = u[0] & 1u;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) (void)subb(u[i], v[i], borrow);
        const uint32_t msw = 0u - (odd & borrow);                // all ones when u is odd and u < v: masked XOR swap
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t tu = (u[i] ^ v[i]) & msw, tx = (x1[i] ^ x2[i]) & msw;
            u[i] ^= tu;  v[i] ^= tu;
            x1[i] ^= tx; x2[i] ^= tx;
        }
        const uint32_t m = 0u - odd;                             // all ones when u is odd
        borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = subb(u[i], v[i] & m, borrow);
        borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x1[i] = subb(x1[i], x2[i] & m, borrow);
        uint32_t carry = 0;
        const uint32_t mp = 0u - borrow;                         // went negative: add p back
#pragma unroll
        for (int i = 0; i < N; i++) x1[i] = addc(x1[i], FP::P[i] & mp, carry);
        // halve u; halve x1 modulo p (add p first when odd; the sum may carry into bit 32N)
#pragma unroll
        for (int i = 0; i < N - 1; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 31);
        u[N - 1] >>= 1;
        const uint32_t mo = 0u - (x1[0] & 1u);
        carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x1[i] = addc(x1[i], FP::P[i] & mo, carry);
#pragma unroll
        for (int i = 0; i < N - 1; i++) x1[i] = (x1[i] >> 1) | (x1[i + 1] << 31);
        x1[N - 1] = (x1[N - 1] >> 1) | (carry << 31);
    }
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = x2[i];
}
template <class FP>
DR_DEV Fe<FP> inv(const Fe<FP>& a) {
    Fe<FP> r, r2;
    inv_words<FP>(a.l, r.l);
#pragma unroll
    for (int i = 0; i < FP::N; i++) r2.l[i] = FP::R2[i];
    return mul(r, mul(r2, r2));                                  // * R^3 (Montgomery): A^-1 -> a^-1 R
}

using Fr = Fe<FrParams>;
using Fq = Fe<FqParams>;

}  // namespace dr
