// Device-side group law of the twisted Edwards curves on the Ring-VRF hot path (BLS12-381 G1: g1.hip.h).
//
//  * Bandersnatch (twisted Edwards a=-5 over Fr; JubJub, a=-1, through the same templates), extended coordinates (X,Y,Z,T) — the formulas the
//    reference runs in dot_ring/curve/native_field/bandersnatch_te.pyx:127-174 (dbl/add-2008-hwcd);
//    a*A is computed as 3p - 5A with additions instead of a Montgomery multiplication.
//  * Coordinates are Fs values (fr29.hip.h: 9 signed limbs of 29 bits, lazily reduced).  Contract of every function here:
//    coordinates come in "normal" (or negated normal: limbs within (-2^29, 2^29), |value| <= 1.5 p) and go out normal.  The
//    comments give (limb bound, value bound in p) of every intermediate: a product needs limb bounds whose product is at most
//    2^59.3 and value bounds whose product is at most 35.
#pragma once
#include "fr29.hip.h"

namespace dr {

// ================================================================= Bandersnatch
struct TePoint {
    Fs x, y, z, t;
};

DR_DEV TePoint te_identity() {
    TePoint p;
    p.x = Fs::zero();
    p.y = Fs::one();
    p.z = Fs::one();
    p.t = Fs::zero();
    return p;
}

// The twisted Edwards curves over Fr the library serves: Bandersnatch (a = -5; the hot path) and JubJub (a = -1;
// dot_ring/curve/specs/jubjub.py:17-29 — same field, other coefficients, order and cofactor).  The curve is a template
// parameter of the group law, so the Bandersnatch kernels compile exactly as before.
enum { CV_BANDERSNATCH = 0, CV_JUBJUB = 1 };

// d in Montgomery form (2^261): 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7 (Bandersnatch),
// 0x2A9318E74BFA2B48F5FD9207E6BD7FD4292D7F6D37579D2601065FD6D6343EB1 (JubJub)
struct TeCurveConsts {
    static constexpr uint32_t D_BANDERSNATCH[9] = {0x1458e5f2u, 0x1ced1bb7u, 0x0c2440c6u, 0x03a6574fu, 0x06ebc6f2u, 0x05d944c8u, 0x185ecb02u, 0x1cdb6c09u, 0x006ed285u};
    static constexpr uint32_t D_JUBJUB[9] = {0x0e9ed5e8u, 0x12245679u, 0x002d9f52u, 0x03bb3367u, 0x0d9bfb3du, 0x18ebb3ccu, 0x1c29ceccu, 0x0a7b6020u, 0x0020d725u};
};
template <int CV = CV_BANDERSNATCH>
DR_DEV Fs te_d_mont() {
    return CV == CV_BANDERSNATCH ? Fs::constant<TeCurveConsts::D_BANDERSNATCH>() : Fs::constant<TeCurveConsts::D_JUBJUB>();
}

// carry for a sum of NON-NEGATIVE low limbs (normal values added up to 7 times: below 2^32 as unsigned); top limb signed
DR_DEV Fs carry_u(const Fs& a) {
    Fs r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L29 - 1; i++) {
        const uint32_t t = (uint32_t)a.l[i] + c;
        r.l[i] = (int32_t)(t & M29);
        c = t >> 29;
    }
    r.l[L29 - 1] = a.l[L29 - 1] + (int32_t)c;
    return r;
}

// -a*A + B shifted by a multiple of p, for A, B products of normal operands (limbs in [0, 2^29), values in (-0.04 p, 1.04 p)):
// Bandersnatch (a = -5): B + 5A - 3p in (-3.3 p, 3.3 p); JubJub (a = -1): B + A - p... kept as B + A (<= 2.1 p).
// Result limbs within (-2^29, 2^29).
template <int CV>
DR_DEV Fs te_b_minus_aA(const Fs& A, const Fs& B) {
    if (CV == CV_JUBJUB) return carry_u(add(A, B));
    Fs t = dbl(dbl(A));                              // limbs < 2^31 as unsigned
    t = add(add(t, A), B);                           // 6 normals: < 3 * 2^30 as unsigned
    return sub_3p(carry_u(t));
}
// a*A shifted by a multiple of p: Bandersnatch 3p - 5A in (-2.2 p, 3.2 p); JubJub -A.  Limbs within (-2^29, 2^29).
template <int CV>
DR_DEV Fs te_aA(const Fs& A) {
    if (CV == CV_JUBJUB) return neg(A);
    Fs t = add(dbl(dbl(A)), A);                      // < 5 * 2^29 as unsigned
    return neg(sub_3p(carry_u(t)));
}
// a*v for an arbitrary normal v (cold paths: curve equation checks)
template <int CV = CV_BANDERSNATCH>
DR_DEV Fs te_mul_a(const Fs& v) { return te_aA<CV>(carry(v)); }

// dbl-2008-hwcd.  WITH_T=false skips T3 (valid when the result is only doubled again).
template <bool WITH_T, int CV = CV_BANDERSNATCH>
DR_DEV TePoint te_dbl(const TePoint& p) {
    const Fs A = sqr(p.x), B = sqr(p.y);             // (2^29, 1.04)
    const Fs C = dbl(sqr(p.z));                      // (2^30, 2.1)
    const Fs E = carry(dbl(mul(p.x, p.y)));          // 2xy = (x+y)^2 - A - B: (2^29, 2.1)
    const Fs D = te_aA<CV>(A);                       // (2^29, 3.2)
    const Fs G = carry(add(D, B));                   // (2^29, 4.3)
    const Fs F = carry(sub(G, C));                   // (2^29, 6.4)
    const Fs H = sub(D, B);                          // (2^30, 4.3)
    TePoint r;
    r.x = mul(E, F);                                 // 13.4
    r.y = mul(G, H);                                 // 18.5
    r.z = mul(F, G);                                 // 27.5
    if (WITH_T) r.t = mul(E, H);
    else r.t = Fs::zero();
    return r;
}

// add-2008-hwcd, unified (also correct for doubling and for the identity).  E = X1 Y2 + Y1 X2 as ONE fused product.
template <int CV = CV_BANDERSNATCH>
DR_DEV TePoint te_add(const TePoint& p, const TePoint& q) {
    const Fs A = mul(p.x, q.x), B = mul(p.y, q.y);
    const Fs C = mul(mul(te_d_mont<CV>(), p.t), q.t);
    const Fs D = mul(p.z, q.z);
    const Fs E = mul2(p.x, q.y, p.y, q.x);           // (2^29, 1.1)
    const Fs F = sub(D, C), G = add(D, C);           // (2^29, 2.1), (2^30, 2.1)
    const Fs H = te_b_minus_aA<CV>(A, B);            // (2^29, 3.3)
    TePoint r;
    r.x = mul(E, F);
    r.y = mul(G, H);
    r.t = mul(E, H);
    r.z = mul(F, G);
    return r;
}

DR_DEV TePoint te_cneg(const TePoint& p, bool negate) {
    TePoint r = p;
    r.x = cneg(p.x, negate);
    r.t = cneg(p.t, negate);
    return r;
}

}  // namespace dr
