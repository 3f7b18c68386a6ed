// Arithmetic bodies of the ring prover's hot kernels, written once over an abstract field type F with the operations of
// fr29.hip.h (add, sub, mul, mul2, carry, carry_u, dbl, neg, F::one()).  On the device F = dr::Fs.  On the host
// tests/native/ring_bounds_check.cpp instantiates the same bodies with an INTERVAL type that tracks, for every intermediate, the
// range of its low limbs and of its value in units of p, and asserts the preconditions of each operation (a product needs limb
// bounds whose product is at most 2^59.3 and value bounds whose product is at most 35; a carry needs limbs within int32; ...):
// the lazy-reduction bookkeeping of these kernels is checked for the WORST case, not for the inputs a parity test happens to draw.
//
// Plain C++ over F: compiles for the device (hipcc) and for the host (g++).
#pragma once

#if defined(__HIPCC__)
#define DR_BODY_FN __device__ __forceinline__
#else
#define DR_BODY_FN static inline
#endif

namespace dr {

// K7: sum_k alpha_k c_k at one point of the 4N domain (constraints.py:83-151, proof_builder.py:175-180).
//   witness values b, ip, ip_n (accip at i and i + 4), x1, x3, y1, y3: the forward NTT's raw output (carried, |value| < 0.51 p)
//   per-ring tables x2, y2, s (px, py, selector), l0, ln (Lagrange rows), nl (x - w^(N-4)): canonical limbs (value in [0, p))
//   a0..a6: the alphas, A and B: k_ring_alpha_aux's two per-proof scalars — normal (products)
// 23 products, ten of them as five fused a b + c d (one reduction each).  The selector form b u + (1 - b) v is v + b (u - v); the three boundary constraints share
// (L0 + Llast)(a5 x + a6 y + a7 accip) - L0 A - Llast B.  Result: limbs within (-2^30, 2^30 + 2^29), |value| < 3.3 p — the
// inverse NTT carries both operands of its first stage on load.
template <int CV, class F>
DR_BODY_FN F body_constraints(const F& b, const F& ip, const F& ip_n, const F& x1, const F& x3, const F& y1, const F& y3, const F& x2,
                              const F& y2, const F& s, const F& l0, const F& ln, const F& nl, const F& a0, const F& a1, const F& a2,
                              const F& a3, const F& a4, const F& a5, const F& a6, const F& A, const F& B) {
    const F omb = sub(F::one(), b);
    const F x1y1 = mul(x1, y1), x2y2 = mul(x2, y2);
    // c1 = (accip' - accip - b s) nl        (the common factor nl of c1..c3 is applied once, after the alphas)
    const F c1 = sub(sub(ip_n, ip), mul(b, s));
    // c2 = (b (x3 (y1 y2 + a x1 x2) - (x1 y1 + x2 y2)) + (1 - b)(x3 - x1)) nl;  y1 y2 + a x1 x2 is ONE fused product with (a x1) as
    // an operand: a = -5: -(4 x1 + x1), carried (limbs back below 2^29, |value| < 2.6 p); a = -1: -x1
    const F ax1 = CV == 1 ? neg(x1) : neg(carry_u(add(dbl(dbl(x1)), x1)));
    const F t2 = sub(mul(x3, mul2(y1, y2, ax1, x2)), add(x1y1, x2y2));
    const F v2 = sub(x3, x1);
    const F c2 = add(v2, mul(b, carry(sub(t2, v2))));
    // c3 = (b (y3 (x1 y2 - x2 y1) - (x1 y1 - x2 y2)) + (1 - b)(y3 - y1)) nl;  x1 y2 - x2 y1 fused
    const F t3 = sub(mul(y3, mul2(x1, y2, neg(x2), y1)), sub(x1y1, x2y2));
    const F v3 = sub(y3, y1);
    const F c3 = add(v3, mul(b, carry(sub(t3, v3))));
    // (a0 c1 + a1 c2 + a2 c3) nl: the first two products fused (operands carried: limbs below 2^29)
    F acc = add(mul2(a0, carry(c1), a1, carry(c2)), mul(a2, c3));
    acc = mul(acc, nl);
    // c4 = b (1 - b)
    acc = add(acc, mul(a3, mul(b, omb)));
    // a5 c5 + a6 c6 + a7 c7 = (L0 + Llast)(a5 x1 + a6 y1 + a7 accip) - (L0 A + Llast B), the last two products fused
    const F lin = add(mul2(a4, x1, a5, y1), mul(a6, ip));
    return add(acc, sub(mul(add(l0, ln), carry(lin)), mul2(l0, A, ln, B)));
}

// K8 quotient coefficient: sum_d tail_d fold_d, fold_d = the sum of up to four STANDARD-form coefficients (canonical limbs), tail_d
// Montgomery: the products are standard form.  Result: sum of two fused products, value in (-0.1 p, 2.3 p).
template <class F>
DR_BODY_FN F body_quotient(const F& t0, const F& f0, const F& t1, const F& f1, const F& t2, const F& f2, const F& t3, const F& f3) {
    return add(mul2(t0, carry(f0), t1, carry(f1)), mul2(t2, carry(f2), t3, carry(f3)));
}

// one Horner step on a lazy accumulator: acc x + c with x normal (Montgomery), c canonical limbs of a standard-form coefficient;
// acc stays within limbs [0, 2^30), value (-0.04 p, 2.04 p)
template <class F>
DR_BODY_FN F body_horner(const F& acc, const F& x, const F& c) {
    return add(mul(acc, x), c);
}

// nu-aggregation of eight standard-form coefficients with Montgomery scalars: four fused products, value in (-0.2 p, 4.2 p)
template <class F>
DR_BODY_FN F body_agg8(const F (&nu)[8], const F (&c)[8]) {
    const F lo = add(mul2(nu[0], c[0], nu[1], c[1]), mul2(nu[2], c[2], nu[3], c[3]));
    const F hi = add(mul2(nu[4], c[4], nu[5], c[5]), mul2(nu[6], c[6], nu[7], c[7]));
    return add(carry_u(lo), hi);                     // limbs below 2^29 + 2^30: the signed carry of canon29 takes it
}
// linearisation polynomial coefficient: k0 c0 + k1 c1 + k2 c2
template <class F>
DR_BODY_FN F body_lin3(const F& k0, const F& c0, const F& k1, const F& c1, const F& k2, const F& c2) {
    return add(mul2(k0, c0, k1, c1), mul(k2, c2));
}

// the value growth of the radix-2 network (kernels_ntt.hip.h): one butterfly, `carry_u` as the kernels schedule it
template <class F>
DR_BODY_FN void body_butterfly(F& u, F& v, const F& w, bool trivial, bool carry_u_too) {
    const F t = trivial ? (carry_u_too ? carry(v) : v) : mul(w, carry(v));
    if (carry_u_too) u = carry(u);
    v = sub(u, t);
    u = add(u, t);
}

}  // namespace dr
