// Device-side group law for the two curves on the Ring-VRF hot path.
//
//  * Bandersnatch (twisted Edwards a=-5 over Fr; JubJub, a=-1, through the same templates), extended coordinates (X,Y,Z,T) — the formulas the
//    reference runs in dot_ring/curve/native_field/bandersnatch_te.pyx:127-174 (dbl/add-2008-hwcd);
//    a*A is computed as -(4A+A) instead of a Montgomery multiplication.
//  * BLS12-381 G1 (y^2 = x^3 + 4 over Fq), XYZZ coordinates (X,Y,ZZ,ZZZ) for bucket sums:
//    mixed add 8M+2S, full add 12M+2S, doubling 6M+4S (EFD "xyzz": madd-2008-s, add-2008-s, dbl-2008-s-1).
//    The reference reaches this arithmetic through blst (dot_ring/ring_proof/pcs/kzg.py:147-175).
#pragma once
#include "field.hip.h"

namespace dr {

// ================================================================= Bandersnatch
struct TePoint {
    Fr x, y, z, t;
};

DR_DEV TePoint te_identity() {
    TePoint p;
    p.x = Fr::zero();
    p.y = Fr::one();
    p.z = Fr::one();
    p.t = Fr::zero();
    return p;
}

// The twisted Edwards curves over Fr the library serves: Bandersnatch (a = -5; the hot path) and JubJub (a = -1;
// dot_ring/curve/specs/jubjub.py:17-29 — same field, other coefficients, order and cofactor).  The curve is a template
// parameter of the group law, so the Bandersnatch kernels compile exactly as before.
enum { CV_BANDERSNATCH = 0, CV_JUBJUB = 1 };

// d in Montgomery form: 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7 (Bandersnatch),
// 0x2A9318E74BFA2B48F5FD9207E6BD7FD4292D7F6D37579D2601065FD6D6343EB1 (JubJub)
template <int CV = CV_BANDERSNATCH>
DR_DEV Fr te_d_mont() {
    Fr d;
    if (CV == CV_BANDERSNATCH) {
        d.l[0] = 0x47a2c730u; d.l[1] = 0xa8dced1bu; d.l[2] = 0xad3cccc7u; d.l[3] = 0x381c065au;
        d.l[4] = 0x188351f8u; d.l[5] = 0x53ff52e1u; d.l[6] = 0x990fe940u; d.l[7] = 0x362e8d63u;
    } else {
        d.l[0] = 0xb974f6b0u; d.l[1] = 0x2a522455u; d.l[2] = 0x0d9acab3u; d.l[3] = 0xfc6cc9efu;
        d.l[4] = 0xc27628d1u; d.l[5] = 0x7a08fb94u; d.l[6] = 0xfe0e262eu; d.l[7] = 0x57f8f6a8u;
    }
    return d;
}

// a*v for a = -5 (Bandersnatch) or a = -1 (JubJub)
template <int CV = CV_BANDERSNATCH>
DR_DEV Fr te_mul_a(const Fr& v) {
    if (CV == CV_JUBJUB) return neg(v);
    Fr t = dbl(v);
    t = dbl(t);
    t = add(t, v);
    return neg(t);
}

// dbl-2008-hwcd.  WITH_T=false skips T3 (valid when the result is only doubled again).
template <bool WITH_T, int CV = CV_BANDERSNATCH>
DR_DEV TePoint te_dbl(const TePoint& p) {
    Fr A = sqr(p.x), B = sqr(p.y);
    Fr C = dbl(sqr(p.z));
    Fr D = te_mul_a<CV>(A);
    Fr E = sub(sub(sqr(add(p.x, p.y)), A), B);
    Fr G = add(D, B), F = sub(G, C), H = sub(D, B);
    TePoint r;
    r.x = mul(E, F);
    r.y = mul(G, H);
    r.z = mul(F, G);
    if (WITH_T) r.t = mul(E, H);
    else r.t = Fr::zero();
    return r;
}

// add-2008-hwcd, unified (also correct for doubling and for the identity).
template <int CV = CV_BANDERSNATCH>
DR_DEV TePoint te_add(const TePoint& p, const TePoint& q) {
    Fr A = mul(p.x, q.x), B = mul(p.y, q.y);
    Fr C = mul(mul(te_d_mont<CV>(), p.t), q.t);
    Fr D = mul(p.z, q.z);
    Fr E = sub(sub(mul(add(p.x, p.y), add(q.x, q.y)), A), B);
    Fr F = sub(D, C), G = add(D, C), H = sub(B, te_mul_a<CV>(A));
    TePoint r;
    r.x = mul(E, F);
    r.y = mul(G, H);
    r.t = mul(E, H);
    r.z = mul(F, G);
    return r;
}

DR_DEV TePoint te_cneg(const TePoint& p, bool negate) {
    TePoint r = p;
    Fr nx = neg(p.x), nt = neg(p.t);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.l[i] = negate ? nx.l[i] : p.x.l[i];
        r.t.l[i] = negate ? nt.l[i] : p.t.l[i];
    }
    return r;
}

// ================================================================= BLS12-381 G1
struct G1Affine {   // Montgomery form; (0,0) encodes the point at infinity (not on the curve: b = 4)
    Fq x, y;
    DR_DEV bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

struct G1Xyzz {     // x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 ; ZZ == 0 encodes infinity
    Fq x, y, zz, zzz;
    DR_DEV bool is_inf() const { return zz.is_zero(); }
};

DR_DEV G1Xyzz g1_inf() {
    G1Xyzz r;
    r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
    return r;
}

DR_DEV G1Xyzz g1_from_affine(const G1Affine& p) {
    G1Xyzz r;
    if (p.is_inf()) return g1_inf();
    r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one();
    return r;
}

// 2*(affine P) -> XYZZ   (mdbl-2008-s-1, a = 0).  Inlined into a cold block of g1_madd: an out-of-line call
// would take its operands by address and push the hot loop's registers through scratch memory every iteration
// (measured: 12 GB of scratch traffic per 2^20-point launch).
DR_DEV G1Xyzz g1_dbl_affine(const G1Affine& p) {
    Fq U = dbl(p.y);
    Fq V = sqr(U);
    Fq W = mul(U, V);
    Fq S = mul(p.x, V);
    Fq X2 = sqr(p.x);
    Fq M = add(dbl(X2), X2);
    G1Xyzz r;
    r.x = sub(sub(sqr(M), S), S);
    r.y = sub(mul(M, sub(S, r.x)), mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2*P in XYZZ (dbl-2008-s-1, a = 0).  Out of line (reduction kernels call it from several sites; keeps
// their code inside the instruction cache).
__device__ __noinline__ G1Xyzz g1_dbl(const G1Xyzz& p) {
    if (p.is_inf()) return p;
    Fq U = dbl(p.y);
    Fq V = sqr(U);
    Fq W = mul(U, V);
    Fq S = mul(p.x, V);
    Fq X2 = sqr(p.x);
    Fq M = add(dbl(X2), X2);
    G1Xyzz r;
    r.x = sub(sub(sqr(M), S), S);
    r.y = sub(mul(M, sub(S, r.x)), mul(W, p.y));
    r.zz = mul(V, p.zz);
    r.zzz = mul(W, p.zzz);
    return r;
}

// acc + (affine q)   (madd-2008-s) with the exceptional cases made explicit
DR_DEV G1Xyzz g1_madd(const G1Xyzz& acc, const G1Affine& q) {
    if (q.is_inf()) return acc;
    if (acc.is_inf()) return g1_from_affine(q);
    Fq U2 = mul(q.x, acc.zz);
    Fq S2 = mul(q.y, acc.zzz);
    Fq P = sub(U2, acc.x);
    Fq R = sub(S2, acc.y);
    if (__builtin_expect(P.is_zero(), 0)) {
        if (R.is_zero()) return g1_dbl_affine(q);
        return g1_inf();
    }
    Fq PP = sqr(P);
    Fq PPP = mul(P, PP);
    Fq Q = mul(acc.x, PP);
    G1Xyzz r;
    r.x = sub(sub(sub(sqr(R), PPP), Q), Q);
    r.y = sub(mul(R, sub(Q, r.x)), mul(acc.y, PPP));
    r.zz = mul(acc.zz, PP);
    r.zzz = mul(acc.zzz, PPP);
    return r;
}

// p + q, both XYZZ (add-2008-s) with the exceptional cases made explicit.  Out of line, as g1_dbl.
__device__ __noinline__ G1Xyzz g1_add(const G1Xyzz& p, const G1Xyzz& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fq U1 = mul(p.x, q.zz), U2 = mul(q.x, p.zz);
    Fq S1 = mul(p.y, q.zzz), S2 = mul(q.y, p.zzz);
    Fq P = sub(U2, U1);
    Fq R = sub(S2, S1);
    if (P.is_zero()) {
        if (R.is_zero()) return g1_dbl(p);
        return g1_inf();
    }
    Fq PP = sqr(P);
    Fq PPP = mul(P, PP);
    Fq Q = mul(U1, PP);
    G1Xyzz r;
    r.x = sub(sub(sub(sqr(R), PPP), Q), Q);
    r.y = sub(mul(R, sub(Q, r.x)), mul(S1, PPP));
    r.zz = mul(mul(p.zz, q.zz), PP);
    r.zzz = mul(mul(p.zzz, q.zzz), PPP);
    return r;
}

// Fully inlined variants for kernels that must stay free of scratch memory: an out-of-line call passes its XYZZ
// operands through scratch, and a kernel that reserves scratch loses resident waves (measured on the comb kernel:
// -25 % when its epilogue stopped calling g1_add).  One call site per kernel, operands muxed by the caller.
DR_DEV G1Xyzz g1_dbl_inl(const G1Xyzz& p) {
    Fq U = dbl(p.y);
    Fq V = sqr(U);
    Fq W = mul(U, V);
    Fq S = mul(p.x, V);
    Fq X2 = sqr(p.x);
    Fq M = add(dbl(X2), X2);
    G1Xyzz r;
    r.x = sub(sub(sqr(M), S), S);
    r.y = sub(mul(M, sub(S, r.x)), mul(W, p.y));
    r.zz = mul(V, p.zz);
    r.zzz = mul(W, p.zzz);
    return r;                                   // an infinite p (zz = 0) stays infinite: zz = V * 0
}
DR_DEV G1Xyzz g1_add_inl(const G1Xyzz& p, const G1Xyzz& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fq U1 = mul(p.x, q.zz), U2 = mul(q.x, p.zz);
    Fq S1 = mul(p.y, q.zzz), S2 = mul(q.y, p.zzz);
    Fq P = sub(U2, U1);
    Fq R = sub(S2, S1);
    if (__builtin_expect(P.is_zero(), 0)) {
        if (R.is_zero()) return g1_dbl_inl(p);
        return g1_inf();
    }
    Fq PP = sqr(P);
    Fq PPP = mul(P, PP);
    Fq Q = mul(U1, PP);
    G1Xyzz r;
    r.x = sub(sub(sub(sqr(R), PPP), Q), Q);
    r.y = sub(mul(R, sub(Q, r.x)), mul(S1, PPP));
    r.zz = mul(mul(p.zz, q.zz), PP);
    r.zzz = mul(mul(p.zzz, q.zzz), PPP);
    return r;
}
DR_DEV G1Xyzz g1_select(bool c, const G1Xyzz& a, const G1Xyzz& b) {      // c ? a : b, branch-free
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        r.x.l[i] = c ? a.x.l[i] : b.x.l[i];
        r.y.l[i] = c ? a.y.l[i] : b.y.l[i];
        r.zz.l[i] = c ? a.zz.l[i] : b.zz.l[i];
        r.zzz.l[i] = c ? a.zzz.l[i] : b.zzz.l[i];
    }
    return r;
}

DR_DEV G1Affine g1_neg_affine(const G1Affine& p, bool negate) {
    G1Affine r = p;
    Fq ny = neg(p.y);
#pragma unroll
    for (int i = 0; i < 12; i++) r.y.l[i] = negate ? ny.l[i] : p.y.l[i];
    return r;
}

}  // namespace dr
