// Device-side group law of the twisted Edwards curves on the Ring-VRF hot path (BLS12-381 G1: g1.hip.h).
//
//  * Bandersnatch (twisted Edwards a=-5 over Fr; JubJub, a=-1, through the same templates), extended coordinates (X,Y,Z,T) — the formulas the
//    reference runs in dot_ring/curve/native_field/bandersnatch_te.pyx:127-174 (dbl/add-2008-hwcd);
//    a*A is computed as -(4A+A) instead of a Montgomery multiplication.
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

}  // namespace dr
