// BLS12-381 G1 in XYZZ coordinates over the unsaturated field of fq28.hip.h — the arithmetic of the MSM's bucket kernels.
// Same formulas as curve.hip.h (EFD madd-2008-s / add-2008-s / dbl-2008-s-1, a = 0); the reference reaches this arithmetic
// through blst (dot_ring/ring_proof/pcs/kzg.py:147-175).  Limb / value bounds are tracked in the comments: "N" = normal
// (a product, or carry()'d), "d" = difference of two N values (|limb| < 2^28), see fq28.hip.h.
#pragma once
#include "fq28.hip.h"

namespace dr {

struct G1Affine28 {
    Fq28 x, y;                       // N (canonical words unpacked)
    bool inf;
};
struct G1Xyzz28 {                    // infinity is a flag here: ZZ == 0 (mod p) would need an exact zero test per step
    Fq28 x, y, zz, zzz;              // x: N, y: d, zz / zzz: N
    bool inf;
};

DR_DEV G1Xyzz28 g1_inf28() {
    G1Xyzz28 r;
    r.x = Fq28::zero(); r.y = Fq28::zero(); r.zz = Fq28::zero(); r.zzz = Fq28::zero();
    r.inf = true;
    return r;
}

DR_DEV G1Affine28 load_affine28(const uint32_t* bases, uint32_t idx) {
    const uint32_t* p = bases + (size_t)idx * 24;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t wx[12], wy[12];
    uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5];
    wx[0] = a.x; wx[1] = a.y; wx[2] = a.z; wx[3] = a.w; wx[4] = b.x; wx[5] = b.y; wx[6] = b.z; wx[7] = b.w;
    wx[8] = c.x; wx[9] = c.y; wx[10] = c.z; wx[11] = c.w;
    wy[0] = d.x; wy[1] = d.y; wy[2] = d.z; wy[3] = d.w; wy[4] = e.x; wy[5] = e.y; wy[6] = e.z; wy[7] = e.w;
    wy[8] = f.x; wy[9] = f.y; wy[10] = f.z; wy[11] = f.w;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= wx[i] | wy[i];
    G1Affine28 r;
    r.x = unpack28(wx);
    r.y = unpack28(wy);
    r.inf = any == 0;
    return r;
}

// 2 * (affine q) -> XYZZ (mdbl-2008-s-1)
DR_DEV G1Xyzz28 g1_dbl_affine28(const G1Affine28& q) {
    Fq28 U = dbl(q.y);                                   // limbs < 2^29
    Fq28 V = sqr(U);
    Fq28 W = mul(U, V);
    Fq28 S = mul(q.x, V);
    Fq28 X2 = sqr(q.x);
    Fq28 M = add(dbl(X2), X2);                           // < 3 * 2^28
    G1Xyzz28 r;
    r.x = carry(sub(sub(sqr(carry(M)), S), S));
    r.y = sub(mul(M, sub(S, r.x)), mul(W, q.y));
    r.zz = V;
    r.zzz = W;
    r.inf = false;
    return r;
}

// acc + (affine q, possibly negated)   (madd-2008-s: 8 products + 2 squarings)
DR_DEV void g1_madd28(G1Xyzz28& acc, const G1Affine28& q) {
    if (q.inf) return;
    if (acc.inf) {
        acc.x = q.x; acc.y = q.y; acc.zz = Fq28::one(); acc.zzz = Fq28::one();
        acc.inf = false;
        return;
    }
    Fq28 U2 = mul(q.x, acc.zz);
    Fq28 S2 = mul(q.y, acc.zzz);
    Fq28 P = sub(U2, acc.x);                             // d
    Fq28 R = sub(S2, acc.y);                             // N - d: (-2^28, 2^29)
    Fq28 PP = sqr(P);
    if (__builtin_expect(maybe_zero_normal(PP), 0)) {    // P = 0 (mod p) => PP = 0 (mod p); exact test only then
        if (is_zero_mod_p(P)) {
            if (is_zero_mod_p(R)) acc = g1_dbl_affine28(q);
            else acc = g1_inf28();
            return;
        }
    }
    Fq28 PPP = mul(P, PP);
    Fq28 Q = mul(acc.x, PP);
    Fq28 X3 = carry(sub(sub(sub(sqr(R), PPP), Q), Q));   // (-3 * 2^28, 2^28) -> N
    Fq28 Y3 = sub(mul(R, sub(Q, X3)), mul(acc.y, PPP));  // d
    acc.zz = mul(acc.zz, PP);
    acc.zzz = mul(acc.zzz, PPP);
    acc.x = X3;
    acc.y = Y3;
}

}  // namespace dr
