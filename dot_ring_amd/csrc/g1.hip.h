// BLS12-381 G1 (y^2 = x^3 + 4 over Fq) in XYZZ coordinates (X, Y, ZZ, ZZZ) over the unsaturated field of fq28.hip.h:
// mixed add 8M+2S, full add 12M+2S, doubling 6M+4S (EFD "xyzz": madd-2008-s, add-2008-s, dbl-2008-s-1, a = 0).
// The reference reaches this arithmetic through blst (dot_ring/ring_proof/pcs/kzg.py:147-175).
//
// Register forms (see fq28.hip.h): "N" = normal (a product, carry()'d, or canonical words unpacked: limbs in [0, 2^28)),
// "d" = difference of two N values (|limb| < 2^28).  A point in registers holds x: N, y: d, zz / zzz: N and an explicit
// infinity flag — testing ZZ == 0 (mod p) on a lazy value would cost a canonicalisation per addition.
// In memory a point is canonical words, Montgomery form with R = 2^392: affine 96 B with (0,0) = infinity, XYZZ 192 B
// with ZZ = 0 (all four coordinates zero) = infinity.
#pragma once
#include "fq28.hip.h"

namespace dr {

struct G1Affine {
    Fq28 x, y;                       // N
    uint32_t inf;                    // 0 / 1 (a word, not a bool: byte-sized members made hipcc keep the flags in LDS)
};
struct G1Xyzz {
    Fq28 x, y, zz, zzz;              // N, d, N, N
    uint32_t inf;
};

DR_DEV G1Xyzz g1_inf() {
    G1Xyzz r;
    r.x = Fq28::zero(); r.y = Fq28::zero(); r.zz = Fq28::zero(); r.zzz = Fq28::zero();
    r.inf = 1;
    return r;
}
DR_DEV G1Xyzz g1_from_affine(const G1Affine& p) {
    G1Xyzz r;
    r.x = p.x; r.y = p.y; r.zz = Fq28::one(); r.zzz = Fq28::one();
    r.inf = p.inf;
    return r;
}
DR_DEV G1Affine g1_neg_affine(const G1Affine& p, bool negate) {
    G1Affine r = p;
    r.y = cneg(p.y, negate);         // d
    return r;
}

// ---------------------------------------------------------------- memory forms
DR_DEV G1Affine load_affine(const uint32_t* bases, size_t idx) {
    const uint32_t* p = bases + idx * 24;
    uint32_t wx[12], wy[12];
    load_words12(p, wx);
    load_words12(p + 12, wy);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= wx[i] | wy[i];
    G1Affine r;
    r.x = unpack28(wx);
    r.y = unpack28(wy);
    r.inf = any == 0 ? 1u : 0u;
    return r;
}
// (fixed-base tables may keep one point per 128-byte line: records of `words` 32-bit words, 24 of them used)
DR_DEV G1Affine load_affine_at(const uint32_t* table, size_t idx, uint32_t words) {
    const uint32_t* p = table + idx * words;
    uint32_t wx[12], wy[12];
    load_words12(p, wx);
    load_words12(p + 12, wy);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= wx[i] | wy[i];
    G1Affine r;
    r.x = unpack28(wx);
    r.y = unpack28(wy);
    r.inf = any == 0 ? 1u : 0u;
    return r;
}
DR_DEV void store_affine(uint32_t* bases, size_t idx, const G1Affine& a) {
    uint32_t* p = bases + idx * 24;
    if (a.inf) {
        uint32_t z[12] = {0};
        store_words12(p, z);
        store_words12(p + 12, z);
        return;
    }
    store_fq28(p, a.x);
    store_fq28(p + 12, a.y);
}
DR_DEV G1Xyzz load_xyzz(const uint32_t* arr, size_t idx) {
    const uint32_t* p = arr + idx * 48;
    uint32_t w[12], any = 0;
    G1Xyzz r;
    load_words12(p + 24, w);
#pragma unroll
    for (int i = 0; i < 12; i++) any |= w[i];
    r.zz = unpack28(w);
    r.inf = any == 0 ? 1u : 0u;
    r.x = load_fq28(p);
    r.y = load_fq28(p + 12);
    r.zzz = load_fq28(p + 36);
    return r;
}
DR_DEV void store_xyzz(uint32_t* arr, size_t idx, const G1Xyzz& v) {
    uint32_t* p = arr + idx * 48;
    if (v.inf) {
        uint32_t z[12] = {0};
        store_words12(p, z); store_words12(p + 12, z); store_words12(p + 24, z); store_words12(p + 36, z);
        return;
    }
    store_fq28(p, v.x); store_fq28(p + 12, v.y); store_fq28(p + 24, v.zz); store_fq28(p + 36, v.zzz);
}

// raw register image (57 words: the limbs as they are, plus the flag) at stride `stride` words — LDS parking slots and
// LDS tree reductions, where a canonicalisation per access would cost more than the addition it feeds
constexpr int XYZZ_RAW_WORDS = 4 * L28 + 1;
DR_DEV void put_raw(uint32_t* slot, uint32_t stride, const G1Xyzz& v) {
#pragma unroll
    for (int i = 0; i < L28; i++) {
        slot[(0 * L28 + i) * stride] = (uint32_t)v.x.l[i];
        slot[(1 * L28 + i) * stride] = (uint32_t)v.y.l[i];
        slot[(2 * L28 + i) * stride] = (uint32_t)v.zz.l[i];
        slot[(3 * L28 + i) * stride] = (uint32_t)v.zzz.l[i];
    }
    slot[4 * L28 * stride] = v.inf;
}
DR_DEV G1Xyzz get_raw(const uint32_t* slot, uint32_t stride) {
    G1Xyzz v;
#pragma unroll
    for (int i = 0; i < L28; i++) {
        v.x.l[i] = (int32_t)slot[(0 * L28 + i) * stride];
        v.y.l[i] = (int32_t)slot[(1 * L28 + i) * stride];
        v.zz.l[i] = (int32_t)slot[(2 * L28 + i) * stride];
        v.zzz.l[i] = (int32_t)slot[(3 * L28 + i) * stride];
    }
    v.inf = slot[4 * L28 * stride];
    return v;
}

DR_DEV G1Xyzz g1_select(bool c, const G1Xyzz& a, const G1Xyzz& b) {      // c ? a : b, branch-free
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < L28; i++) {
        r.x.l[i] = c ? a.x.l[i] : b.x.l[i];
        r.y.l[i] = c ? a.y.l[i] : b.y.l[i];
        r.zz.l[i] = c ? a.zz.l[i] : b.zz.l[i];
        r.zzz.l[i] = c ? a.zzz.l[i] : b.zzz.l[i];
    }
    r.inf = c ? a.inf : b.inf;
    return r;
}

// ---------------------------------------------------------------- group law
// 2 * (affine q) -> XYZZ (mdbl-2008-s-1).  Cold: reached from g1_madd when a bucket meets the same point twice.
DR_DEV G1Xyzz g1_dbl_affine(const G1Affine& q) {
    Fq28 U = dbl(q.y);                                   // |limb| < 2^29
    Fq28 V = sqr(U);
    Fq28 W = mul(U, V);
    Fq28 S = mul(q.x, V);
    Fq28 X2 = sqr(q.x);
    Fq28 M = carry(add(dbl(X2), X2));                    // 3 X2 -> N
    G1Xyzz r;
    r.x = carry(sub(sub(sqr(M), S), S));
    r.y = sub(mul(M, sub(S, r.x)), mul(W, q.y));
    r.zz = V;
    r.zzz = W;
    r.inf = 0;
    return r;
}

// 2 * p in XYZZ (dbl-2008-s-1).  Inlined everywhere: an out-of-line call passes its operands through scratch memory, and a
// kernel that reserves scratch loses resident waves (measured in round 1: -21 % on the level reduction, -25 % on the comb).
DR_DEV G1Xyzz g1_dbl(const G1Xyzz& p) {
    Fq28 U = dbl(p.y);                                   // |limb| < 2^29
    Fq28 V = sqr(U);
    Fq28 W = mul(U, V);
    Fq28 S = mul(p.x, V);
    Fq28 X2 = sqr(p.x);
    Fq28 M = carry(add(dbl(X2), X2));
    G1Xyzz r;
    r.x = carry(sub(sub(sqr(M), S), S));
    r.y = sub(mul(M, sub(S, r.x)), mul(W, p.y));
    r.zz = mul(V, p.zz);
    r.zzz = mul(W, p.zzz);
    r.inf = p.inf;                                       // (y = 0 never occurs: the curve group has odd order h r)
    return r;
}

// acc + (affine q)   (madd-2008-s) with the exceptional cases made explicit
DR_DEV G1Xyzz g1_madd(const G1Xyzz& acc, const G1Affine& q) {
    if (q.inf) return acc;
    if (acc.inf) return g1_from_affine(q);
    Fq28 U2 = mul(q.x, acc.zz);
    Fq28 S2 = mul(q.y, acc.zzz);
    Fq28 P = sub(U2, acc.x);                             // d
    Fq28 R = sub(S2, acc.y);                             // N - d: (-2^28, 2^29)
    Fq28 PP = sqr(P);
    if (__builtin_expect(maybe_zero_normal(PP), 0)) {    // P = 0 (mod p) => PP = 0 (mod p); the exact test only then
        if (is_zero_mod_p(P)) {
            if (is_zero_mod_p(R)) return g1_dbl_affine(q);
            return g1_inf();
        }
    }
    Fq28 PPP = mul(P, PP);
    Fq28 Q = mul(acc.x, PP);
    G1Xyzz r;
    r.x = carry(sub(sub(sub(sqr(R), PPP), Q), Q));       // (-3 * 2^28, 2^28) -> N
    r.y = mul2(R, sub(Q, r.x), neg(acc.y), PPP);         // R (Q - X3) - Y1 PPP with one reduction: N (so also d)
    r.zz = mul(acc.zz, PP);
    r.zzz = mul(acc.zzz, PPP);
    r.inf = 0;
    return r;
}

// p + q, both XYZZ (add-2008-s) with the exceptional cases made explicit
DR_DEV G1Xyzz g1_add(const G1Xyzz& p, const G1Xyzz& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    Fq28 U1 = mul(p.x, q.zz), U2 = mul(q.x, p.zz);
    Fq28 S1 = mul(p.y, q.zzz), S2 = mul(q.y, p.zzz);
    Fq28 P = sub(U2, U1);                                // d
    Fq28 R = sub(S2, S1);                                // d
    Fq28 PP = sqr(P);
    if (__builtin_expect(maybe_zero_normal(PP), 0)) {
        if (is_zero_mod_p(P)) {
            if (is_zero_mod_p(R)) return g1_dbl(p);
            return g1_inf();
        }
    }
    Fq28 PPP = mul(P, PP);
    Fq28 Q = mul(U1, PP);
    G1Xyzz r;
    r.x = carry(sub(sub(sub(sqr(R), PPP), Q), Q));
    r.y = sub(mul(R, sub(Q, r.x)), mul(S1, PPP));       // (two products here: the fused form needs 70 operand registers at once and
    r.zz = mul(mul(p.zz, q.zz), PP);                     //  pushes the reduction kernels past 256 VGPRs = one resident wave)
    r.zzz = mul(mul(p.zzz, q.zzz), PPP);
    r.inf = 0;
    return r;
}

// XYZZ -> affine (Montgomery, N); infinity stays infinity
DR_DEV G1Affine g1_to_affine_dev(const G1Xyzz& p) {
    G1Affine a;
    a.inf = p.inf;
    if (p.inf) { a.x = Fq28::zero(); a.y = Fq28::zero(); return a; }
    Fq28 zi3 = inv(p.zzz);
    Fq28 t = mul(p.zz, zi3);
    a.x = mul(p.x, sqr(t));
    a.y = mul(p.y, zi3);
    return a;
}

}  // namespace dr
