// The G1 arithmetic of round 1 — XYZZ formulas over SATURATED 32-bit limbs (Fe<FqParams>, field.hip.h: one
// v_mad_u64_u32 + one v_addc_co_u32 per partial product) — kept for tools/ubench_limbs.hip and tools/ubench_affine.hip,
// which measure it against the unsaturated 14 x 28-bit field the library uses since round 2 (fq28.hip.h, g1.hip.h).
#pragma once
#include "field.hip.h"

namespace legacy {
using namespace dr;

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


DR_DEV Fq load_fq(const uint32_t* p) {
    Fq r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
DR_DEV void store_fq(uint32_t* p, const Fq& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
DR_DEV G1Affine load_affine(const uint32_t* bases, uint32_t idx) {
    const uint32_t* p = bases + (size_t)idx * 24;
    G1Affine a;
    a.x = load_fq(p);
    a.y = load_fq(p + 12);
    return a;
}
DR_DEV G1Xyzz load_xyzz(const uint32_t* arr, size_t idx) {
    const uint32_t* p = arr + idx * 48;
    G1Xyzz r;
    r.x = load_fq(p); r.y = load_fq(p + 12); r.zz = load_fq(p + 24); r.zzz = load_fq(p + 36);
    return r;
}
DR_DEV void store_xyzz(uint32_t* arr, size_t idx, const G1Xyzz& v) {
    uint32_t* p = arr + idx * 48;
    store_fq(p, v.x); store_fq(p + 12, v.y); store_fq(p + 24, v.zz); store_fq(p + 36, v.zzz);
}


// standard-form little-endian limbs -> Montgomery, in place (SRS load). (0,0) stays (0,0).
__global__ void k_g1_bases_to_mont(uint32_t* bases, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* p = bases + (size_t)i * 24;
    store_fq(p, to_mont(load_fq(p)));
    store_fq(p + 12, to_mont(load_fq(p + 12)));
}


// XYZZ -> affine (Montgomery); infinity -> (0,0)
DR_DEV G1Affine g1_to_affine_dev(const G1Xyzz& p) {
    G1Affine a;
    if (p.is_inf()) { a.x = Fq::zero(); a.y = Fq::zero(); return a; }
    Fq zi3 = inv(p.zzz);
    Fq t = mul(p.zz, zi3);
    a.x = mul(p.x, sqr(t));
    a.y = mul(p.y, zi3);
    return a;
}

// synthetic bases for full-size measurements: bases[i] = (first + i) * seed   (seed affine, Montgomery)
__global__ void k_g1_synth_bases(uint32_t* bases, uint32_t n, uint32_t first, const uint32_t* seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine s = load_affine(seed, 0);
    uint32_t k = first + i;
    G1Xyzz acc = g1_inf();
#pragma unroll 1
    for (int bit = 31 - __clz(k | 1); bit >= 0; bit--) {
        acc = g1_dbl(acc);
        if ((k >> bit) & 1) acc = g1_madd(acc, s);
    }
    G1Affine a = g1_to_affine_dev(acc);
    uint32_t* p = bases + (size_t)i * 24;
    store_fq(p, a.x);
    store_fq(p + 12, a.y);
}


}  // namespace legacy
