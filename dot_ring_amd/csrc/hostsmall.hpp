// Host-side versions of three GPU steps for SINGLE proofs (and pairs): a kernel launch chain costs 0.7 - 2 ms of latency
// whatever the batch size, while one x86 core decodes a point in ~80 us and folds an 11-point G1 MSM in ~0.6 ms.  The batch
// verifier uses these up to DOTRING_VERIFY_HOST_MAX proofs (default 8) so that RingVRF.verify of one proof is not slower than
// the reference's CPU verifier (3.99 ms, docs/BENCHMARK.md:73); every larger batch takes the kernels.
//   te_decode_checked   dec_point (dot_ring/curve/point.py:150-214, vrf/codec.py:39-45, curve/curve.py:56-67): decompress,
//                       then non-identity member of the prime-order subgroup — same verdicts as k_bsn_decode_points
//   g1_msm_small        sum k_i P_i for a handful of G1 points (Straus, 4-bit windows) — what the verifier's two folds need
#pragma once
#include <thread>
#include <vector>

#include "hostmath.hpp"
#include "hostproto.hpp"

namespace drh {

struct TeExt {                      // extended twisted Edwards coordinates over the host field (Montgomery form)
    Fr x, y, z, t;
};
inline TeExt te_ext_add(const TeExt& p, const TeExt& q, const Fr& d, const Fr& neg_a) {      // add-2008-hwcd, unified
    Fr A = p.x * q.x, B = p.y * q.y, C = p.t * d * q.t, D = p.z * q.z;
    Fr E = (p.x + p.y) * (q.x + q.y) - A - B, F = D - C, G = D + C, H = B + A * neg_a;
    return {E * F, G * H, F * G, E * H};
}

inline bool te_decode_checked(const TeCurveHost& cv, const uint8_t enc[32], uint8_t out_xy[64]) {
    uint8_t yb[32];
    std::memcpy(yb, enc, 32);
    const bool sign = (yb[31] & 0x80) != 0;
    yb[31] &= 0x7f;
    Fr y;
    if (!Fr::load_le(y, yb)) return false;                          // y >= p
    uint8_t d_le[32];
    store_le32(cv.d, d_le);
    Fr d;
    if (!Fr::load_le(d, d_le)) return false;
    const Fr one = Fr::one(), neg_a = Fr::from_u64(cv.neg_a[0]);
    const Fr y2 = y.sqr();
    const Fr den = neg_a.neg() - d * y2;                            // a - d y^2
    if (den.is_zero()) return false;
    const Fr x2 = (one - y2) * den.inv();
    Fr x;
    if (!fr_sqrt(x, x2)) return false;
    if (Fr::gt_std(x.from_mont(), x.neg().from_mont()) != sign) x = x.neg();      // sign bit: the larger of (x, p - x)
    if (x.is_zero() && y == one) return false;                      // identity
    // prime-order subgroup: [n] P = O
    const TeExt P{x, y, one, x * y};
    TeExt acc{Fr::zero(), one, one, Fr::zero()};
    bool started = false;
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            if (started) acc = te_ext_add(acc, acc, d, neg_a);
            if ((cv.n.m[i] >> b) & 1) {
                acc = started ? te_ext_add(acc, P, d, neg_a) : P;
                started = true;
            }
        }
    if (!(acc.x.is_zero() && acc.y == acc.z)) return false;
    x.store_le(out_xy);
    y.store_le(out_xy + 32);
    return true;
}

// ---- Elligator 2 and one scalar multiplication on the host, for the head of a prover call of a few dozen proofs: the two kernels
// (k_bsn_encode_to_curve, k_bsn_scalar_mul_glv) are dependent chains of ~1.2 ms whatever the batch; one x86 core does both in ~0.06 ms per input.
// The map follows ell2_prepare / ell2_finish of kernels_bsn.hip.h step by step (te_curve.py:48-95: RFC 9380 6.7.1 with Z = 5 on the
// Montgomery model, then the birational map), so the two give the same point for every input, exceptional cases included.
struct Ell2ConstsHost {
    Fr d, neg_a, aob, inv_b2, mont_b, five;
};
inline const Ell2ConstsHost& ell2_consts_bandersnatch() {
    static const Ell2ConstsHost c = [] {
        Ell2ConstsHost k;
        uint8_t d_le[32];
        store_le32(te_curve(0)->d, d_le);
        (void)Fr::load_le(k.d, d_le);
        k.five = Fr::from_u64(5);
        k.neg_a = k.five;
        const Fr a = k.five.neg(), inv_den = (a - k.d).inv();
        const Fr mont_a = (a + k.d).dbl() * inv_den;
        k.mont_b = Fr::from_u64(4) * inv_den;
        k.aob = mont_a * k.mont_b.inv();
        k.inv_b2 = k.mont_b.sqr().inv();
        return k;
    }();
    return c;
}
// one Elligator map: u (Montgomery form) and 1 / (1 + 5 u^2) -> a point in extended coordinates
inline TeExt ell2_map_host(const Ell2ConstsHost& k, const Fr& u, const Fr& tv1, const Fr& inv_den) {
    const Fr one = Fr::one();
    const Fr x1 = (k.aob * inv_den).neg();
    const Fr gx1 = ((x1 + k.aob) * x1 + k.inv_b2) * x1;
    Fr y;
    const bool e2 = fr_sqrt(y, gx1);                       // sqrt(g(x1)), or sqrt(Z g(x1)) when g(x1) is not a square
    Fr x = x1;
    if (!e2) {
        Fr r;
        (void)fr_sqrt(r, k.five * gx1);                    // Z g(x1) is a square when g(x1) is not
        x = x1.neg() - k.aob;
        y = tv1.is_zero() ? Fr::zero() : u * r;            // sqrt(Z u^2 g(x1))
    }
    const bool odd = (y.from_mont().l[0] & 1u) != 0;
    if (e2 != odd) y = y.neg();                            // e2 XOR sgn0(y) -> negate
    const Fr s = x * k.mont_b, t = y * k.mont_b;
    const Fr sp1 = s + one, Z = sp1 * t;
    if (Z.is_zero()) return {Fr::zero(), one, one, Fr::zero()};
    const Fr X = s * sp1, Y = (s - one) * t;
    return {X * Z, Y * Z, Z.sqr(), X * Y};
}
// encode_to_curve of Bandersnatch from the two field elements of hash_to_field (32-byte little-endian each): map both, add, clear the
// cofactor 4; affine x || y out
inline bool te_encode_to_curve_host(const uint8_t u2[64], uint8_t out_xy[64]) {
    const Ell2ConstsHost& k = ell2_consts_bandersnatch();
    const Fr one = Fr::one();
    Fr u[2], tv1[2], den[2];
    for (int h = 0; h < 2; h++) {
        if (!Fr::load_le(u[h], u2 + 32 * h)) return false;
        tv1[h] = k.five * u[h].sqr();
        if ((tv1[h] + one).is_zero()) tv1[h] = Fr::zero();
        den[h] = tv1[h] + one;
    }
    const Fr both = (den[0] * den[1]).inv();               // never zero (tv1 = -1 was mapped to 0)
    const TeExt q0 = ell2_map_host(k, u[0], tv1[0], both * den[1]), q1 = ell2_map_host(k, u[1], tv1[1], both * den[0]);
    TeExt r = te_ext_add(q0, q1, k.d, k.neg_a);
    r = te_ext_add(r, r, k.d, k.neg_a);
    r = te_ext_add(r, r, k.d, k.neg_a);
    const Fr zi = r.z.inv();
    (r.x * zi).store_le(out_xy);
    (r.y * zi).store_le(out_xy + 32);
    return true;
}
// k * P on Bandersnatch, P affine x || y, k a 256-bit little-endian scalar (plain double-and-add: ~0.12 ms); affine out
inline bool te_scalar_mul_host(const uint8_t p_xy[64], const uint8_t k_le[32], uint8_t out_xy[64]) {
    const Ell2ConstsHost& c = ell2_consts_bandersnatch();
    const Fr one = Fr::one();
    Fr x, y;
    if (!Fr::load_le(x, p_xy) || !Fr::load_le(y, p_xy + 32)) return false;
    const TeExt P{x, y, one, x * y};
    TeExt acc{Fr::zero(), one, one, Fr::zero()};
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        if (started) acc = te_ext_add(acc, acc, c.d, c.neg_a);
        if ((k_le[i >> 3] >> (i & 7)) & 1) {
            acc = started ? te_ext_add(acc, P, c.d, c.neg_a) : P;
            started = true;
        }
    }
    const Fr zi = acc.z.inv();
    (acc.x * zi).store_le(out_xy);
    (acc.y * zi).store_le(out_xy + 32);
    return true;
}

struct G1AffineHost {
    Fq x, y;
    bool inf;
};

// sum_i k_i P_i, k_i little-endian 32 bytes (any value below 2^256): Straus with 4-bit windows and a 15-entry table per
// point; `threads` > 1 splits the points (each thread its own doubling chain: worth it from ~6 points)
inline G1 g1_msm_small(const G1AffineHost* pts, const uint8_t* scalars, size_t n, unsigned threads = 1) {
    auto part = [&](size_t lo, size_t hi) {
        std::vector<G1> table((hi - lo) * 15);
        for (size_t i = lo; i < hi; i++) {
            G1* t = &table[(i - lo) * 15];
            if (pts[i].inf) { for (int e = 0; e < 15; e++) t[e] = G1::inf(); continue; }
            t[0].x = pts[i].x; t[0].y = pts[i].y; t[0].zz = Fq::one(); t[0].zzz = Fq::one();
            t[1] = g1_dbl(t[0]);
            for (int e = 2; e < 15; e++) t[e] = g1_add(t[e - 1], t[0]);
        }
        G1 acc = G1::inf();
        for (int w = 63; w >= 0; w--) {
            if (!acc.is_inf()) for (int j = 0; j < 4; j++) acc = g1_dbl(acc);
            for (size_t i = lo; i < hi; i++) {
                const unsigned nib = (scalars[32 * i + (w >> 1)] >> (4 * (w & 1))) & 15u;
                if (nib) acc = g1_add(acc, table[(i - lo) * 15 + nib - 1]);
            }
        }
        return acc;
    };
    if (threads <= 1 || n < 6) return part(0, n);
    if (threads > n) threads = (unsigned)n;
    std::vector<G1> res(threads);
    parallel_for(threads, [&](size_t k) { res[k] = part(n * k / threads, n * (k + 1) / threads); }, 1);      // (the worker pool: no thread is started here)
    G1 acc = res[0];
    for (unsigned k = 1; k < threads; k++) acc = g1_add(acc, res[k]);
    return acc;
}

}  // namespace drh
