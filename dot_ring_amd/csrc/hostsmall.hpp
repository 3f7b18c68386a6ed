// Host-side versions of three GPU steps for SINGLE proofs (and pairs): a kernel launch chain costs 0.7 - 2 ms of latency
// whatever the batch size, while one x86 core decodes a point in ~80 us and folds an 11-point G1 MSM in ~0.6 ms.  The batch
// verifier uses these up to DOTRING_VERIFY_HOST_MAX proofs (default 6) so that RingVRF.verify of one proof is not slower than
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
    std::vector<std::thread> pool;
    for (unsigned k = 1; k < threads; k++) pool.emplace_back([&, k] { res[k] = part(n * k / threads, n * (k + 1) / threads); });
    res[0] = part(0, n / threads);
    for (auto& th : pool) th.join();
    G1 acc = res[0];
    for (unsigned k = 1; k < threads; k++) acc = g1_add(acc, res[k]);
    return acc;
}

}  // namespace drh
