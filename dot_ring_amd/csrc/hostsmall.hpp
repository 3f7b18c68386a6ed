// Host-side versions of three GPU steps for SINGLE proofs (and pairs): a kernel launch chain costs 0.7 - 2 ms of latency
// whatever the batch size, while one x86 core decodes a point in ~80 us and folds an 11-point G1 MSM in ~0.6 ms.  The batch
// verifier uses these up to DOTRING_VERIFY_HOST_MAX proofs (default 8) so that RingVRF.verify of one proof is not slower than
// the reference's CPU verifier (3.99 ms, docs/BENCHMARK.md:73); every larger batch takes the kernels.
//   te_decode_checked   dec_point (dot_ring/curve/point.py:150-214, vrf/codec.py:39-45, curve/curve.py:56-67): decompress,
//                       then non-identity member of the prime-order subgroup — same verdicts as k_bsn_decode_points
//   g1_msm_small        sum k_i P_i for a handful of G1 points (Straus, 4-bit windows) — what the verifier's two folds need
#pragma once
#include <array>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "hostmath.hpp"
#include "hostproto.hpp"

namespace drh {

// calls of up to this many items take the host versions below instead of a kernel launch chain (DOTRING_SMALL_HOST_MAX, default 64; 0 = never)
inline size_t small_host_max() {
    static const size_t v = std::getenv("DOTRING_SMALL_HOST_MAX") ? (size_t)std::atol(std::getenv("DOTRING_SMALL_HOST_MAX")) : 64;
    return v;
}

struct TeExt {                      // extended twisted Edwards coordinates over the host field (Montgomery form)
    Fr x, y, z, t;
};
inline TeExt te_ext_add(const TeExt& p, const TeExt& q, const Fr& d, const Fr& neg_a) {      // add-2008-hwcd, unified
    Fr A = p.x * q.x, B = p.y * q.y, C = p.t * d * q.t, D = p.z * q.z;
    Fr E = (p.x + p.y) * (q.x + q.y) - A - B, F = D - C, G = D + C, H = B + A * neg_a;
    return {E * F, G * H, F * G, E * H};
}

// The same group law with the curve's small -a (5 on Bandersnatch, 1 on JubJub) applied by additions, and a dedicated doubling
// (dbl-2008-hwcd: 4 squarings + 3 products, + 1 for T): what the scalar multiplications below are made of.  `want_t`: the T coordinate
// is only read by an ADDITION, so the doublings inside a window skip it.
struct TeHostParams {
    Fr d, neg_a;
    unsigned neg_a_small;
};
inline TeHostParams te_host_params(const TeCurveHost& cv) {
    TeHostParams k;
    uint8_t d_le[32];
    store_le32(cv.d, d_le);
    (void)Fr::load_le(k.d, d_le);
    k.neg_a = Fr::from_u64(cv.neg_a[0]);
    k.neg_a_small = (unsigned)cv.neg_a[0];
    return k;
}
inline Fr te_times_neg_a(const Fr& v, const TeHostParams& c) {
    if (c.neg_a_small == 1) return v;
    if (c.neg_a_small == 5) return v.dbl().dbl() + v;
    return v * c.neg_a;
}
inline TeExt te_add(const TeExt& p, const TeExt& q, const TeHostParams& c) {                // add-2008-hwcd, unified: 9 products
    const Fr A = p.x * q.x, B = p.y * q.y, C = p.t * c.d * q.t, D = p.z * q.z;
    const Fr E = (p.x + p.y) * (q.x + q.y) - A - B, F = D - C, G = D + C, H = B + te_times_neg_a(A, c);
    return {E * F, G * H, F * G, E * H};
}
inline TeExt te_dbl(const TeExt& p, const TeHostParams& c, bool want_t = true) {
    const Fr A = p.x.sqr(), B = p.y.sqr(), C = p.z.sqr().dbl(), D = te_times_neg_a(A, c).neg();    // D = a X^2
    const Fr E = (p.x + p.y).sqr() - A - B, G = D + B, F = G - C, H = D - B;
    return {E * F, G * H, F * G, want_t ? E * H : Fr::zero()};
}
// signed binary digits (non-adjacent form, least significant first) of a 256-bit constant: a third of them non-zero
inline std::vector<int8_t> naf_digits(const uint64_t k[4]) {
    uint64_t v[5] = {k[0], k[1], k[2], k[3], 0};
    std::vector<int8_t> out;
    auto nonzero = [&] { return (v[0] | v[1] | v[2] | v[3] | v[4]) != 0; };
    while (nonzero()) {
        int8_t dgt = 0;
        if (v[0] & 1) {
            dgt = (v[0] & 3) == 3 ? -1 : 1;
            if (dgt == 1) v[0] -= 1;                                   // (odd: no borrow)
            else { for (int i = 0; i < 5; i++) { if (++v[i] != 0) break; } }
        }
        out.push_back(dgt);
        for (int i = 0; i < 4; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
        v[4] >>= 1;
    }
    return out;
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
    // prime-order subgroup: [n] P = O — n in non-adjacent form (a third of the digits non-zero), doublings without T unless an addition follows
    static const std::vector<int8_t> naf[2] = {naf_digits(te_curve(0)->n.m), naf_digits(te_curve(1)->n.m)};
    const std::vector<int8_t>& dg = naf[cv.id == 1 ? 1 : 0];
    const TeHostParams hp = te_host_params(cv);
    const TeExt P{x, y, one, x * y}, Pn{x.neg(), y, one, P.t.neg()};
    TeExt acc = P;                                                   // the top digit of n's form is +1
    for (size_t i = dg.size() - 1; i-- > 0;) {
        acc = te_dbl(acc, hp, dg[i] != 0);
        if (dg[i]) acc = te_add(acc, dg[i] > 0 ? P : Pn, hp);
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
// ---- the sigma protocols of ONE proof on host cores (DOTRING_SMALL_HOST_MAX): scalar multiplications of secret scalars on a fixed
// schedule, Straus for the verifier's public combinations, window tables for the suite's constant bases.
inline TeExt te_identity() { return {Fr::zero(), Fr::one(), Fr::one(), Fr::zero()}; }
inline bool te_load_affine(const uint8_t xy[64], TeExt& out) {
    Fr x, y;
    if (!Fr::load_le(x, xy) || !Fr::load_le(y, xy + 32)) return false;
    out = {x, y, Fr::one(), x * y};
    return true;
}
inline void te_store_affine(const TeExt& p, uint8_t out_xy[64]) {
    const Fr zi = p.z.inv();
    (p.x * zi).store_le(out_xy);
    (p.y * zi).store_le(out_xy + 32);
}
inline TeExt te_neg(const TeExt& p) { return {p.x.neg(), p.y, p.z, p.t.neg()}; }
inline bool te_equal(const TeExt& p, const TeExt& q) { return p.x * q.z == q.x * p.z && p.y * q.z == q.y * p.z; }
// entry `idx` of a table of `count` points without an index-dependent access: every entry is read, the wanted one kept by mask
inline TeExt te_table_pick(const TeExt* table, unsigned count, unsigned idx) {
    TeExt r;
    std::memset(&r, 0, sizeof r);
    uint64_t* out = reinterpret_cast<uint64_t*>(&r);
    for (unsigned e = 0; e < count; e++) {
        const uint64_t mask = 0 - (uint64_t)(e == idx);
        const uint64_t* in = reinterpret_cast<const uint64_t*>(&table[e]);
        for (unsigned w = 0; w < sizeof(TeExt) / 8; w++) out[w] |= in[w] & mask;
    }
    return r;
}
// k * P for a SECRET scalar k (four 64-bit limbs, any value): 4-bit windows from the top, four doublings and one addition per window
// whatever the digits (digit 0 adds the identity: the unified formulas take it), table entries picked by mask.  The sequence of group
// operations and memory accesses does not depend on k.
inline TeExt te_mul_secret(const TeExt& P, const uint64_t k[4], const TeHostParams& c) {
    TeExt table[16];
    table[0] = te_identity();
    table[1] = P;
    for (int e = 2; e < 16; e++) table[e] = te_add(table[e - 1], P, c);
    TeExt acc = te_identity();
    for (int w = 63; w >= 0; w--) {
        for (int j = 0; j < 4; j++) acc = te_dbl(acc, c, j == 3);
        acc = te_add(acc, te_table_pick(table, 16, (unsigned)(k[w >> 4] >> (4 * (w & 15))) & 15u), c);
    }
    explicit_bzero(table, sizeof table);
    return acc;
}
// window table of a constant base: entry [w][e] = e * 16^w * P (e = 0: the identity); a multiplication is 64 additions, no doubling
struct TeFixedTable {
    std::vector<TeExt> t;       // 64 x 16
};
inline TeFixedTable te_fixed_table(const TeExt& P, const TeHostParams& c) {
    TeFixedTable ft;
    ft.t.resize(64 * 16);
    TeExt base = P;
    for (int w = 0; w < 64; w++) {
        TeExt* row = &ft.t[16 * w];
        row[0] = te_identity();
        row[1] = base;
        for (int e = 2; e < 16; e++) row[e] = te_add(row[e - 1], base, c);
        base = te_add(row[15], base, c);
    }
    return ft;
}
// k * Base from its table; `secret`: entries picked by mask (prover), otherwise indexed (verifier)
inline TeExt te_mul_fixed(const TeFixedTable& ft, const uint64_t k[4], const TeHostParams& c, bool secret) {
    TeExt acc = te_identity();
    for (int w = 0; w < 64; w++) {
        const unsigned nib = (unsigned)(k[w >> 4] >> (4 * (w & 15))) & 15u;
        if (secret) acc = te_add(acc, te_table_pick(&ft.t[16 * w], 16, nib), c);
        else if (nib) acc = te_add(acc, ft.t[16 * w + nib], c);
    }
    return acc;
}
// sum_i k_i * P_i for PUBLIC scalars (the verifier's combinations): Straus, 4-bit windows, one doubling chain for all points
inline TeExt te_msm_public(const TeExt* pts, const uint64_t (*ks)[4], size_t n, const TeHostParams& c) {
    std::vector<TeExt> table(n * 15);
    int top = -1;
    for (size_t i = 0; i < n; i++) {
        TeExt* t = &table[15 * i];
        t[0] = pts[i];
        for (int e = 1; e < 15; e++) t[e] = te_add(t[e - 1], pts[i], c);
        for (int w = 63; w > top; w--)
            if ((ks[i][w >> 4] >> (4 * (w & 15))) & 15u) { top = w; break; }
    }
    TeExt acc = te_identity();
    bool started = false;
    for (int w = top; w >= 0; w--) {
        if (started) for (int j = 0; j < 4; j++) acc = te_dbl(acc, c, j == 3);
        for (size_t i = 0; i < n; i++) {
            const unsigned nib = (unsigned)(ks[i][w >> 4] >> (4 * (w & 15))) & 15u;
            if (nib) { acc = te_add(acc, table[15 * i + nib - 1], c); started = true; }
        }
    }
    return acc;
}
// window tables of constant bases, built once per (curve, point) and shared (a table in use outlives its eviction from the cache)
inline std::shared_ptr<const TeFixedTable> te_fixed_table_cached(const TeCurveHost& cv, const uint8_t base_xy[64]) {
    static std::mutex m;
    static std::vector<std::pair<std::array<uint8_t, 65>, std::shared_ptr<const TeFixedTable>>> cache;
    std::array<uint8_t, 65> key;
    std::memcpy(key.data(), base_xy, 64);
    key[64] = (uint8_t)cv.id;
    std::lock_guard<std::mutex> lk(m);
    for (auto& e : cache)
        if (e.first == key) return e.second;
    TeExt P;
    if (!te_load_affine(base_xy, P)) return nullptr;
    auto t = std::make_shared<TeFixedTable>(te_fixed_table(P, te_host_params(cv)));
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.emplace_back(key, t);
    return t;
}
// the suite's two constant bases (generator, Pedersen blinding base) with their tables
struct TeSuiteTables {
    TeHostParams c;
    TeExt g, b;
    std::shared_ptr<const TeFixedTable> hold_g, hold_b;
    const TeFixedTable &tg, &tb;
    TeSuiteTables(const TeHostParams& c_, const TeExt& g_, const TeExt& b_, std::shared_ptr<const TeFixedTable> hg, std::shared_ptr<const TeFixedTable> hb)
        : c(c_), g(g_), b(b_), hold_g(std::move(hg)), hold_b(std::move(hb)), tg(*hold_g), tb(*hold_b) {}
};
inline std::shared_ptr<const TeSuiteTables> te_suite_tables(const VrfSuite& su) {
    TeExt g, b;
    if (!te_load_affine(su.generator, g) || !te_load_affine(su.blinding_base, b)) return nullptr;
    auto hg = te_fixed_table_cached(*su.cv, su.generator), hb = te_fixed_table_cached(*su.cv, su.blinding_base);
    if (!hg || !hb) return nullptr;
    return std::make_shared<const TeSuiteTables>(te_host_params(*su.cv), g, b, std::move(hg), std::move(hb));
}

// k * P on Bandersnatch for a SECRET k (the prover's x * I), P affine x || y, k 32 little-endian bytes; affine out.  Fixed schedule
// (te_mul_secret): the run time does not depend on the key's bit length or weight.
inline bool te_scalar_mul_host(const uint8_t p_xy[64], const uint8_t k_le[32], uint8_t out_xy[64]) {
    static const TeHostParams c = te_host_params(*te_curve(0));
    TeExt P;
    if (!te_load_affine(p_xy, P)) return false;
    uint64_t k[4];
    load_le32(k_le, k);
    TeExt r = te_mul_secret(P, k, c);
    te_store_affine(r, out_xy);
    explicit_bzero(k, sizeof k);
    explicit_bzero(&r, sizeof r);
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
