// Host-side (x86-64) field and G1 arithmetic for the thin CPU steps that surround the GPU kernels:
//   * combining the per-window sums of an MSM  (<= 17 points, 255 doublings — a serial chain that a single
//     GPU lane runs ~50x slower than one host core),
//   * projective -> affine conversion and byte encodings of single results,
//   * zcash G1 compression / decompression, Fr square roots for point decoding.
// 64-bit limbs, Montgomery form with R = 2^(64*N) — the same R as the device's 32-bit-limb Fr layout, so device Fr
// buffers are reinterpreted in place (little-endian limbs).
#pragma once
#include <cstdint>
#include <cstring>

namespace drh {

typedef unsigned __int128 u128;

template <int N>
struct FieldParams;   // P, R2, N0

template <>
struct FieldParams<4> {   // Fr
    static constexpr uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
    static constexpr uint64_t N0 = 0xfffffffeffffffffULL;
};
template <>
struct FieldParams<6> {   // Fq
    static constexpr uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                      0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static constexpr uint64_t R2[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                                       0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
    static constexpr uint64_t N0 = 0x89f3fffcfffcfffdULL;
};

template <int N>
struct Fe {
    uint64_t l[N];
    using FP = FieldParams<N>;

    static Fe zero() { Fe r; std::memset(r.l, 0, sizeof r.l); return r; }
    static Fe from_u64(uint64_t v) { Fe r = zero(); r.l[0] = v; return r.to_mont(); }
    static Fe one() { return from_u64(1); }
    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
    bool operator==(const Fe& o) const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i] ^ o.l[i]; return a == 0; }
    bool operator!=(const Fe& o) const { return !(*this == o); }

    static bool geq_p(const uint64_t* a) {
        for (int i = N - 1; i >= 0; i--) { if (a[i] > FP::P[i]) return true; if (a[i] < FP::P[i]) return false; }
        return true;
    }
    Fe operator+(const Fe& o) const {
        Fe r; u128 c = 0;
        for (int i = 0; i < N; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        if (c || geq_p(r.l)) r.sub_p();
        return r;
    }
    Fe operator-(const Fe& o) const {
        Fe r; uint64_t b = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)l[i] - o.l[i] - b; r.l[i] = (uint64_t)d; b = (uint64_t)(d >> 127); }
        if (b) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)r.l[i] + FP::P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    Fe neg() const { return is_zero() ? *this : zero() - *this; }
    void sub_p() {
        uint64_t b = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)l[i] - FP::P[i] - b; l[i] = (uint64_t)d; b = (uint64_t)(d >> 127); }
    }
    Fe operator*(const Fe& o) const {   // CIOS
        uint64_t t[N + 2] = {0};
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
            for (int j = 0; j < N; j++) { u128 p = (u128)l[j] * o.l[i] + t[j] + c; t[j] = (uint64_t)p; c = (uint64_t)(p >> 64); }
            u128 top = (u128)t[N] + c; t[N] = (uint64_t)top; t[N + 1] = (uint64_t)(top >> 64);
            uint64_t m = t[0] * FP::N0;
            u128 p = (u128)m * FP::P[0] + t[0]; c = (uint64_t)(p >> 64);
            for (int j = 1; j < N; j++) { p = (u128)m * FP::P[j] + t[j] + c; t[j - 1] = (uint64_t)p; c = (uint64_t)(p >> 64); }
            top = (u128)t[N] + c; t[N - 1] = (uint64_t)top; t[N] = t[N + 1] + (uint64_t)(top >> 64);
        }
        Fe r; std::memcpy(r.l, t, sizeof r.l);
        if (t[N] || geq_p(r.l)) r.sub_p();
        return r;
    }
    Fe sqr() const { return *this * *this; }
    Fe dbl() const { return *this + *this; }
    Fe to_mont() const { Fe r2; std::memcpy(r2.l, FP::R2, sizeof r2.l); return *this * r2; }
    Fe from_mont() const { Fe o = zero(); o.l[0] = 1; return *this * o; }
    // exponent: plain little-endian limbs
    Fe pow(const uint64_t* e, int el) const {
        Fe r = one(); bool started = false;
        for (int i = el - 1; i >= 0; i--)
            for (int b = 63; b >= 0; b--) {
                if (started) r = r.sqr();
                if ((e[i] >> b) & 1) { r = started ? r * *this : *this; started = true; }
            }
        return r;
    }
    Fe inv_fermat() const {
        uint64_t e[N]; std::memcpy(e, FP::P, sizeof e);
        uint64_t b = 2;   // p - 2
        for (int i = 0; i < N && b; i++) { uint64_t old = e[i]; e[i] -= b; b = old < b ? 1 : 0; }
        return pow(e, N);
    }
    // binary extended Euclid on the stored (Montgomery) value A = aR: gives A^-1 = a^-1 R^-1, then one Montgomery
    // product with R^3 brings it back to a^-1 R.  ~5x faster than the Fermat power; 0 -> 0.
    Fe inv() const {
        if (is_zero()) return *this;
        auto even = [](const uint64_t* a) { return (a[0] & 1) == 0; };
        auto shr1 = [](uint64_t* a, uint64_t top) { for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[N - 1] = (a[N - 1] >> 1) | (top << 63); };
        auto halve = [&](uint64_t* x) {                    // x/2 mod p
            uint64_t top = 0;
            if (x[0] & 1) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)x[i] + FP::P[i]; x[i] = (uint64_t)c; c >>= 64; } top = (uint64_t)c; }
            shr1(x, top);
        };
        auto geq = [](const uint64_t* a, const uint64_t* b) { for (int i = N - 1; i >= 0; i--) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; } return true; };
        auto sub = [](uint64_t* a, const uint64_t* b) { uint64_t bw = 0; for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - b[i] - bw; a[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); } return bw; };
        auto submod = [&](uint64_t* a, const uint64_t* b) { if (sub(a, b)) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)a[i] + FP::P[i]; a[i] = (uint64_t)c; c >>= 64; } } };
        auto is_one = [](const uint64_t* a) { uint64_t r = a[0] ^ 1; for (int i = 1; i < N; i++) r |= a[i]; return r == 0; };
        uint64_t u[N], v[N], x1[N] = {1}, x2[N] = {0};
        std::memcpy(u, l, sizeof u);
        std::memcpy(v, FP::P, sizeof v);
        while (!is_one(u) && !is_one(v)) {
            while (even(u)) { shr1(u, 0); halve(x1); }
            while (even(v)) { shr1(v, 0); halve(x2); }
            if (geq(u, v)) { sub(u, v); submod(x1, x2); } else { sub(v, u); submod(x2, x1); }
        }
        Fe r; std::memcpy(r.l, is_one(u) ? x1 : x2, sizeof r.l);
        static const Fe r3 = [] { Fe r2; std::memcpy(r2.l, FP::R2, sizeof r2.l); return r2 * r2; }();
        return r * r3;
    }
    // standard-form (non-Montgomery) comparison helper: returns true if a > b as integers
    static bool gt_std(const Fe& a, const Fe& b) {
        for (int i = N - 1; i >= 0; i--) { if (a.l[i] > b.l[i]) return true; if (a.l[i] < b.l[i]) return false; }
        return false;
    }
    // byte I/O on the standard form
    static bool load_le(Fe& out, const uint8_t* in) {   // returns false if >= p
        Fe t;
        for (int i = 0; i < N; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w |= (uint64_t)in[8 * i + j] << (8 * j); t.l[i] = w; }
        if (geq_p(t.l)) return false;
        out = t.to_mont(); return true;
    }
    static bool load_be(Fe& out, const uint8_t* in) {
        uint8_t le[8 * N];
        for (int i = 0; i < 8 * N; i++) le[i] = in[8 * N - 1 - i];
        return load_le(out, le);
    }
    void store_le(uint8_t* out) const {
        Fe s = from_mont();
        for (int i = 0; i < N; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(s.l[i] >> (8 * j));
    }
    void store_be(uint8_t* out) const {
        uint8_t le[8 * N]; store_le(le);
        for (int i = 0; i < 8 * N; i++) out[i] = le[8 * N - 1 - i];
    }
};

using Fr = Fe<4>;
using Fq = Fe<6>;

// ---- Fr square root: Tonelli-Shanks, p - 1 = Q * 2^32, non-residue 5
// (the reference's sqrt_mod_bls_scalar_cy, dot_ring/curve/native_field/bandersnatch_te.pyx:421-477)
inline bool fr_sqrt(Fr& out, const Fr& x) {
    if (x.is_zero()) { out = x; return true; }
    static const uint64_t Q[4] = {0xfffe5bfeffffffffULL, 0x09a1d80553bda402ULL, 0x299d7d483339d808ULL, 0x0000000073eda753ULL};
    static const uint64_t Q1H[4] = {0x7fff2dff80000000ULL, 0x04d0ec02a9ded201ULL, 0x94cebea4199cec04ULL, 0x0000000039f6d3a9ULL};
    Fr one = Fr::one();
    Fr t = x.pow(Q, 4), R = x.pow(Q1H, 4);
    Fr c = Fr::from_u64(5).pow(Q, 4);
    int M = 32;
    for (;;) {
        if (t == one) { out = R; return true; }
        int i = 1;
        Fr tmp = t.sqr();
        while (tmp != one) { tmp = tmp.sqr(); if (++i >= M) return false; }
        Fr b = c;
        for (int j = 0; j < M - i - 1; j++) b = b.sqr();
        M = i; c = b.sqr(); t = t * c; R = R * b;
    }
}

// ---- G1 in XYZZ coordinates (host copy of the device formulas, curve.hip.h)
struct G1 {
    Fq x, y, zz, zzz;
    bool is_inf() const { return zz.is_zero(); }
    static G1 inf() { G1 r; r.x = r.y = r.zz = r.zzz = Fq::zero(); return r; }
};

inline G1 g1_dbl(const G1& p) {
    if (p.is_inf()) return p;
    Fq U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V, X2 = p.x.sqr(), M = X2.dbl() + X2;
    G1 r;
    r.x = M.sqr() - S - S;
    r.y = M * (S - r.x) - W * p.y;
    r.zz = V * p.zz;
    r.zzz = W * p.zzz;
    return r;
}
inline G1 g1_add(const G1& p, const G1& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fq U1 = p.x * q.zz, U2 = q.x * p.zz, S1 = p.y * q.zzz, S2 = q.y * p.zzz;
    Fq P = U2 - U1, R = S2 - S1;
    if (P.is_zero()) return R.is_zero() ? g1_dbl(p) : G1::inf();
    Fq PP = P.sqr(), PPP = P * PP, Q = U1 * PP;
    G1 r;
    r.x = R.sqr() - PPP - Q - Q;
    r.y = R * (Q - r.x) - S1 * PPP;
    r.zz = p.zz * q.zz * PP;
    r.zzz = p.zzz * q.zzz * PPP;
    return r;
}
// affine (Montgomery) coordinates; returns false for infinity
inline bool g1_to_affine(const G1& p, Fq& ax, Fq& ay) {
    if (p.is_inf()) return false;
    Fq zi3 = p.zzz.inv();            // 1/ZZZ
    Fq zi2 = (p.zz * zi3).sqr();      // (ZZ/ZZZ)^2 = ZZ^2/ZZ^3 = 1/ZZ   (ZZ^3 = ZZZ^2)
    ax = p.x * zi2;
    ay = p.y * zi3;
    return true;
}
inline bool g1_on_curve(const Fq& x, const Fq& y) {
    return y.sqr() == x.sqr() * x + Fq::from_u64(4);
}

}  // namespace drh
