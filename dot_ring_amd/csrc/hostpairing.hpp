// Host-side BLS12-381 optimal-ate pairing product check:  prod_i e(P_i, Q_i) == 1.
// The KZG verifier needs two Miller loops and one final exponentiation per *batch* (reference:
// dot_ring/ring_proof/pcs/kzg.py:216-230,298-301,335-338 through blst's PT / finalverify, pcs/pairing.py:24-31);
// SURVEY §8 keeps this on the CPU.  Tower Fq2 = Fq[u]/(u^2+1),
// Fq6 = Fq2[v]/(v^3 - (u+1)), Fq12 = Fq6[w]/(w^2 - v).  The check itself runs on prepared G2 points (line coefficients
// precomputed per fixed Q), sparse line products and a final exponentiation by Frobenius maps + cyclotomic squarings;
// the affine Miller loop and the plain square-and-multiply exponentiation ((f^(p^6-1))^(p^2+1))^((p^4-p^2+1)/r)
// (pairing_consts.hpp, generated from the curve parameters with Python big ints) stay as the references the self-check
// compares them with.
//
// Line functions: for the M-type twist E': y^2 = x^3 + 4(u+1) and T = (xT, yT) in E'(Fq2), slope s, P = (xP, yP):
//   l * w^3 = (s*xT - yT) + (-s*xP) * v + yP * v*w          (w^3 lies in Fq4, killed by the final exponentiation)
#pragma once
#include <vector>

#include "hostmath.hpp"
#include "pairing_consts.hpp"

namespace drh {

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    Fq2 operator*(const Fq2& o) const {
        Fq a = c0 * o.c0, b = c1 * o.c1;
        return {a - b, (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    Fq2 sqr() const {                                          // (c0 + c1)(c0 - c1) + 2 c0 c1 u
        Fq m = c0 * c1;
        return {(c0 + c1) * (c0 - c1), m + m};
    }
    Fq2 scale(const Fq& k) const { return {c0 * k, c1 * k}; }
    Fq2 mul_xi() const { return {c0 - c1, c0 + c1}; }          // * (1 + u)
    Fq2 conj() const { return {c0, c1.neg()}; }
    Fq2 inv() const {
        Fq n = (c0.sqr() + c1.sqr()).inv();
        return {c0 * n, (c1 * n).neg()};
    }
};

struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fq6 operator+(const Fq6& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    Fq6 operator*(const Fq6& o) const {      // schoolbook with v^3 = xi
        Fq2 t0 = c0 * o.c0, t1 = c1 * o.c1, t2 = c2 * o.c2;
        Fq2 r0 = t0 + ((c1 + c2) * (o.c1 + o.c2) - t1 - t2).mul_xi();
        Fq2 r1 = (c0 + c1) * (o.c0 + o.c1) - t0 - t1 + t2.mul_xi();
        Fq2 r2 = (c0 + c2) * (o.c0 + o.c2) - t0 - t2 + t1;
        return {r0, r1, r2};
    }
    Fq6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    Fq6 inv() const {
        Fq2 a = c0.sqr() - (c1 * c2).mul_xi();
        Fq2 b = c2.sqr().mul_xi() - c0 * c1;
        Fq2 c = c1.sqr() - c0 * c2;
        Fq2 d = ((c2 * b + c1 * c).mul_xi() + c0 * a).inv();
        return {a * d, b * d, c * d};
    }
};

struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq12 operator*(const Fq12& o) const {
        Fq6 a = c0 * o.c0, b = c1 * o.c1;
        return {a + b.mul_v(), (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    Fq12 sqr() const {                                           // complex squaring: 2 Fq6 products
        Fq6 ab = c0 * c1;
        Fq6 re = (c0 + c1) * (c0 + c1.mul_v()) - ab - ab.mul_v();
        return {re, ab + ab};
    }
    // this * (A + B v + C v w) with A, B in Fq2 and C in Fq — the shape of a Miller-loop line: 12 Fq2 products + 6 Fq
    // products instead of 18 Fq2 products
    Fq12 mul_by_line(const Fq2& A, const Fq2& B, const Fq& C) const {
        auto mul_ab0 = [](const Fq6& f, const Fq2& a, const Fq2& b) {          // f * (a + b v)
            Fq2 t0 = f.c0 * a, t1 = f.c1 * b;
            return Fq6{t0 + (f.c2 * b).mul_xi(), (f.c0 + f.c1) * (a + b) - t0 - t1, t1 + f.c2 * a};
        };
        Fq6 t0 = mul_ab0(c0, A, B);
        Fq6 t1{c1.c2.scale(C).mul_xi(), c1.c0.scale(C), c1.c1.scale(C)};       // c1 * (C v)
        Fq2 BC{B.c0 + C, B.c1};
        Fq6 t2 = mul_ab0(c0 + c1, A, BC);
        return {t0 + t1.mul_v(), t2 - t0 - t1};
    }
    // squaring in the cyclotomic subgroup (after the easy part of the final exponentiation; Granger-Scott 2010: the element
    // is three Fq4 pairs, each squared with 3 Fq2 squarings): 18 Fq products against 36 for sqr()
    Fq12 cyclotomic_sqr() const {
        auto fq4_sqr = [](const Fq2& a, const Fq2& b, Fq2& r0, Fq2& r1) {      // (a + b t)^2 with t^2 = xi
            Fq2 t0 = a.sqr(), t1 = b.sqr();
            r0 = t1.mul_xi() + t0;
            r1 = (a + b).sqr() - t0 - t1;
        };
        Fq2 z0 = c0.c0, z4 = c0.c1, z3 = c0.c2, z2 = c1.c0, z1 = c1.c1, z5 = c1.c2, t0, t1, t2, t3;
        fq4_sqr(z0, z1, t0, t1);
        z0 = t0 - z0; z0 = z0 + z0 + t0;
        z1 = t1 + z1; z1 = z1 + z1 + t1;
        fq4_sqr(z2, z3, t0, t1);
        fq4_sqr(z4, z5, t2, t3);
        z4 = t0 - z4; z4 = z4 + z4 + t0;
        z5 = t1 + z5; z5 = z5 + z5 + t1;
        t0 = t3.mul_xi();
        z2 = t0 + z2; z2 = z2 + z2 + t0;
        z3 = t2 - z3; z3 = z3 + z3 + t2;
        return {Fq6{z0, z4, z3}, Fq6{z2, z1, z5}};
    }
    Fq12 conj() const { return {c0, c1.neg()}; }               // = x^(p^6)
    Fq12 inv() const {
        Fq6 d = (c0 * c0 - (c1 * c1).mul_v()).inv();
        return {c0 * d, (c1 * d).neg()};
    }
    Fq12 pow(const uint64_t* e, int len) const {
        Fq12 r = one();
        bool started = false;
        for (int i = len - 1; i >= 0; i--)
            for (int b = 63; b >= 0; b--) {
                if (started) r = r.sqr();
                if ((e[i] >> b) & 1) { r = started ? r * *this : *this; started = true; }
            }
        return r;
    }
};

struct G2Affine {
    Fq2 x, y;
    bool inf;
};

inline bool g2_on_curve(const G2Affine& q) {
    if (q.inf) return true;
    Fq2 b = Fq2{Fq::from_u64(4), Fq::from_u64(4)};
    return q.y.sqr() == q.x.sqr() * q.x + b;
}

// affine group law on the twist (setup-time only: tau * G2 for a known-tau test SRS)
inline G2Affine g2_add(const G2Affine& a, const G2Affine& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fq2 s;
    if (a.x == b.x) {
        if (!(a.y == b.y) || a.y.is_zero()) return {Fq2::zero(), Fq2::zero(), true};
        s = a.x.sqr().scale(Fq::from_u64(3)) * (a.y + a.y).inv();
    } else {
        s = (b.y - a.y) * (b.x - a.x).inv();
    }
    Fq2 x = s.sqr() - a.x - b.x;
    return {x, s * (a.x - x) - a.y, false};
}
inline G2Affine g2_mul(const G2Affine& q, const uint8_t scalar_le[32]) {
    G2Affine r{Fq2::zero(), Fq2::zero(), true};
    for (int bit = 255; bit >= 0; bit--) {
        r = g2_add(r, r);
        if ((scalar_le[bit >> 3] >> (bit & 7)) & 1) r = g2_add(r, q);
    }
    return r;
}

// one Miller loop f_{|x|,Q}(P), conjugated for the negative BLS parameter x = -0xd201000000010000
inline Fq12 miller_loop(const Fq& px, const Fq& py, const G2Affine& q) {
    const uint64_t X = 0xd201000000010000ULL;
    Fq2 tx = q.x, ty = q.y;
    Fq12 f = Fq12::one();
    auto line = [&](const Fq2& s, const Fq2& lx, const Fq2& ly) {
        Fq12 l;
        l.c0 = {s * lx - ly, s.scale(px).neg(), Fq2::zero()};
        l.c1 = {Fq2::zero(), Fq2{py, Fq::zero()}, Fq2::zero()};
        return l;
    };
    Fq three = Fq::from_u64(3);
    for (int b = 62; b >= 0; b--) {
        Fq2 s = tx.sqr().scale(three) * (ty + ty).inv();
        f = f.sqr() * line(s, tx, ty);
        Fq2 nx = s.sqr() - tx - tx;
        ty = s * (tx - nx) - ty;
        tx = nx;
        if ((X >> b) & 1) {
            Fq2 s2 = (q.y - ty) * (q.x - tx).inv();
            f = f * line(s2, tx, ty);
            Fq2 ax = s2.sqr() - tx - q.x;
            ty = s2 * (tx - ax) - ty;
            tx = ax;
        }
    }
    return f.conj();
}

// product of Miller loops with ONE accumulator (one Fq12 squaring per bit for all pairs) and the slope denominators
// of all pairs inverted together (Montgomery's trick: one Fq2 inversion per step instead of one per pair)
inline Fq12 multi_miller_loop(const Fq* px, const Fq* py, const G2Affine* qs, size_t n) {
    const uint64_t X = 0xd201000000010000ULL;
    Fq12 f = Fq12::one();
    if (n == 0) return f;
    std::vector<Fq2> tx(n), ty(n), den(n), pre(n);
    for (size_t i = 0; i < n; i++) { tx[i] = qs[i].x; ty[i] = qs[i].y; }
    auto invert_all = [&]() {                              // den[i] <- 1/den[i]
        Fq2 run = Fq2::one();
        for (size_t i = 0; i < n; i++) { pre[i] = run; run = run * den[i]; }
        Fq2 inv = run.inv();
        for (size_t i = n; i-- > 0;) { Fq2 d = den[i]; den[i] = inv * pre[i]; inv = inv * d; }
    };
    auto line = [&](const Fq2& s, const Fq2& lx, const Fq2& ly, const Fq& x, const Fq& y) {
        Fq12 l;
        l.c0 = {s * lx - ly, s.scale(x).neg(), Fq2::zero()};
        l.c1 = {Fq2::zero(), Fq2{y, Fq::zero()}, Fq2::zero()};
        return l;
    };
    Fq three = Fq::from_u64(3);
    for (int b = 62; b >= 0; b--) {
        f = f.sqr();
        for (size_t i = 0; i < n; i++) den[i] = ty[i] + ty[i];
        invert_all();
        for (size_t i = 0; i < n; i++) {
            Fq2 s = tx[i].sqr().scale(three) * den[i];
            f = f * line(s, tx[i], ty[i], px[i], py[i]);
            Fq2 nx = s.sqr() - tx[i] - tx[i];
            ty[i] = s * (tx[i] - nx) - ty[i];
            tx[i] = nx;
        }
        if ((X >> b) & 1) {
            for (size_t i = 0; i < n; i++) den[i] = qs[i].x - tx[i];
            invert_all();
            for (size_t i = 0; i < n; i++) {
                Fq2 s = (qs[i].y - ty[i]) * den[i];
                f = f * line(s, tx[i], ty[i], px[i], py[i]);
                Fq2 ax = s.sqr() - tx[i] - qs[i].x;
                ty[i] = s * (tx[i] - ax) - ty[i];
                tx[i] = ax;
            }
        }
    }
    return f.conj();
}

// The verifier's G2 points are fixed (the SRS's H and tau H): the slopes and the constant terms of all 68 lines of a
// Miller loop depend on Q alone, so they are computed once per Q (G2Prepared) and a pairing is then 63 squarings and
// 68 sparse products per pair with no inversion and no curve arithmetic: line_k(P) = A_k + (-s_k xP) v + yP v w.
struct G2Prepared {
    std::vector<Fq2> s, a;             // per step (doublings and additions in loop order): slope, s * xT - yT
};
inline G2Prepared g2_prepare(const G2Affine& q) {
    const uint64_t X = 0xd201000000010000ULL;
    G2Prepared pr;
    Fq2 tx = q.x, ty = q.y;
    Fq three = Fq::from_u64(3);
    for (int b = 62; b >= 0; b--) {
        Fq2 s = tx.sqr().scale(three) * (ty + ty).inv();
        pr.s.push_back(s); pr.a.push_back(s * tx - ty);
        Fq2 nx = s.sqr() - tx - tx;
        ty = s * (tx - nx) - ty;
        tx = nx;
        if ((X >> b) & 1) {
            Fq2 s2 = (q.y - ty) * (q.x - tx).inv();
            pr.s.push_back(s2); pr.a.push_back(s2 * tx - ty);
            Fq2 ax = s2.sqr() - tx - q.x;
            ty = s2 * (tx - ax) - ty;
            tx = ax;
        }
    }
    return pr;
}
inline Fq12 multi_miller_loop_prepared(const Fq* px, const Fq* py, const G2Prepared* const* qs, size_t n) {
    const uint64_t X = 0xd201000000010000ULL;
    Fq12 f = Fq12::one();
    if (n == 0) return f;
    std::vector<Fq> nx(n);
    for (size_t i = 0; i < n; i++) nx[i] = px[i].neg();
    size_t k = 0;
    for (int b = 62; b >= 0; b--) {
        f = f.sqr();
        for (size_t i = 0; i < n; i++) f = f.mul_by_line(qs[i]->a[k], qs[i]->s[k].scale(nx[i]), py[i]);
        k++;
        if ((X >> b) & 1) {
            for (size_t i = 0; i < n; i++) f = f.mul_by_line(qs[i]->a[k], qs[i]->s[k].scale(nx[i]), py[i]);
            k++;
        }
    }
    return f.conj();
}

// reference form (plain square-and-multiply): f^((p^12 - 1)/r)
inline Fq12 final_exponentiation(const Fq12& f) {
    Fq12 t = f.conj() * f.inv();                       // f^(p^6 - 1)
    t = t.pow(FE_EASY2, FE_EASY2_LEN);                 // ^(p^2 + 1)
    return t.pow(FE_HARD, FE_HARD_LEN);                // ^((p^4 - p^2 + 1)/r)
}

// ---- fast form used by the pairing check --------------------------------------------------------------------------
// Frobenius: with Fq12 = Fq2[w]/(w^6 - xi), an element is sum_k a_k w^k (a_0 = c0.c0, a_1 = c1.c0, a_2 = c0.c1,
// a_3 = c1.c1, a_4 = c0.c2, a_5 = c1.c2) and a^p = sum_k conj(a_k) * gamma^k * w^k with gamma = xi^((p-1)/6).
inline Fq2 fq2_pow(const Fq2& a, const uint64_t* e, int len) {
    Fq2 r = Fq2::one();
    for (int i = len - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            r = r.sqr();
            if ((e[i] >> b) & 1) r = r * a;
        }
    return r;
}
struct FrobeniusConsts {
    Fq2 g[6];
    FrobeniusConsts() {
        uint64_t e[6];
        std::memcpy(e, FieldParams<6>::P, sizeof e);
        e[0] -= 1;                                         // p - 1 (p is odd: no borrow)
        u128 rem = 0;                                      // / 6
        for (int i = 5; i >= 0; i--) { u128 cur = (rem << 64) | e[i]; e[i] = (uint64_t)(cur / 6); rem = cur % 6; }
        Fq2 xi{Fq::one(), Fq::one()};
        g[0] = Fq2::one();
        g[1] = fq2_pow(xi, e, 6);
        for (int k = 2; k < 6; k++) g[k] = g[k - 1] * g[1];
    }
};
inline const FrobeniusConsts& frobenius_consts() { static FrobeniusConsts c; return c; }
inline Fq12 frobenius(const Fq12& a) {
    const FrobeniusConsts& fc = frobenius_consts();
    Fq12 r;
    r.c0.c0 = a.c0.c0.conj();
    r.c1.c0 = a.c1.c0.conj() * fc.g[1];
    r.c0.c1 = a.c0.c1.conj() * fc.g[2];
    r.c1.c1 = a.c1.c1.conj() * fc.g[3];
    r.c0.c2 = a.c0.c2.conj() * fc.g[4];
    r.c1.c2 = a.c1.c2.conj() * fc.g[5];
    return r;
}
// a^x for the (negative) BLS parameter x and a in the cyclotomic subgroup (inverse = conjugate)
inline Fq12 cyclotomic_exp_x(const Fq12& a) {
    const uint64_t X = 0xd201000000010000ULL;
    Fq12 r = a;
    for (int b = 62; b >= 0; b--) {
        r = r.cyclotomic_sqr();
        if ((X >> b) & 1) r = r * a;
    }
    return r.conj();
}
// f^(3 (p^12 - 1)/r): easy part with Frobenius maps, hard part through
//   3 (p^4 - p^2 + 1)/r = (x - 1)^2 (x + p) (x^2 + p^2 - 1) + 3
// (five exponentiations by x).  The extra factor 3 is coprime to r, so "== 1" is unchanged.
inline Fq12 final_exponentiation_check(const Fq12& f) {
    Fq12 t = f.conj() * f.inv();                       // f^(p^6 - 1)
    t = frobenius(frobenius(t)) * t;                   // ^(p^2 + 1): now in the cyclotomic subgroup
    Fq12 a = cyclotomic_exp_x(t) * t.conj();           // t^(x-1)
    a = cyclotomic_exp_x(a) * a.conj();                // t^((x-1)^2)
    Fq12 b = cyclotomic_exp_x(a) * frobenius(a);       // ^(x + p)
    Fq12 c = cyclotomic_exp_x(cyclotomic_exp_x(b)) * frobenius(frobenius(b)) * b.conj();    // ^(x^2 + p^2 - 1)
    return c * t.cyclotomic_sqr() * t;
}

}  // namespace drh
