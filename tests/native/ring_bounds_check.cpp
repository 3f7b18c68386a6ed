// Interval check of the lazy-reduction bookkeeping in dot_ring_amd/csrc/ring_body.hip.h (host, g++).
// FsB stands in for dr::Fs: instead of nine limbs it carries the RANGE of the low limbs (0..7) and of the value in units of p,
// every operation asserts the preconditions fr29.hip.h states for it and returns the range of its result.  The bodies the
// device kernels run (constraint point, quotient / aggregation / Horner steps, 24 NTT stages) are instantiated with worst-case
// input ranges; any violated precondition aborts with the operation and the offending numbers.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace dr {

static const double TWO29 = 536870912.0, TWO31 = 2147483648.0, TWO32 = 4294967296.0;
static const double MUL_LIMB_LIMIT = std::pow(2.0, 59.3), MUL_VALUE_LIMIT = 35.0, R_OVER_P = 70.7;
static const double P_TOP = 7597479.33;          // p / 2^232: the top limb of a value v p is about v * P_TOP

static const char* g_where = "";
static void fail(const char* op, const char* what, double a, double b) {
    std::fprintf(stderr, "BOUND VIOLATION in %s: %s: %s (%.6g, %.6g)\n", g_where, op, what, a, b);
    std::exit(1);
}

struct FsB {
    double lmin, lmax;               // low limbs 0..7
    double vmin, vmax;               // value / p
    static FsB make(double l0, double l1, double v0, double v1) { return FsB{l0, l1, v0, v1}; }
    static FsB one() { return make(0, TWO29 - 1, 0, 1); }                       // 2^261 mod p, canonical limbs
    double labs() const { return std::max(std::fabs(lmin), std::fabs(lmax)); }
    double vabs() const { return std::max(std::fabs(vmin), std::fabs(vmax)); }
    // magnitude of the top limb: the value's share plus what uncarried low limbs can push into it
    double top() const { return vabs() * P_TOP + labs() / TWO29 + 2; }
    double maxlimb() const { return std::max(labs(), top()); }
};

static void check_int32(const char* op, const FsB& r) {
    if (r.lmin <= -TWO31 || r.lmax >= TWO31) fail(op, "limb leaves int32", r.lmin, r.lmax);
    if (r.top() >= TWO31) fail(op, "top limb leaves int32", r.top(), 0);
}
// sums of non-negative limbs may use all 32 bits (the device adds are the same instructions; only carry_u may read such a value)
static void check_sum(const char* op, const FsB& r) {
    if (r.lmin >= 0 && r.lmax < TWO32 && r.top() < TWO31) return;
    check_int32(op, r);
}
static FsB add(const FsB& a, const FsB& b) { FsB r = FsB::make(a.lmin + b.lmin, a.lmax + b.lmax, a.vmin + b.vmin, a.vmax + b.vmax); check_sum("add", r); return r; }
static FsB sub(const FsB& a, const FsB& b) {
    check_int32("sub (operand)", a); check_int32("sub (operand)", b);
    FsB r = FsB::make(a.lmin - b.lmax, a.lmax - b.lmin, a.vmin - b.vmax, a.vmax - b.vmin); check_int32("sub", r); return r;
}
static FsB dbl(const FsB& a) { return add(a, a); }
static FsB neg(const FsB& a) { check_int32("neg (operand)", a); return FsB::make(-a.lmax, -a.lmin, -a.vmax, -a.vmin); }
static FsB carry(const FsB& a) {
    check_int32("carry (operand)", a);
    FsB r = FsB::make(0, TWO29 - 1, a.vmin, a.vmax);
    if (r.top() >= TWO31) fail("carry", "top limb leaves int32", r.top(), 0);
    return r;
}
static FsB carry_u(const FsB& a) {                 // unsigned carry pass: limbs must be non-negative and below 2^32
    if (a.lmin < 0 || a.lmax >= TWO32) fail("carry_u", "limbs outside [0, 2^32)", a.lmin, a.lmax);
    return FsB::make(0, TWO29 - 1, a.vmin, a.vmax);
}
static FsB sub_3p(const FsB& a) { check_int32("sub_3p (operand)", a); return FsB::make(a.lmin - (TWO29 - 1), a.lmax, a.vmin - 3, a.vmax - 3); }
static FsB product(const char* op, double lo, double hi) {
    if (std::max(std::fabs(lo), std::fabs(hi)) > MUL_VALUE_LIMIT) fail(op, "|a b| exceeds 35 p^2", lo, hi);
    return FsB::make(0, TWO29 - 1, lo / R_OVER_P, hi / R_OVER_P + 1);
}
static void range_of_product(const FsB& a, const FsB& b, double& lo, double& hi) {
    const double c[4] = {a.vmin * b.vmin, a.vmin * b.vmax, a.vmax * b.vmin, a.vmax * b.vmax};
    lo = *std::min_element(c, c + 4);
    hi = *std::max_element(c, c + 4);
}
static FsB mul(const FsB& a, const FsB& b) {
    check_int32("mul (operand)", a); check_int32("mul (operand)", b);
    if (a.maxlimb() * b.maxlimb() > MUL_LIMB_LIMIT) fail("mul", "limb bounds exceed 2^59.3", a.maxlimb(), b.maxlimb());
    double lo, hi;
    range_of_product(a, b, lo, hi);
    return product("mul", lo, hi);
}
static FsB mul2(const FsB& a, const FsB& b, const FsB& c, const FsB& d) {      // a b + c d with one reduction: every limb below 2^29
    for (const FsB* x : {&a, &b, &c, &d})
        if (x->maxlimb() >= TWO29 + 1) fail("mul2", "operand limb reaches 2^29", x->maxlimb(), 0);
    double l1, h1, l2, h2;
    range_of_product(a, b, l1, h1);
    range_of_product(c, d, l2, h2);
    return product("mul2", l1 + l2, h1 + h2);
}
static FsB reduce_small(const FsB& a) {
    check_int32("reduce_small (operand)", a);
    if (a.vabs() >= 30) fail("reduce_small", "|value| reaches 30 p", a.vmin, a.vmax);
    return FsB::make(0, TWO29 - 1, -0.51, 0.51);
}

// the two ways into canonical words
static void canon29(const FsB& a) {
    check_int32("canon29 (operand)", a);
    if (a.vabs() >= 8) fail("canon29", "|value| reaches 8 p", a.vmin, a.vmax);
}
static void canon29_small(const FsB& a) {
    check_int32("canon29_small (operand)", a);
    if (a.vmin <= -1 || a.vmax >= 3) fail("canon29_small", "value outside (-p, 3p)", a.vmin, a.vmax);
}

}  // namespace dr

#include "ring_body.hip.h"

using dr::FsB;

static void expect_within(const char* what, const FsB& r, double labs, double vabs) {
    if (r.labs() > labs || r.vabs() > vabs) {
        std::fprintf(stderr, "RESULT RANGE of %s wider than stated: limbs (%.6g, %.6g), value (%.4g, %.4g)\n", what, r.lmin, r.lmax, r.vmin, r.vmax);
        std::exit(1);
    }
    std::printf("%-28s limbs (%.4g, %.4g)  value (%.3f, %.3f) p\n", what, r.lmin, r.lmax, r.vmin, r.vmax);
}

int main() {
    const FsB canonical = FsB::make(0, dr::TWO29 - 1, 0, 1);                  // table entries, unpacked standard-form coefficients
    const FsB normal = FsB::make(0, dr::TWO29 - 1, -0.04, 1.04);              // products of normal operands
    const FsB reduced = FsB::make(0, dr::TWO29 - 1, -0.51, 0.51);             // reduce_small output (the forward NTT's FS9 results)
    const FsB lazy_sum = FsB::make(0, dr::TWO29 * 2, -0.04, 2.04);            // a Horner accumulator
    for (int cv = 0; cv < 2; cv++) {
        dr::g_where = cv ? "constraints (JubJub)" : "constraints (Bandersnatch)";
        const FsB r = cv ? dr::body_constraints<1>(reduced, reduced, reduced, reduced, reduced, reduced, reduced, canonical, canonical, canonical,
                                                   canonical, canonical, canonical, normal, normal, normal, normal, normal, normal, normal, normal, normal)
                         : dr::body_constraints<0>(reduced, reduced, reduced, reduced, reduced, reduced, reduced, canonical, canonical, canonical,
                                                   canonical, canonical, canonical, normal, normal, normal, normal, normal, normal, normal, normal, normal);
        expect_within(dr::g_where, r, dr::TWO29 * 3, 3.3);
    }
    {   // the hidden rows of coset 0 (k_ring_hidden_rows): ring point converted from the 2^256 form (normal, not canonical), zero selector / Lagrange rows
        dr::g_where = "constraints (hidden rows)";
        const FsB zero = FsB::make(0, 0, 0, 0);
        expect_within(dr::g_where, dr::body_constraints<0>(reduced, reduced, reduced, reduced, reduced, reduced, reduced, normal, normal, zero, zero, zero, canonical,
                                                           normal, normal, normal, normal, normal, normal, normal, zero, zero), dr::TWO29 * 3, 3.3);
    }
    dr::g_where = "quotient";
    {
        const FsB fold = FsB::make(0, 4 * (dr::TWO29 - 1), 0, 4);            // up to four canonical coefficients added
        const FsB q = dr::body_quotient(normal, fold, normal, fold, normal, fold, normal, fold);
        expect_within("quotient", q, dr::TWO29 * 2, 2.3);
        dr::canon29_small(q);
    }
    dr::g_where = "horner";
    {
        FsB acc = FsB::make(0, 0, 0, 0);
        for (int i = 0; i < 64; i++) acc = dr::body_horner(acc, normal, canonical);
        expect_within("horner (64 steps)", acc, lazy_sum.lmax, 2.04);
        const FsB part = dr::mul(acc, normal);                                 // times x^lo: a normal value again
        expect_within("horner * x^lo", part, dr::TWO29, 1.04);
        // block reduction of 256 partial values (k_ring_eval): a tree of additions with a carry per level, reduce_small after the
        // fourth and the last level, then canonical words
        dr::g_where = "horner tree";
        FsB sum = part;
        for (int level = 0; level < 8; level++) {
            sum = dr::carry(dr::add(sum, sum));
            if (level == 3 || level == 7) sum = dr::reduce_small(sum);
        }
        dr::canon29_small(sum);
    }
    dr::g_where = "agg8 / lin3";
    {
        const FsB nu[8] = {normal, normal, normal, normal, normal, normal, normal, normal};
        const FsB c[8] = {canonical, canonical, canonical, canonical, canonical, canonical, canonical, canonical};
        const FsB a8 = dr::body_agg8(nu, c), l3 = dr::body_lin3(normal, canonical, normal, canonical, normal, canonical);
        expect_within("agg8", a8, dr::TWO29 * 3, 4.2);
        expect_within("lin3", l3, dr::TWO29 * 2, 2.1);
        dr::canon29(a8);
        dr::canon29_small(l3);
    }
    // the NTT network: worst case = an element that is the upper operand in every stage.  Inputs: fs_from_std products (normal) or
    // the constraint kernel's raw sums (limbs within (-2^30, 2^30 + 2^29), |value| < 3.3 p); passes of <= 10 and <= 4 stages, the
    // first stage of every pass and every second one after it carry both operands.
    for (int variant = 0; variant < 2; variant++) {
        dr::g_where = variant ? "ntt (constraint-kernel input)" : "ntt (normal input)";
        const FsB in = variant ? FsB::make(-dr::TWO29 * 2, dr::TWO29 * 3, -3.3, 3.3) : normal;
        FsB u = in, v = in;
        // normal input: 24 stages (the C ABI's largest transform); the constraint kernel's sums only enter the prover's own
        // 4N-point inverse transform (14 stages at N = 4096): 16 stages
        const int pass_len[] = {10, 4, variant ? 2 : 4, variant ? 0 : 4, variant ? 0 : 2};
        int stage = 0;
        for (int pass = 0; pass < 5; pass++)
            for (int s = 1; s <= pass_len[pass]; s++, stage++) {
                FsB a = u, b = v;                                              // both candidates for the next stage's operands
                dr::body_butterfly(a, b, normal, stage == 0, (s & 1) != 0);
                // the next stage may pick either output as its upper or lower operand: keep the union of the ranges
                const FsB w = FsB::make(std::min(a.lmin, b.lmin), std::max(a.lmax, b.lmax), std::min(a.vmin, b.vmin), std::max(a.vmax, b.vmax));
                u = w;
                v = w;
            }
        expect_within(dr::g_where, u, dr::TWO31, 33.0);
        // the way out of the last stage of a 2^24-point transform: times the (standard-form or Montgomery) factor
        dr::g_where = "ntt store";
        (void)dr::mul(dr::carry(u), canonical);
    }
    // raw FS9 output without a factor leaves through reduce_small (|value| < 30 p): ntt_run admits it up to 2^16 points
    {
        dr::g_where = "ntt (16 stages, raw output)";
        FsB u = normal, v = normal;
        const int pass_len[] = {10, 4, 2};
        int stage = 0;
        for (int pass = 0; pass < 3; pass++)
            for (int s = 1; s <= pass_len[pass]; s++, stage++) {
                FsB a = u, b = v;
                dr::body_butterfly(a, b, normal, stage == 0, (s & 1) != 0);
                const FsB w = FsB::make(std::min(a.lmin, b.lmin), std::max(a.lmax, b.lmax), std::min(a.vmin, b.vmin), std::max(a.vmax, b.vmax));
                u = w;
                v = w;
            }
        expect_within(dr::g_where, u, dr::TWO31, 21.0);
        (void)dr::reduce_small(u);
    }
    std::printf("all bounds hold\n");
    return 0;
}
