// Bandersnatch kernels (seam A), part 2: everything that reads the per-context constant block (Elligator 2, Tonelli-Shanks,
// the GLV endomorphism, point decoding).  Part 1 (curve-templated scalar multiplication, shared helpers): kernels_te.hip.h.
#pragma once
#include "kernels_te.hip.h"

namespace dr {

// Constants of the Elligator map and of Tonelli-Shanks, in Montgomery form (2^261: the host multiplies its 2^256 form by 32), computed once per context on the host
// (capi_core.hip: bsn_consts_init) instead of by every lane: the Montgomery-model coefficients derived from a = -5 and d
// (A_M = 2(a+d)/(a-d), B_M = 4/(a-d)) and c_pow[j] = (5^Q)^(2^j), 5 the non-residue, p - 1 = Q * 2^32.
struct BsnConsts {
    uint32_t mont_b[8], a_over_b[8], inv_b2[8];
    uint32_t c_pow[32][8];
    uint32_t glv_b[8], glv_c[8];         // endomorphism coefficients (bandersnatch.py:58-67), Montgomery form
    uint32_t z_q1h[8];                   // 5^((Q+1)/2): turns x^((Q+1)/2) into (5 x)^((Q+1)/2) (fr_sqrt_or_5x)
    // discrete logarithms in the 2-Sylow subgroup <c>, c = 5^Q of order 2^32, by 8-bit windows (fr_sqrt_core):
    uint32_t dl_mul[4][256][8];          // c^(-k 2^(8j))
    uint32_t dl_half[4][256][8];         // c^(-k 2^(8j) / 2)   (j = 0: even k only)
    uint32_t dl_hash_mul;                // perfect hash of the 256 elements of <c^(2^24)>: map[(low word * dl_hash_mul) >> 16] = exponent
    uint8_t dl_map[65536];
};
__device__ BsnConsts g_bsn_consts;
DR_DEV Fs bsn_const(const uint32_t (&w)[8]) { return unpack29(w); }

// Square roots in Fr, p - 1 = Q 2^32.  The reference's sqrt_mod_bls_scalar_cy (bandersnatch_te.pyx:421-477) is Tonelli-Shanks: with
// w = x^((Q-1)/2), R = w x, t = R w = x^Q  (R^2 = x t), it walks t down to 1 with ~250 squarings on average and 500 at worst, a
// different number in every lane.  t lies in the cyclic group <c> of order 2^32, c = 5^Q, so here its logarithm e (t = c^e) is read
// off in four 8-bit windows instead: t^(2^24) is one of the 256 elements of <c^(2^24)> — a perfect-hash table gives e mod 2^8 —,
// t c^(-e0) raised to 2^16 the next window, and so on: 48 squarings, 7 products, 4 table look-ups, the same in every lane.
// x is a square iff e is even, and then x = (R c^(-e/2))^2.  Which of the two roots comes out differs from Tonelli-Shanks in general;
// every caller fixes the sign itself (Elligator: sgn0; point decoding: the sign bit; dr_fr_sqrt: host code).
// OR_5X (Elligator 2 needs sqrt(g) when g is a square and sqrt(Z u^2 g) = u sqrt(Z g) otherwise, Z = 5 — the non-residue c is built
// on): when e is odd, (R, t) <- (R 5^((Q+1)/2), t c) are the same quantities for 5 g, whose logarithm e + 1 is even.  ONE
// exponentiation serves both cases.  Returns whether x was a square; root = sqrt(x), or sqrt(5 x) (OR_5X), else unspecified.
DR_DEV uint32_t fr_dlog_window(const Fs& v) {
    const Fr c = pack(v);
    return g_bsn_consts.dl_map[(c.l[0] * g_bsn_consts.dl_hash_mul) >> 16];
}
template <bool OR_5X>
DR_DEV bool fr_sqrt_core(const Fs& x, Fs& root) {
    root = x;
    if (is_zero(x)) return true;
    constexpr uint32_t QM1H[8] = {0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u, 0u};   // (Q-1)/2
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = QM1H[i];
    const Fs w = fr_pow_limbs(x, e);
    Fs R = mul(w, x);
    Fs t = mul(R, w);
    bool square = true;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        Fs v = t;
#pragma unroll 1
        for (int k = 0; k < 24 - 8 * j; k++) v = sqr(v);
        uint32_t ej = fr_dlog_window(v);
        if (j == 0 && (ej & 1u)) {
            square = false;
            if (!OR_5X) return false;
            t = mul(t, bsn_const(g_bsn_consts.c_pow[0]));
            R = mul(R, bsn_const(g_bsn_consts.z_q1h));
            ej = (ej + 1u) & 255u;
        }
        if (j < 3) t = mul(t, bsn_const(g_bsn_consts.dl_mul[j][ej]));
        R = mul(R, bsn_const(g_bsn_consts.dl_half[j][ej]));
    }
    root = R;
    return square;
}
DR_DEV bool fr_sqrt(const Fs& x, Fs& root) { return fr_sqrt_core<false>(x, root); }
DR_DEV bool fr_sqrt_or_5x(const Fs& x, Fs& root) { return fr_sqrt_core<true>(x, root); }

// Diagnostic (dr_fr_ops_selftest): the unsaturated field arithmetic of fr29.hip.h on its own, one lane per (a, b) pair of
// standard-form elements.  out[i] = twelve 32-byte standard-form records: a b, a^2, a + b, a - b, a^-1 (0 for 0),
// (a + b)(a - b) through two lazy operands, a * (curve coefficient -5) through the shifted addition chain of the group
// law (3p - 5a), a b + b a through the fused product, sqrt(a) (zero when a is not a square); flag[i] = 1 iff a is a square;
// then the helpers of the NTT / polynomial kernels: a b + a through canon29_small, a + 27 b and a - 28 b through reduce_small
// (lazy sums of up to 28 p) and canon29_small.
__global__ void k_fr_ops_selftest(const uint32_t* __restrict__ a_std, const uint32_t* __restrict__ b_std, uint32_t n, uint32_t* __restrict__ out,
                                  uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fs a = fs_from_std(load_fr_std(a_std + (size_t)i * 8)), b = fs_from_std(load_fr_std(b_std + (size_t)i * 8));
    uint32_t* o = out + (size_t)i * 96;
    store_fr_std(o, fs_to_std(mul(a, b)));
    store_fr_std(o + 8, fs_to_std(sqr(a)));
    store_fr_std(o + 16, fs_to_std(add(a, b)));
    store_fr_std(o + 24, fs_to_std(sub(a, b)));
    store_fr_std(o + 32, fs_to_std(inv(a)));
    store_fr_std(o + 40, fs_to_std(mul(add(a, b), sub(a, b))));
    store_fr_std(o + 48, fs_to_std(te_aA<CV_BANDERSNATCH>(a)));
    store_fr_std(o + 56, fs_to_std(mul2(a, b, b, a)));
    Fs r;
    const bool square = fr_sqrt(a, r);
    store_fr_std(o + 64, fs_to_std(square ? r : Fs::zero()));
    flag[i] = square ? 1u : 0u;
    // records 9..11 are written in the Montgomery form they are computed in (the host divides by 2^261)
    Fr w;
    canon29_small(add(mul(a, b), a), w.l);
    store_fr_std(o + 72, w);
    Fs up = a, down = a;
#pragma unroll 1
    for (int k = 0; k < 28; k++) {
        if (k < 27) up = add(up, b);
        down = sub(down, b);
        if ((k & 1) == 1) { up = carry(up); down = carry(down); }
    }
    canon29_small(reduce_small(up), w.l);
    store_fr_std(o + 80, w);
    canon29_small(reduce_small(down), w.l);
    store_fr_std(o + 88, w);
}

// Elligator 2 onto the Montgomery model up to the point (s, t) = (x B_M, y B_M); the inversion 1/(1 + Z u^2) is
// supplied by the caller so that the two maps of one input share ONE inversion (Montgomery's trick).
struct EllHalf { Fs u, tv1, den; };
DR_DEV EllHalf ell2_prepare(const Fs& u) {
    Fr five_std = Fr::zero(); five_std.l[0] = 5;
    const Fs five = fs_from_std(five_std);
    EllHalf h;
    h.u = u;
    h.tv1 = mul(five, sqr(u));                         // Z = 5
    if (is_zero(add(h.tv1, Fs::one()))) h.tv1 = Fs::zero();
    h.den = carry(add(h.tv1, Fs::one()));              // carried: it meets the partner's denominator in a product
    return h;
}
// second half: the birational map to the twisted Edwards model, inversion-free (extended coordinates)
DR_DEV TePoint ell2_finish(const EllHalf& h, const Fs& inv_den) {
    const Fs aob = bsn_const(g_bsn_consts.a_over_b), inv_b2 = bsn_const(g_bsn_consts.inv_b2), mont_b = bsn_const(g_bsn_consts.mont_b);
    Fs x1 = neg(mul(aob, inv_den));
    Fs gx1 = mul(add(mul(add(x1, aob), x1), inv_b2), x1);
    Fs y;
    const bool e2 = fr_sqrt_or_5x(gx1, y);             // sqrt(g(x1)), or sqrt(Z g(x1)) when g(x1) is not a square
    Fs x = x1;
    if (!e2) {
        x = sub(neg(x1), aob);
        y = is_zero(h.tv1) ? Fs::zero() : mul(h.u, y); // sqrt(Z u^2 g(x1)) = u sqrt(Z g(x1)); tv1 = 0 (u = 0, or Z u^2 = -1 mapped to 0): the root of 0
    }
    bool odd = (fs_to_std(y).l[0] & 1u) != 0;
    if (e2 != odd) y = neg(y);                          // e2 XOR e3 -> negate
    Fs s = mul(x, mont_b), t = mul(y, mont_b);
    // (s,t) -> (v,w) = (s/t, (s-1)/(s+1)); with Z = (s+1) t:  X = s (s+1), Y = (s-1) t; exceptional case -> (0,1)
    Fs sp1 = add(s, Fs::one());
    Fs Z = mul(sp1, t);
    if (is_zero(Z)) return te_identity();
    Fs X = mul(s, sp1), Y = mul(sub(s, Fs::one()), t);
    TePoint r;
    r.x = mul(X, Z); r.y = mul(Y, Z); r.z = sqr(Z); r.t = mul(X, Y);
    return r;
}

// Two lanes per input, one Elligator map each: the two square roots — the long part of the chain — run side by side;
// the partner's denominator and point cross over by shuffles (lanes 2i and 2i+1 sit in the same wave).
__global__ __launch_bounds__(64) void k_bsn_encode_to_curve(const uint32_t* __restrict__ us /* n*2*8 std */, uint32_t* __restrict__ out /* n*16 std */,
                                                            uint32_t n) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = gid >> 1;
    const uint32_t half = gid & 1u;
    const bool live = i < n;
    if (!live) i = n - 1;                                  // keep the pair (and the shuffles) converged
    EllHalf h = ell2_prepare(fs_from_std(load_fr_std(us + (size_t)i * 16 + 8 * half)));
    Fs other;
#pragma unroll
    for (int t = 0; t < L29; t++) other.l[t] = __shfl_xor(h.den.l[t], 1, 64);
    Fs both = inv(mul(h.den, other));                      // den = 1 + Z u^2 is never zero (tv1 = -1 was mapped to 0)
    TePoint q = ell2_finish(h, mul(both, other));
    TePoint p;
#pragma unroll
    for (int t = 0; t < L29; t++) {
        p.x.l[t] = __shfl_xor(q.x.l[t], 1, 64);
        p.y.l[t] = __shfl_xor(q.y.l[t], 1, 64);
        p.z.l[t] = __shfl_xor(q.z.l[t], 1, 64);
        p.t.l[t] = __shfl_xor(q.t.l[t], 1, 64);
    }
    TePoint r = te_add(q, p);
    r = te_dbl<false>(r);
    r = te_dbl<false>(r);
    if (live && half == 0) te_store_affine(out + (size_t)i * 16, r);
}

// ---- GLV on lane pairs (dot_ring/curve/glv.py:128-189, specs/bandersnatch.py:177-191) ----------------------------------
// k*P = k1*P + k2*psi(P) with |k1|, |k2| < 2^128 (split on the host, hostproto.hpp: glv_decompose).  The reference feeds the
// two half-length scalars to a joint 2-bit window kernel; here the two halves go to two adjacent LANES, each running the
// same signed 4-bit window core over 33 windows instead of 64, and one shuffle adds them up: the dependent chain — which
// is what a launch of a few thousand scalar multiplications costs — is half as long, the total work unchanged.
// psi(x, y) = (f h : g x y : h x y), f = c (1 - y^2), g = b (y^2 + b), h = y^2 - b, returned in extended coordinates.
DR_DEV TePoint bsn_endomorphism(const Fs& x, const Fs& y) {
    const Fs b = bsn_const(g_bsn_consts.glv_b), c = bsn_const(g_bsn_consts.glv_c);
    Fs y2 = sqr(y), xy = mul(x, y);
    Fs f = mul(c, sub(Fs::one(), y2)), g = mul(b, add(y2, b)), h = sub(y2, b);
    Fs X = mul(f, h), Y = mul(g, xy), Z = mul(h, xy);
    if (is_zero(Z)) return te_identity();            // x y = 0: the identity (or 2-/4-torsion, never a subgroup point)
    TePoint r;
    r.x = mul(X, Z); r.y = mul(Y, Z); r.z = sqr(Z); r.t = mul(X, Y);
    return r;
}
// the signed 4-bit window core over NW windows (NW*4 scalar bits, k given as ceil(NW/8) words), base in extended coordinates
template <int NW>
DR_DEV TePoint bsn_window_core(uint32_t* tab, int lane, const TePoint& P, const uint32_t (&k)[(NW + 7) / 8]) {
    constexpr int WORDS = (NW + 7) / 8;
    lds_store_point(tab, 0, lane, P);
    TePoint Q = te_dbl<true>(P);
    lds_store_point(tab, 1, lane, Q);
#pragma unroll 1
    for (int e = 2; e < BSN_TABLE; e++) {
        Q = te_add(Q, P);
        lds_store_point(tab, e, lane, Q);
    }
    uint32_t dig[WORDS];
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < WORDS; w++) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = ((k[w] >> (4 * j)) & 15u) + carry;
            carry = v >= 8u ? 1u : 0u;
            packed |= ((v + 8u) & 15u) << (4 * j);
        }
        dig[w] = packed;
    }
    TePoint acc = te_identity();
#pragma unroll 1
    for (int w = NW - 1; w >= 0; w--) {
#pragma unroll 1
        for (int j = 0; j < 3; j++) acc = te_dbl<false>(acc);
        acc = te_dbl<true>(acc);
        int d = (int)((dig[w >> 3] >> (4 * (w & 7))) & 15u) - 8;
        int mag = d < 0 ? -d : d;
        TePoint T = lds_load_point(tab, mag == 0 ? 0 : mag - 1, lane);
        T = te_cneg(T, d < 0);
        if (mag == 0) T = te_identity();
        acc = te_add(acc, T);
    }
    return acc;
}
// ---- GLV decomposition on the device (dot_ring/curve/glv.py:128-163; the host's glv_decompose of hostproto.hpp word for word):
// k = k1 + k2 * lambda (mod n), |k1|, |k2| < 2^128, from the lattice basis v1 = (a1, b1), v2 = (a2, -a1):
//   c1 = floor(k * G1 / 2^256), c2 = floor(k * G2 / 2^256)  (G1 = floor(2^256 a1 / n), G2 = floor(2^256 b1 / n)),
//   k1 = k - (c1 a1 + c2 a2), k2 = c2 a1 - c1 b1 — 256-bit integers in 32-bit words with 64-bit multiply-adds.
template <int NA, int NB>
DR_DEV void mul_words(const uint32_t (&a)[NA], const uint32_t (&b)[NB], uint32_t (&out)[NA + NB]) {
#pragma unroll
    for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; i++) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + c;
            out[i + j] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
        out[i + NB] = c;
    }
}
// signed 256-bit (d, borrow of the subtraction that produced it) -> 128-bit magnitude + sign; false when it does not fit
DR_DEV bool glv_finish(uint32_t (&d)[8], uint32_t borrow, uint32_t (&mag)[4], uint32_t& neg) {
    neg = borrow ? 1u : 0u;
    if (borrow) {
        uint32_t c = 1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t t = (uint64_t)(~d[i]) + c;
            d[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) mag[i] = d[i];
    return (d[4] | d[5] | d[6] | d[7]) == 0;
}
// k < n in, (|k1|, |k2|, signs) out; false only if a half does not fit 128 bits (cannot happen for k < n with this basis)
DR_DEV bool glv_split_dev(const uint32_t (&k)[8], uint32_t (&k1)[4], uint32_t (&k2)[4], uint32_t& neg1, uint32_t& neg2) {
    constexpr uint32_t A1[4] = {0x9789181fu, 0x4b02f94au, 0x4be6928eu, 0x555fe200u};
    constexpr uint32_t B1[4] = {0x23d61f44u, 0xf8e2591au, 0xe55e8f5du, 0x0814b3eeu};
    constexpr uint32_t A2[4] = {0x47ac3e88u, 0xf1c4b234u, 0xcabd1ebbu, 0x102967ddu};
    constexpr uint32_t G1[5] = {0x3f4747c1u, 0xdebac77au, 0x541cf632u, 0xf21df5b0u, 0x00000002u};   // floor(2^256 a1 / n)
    constexpr uint32_t G2[4] = {0x547768aau, 0x993b75e7u, 0xd8767bdeu, 0x4760f127u};                // floor(2^256 b1 / n)
    uint32_t t13[13], t12[12], c1[4], c2[4];
    mul_words<8, 5>(k, G1, t13);
    mul_words<8, 4>(k, G2, t12);
    bool ok = t13[12] == 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { c1[i] = t13[8 + i]; c2[i] = t12[8 + i]; }
    uint32_t p1[8], p2[8], s[8], d[8];
    mul_words<4, 4>(c1, A1, p1);
    mul_words<4, 4>(c2, A2, p2);
    uint32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = addc(p1[i], p2[i], cy);
    ok = ok && cy == 0;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = subb(k[i], s[i], bw);
    ok = glv_finish(d, bw, k1, neg1) && ok;
    mul_words<4, 4>(c2, A1, p1);
    mul_words<4, 4>(c1, B1, p2);
    bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = subb(p1[i], p2[i], bw);
    ok = glv_finish(d, bw, k2, neg2) && ok;
    return ok;
}

// one half of a GLV pair: the lane's base (P or psi(P), negated when its half-scalar is negative) times |k_half|
// split: per term 12 words — |k1| (4), |k2| (4), neg1, neg2, 2 pad; RAW: `split` holds the n raw 8-word scalars instead and
// every lane reduces its term's scalar mod n and decomposes it itself (both lanes of a pair: ~400 instructions against the
// ~300 000 of the window chain — and no host pass over device-resident scalars)
template <bool RAW = false>
DR_DEV TePoint bsn_glv_half(uint32_t* tab, int lane, const uint32_t* __restrict__ pts, const uint32_t* __restrict__ split, size_t term, bool second,
                            uint32_t* __restrict__ err = nullptr) {
    Fs px = fs_from_std(load_fr_std(pts + term * 16));
    Fs py = fs_from_std(load_fr_std(pts + term * 16 + 8));
    TePoint base;
    if (second) base = bsn_endomorphism(px, py);
    else { base.x = px; base.y = py; base.z = Fs::one(); base.t = mul(px, py); }
    uint32_t k[5];
    if constexpr (RAW) {
        uint32_t kk[8], k1[4], k2[4], n1, n2;
#pragma unroll
        for (int j = 0; j < 8; j++) kk[j] = split[term * 8 + j];
        reduce_mod_order<CV_BANDERSNATCH>(kk);
        if (!glv_split_dev(kk, k1, k2, n1, n2) && err) atomicOr(err, 1u);
        base = te_cneg(base, (second ? n2 : n1) != 0);
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = second ? k2[j] : k1[j];
    } else {
        const uint32_t* s = split + term * 12;
        base = te_cneg(base, s[8 + (second ? 1 : 0)] != 0);
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = s[(second ? 4 : 0) + j];
    }
    k[4] = 0;
    return bsn_window_core<33>(tab, lane, base, k);
}
// out[i] = k[i] * P[i] from the split scalars; lanes 2i, 2i+1 share one scalar multiplication
template <bool RAW = false>
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_scalar_mul_glv(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ split,
                                                                  uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ err = nullptr) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    uint32_t i = (blockIdx.x * BSN_BLOCK + lane) >> 1;
    const bool live = i < n;
    if (!live) i = n - 1;
    TePoint acc = bsn_glv_half<RAW>(tab, lane, pts, split, i, (lane & 1) != 0, err);
    TePoint o = te_shfl_down(acc, 1);
    acc = te_add(acc, o);
    if (live && !(lane & 1)) te_store_affine(out + (size_t)i * 16, acc);
}
// out[g] = sum_{j<m} k[g*m+j] * P[g*m+j], m <= 32: 2m lanes per group (mpad2 = 2m rounded up to a power of two)
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_msm_groups_glv(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ split,
                                                                  uint32_t* __restrict__ out, uint32_t groups, uint32_t m, uint32_t mpad2) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    const uint32_t per_block = BSN_BLOCK / mpad2;
    const uint32_t g = blockIdx.x * per_block + lane / mpad2;
    const uint32_t slot = lane % mpad2, j = slot >> 1;
    const bool live = g < groups && j < m;
    const size_t term = live ? (size_t)g * m + j : 0;
    TePoint r = bsn_glv_half(tab, lane, pts, split, term, (slot & 1) != 0);
    TePoint acc = live ? r : te_identity();
#pragma unroll 1
    for (uint32_t s = mpad2 >> 1; s > 0; s >>= 1) {
        TePoint o = te_shfl_down(acc, s);
        acc = te_add(acc, o);
    }
    if (g < groups && slot == 0) te_store_affine(out + (size_t)g * 16, acc);
}

// ---- point decoding for verifiers --------------------------------------------------------------------------------
// dec_point for a batch (dot_ring/curve/point.py:150-214, te_affine_point.py:297-316, vrf/codec.py:39-45,
// curve/curve.py:56-67): y = the 255 low bits (rejected when >= p), x^2 = (1 - y^2) / (a - d y^2), the sign bit picks
// the larger of (x, p - x); the point must be a non-identity member of the prime-order subgroup.  Subgroup test
// without an unreduced scalar: Q = 4P must not be the identity and [4^-1 mod n] Q must give back P (a torsion
// component of P is killed by the 4 and would not come back).  One lane per point, ok[i] = 1 when valid.
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_decode_points(const uint32_t* __restrict__ enc /* n*8 */, uint32_t* __restrict__ out_xy /* n*16 std */,
                                                                 uint32_t* __restrict__ ok, uint32_t n) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    uint32_t i = (blockIdx.x * BSN_BLOCK + lane) >> 1;           // two lanes per point: the GLV halves of the subgroup check
    const bool half = (lane & 1) != 0;
    const bool live = i < n;
    if (!live) i = n - 1;
    Fr ys = load_fr_std(enc + (size_t)i * 8);
    const bool sign = (ys.l[7] >> 31) != 0;
    ys.l[7] &= 0x7fffffffu;
    bool valid = true;
    {   // y < p
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) (void)subb(ys.l[j], FrParams::P[j], borrow);
        valid = borrow != 0;
    }
    if (!valid) ys = Fr::zero();
    const Fs one = Fs::one();
    Fs y = fs_from_std(ys);
    Fs y2 = sqr(y);
    Fs den = sub(te_mul_a(one), mul(te_d_mont(), y2));
    if (is_zero(den)) { valid = false; den = one; }
    Fs x2 = mul(sub(one, y2), inv(den));
    Fs x;
    if (!fr_sqrt(x2, x)) { valid = false; x = one; }
    {
        Fr xs = fs_to_std(x), nxs = fs_to_std(neg(x));
        bool x_larger = false;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            if (xs.l[j] != nxs.l[j]) { x_larger = xs.l[j] > nxs.l[j]; break; }
        }
        if (x_larger != sign) x = neg(x);          // sign set: the larger root, otherwise the smaller
    }
    TePoint P;
    P.x = x; P.y = y; P.z = one; P.t = mul(x, y);
    TePoint Q = te_dbl<false>(te_dbl<false>(P));
    if (is_zero(Q.x)) { valid = false; Q = P; }    // 4P = O (x = 0 also covers the order-2 point (0,-1), which 4 kills anyway)
    Fs zi = inv(is_zero(Q.z) ? one : Q.z);
    Fs qx = mul(Q.x, zi), qy = mul(Q.y, zi);
    // [4^-1 mod n] Q by GLV on the lane pair (Q = 4P lies in the prime-order subgroup, where psi acts as lambda):
    // 4^-1 = k1 + k2 lambda with k1 > 0 > k2, both below 2^127
    TePoint base;
    uint32_t kh[5];
    if (!half) {
        base.x = qx; base.y = qy; base.z = one; base.t = mul(qx, qy);
        kh[0] = 0xed8e8490u; kh[1] = 0x84857086u; kh[2] = 0xddb6c35fu; kh[3] = 0x2581605du;
    } else {
        base = te_cneg(bsn_endomorphism(qx, qy), true);
        kh[0] = 0x0e93904eu; kh[1] = 0xccca6304u; kh[2] = 0x928eeeb6u; kh[3] = 0x535ab504u;
    }
    kh[4] = 0;
    TePoint R = bsn_window_core<33>(tab, lane, base, kh);
    R = te_add(R, te_shfl_down(R, 1));
    if (!equal(R.x, mul(x, R.z)) || !equal(R.y, mul(y, R.z))) valid = false;
    if (live && !half) {
        store_fr_std(out_xy + (size_t)i * 16, fs_to_std(x));
        store_fr_std(out_xy + (size_t)i * 16 + 8, fs_to_std(y));
        ok[i] = valid ? 1u : 0u;
    }
}

// The same decoding for a curve without an endomorphism (JubJub, cofactor 8): one lane per point, Q = hP by log2(h)
// doublings, then [h^-1 mod n] Q through the plain 64-window core.  TAI = true is the device half of try-and-increment
// hash-to-curve (dot_ring/curve/point.py:252-296): the candidate only has to decompress; the output is hP and ok says
// "decompressed and hP is not the identity".
template <int CV, bool TAI>
__global__ __launch_bounds__(BSN_BLOCK) void k_te_decode_points(const uint32_t* __restrict__ enc /* n*8 */, uint32_t* __restrict__ out_xy /* n*16 std */,
                                                                uint32_t* __restrict__ ok, uint32_t n) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    uint32_t i = blockIdx.x * BSN_BLOCK + lane;
    const bool live = i < n;
    if (!live) i = n - 1;
    Fr ys = load_fr_std(enc + (size_t)i * 8);
    const bool sign = (ys.l[7] >> 31) != 0;
    ys.l[7] &= 0x7fffffffu;
    bool valid = true;
    {   // y < p
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) (void)subb(ys.l[j], FrParams::P[j], borrow);
        valid = borrow != 0;
    }
    if (!valid) ys = Fr::zero();
    const Fs one = Fs::one();
    Fs y = fs_from_std(ys);
    Fs y2 = sqr(y);
    Fs den = sub(te_mul_a<CV>(one), mul(te_d_mont<CV>(), y2));
    if (is_zero(den)) { valid = false; den = one; }
    Fs x2 = mul(sub(one, y2), inv(den));
    Fs x;
    if (!fr_sqrt(x2, x)) { valid = false; x = one; }
    {
        Fr xs = fs_to_std(x), nxs = fs_to_std(neg(x));
        bool x_larger = false;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            if (xs.l[j] != nxs.l[j]) { x_larger = xs.l[j] > nxs.l[j]; break; }
        }
        if (x_larger != sign) x = neg(x);
    }
    TePoint P;
    P.x = x; P.y = y; P.z = one; P.t = mul(x, y);
    constexpr int LOG2_H = CV == CV_JUBJUB ? 3 : 2;
    TePoint Q = P;
#pragma unroll 1
    for (int j = 0; j < LOG2_H; j++) Q = te_dbl<true, CV>(Q);
    if (is_zero(Q.x)) { valid = false; Q = P; }    // hP = O (x = 0 also covers the order-2 point (0,-1), which h kills anyway)
    Fs zi = inv(is_zero(Q.z) ? one : Q.z);
    Fs qx = mul(Q.x, zi), qy = mul(Q.y, zi);
    if (TAI) {
        if (live) {
            store_fr_std(out_xy + (size_t)i * 16, fs_to_std(qx));
            store_fr_std(out_xy + (size_t)i * 16 + 8, fs_to_std(qy));
            ok[i] = valid ? 1u : 0u;
        }
        return;
    }
    // h^-1 mod n
    constexpr uint32_t HINV_B[8] = {0xde592de9u, 0x17bdc507u, 0x5712c355u, 0xbfaba540u, 0x81ce5880u, 0x899ad881u, 0x97cd877du, 0x15bc8f5fu};
    constexpr uint32_t HINV_J[8] = {0xdadee597u, 0x5a12e1cbu, 0x79990210u, 0x14cd0412u, 0x20268760u, 0x20cce760u, 0x4ca675f5u, 0x01cfb69du};
    uint32_t k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = CV == CV_JUBJUB ? HINV_J[j] : HINV_B[j];
    TePoint R = bsn_scalar_mul_core<CV>(tab, lane, qx, qy, k);
    if (!equal(R.x, mul(x, R.z)) || !equal(R.y, mul(y, R.z))) valid = false;
    if (live) {
        store_fr_std(out_xy + (size_t)i * 16, fs_to_std(x));
        store_fr_std(out_xy + (size_t)i * 16 + 8, fs_to_std(y));
        ok[i] = valid ? 1u : 0u;
    }
}

}  // namespace dr
