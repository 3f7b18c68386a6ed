/*
 * ORACLE — test infrastructure only.  CPU restatement (plain C, gcc) of the arithmetic on
 * dot-ring's Ring-VRF hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (dot_ring_amd) never does.
 *
 * What is restated, with the reference lines each block follows:
 *   Fr-255 Montgomery field        dot_ring/curve/native_field/bls12_381_scalar.c:43-264  (via mont_tmpl.h)
 *   Tonelli-Shanks sqrt in Fr      dot_ring/curve/native_field/bandersnatch_te.pyx:421-477
 *   extended twisted-Edwards       dot_ring/curve/native_field/bandersnatch_te.pyx:127-174 (dbl/add-2008-hwcd)
 *   joint 2-bit window k1P1+k2P2   dot_ring/curve/native_field/bandersnatch_te.pyx:480-554
 *   GLV split + endomorphism       dot_ring/curve/glv.py:128-189, dot_ring/curve/specs/bandersnatch.py:177-191
 *   signed-digit Pippenger (TE)    dot_ring/curve/native_field/bandersnatch_te.pyx:257-418,
 *                                  window rule dot_ring/curve/specs/bandersnatch.py:23-36, centring :270-284
 *   radix-2 DIT NTT                dot_ring/ring_proof/polynomial/ntt.pyx:116-163,
 *                                  dot_ring/curve/native_field/bls12_381_scalar.c:333-356
 *   G1 MSM (KZG commit)            call site dot_ring/ring_proof/pcs/kzg.py:152-175.  The arithmetic lives in the
 *                                  third-party blst fork (github.com/Chainscore/blst, branch
 *                                  fix/python-as-memory-refcount, no commit pin; not in /root/reference).  Restated
 *                                  here from the published algorithm: Jacobian coordinates over Fp-381 (a=0, b=4),
 *                                  signed-digit bucket Pippenger.  Pinned by the reference's ring KATs
 *                                  (tests/golden: ring_pks_com / ring_proof commitments, safrole ring root).
 *   zcash G1 (de)compression       call sites dot_ring/ring_proof/pcs/kzg.py:129-144 (blst compress/serialize).
 *
 * Byte conventions at this C boundary: every field element is LITTLE-endian, standard (non-Montgomery) form;
 * TE affine point = x(32)||y(32); G1 affine point = x(48)||y(48), infinity = all-zero bytes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "consts.h"

#define MT_NL 4
#define MT_(n) fr_##n
#include "mont_tmpl.h"
#undef MT_NL
#undef MT_

#define MT_NL 6
#define MT_(n) fp_##n
#include "mont_tmpl.h"
#undef MT_NL
#undef MT_

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ Fr helpers */

typedef struct { uint64_t v[4]; } fr_t;

static fr_t FR_ONE, FR_A, FR_D, FR_GLV_B, FR_GLV_C, FR_TS_C;
static uint64_t FP_ONE[6], FP_B4[6], FP_B12[6];
static int g_init_done = 0;

static void orc_init(void) {
    if (g_init_done) return;
    fr_one_mont(FR_ONE.v);
    fr_to_mont(FR_A.v, bsn_A);
    fr_to_mont(FR_D.v, bsn_D);
    fr_to_mont(FR_GLV_B.v, bsn_GLV_B);
    fr_to_mont(FR_GLV_C.v, bsn_GLV_C);
    fr_to_mont(FR_TS_C.v, fr_TS_C);
    fp_one_mont(FP_ONE);
    uint64_t four[6] = {4, 0, 0, 0, 0, 0}, twelve[6] = {12, 0, 0, 0, 0, 0};
    fp_to_mont(FP_B4, four);
    fp_to_mont(FP_B12, twelve);
    g_init_done = 1;
}

EXPORT void orc_fr_add(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    uint64_t x[4], y[4];
    fr_from_le(x, a); fr_from_le(y, b);
    fr_add(x, x, y);
    fr_to_le(o, x);
}
EXPORT void orc_fr_sub(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    uint64_t x[4], y[4];
    fr_from_le(x, a); fr_from_le(y, b);
    fr_sub(x, x, y);
    fr_to_le(o, x);
}
EXPORT void orc_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    uint64_t x[4], y[4];
    fr_load(x, a); fr_load(y, b);
    fr_mul(x, x, y);
    fr_store(o, x);
}
/* raw Montgomery product a*b*R^-1 on plain limbs — lets tests compare with the reference's
 * bls_scalar_mul_mont (oracle/_ref) limb for limb */
EXPORT void orc_fr_mul_mont_raw(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    uint64_t x[4], y[4];
    fr_from_le(x, a); fr_from_le(y, b);
    fr_mul(x, x, y);
    fr_to_le(o, x);
}
EXPORT void orc_fr_inv(const uint8_t *a, uint8_t *o) {
    uint64_t x[4];
    fr_load(x, a);
    fr_inv(x, x);
    fr_store(o, x);
}
EXPORT void orc_fr_pow(const uint8_t *a, const uint8_t *e, uint8_t *o) {
    uint64_t x[4], ee[4];
    fr_load(x, a); fr_from_le(ee, e);
    fr_pow(x, x, ee, 4);
    fr_store(o, x);
}

/* bandersnatch_te.pyx:421 — Tonelli-Shanks with p-1 = Q*2^32, non-residue 5. Returns 0 on non-square. */
static int fr_sqrt_mont(uint64_t *out, const uint64_t *xm) {
    orc_init();
    if (fr_is_zero(xm)) { fr_zero(out); return 1; }
    uint64_t R[4], t[4], c[4], b[4], tmp[4];
    fr_pow(t, xm, fr_TS_Q, 4);
    fr_pow(R, xm, fr_TS_Q1H, 4);
    fr_copy(c, FR_TS_C.v);
    int M = 32;
    for (;;) {
        if (fr_eq(t, FR_ONE.v)) { fr_copy(out, R); return 1; }
        int i = 1;
        fr_sqr(tmp, t);
        while (!fr_eq(tmp, FR_ONE.v)) {
            fr_sqr(tmp, tmp);
            i++;
            if (i >= M) return 0;
        }
        fr_copy(b, c);
        for (int j = 0; j < M - i - 1; j++) fr_sqr(b, b);
        M = i;
        fr_sqr(c, b);
        fr_mul(t, t, c);
        fr_mul(R, R, b);
    }
}
EXPORT int orc_fr_sqrt(const uint8_t *a, uint8_t *o) {
    uint64_t x[4];
    fr_load(x, a);
    if (!fr_sqrt_mont(x, x)) return 0;
    fr_store(o, x);
    return 1;
}

/* ------------------------------------------------------------------ Bandersnatch, extended TE */

typedef struct { uint64_t x[4], y[4], z[4], t[4]; } te_t;

static void te_identity(te_t *o) {          /* bandersnatch_te.pyx:104 — (0,1,1,0) */
    orc_init();
    fr_zero(o->x); fr_copy(o->y, FR_ONE.v); fr_copy(o->z, FR_ONE.v); fr_zero(o->t);
}
static void te_neg(te_t *o, const te_t *p) { /* :118 */
    fr_neg(o->x, p->x); fr_copy(o->y, p->y); fr_copy(o->z, p->z); fr_neg(o->t, p->t);
}
/* :127 dbl-2008-hwcd: A=X^2 B=Y^2 C=2Z^2 D=aA E=(X+Y)^2-A-B G=D+B F=G-C H=D-B */
static void te_dbl(te_t *o, const te_t *p) {
    uint64_t A[4], B[4], C[4], D[4], E[4], F[4], G[4], H[4], s[4];
    fr_sqr(A, p->x); fr_sqr(B, p->y);
    fr_sqr(C, p->z); fr_add(C, C, C);
    fr_mul(D, FR_A.v, A);
    fr_add(s, p->x, p->y); fr_sqr(E, s); fr_sub(E, E, A); fr_sub(E, E, B);
    fr_add(G, D, B); fr_sub(F, G, C); fr_sub(H, D, B);
    fr_mul(o->x, E, F); fr_mul(o->y, G, H); fr_mul(o->t, E, H); fr_mul(o->z, F, G);
}
/* :148 add-2008-hwcd (unified): A=X1X2 B=Y1Y2 C=d T1T2 D=Z1Z2 E=(X1+Y1)(X2+Y2)-A-B F=D-C G=D+C H=B-aA */
static void te_add(te_t *o, const te_t *p, const te_t *q) {
    uint64_t A[4], B[4], C[4], D[4], E[4], F[4], G[4], H[4], s1[4], s2[4];
    fr_mul(A, p->x, q->x); fr_mul(B, p->y, q->y);
    fr_mul(C, FR_D.v, p->t); fr_mul(C, C, q->t);
    fr_mul(D, p->z, q->z);
    fr_add(s1, p->x, p->y); fr_add(s2, q->x, q->y);
    fr_mul(E, s1, s2); fr_sub(E, E, A); fr_sub(E, E, B);
    fr_sub(F, D, C); fr_add(G, D, C);
    fr_mul(s1, FR_A.v, A); fr_sub(H, B, s1);
    fr_mul(o->x, E, F); fr_mul(o->y, G, H); fr_mul(o->t, E, H); fr_mul(o->z, F, G);
}
static void te_from_affine_bytes(te_t *o, const uint8_t *xy) {
    orc_init();
    fr_load(o->x, xy); fr_load(o->y, xy + 32);
    fr_copy(o->z, FR_ONE.v);
    fr_mul(o->t, o->x, o->y);
}
/* bandersnatch_te.pyx:91,244 — z==0 maps to (0,1) */
static void te_to_affine_bytes(uint8_t *xy, const te_t *p) {
    uint64_t zi[4], ax[4], ay[4];
    if (fr_is_zero(p->z)) {
        memset(xy, 0, 64); xy[32] = 1; return;
    }
    fr_inv(zi, p->z);
    fr_mul(ax, p->x, zi); fr_mul(ay, p->y, zi);
    fr_store(xy, ax); fr_store(xy + 32, ay);
}

static inline unsigned bits_at(const uint64_t *k, int nl, int pos, int w) {
    int li = pos >> 6, sh = pos & 63;
    if (li >= nl) return 0;
    uint64_t v = k[li] >> sh;
    if (sh + w > 64 && li + 1 < nl) v |= k[li + 1] << (64 - sh);
    return (unsigned)(v & ((1ULL << w) - 1));
}
static inline int bit_length(const uint64_t *k, int nl) {
    for (int i = nl - 1; i >= 0; i--)
        if (k[i]) return i * 64 + 64 - __builtin_clzll(k[i]);
    return 0;
}

EXPORT void orc_te_add(const uint8_t *p, const uint8_t *q, uint8_t *out) {
    te_t a, b, r;
    te_from_affine_bytes(&a, p); te_from_affine_bytes(&b, q);
    te_add(&r, &a, &b);
    te_to_affine_bytes(out, &r);
}

/* plain MSB-first double-and-add: the simplest statement of k*P (ground truth for everything else) */
EXPORT void orc_te_mul_naive(const uint8_t *p, const uint8_t *k, uint8_t *out) {
    te_t P, R;
    uint64_t kk[4];
    te_from_affine_bytes(&P, p);
    fr_from_le(kk, k);
    te_identity(&R);
    for (int i = bit_length(kk, 4) - 1; i >= 0; i--) {
        te_dbl(&R, &R);
        if ((kk[i >> 6] >> (i & 63)) & 1) te_add(&R, &R, &P);
    }
    te_to_affine_bytes(out, &R);
}

/* bandersnatch_te.pyx:480 — k1*P1 + k2*P2, joint 2-bit windows over a 4x4 table */
static void te_mul2_w2(te_t *R, const uint64_t *k1, const uint64_t *k2, const te_t *P1, const te_t *P2) {
    te_t m1[4], m2[4], tab[4][4];
    te_identity(&m1[0]); m1[1] = *P1; te_add(&m1[2], &m1[1], P1); te_add(&m1[3], &m1[2], P1);
    te_identity(&m2[0]); m2[1] = *P2; te_add(&m2[2], &m2[1], P2); te_add(&m2[3], &m2[2], P2);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (i == 0) tab[i][j] = m2[j];
            else if (j == 0) tab[i][j] = m1[i];
            else te_add(&tab[i][j], &m1[i], &m2[j]);
        }
    int b1 = bit_length(k1, 4), b2 = bit_length(k2, 4);
    int mb = b1 > b2 ? b1 : b2;
    if (mb == 0) mb = 1;
    int nw = (mb + 1) >> 1;
    te_identity(R);
    for (int i = nw - 1; i >= 0; i--) {
        te_dbl(R, R); te_dbl(R, R);
        unsigned w1 = bits_at(k1, 4, 2 * i, 2), w2 = bits_at(k2, 4, 2 * i, 2);
        if (w1 | w2) te_add(R, R, &tab[w1][w2]);
    }
}

/* ---- small signed multi-limb helpers for the GLV split (glv.py:128-160) */
static void mul_limbs(uint64_t *o, const uint64_t *a, int na, const uint64_t *b, int nb) {
    memset(o, 0, (na + nb) * 8);
    for (int i = 0; i < na; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < nb; j++) {
            c += (unsigned __int128)a[i] * b[j] + o[i + j];
            o[i + j] = (uint64_t)c;
            c >>= 64;
        }
        o[i + nb] = (uint64_t)c;
    }
}
/* q = round(num / n) for a 6-limb num and the 4-limb group order; bitwise long division */
static void div_round_n(uint64_t *q, const uint64_t *num6) {
    uint64_t num[7], rem[5] = {0, 0, 0, 0, 0};
    memcpy(num, num6, 48); num[6] = 0;
    /* + n/2 for rounding */
    unsigned __int128 c = 0;
    for (int i = 0; i < 7; i++) {
        c += (unsigned __int128)num[i] + (i < 4 ? bsn_NHALF[i] : 0);
        num[i] = (uint64_t)c; c >>= 64;
    }
    memset(q, 0, 3 * 8);
    for (int bit = 7 * 64 - 1; bit >= 0; bit--) {
        /* rem = rem*2 + bit */
        for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] = (rem[0] << 1) | ((num[bit >> 6] >> (bit & 63)) & 1);
        int ge = rem[4] != 0;
        if (!ge) {
            ge = 1;
            for (int i = 3; i >= 0; i--) {
                if (rem[i] > bsn_N[i]) break;
                if (rem[i] < bsn_N[i]) { ge = 0; break; }
            }
        }
        if (ge) {
            uint64_t br = 0;
            for (int i = 0; i < 5; i++) {
                unsigned __int128 d = (unsigned __int128)rem[i] - (i < 4 ? bsn_N[i] : 0) - br;
                rem[i] = (uint64_t)d; br = (uint64_t)(d >> 127);
            }
            if (bit < 192) q[bit >> 6] |= 1ULL << (bit & 63);
        }
    }
}
/* signed 320-bit accumulator (magnitude + sign) */
typedef struct { uint64_t m[5]; int neg; } sbig;
static void sbig_add(sbig *acc, const uint64_t *mag5, int neg) {
    if (neg == acc->neg) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 5; i++) { c += (unsigned __int128)acc->m[i] + mag5[i]; acc->m[i] = (uint64_t)c; c >>= 64; }
        return;
    }
    int ge = 1;
    for (int i = 4; i >= 0; i--) { if (acc->m[i] > mag5[i]) break; if (acc->m[i] < mag5[i]) { ge = 0; break; } }
    const uint64_t *big = ge ? acc->m : mag5, *small = ge ? mag5 : acc->m;
    uint64_t r[5], br = 0;
    for (int i = 0; i < 5; i++) { unsigned __int128 d = (unsigned __int128)big[i] - small[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 127); }
    memcpy(acc->m, r, 40);
    if (!ge) acc->neg = neg;
}
static void sbig_addmul(sbig *acc, const uint64_t *a3, int aneg, const uint64_t *b2, int bneg) {
    uint64_t prod[5];
    mul_limbs(prod, a3, 3, b2, 2);
    sbig_add(acc, prod, aneg ^ bneg);
}
/* k (in [0,n)) -> (k1,k2) with k = k1 + k2*lambda mod n, |k1|,|k2| ~ 2^127 */
static void glv_decompose(uint64_t *k1, int *k1neg, uint64_t *k2, int *k2neg, const uint64_t *k) {
    uint64_t num[6], b1[3], b2[3];
    /* b1 = round(k*v2_1/det), b2 = round(-k*v1_1/det) — sign bookkeeping on magnitudes */
    mul_limbs(num, k, 4, bsn_GLV_V21, 2); div_round_n(b1, num);
    int b1neg = bsn_GLV_V21_NEG ^ bsn_GLV_DET_NEG;
    mul_limbs(num, k, 4, bsn_GLV_V11, 2); div_round_n(b2, num);
    int b2neg = 1 ^ bsn_GLV_V11_NEG ^ bsn_GLV_DET_NEG;
    /* vx = b1*v1_0 + b2*v2_0 ; vy = b1*v1_1 + b2*v2_1 ; k1 = k - vx ; k2 = -vy */
    sbig vx = {{0}, 0}, vy = {{0}, 0};
    sbig_addmul(&vx, b1, b1neg, bsn_GLV_V10, bsn_GLV_V10_NEG);
    sbig_addmul(&vx, b2, b2neg, bsn_GLV_V20, bsn_GLV_V20_NEG);
    sbig_addmul(&vy, b1, b1neg, bsn_GLV_V11, bsn_GLV_V11_NEG);
    sbig_addmul(&vy, b2, b2neg, bsn_GLV_V21, bsn_GLV_V21_NEG);
    sbig a = vx;
    a.neg ^= 1;
    uint64_t kmag[5] = {k[0], k[1], k[2], k[3], 0};
    sbig_add(&a, kmag, 0);
    memcpy(k1, a.m, 32); *k1neg = a.neg && !fr_is_zero(k1);
    memcpy(k2, vy.m, 32); *k2neg = (vy.neg ^ 1) && !fr_is_zero(k2);
}
/* glv.py:165 — phi(x,y) = (f*h, g*x*y, h*x*y) with f=c(1-y^2) g=b(y^2+b) h=y^2-b; kept projective */
static void te_endo(te_t *o, const te_t *paff) {
    uint64_t y2[4], xy[4], f[4], g[4], h[4], t[4];
    fr_sqr(y2, paff->y); fr_mul(xy, paff->x, paff->y);
    fr_sub(t, FR_ONE.v, y2); fr_mul(f, FR_GLV_C.v, t);
    fr_add(t, y2, FR_GLV_B.v); fr_mul(g, FR_GLV_B.v, t);
    fr_sub(h, y2, FR_GLV_B.v);
    /* projective (X,Y,Z) = (f*h, g*xy, h*xy); extended needs T = X*Y/Z: scale to (X*Z, Y*Z, Z^2, X*Y) */
    uint64_t X[4], Y[4], Z[4];
    fr_mul(X, f, h); fr_mul(Y, g, xy); fr_mul(Z, h, xy);
    fr_mul(o->x, X, Z); fr_mul(o->y, Y, Z); fr_sqr(o->z, Z); fr_mul(o->t, X, Y);
}
/* bandersnatch.py:177 — GLV scalar multiplication: k*P = k1*P + k2*phi(P) through the w2 kernel */
static void te_mul_glv(te_t *R, const te_t *Paff, const uint64_t *k) {
    uint64_t k1[4], k2[4];
    int n1, n2;
    te_t P1 = *Paff, P2;
    glv_decompose(k1, &n1, k2, &n2, k);
    te_endo(&P2, Paff);
    if (n1) te_neg(&P1, &P1);
    if (n2) te_neg(&P2, &P2);
    te_mul2_w2(R, k1, k2, &P1, &P2);
}
EXPORT void orc_te_mul_glv(const uint8_t *p, const uint8_t *k, uint8_t *out) {
    te_t P, R;
    uint64_t kk[4];
    te_from_affine_bytes(&P, p);
    fr_from_le(kk, k);
    te_mul_glv(&R, &P, kk);
    te_to_affine_bytes(out, &R);
}
/* batch of independent variable-base scalar multiplications (config 2 workload; CPU baseline leg) */
EXPORT void orc_te_mul_batch(const uint8_t *pts, const uint8_t *ks, size_t n, uint8_t *out, int use_glv) {
    for (size_t i = 0; i < n; i++) {
        if (use_glv) orc_te_mul_glv(pts + 64 * i, ks + 32 * i, out + 64 * i);
        else orc_te_mul_naive(pts + 64 * i, ks + 32 * i, out + 64 * i);
    }
}
EXPORT void orc_te_mul2_w2(const uint8_t *p1, const uint8_t *k1, const uint8_t *p2, const uint8_t *k2, uint8_t *out) {
    te_t P1, P2, R;
    uint64_t a[4], b[4];
    te_from_affine_bytes(&P1, p1); te_from_affine_bytes(&P2, p2);
    fr_from_le(a, k1); fr_from_le(b, k2);
    te_mul2_w2(&R, a, b, &P1, &P2);
    te_to_affine_bytes(out, &R);
}

/* bandersnatch.py:23 */
EXPORT int orc_te_pippenger_window(size_t n) {
    if (n < 8) return 2;
    if (n < 96) return 3;
    if (n < 192) return 4;
    if (n < 384) return 5;
    if (n < 768) return 6;
    if (n < 1024) return 7;
    return 8;
}
/* bandersnatch_te.pyx:257 — signed-digit Pippenger; scalars arrive in [0,n) and are centred to
 * (-n/2, n/2] as bandersnatch.py:270-284 does before the call. window_bits<=0 -> the reference rule. */
EXPORT void orc_te_msm(const uint8_t *pts, const uint8_t *ks, size_t n, int window_bits, uint8_t *out) {
    orc_init();
    te_t result;
    te_identity(&result);
    if (n == 0) { te_to_affine_bytes(out, &result); return; }
    if (window_bits <= 0) window_bits = orc_te_pippenger_window(n);
    int bucket_count = 1 << window_bits, half = bucket_count >> 1;
    te_t *P = malloc(n * sizeof(te_t)), *NP = malloc(n * sizeof(te_t));
    uint64_t *K = malloc(n * 32);
    int max_bits = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4];
        te_t base;
        fr_from_le(k, ks + 32 * i);
        te_from_affine_bytes(&base, pts + 64 * i);
        int neg = 0;
        /* k > n/2  ->  k - n (negative): magnitude n-k, negated point */
        int gt = 0;
        for (int j = 3; j >= 0; j--) { if (k[j] > bsn_NHALF[j]) { gt = 1; break; } if (k[j] < bsn_NHALF[j]) break; }
        if (gt) { fr_raw_sub(k, bsn_N, k); neg = 1; }
        memcpy(K + 4 * i, k, 32);
        int b = bit_length(k, 4);
        if (b > max_bits) max_bits = b;
        if (neg) { te_neg(&P[i], &base); NP[i] = base; } else { P[i] = base; te_neg(&NP[i], &base); }
    }
    if (max_bits == 0) { te_to_affine_bytes(out, &result); goto done; }
    {
        int base_windows = (max_bits + window_bits - 1) / window_bits, nw = base_windows + 1;
        int16_t *dig = malloc(n * nw * sizeof(int16_t));
        te_t *buckets = malloc((half + 1) * sizeof(te_t));
        int highest = -1;
        for (size_t i = 0; i < n; i++) {
            int carry = 0;
            for (int w = 0; w < base_windows; w++) {
                int d = (int)bits_at(K + 4 * i, 4, w * window_bits, window_bits) + carry;
                if (d >= half) { d -= bucket_count; carry = 1; } else carry = 0;
                dig[i * nw + w] = (int16_t)d;
                if (d != 0 && w > highest) highest = w;
            }
            dig[i * nw + base_windows] = (int16_t)carry;
            if (carry && base_windows > highest) highest = base_windows;
        }
        for (int w = highest; w >= 0; w--) {
            if (w != highest) for (int j = 0; j < window_bits; j++) te_dbl(&result, &result);
            for (int b = 1; b <= half; b++) te_identity(&buckets[b]);
            for (size_t i = 0; i < n; i++) {
                int d = dig[i * nw + w];
                if (d > 0) te_add(&buckets[d], &buckets[d], &P[i]);
                else if (d < 0) te_add(&buckets[-d], &buckets[-d], &NP[i]);
            }
            te_t running;
            te_identity(&running);
            for (int b = half; b > 0; b--) {
                te_add(&running, &running, &buckets[b]);
                te_add(&result, &result, &running);
            }
        }
        free(dig); free(buckets);
        te_to_affine_bytes(out, &result);
    }
done:
    free(P); free(NP); free(K);
}

/* ------------------------------------------------------------------ NTT over Fr */

/* ntt.pyx:141-160 + bls12_381_scalar.c:333: bit-reverse gather, log2(n) DIT rounds with per-stage
 * twiddles w_m^j (fft.py:30-55), optional output scaling (transform_scaled). In place, LE standard form. */
EXPORT int orc_ntt(uint8_t *data, size_t n, const uint8_t *omega, const uint8_t *scale) {
    orc_init();
    if (n < 2 || (n & (n - 1))) return -1;
    int lg = __builtin_ctzll(n);
    fr_t *a = malloc(n * sizeof(fr_t)), *tw = malloc((n / 2) * sizeof(fr_t));
    uint64_t w[4], sc[4];
    fr_load(w, omega);
    for (size_t i = 0; i < n; i++) {
        size_t r = 0;
        for (int b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b);
        fr_load(a[i].v, data + 32 * r);
    }
    for (size_t m = 2; m <= n; m <<= 1) {
        size_t half = m >> 1, stride = n / m;
        /* w_step = omega^stride */
        uint64_t step[4], e[4] = {stride, 0, 0, 0};
        fr_pow(step, w, e, 1);
        fr_copy(tw[0].v, FR_ONE.v);
        for (size_t j = 1; j < half; j++) fr_mul(tw[j].v, tw[j - 1].v, step);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                uint64_t t[4];
                fr_mul(t, tw[j].v, a[k + j + half].v);
                fr_sub(a[k + j + half].v, a[k + j].v, t);
                fr_add(a[k + j].v, a[k + j].v, t);
            }
    }
    if (scale) fr_load(sc, scale);
    for (size_t i = 0; i < n; i++) {
        if (scale) fr_mul(a[i].v, a[i].v, sc);
        fr_store(data + 32 * i, a[i].v);
    }
    free(a); free(tw);
    return 0;
}

/* ------------------------------------------------------------------ BLS12-381 G1 (Jacobian, a=0, b=4) */

typedef struct { uint64_t x[6], y[6], z[6]; } g1_t;   /* z==0 : infinity */
typedef struct { uint64_t x[6], y[6]; int inf; } g1a_t;

static void g1_set_inf(g1_t *o) { memset(o, 0, sizeof *o); }
static int g1_is_inf(const g1_t *p) { return fp_is_zero(p->z); }

/* dbl-2009-l */
static void g1_dbl(g1_t *o, const g1_t *p) {
    if (g1_is_inf(p)) { g1_set_inf(o); return; }
    uint64_t A[6], B[6], C[6], D[6], E[6], F[6], t[6], Z3[6];
    fp_sqr(A, p->x); fp_sqr(B, p->y); fp_sqr(C, B);
    fp_add(t, p->x, B); fp_sqr(t, t); fp_sub(t, t, A); fp_sub(t, t, C); fp_add(D, t, t);
    fp_add(E, A, A); fp_add(E, E, A);
    fp_sqr(F, E);
    fp_mul(Z3, p->y, p->z); fp_add(Z3, Z3, Z3);
    fp_sub(o->x, F, D); fp_sub(o->x, o->x, D);
    fp_sub(t, D, o->x); fp_mul(t, E, t);
    fp_add(C, C, C); fp_add(C, C, C); fp_add(C, C, C);
    fp_sub(o->y, t, C);
    fp_copy(o->z, Z3);
}
/* add-2007-bl with the doubling / inverse cases made explicit */
static void g1_add(g1_t *o, const g1_t *p, const g1_t *q) {
    if (g1_is_inf(p)) { *o = *q; return; }
    if (g1_is_inf(q)) { *o = *p; return; }
    uint64_t Z1Z1[6], Z2Z2[6], U1[6], U2[6], S1[6], S2[6], H[6], I[6], J[6], r[6], V[6], t[6];
    fp_sqr(Z1Z1, p->z); fp_sqr(Z2Z2, q->z);
    fp_mul(U1, p->x, Z2Z2); fp_mul(U2, q->x, Z1Z1);
    fp_mul(S1, p->y, q->z); fp_mul(S1, S1, Z2Z2);
    fp_mul(S2, q->y, p->z); fp_mul(S2, S2, Z1Z1);
    if (fp_eq(U1, U2)) {
        if (fp_eq(S1, S2)) { g1_dbl(o, p); return; }
        g1_set_inf(o); return;
    }
    fp_sub(H, U2, U1);
    fp_add(I, H, H); fp_sqr(I, I);
    fp_mul(J, H, I);
    fp_sub(r, S2, S1); fp_add(r, r, r);
    fp_mul(V, U1, I);
    g1_t R;
    fp_sqr(R.x, r); fp_sub(R.x, R.x, J); fp_sub(R.x, R.x, V); fp_sub(R.x, R.x, V);
    fp_sub(t, V, R.x); fp_mul(t, r, t);
    fp_mul(S1, S1, J); fp_add(S1, S1, S1);
    fp_sub(R.y, t, S1);
    fp_add(t, p->z, q->z); fp_sqr(t, t); fp_sub(t, t, Z1Z1); fp_sub(t, t, Z2Z2);
    fp_mul(R.z, t, H);
    *o = R;
}
static void g1_from_affine(g1_t *o, const g1a_t *a) {
    if (a->inf) { g1_set_inf(o); return; }
    fp_copy(o->x, a->x); fp_copy(o->y, a->y); fp_copy(o->z, FP_ONE);
}
static void g1_neg(g1_t *o, const g1_t *p) { fp_copy(o->x, p->x); fp_neg(o->y, p->y); fp_copy(o->z, p->z); }

static void g1a_load(g1a_t *o, const uint8_t *xy) {
    orc_init();
    int allz = 1;
    for (int i = 0; i < 96; i++) if (xy[i]) { allz = 0; break; }
    o->inf = allz;
    if (allz) { fp_zero(o->x); fp_zero(o->y); return; }
    fp_load(o->x, xy); fp_load(o->y, xy + 48);
}
static void g1_store_affine(uint8_t *xy, const g1_t *p) {
    if (g1_is_inf(p)) { memset(xy, 0, 96); return; }
    uint64_t zi[6], zi2[6], ax[6], ay[6];
    fp_inv(zi, p->z); fp_sqr(zi2, zi);
    fp_mul(ax, p->x, zi2); fp_mul(zi2, zi2, zi); fp_mul(ay, p->y, zi2);
    fp_store(xy, ax); fp_store(xy + 48, ay);
}

EXPORT void orc_g1_add(const uint8_t *p, const uint8_t *q, uint8_t *out) {
    g1a_t a, b; g1_t A, B, R;
    g1a_load(&a, p); g1a_load(&b, q);
    g1_from_affine(&A, &a); g1_from_affine(&B, &b);
    g1_add(&R, &A, &B);
    g1_store_affine(out, &R);
}
EXPORT void orc_g1_mul(const uint8_t *p, const uint8_t *k, uint8_t *out) {
    g1a_t a; g1_t P, R;
    uint64_t kk[4];
    g1a_load(&a, p); g1_from_affine(&P, &a);
    fr_from_le(kk, k);
    g1_set_inf(&R);
    for (int i = bit_length(kk, 4) - 1; i >= 0; i--) {
        g1_dbl(&R, &R);
        if ((kk[i >> 6] >> (i & 63)) & 1) g1_add(&R, &R, &P);
    }
    g1_store_affine(out, &R);
}
EXPORT int orc_g1_on_curve(const uint8_t *p) {
    g1a_t a;
    g1a_load(&a, p);
    if (a.inf) return 1;
    uint64_t l[6], r[6];
    fp_sqr(l, a.y);
    fp_sqr(r, a.x); fp_mul(r, r, a.x); fp_add(r, r, FP_B4);
    return fp_eq(l, r);
}
/* y from x: y^2 = x^3+4, p = 3 mod 4 -> y = (x^3+4)^((p+1)/4); returns 0 if not a square.
 * `larger` selects the lexicographically larger root (zcash "sort" flag: y > p-y). */
EXPORT int orc_g1_recover_y(const uint8_t *x48, int larger, uint8_t *y48) {
    orc_init();
    uint64_t x[6], r[6], y[6], chk[6], ny[6], ys[6], nys[6];
    fp_load(x, x48);
    fp_sqr(r, x); fp_mul(r, r, x); fp_add(r, r, FP_B4);
    fp_pow(y, r, fp_SQRT_E, 6);
    fp_sqr(chk, y);
    if (!fp_eq(chk, r)) return 0;
    fp_neg(ny, y);
    fp_from_mont(ys, y); fp_from_mont(nys, ny);
    int y_larger = fp_geq(ys, nys) && !fp_eq(ys, nys);
    fp_to_le(y48, (y_larger == (larger != 0)) ? ys : nys);
    return 1;
}

/* window choice for the CPU Pippenger: minimise  ceil(255/c) * (n + 2^(c-1))  */
static int g1_window(size_t n) {
    int best = 2; double bestc = 1e300;
    for (int c = 2; c <= 16; c++) {
        double cost = (double)((255 + c - 1) / c + 0) * ((double)n + (double)(1u << (c - 1)));
        if (cost < bestc) { bestc = cost; best = c; }
    }
    return best;
}
/* Signed-digit bucket Pippenger, high window to low, running-sum bucket reduction.
 * Scalars: n x 32 B LE, any value < 2^256 (reduced mod r by the caller or not — the group order divides out). */
EXPORT void orc_g1_msm(const uint8_t *pts, const uint8_t *ks, size_t n, int window_bits, uint8_t *out, int *is_inf) {
    orc_init();
    g1_t result;
    g1_set_inf(&result);
    if (n) {
        int c = window_bits > 0 ? window_bits : g1_window(n);
        int half = 1 << (c - 1), nw = (256 + c - 1) / c + 1;
        g1a_t *A = malloc(n * sizeof(g1a_t));
        int32_t *dig = malloc(n * (size_t)nw * sizeof(int32_t));
        g1_t *buckets = malloc((size_t)(half + 1) * sizeof(g1_t));
        for (size_t i = 0; i < n; i++) {
            uint64_t k[4];
            g1a_load(&A[i], pts + 96 * i);
            fr_from_le(k, ks + 32 * i);
            int carry = 0;
            for (int w = 0; w < nw; w++) {
                int d = (int)bits_at(k, 4, w * c, c) + carry;
                if (d > half) { d -= (1 << c); carry = 1; } else carry = 0;
                dig[i * nw + w] = d;
            }
        }
        for (int w = nw - 1; w >= 0; w--) {
            for (int j = 0; j < c; j++) g1_dbl(&result, &result);
            for (int b = 1; b <= half; b++) g1_set_inf(&buckets[b]);
            for (size_t i = 0; i < n; i++) {
                int d = dig[i * nw + w];
                if (d == 0 || A[i].inf) continue;
                g1_t P;
                g1_from_affine(&P, &A[i]);
                if (d < 0) { g1_neg(&P, &P); d = -d; }
                g1_add(&buckets[d], &buckets[d], &P);
            }
            g1_t running;
            g1_set_inf(&running);
            for (int b = half; b > 0; b--) {
                g1_add(&running, &running, &buckets[b]);
                g1_add(&result, &result, &running);
            }
        }
        free(A); free(dig); free(buckets);
    }
    if (is_inf) *is_inf = g1_is_inf(&result);
    g1_store_affine(out, &result);
}
/* the most literal statement of an MSM: sum of double-and-add products (ground truth for orc_g1_msm) */
EXPORT void orc_g1_msm_naive(const uint8_t *pts, const uint8_t *ks, size_t n, uint8_t *out) {
    g1_t acc;
    orc_init();
    g1_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        uint8_t t[96];
        g1a_t a; g1_t T;
        orc_g1_mul(pts + 96 * i, ks + 32 * i, t);
        g1a_load(&a, t); g1_from_affine(&T, &a);
        g1_add(&acc, &acc, &T);
    }
    g1_store_affine(out, &acc);
}
