/*
 * ORACLE — test infrastructure only (never shipped, never on the product path).
 *
 * Generic N x 64-bit-limb Montgomery field template.  Include with
 *   #define MT_NL      <limbs>
 *   #define MT_(name)  <prefix>##name
 * and the constants MT_(P)[MT_NL], MT_(R2)[MT_NL], MT_(N0) (= -P^-1 mod 2^64).
 *
 * Restates the CIOS Montgomery multiplication of the reference
 *   dot_ring/curve/native_field/bls12_381_scalar.c:99-203  (mul_mont, 4 limbs, unrolled)
 *   dot_ring/curve/native_field/bls12_381_scalar.c:43-97   (add / sub with conditional correction)
 *   dot_ring/curve/native_field/bls12_381_scalar.c:205-264 (to/from mont, exp, Fermat inverse)
 * in loop form so that the same text serves Fr (4 limbs) and Fp-381 (6 limbs; the
 * reference gets Fp-381 from the external blst library).
 * Results are always fully reduced (< P), like the reference.
 */
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 MT_(u128);

static inline int MT_(is_zero)(const uint64_t *a) {
    uint64_t acc = 0;
    for (int i = 0; i < MT_NL; i++) acc |= a[i];
    return acc == 0;
}

static inline int MT_(eq)(const uint64_t *a, const uint64_t *b) {
    uint64_t acc = 0;
    for (int i = 0; i < MT_NL; i++) acc |= a[i] ^ b[i];
    return acc == 0;
}

static inline void MT_(copy)(uint64_t *o, const uint64_t *a) { memcpy(o, a, MT_NL * 8); }
static inline void MT_(zero)(uint64_t *o) { memset(o, 0, MT_NL * 8); }

/* returns 1 when a >= b */
static inline int MT_(geq)(const uint64_t *a, const uint64_t *b) {
    for (int i = MT_NL - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}

static inline uint64_t MT_(raw_add)(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    MT_(u128) c = 0;
    for (int i = 0; i < MT_NL; i++) {
        c += (MT_(u128))a[i] + b[i];
        o[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}

static inline uint64_t MT_(raw_sub)(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < MT_NL; i++) {
        MT_(u128) d = (MT_(u128))a[i] - b[i] - borrow;
        o[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 127);
    }
    return borrow;
}

/* bls12_381_scalar.c:43 — add then subtract P when the sum overflowed or is >= P */
static inline void MT_(add)(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    uint64_t s[MT_NL], t[MT_NL];
    uint64_t carry = MT_(raw_add)(s, a, b);
    uint64_t borrow = MT_(raw_sub)(t, s, MT_(P));
    MT_(copy)(o, (carry || !borrow) ? t : s);
}

/* bls12_381_scalar.c:74 — subtract then add P back on borrow */
static inline void MT_(sub)(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    uint64_t d[MT_NL], t[MT_NL];
    uint64_t borrow = MT_(raw_sub)(d, a, b);
    MT_(raw_add)(t, d, MT_(P));
    MT_(copy)(o, borrow ? t : d);
}

static inline void MT_(neg)(uint64_t *o, const uint64_t *a) {
    if (MT_(is_zero)(a)) { MT_(zero)(o); return; }
    MT_(raw_sub)(o, MT_(P), a);
}

/* bls12_381_scalar.c:99 — CIOS: for each b-limb, r += a*b[i]; m = r[0]*N0; r = (r + m*P) >> 64 */
static inline void MT_(mul)(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    uint64_t r[MT_NL + 2];
    memset(r, 0, sizeof r);
    for (int i = 0; i < MT_NL; i++) {
        uint64_t u = 0;
        for (int j = 0; j < MT_NL; j++) {
            MT_(u128) prod = (MT_(u128))a[j] * b[i] + r[j] + u;
            r[j] = (uint64_t)prod;
            u = (uint64_t)(prod >> 64);
        }
        MT_(u128) top = (MT_(u128))r[MT_NL] + u;
        r[MT_NL] = (uint64_t)top;
        r[MT_NL + 1] = (uint64_t)(top >> 64);

        uint64_t m = r[0] * MT_(N0);
        MT_(u128) prod = (MT_(u128))m * MT_(P)[0] + r[0];
        u = (uint64_t)(prod >> 64);
        for (int j = 1; j < MT_NL; j++) {
            prod = (MT_(u128))m * MT_(P)[j] + r[j] + u;
            r[j - 1] = (uint64_t)prod;
            u = (uint64_t)(prod >> 64);
        }
        top = (MT_(u128))r[MT_NL] + u;
        r[MT_NL - 1] = (uint64_t)top;
        r[MT_NL] = r[MT_NL + 1] + (uint64_t)(top >> 64);
    }
    uint64_t t[MT_NL];
    uint64_t borrow = MT_(raw_sub)(t, r, MT_(P));
    MT_(copy)(o, (r[MT_NL] || !borrow) ? t : r);
}

static inline void MT_(sqr)(uint64_t *o, const uint64_t *a) { MT_(mul)(o, a, a); }

/* bls12_381_scalar.c:205,209 */
static inline void MT_(to_mont)(uint64_t *o, const uint64_t *a) { MT_(mul)(o, a, MT_(R2)); }
static inline void MT_(from_mont)(uint64_t *o, const uint64_t *a) {
    uint64_t one[MT_NL];
    MT_(zero)(one);
    one[0] = 1;
    MT_(mul)(o, a, one);
}

static inline void MT_(one_mont)(uint64_t *o) {
    uint64_t one[MT_NL];
    MT_(zero)(one);
    one[0] = 1;
    MT_(to_mont)(o, one);
}

/* bls12_381_scalar.c:224 — LSB-first square-and-multiply; base in Montgomery form, exp a plain integer */
static inline void MT_(pow)(uint64_t *o, const uint64_t *base, const uint64_t *e, int elimbs) {
    uint64_t res[MT_NL], b[MT_NL];
    MT_(one_mont)(res);
    MT_(copy)(b, base);
    for (int i = 0; i < elimbs; i++) {
        uint64_t w = e[i];
        for (int j = 0; j < 64; j++) {
            if (w & 1) MT_(mul)(res, res, b);
            MT_(sqr)(b, b);
            w >>= 1;
        }
    }
    MT_(copy)(o, res);
}

/* bls12_381_scalar.c:245 — Fermat inverse a^(P-2) */
static inline void MT_(inv)(uint64_t *o, const uint64_t *a) {
    uint64_t e[MT_NL], two[MT_NL];
    MT_(zero)(two);
    two[0] = 2;
    MT_(raw_sub)(e, MT_(P), two);
    MT_(pow)(o, a, e, MT_NL);
}

/* little-endian byte I/O (bls12_381_scalar.c:266,276) */
static inline void MT_(from_le)(uint64_t *o, const uint8_t *in) {
    for (int i = 0; i < MT_NL; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w |= (uint64_t)in[i * 8 + j] << (8 * j);
        o[i] = w;
    }
}
static inline void MT_(to_le)(uint8_t *out, const uint64_t *a) {
    for (int i = 0; i < MT_NL; i++)
        for (int j = 0; j < 8; j++) out[i * 8 + j] = (uint8_t)(a[i] >> (8 * j));
}
/* standard-form LE bytes -> Montgomery, and back */
static inline void MT_(load)(uint64_t *o, const uint8_t *in) {
    uint64_t t[MT_NL];
    MT_(from_le)(t, in);
    MT_(to_mont)(o, t);
}
static inline void MT_(store)(uint8_t *out, const uint64_t *a) {
    uint64_t t[MT_NL];
    MT_(from_mont)(t, a);
    MT_(to_le)(out, t);
}
