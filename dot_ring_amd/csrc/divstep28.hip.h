// Modular inversion by Bernstein-Yang division steps ("Fast constant-time gcd computation and modular inversion", 2019) on
// signed unsaturated limbs — 14 x 28 bits for Fq (fq28.hip.h), 9 x 29 bits for Fr (fr29.hip.h): the radix the fields compute
// in, so no repacking on the way in or out.  The text below speaks of the Fq instance.
//
//   divstep(delta, f, g) = (1 - delta, g, (g - f) / 2)          if delta > 0 and g odd
//                          (1 + delta, f, (g + (g mod 2) f) / 2) otherwise                          (f odd throughout)
//
// From (f, g) = (p, x) the g component reaches 0 after at most floor((49 * 381 + 57) / 17) = 1101 steps (their Theorem 11.2
// for 381-bit inputs) and f is then +-gcd = +-1.  Steps are taken 28 at a time: the low 32 bits of f and g decide a batch, its
// effect on the full numbers is a 2 x 2 integer matrix t = (u v; q r) with 2^28 (f', g') = t (f, g) and |u| + |v|, |q| + |r| <=
// 2^28.  The same matrix updates (d, e), kept with d x = f and e x = g (mod p); the division by 2^28 is made exact by adding the
// multiple of p that cancels the low limb (the Montgomery step, centred so that |d|, |e| grow by at most p / 2 per batch:
// < 21 p after the 40 batches that cover 1101 steps).  At the end x^-1 = f d.
//
// Per batch: 28 branch-free steps of ~17 integer operations, 56 multiply-adds for (f, g), 84 for (d, e): ~750 instructions
// against ~3400 for the same number of steps of the word-wise binary Euclid in field.hip.h (inv_words) — the affine
// conversion of MSM results (k_g1_results_affine) is ONE inversion per lane, pure latency.
//
// The routine is plain integer C++: it compiles for the device (hipcc) and for the host (g++, tests/test_divstep_cpu.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DR_DIVSTEP_FN __host__ __device__ __forceinline__
#else
#define DR_DIVSTEP_FN static inline
#endif

namespace dr {

// N limbs of BITS bits; P: the modulus (odd; 21 p — or MAX_BATCHES / 2 + 1 times p — must fit the signed top limb);
// n0 = -p^-1 mod 2^BITS; MAX_BATCHES * BITS >= floor((49 bitlen(p) + 57) / 17).
// x: limbs 0..N-2 in [0, 2^BITS), 0 <= value < p.  out: x^-1 mod p up to size — a signed value with |out| < (MAX_BATCHES / 2 + 1) p
// and out * x = 1 (mod p) (0 for x = 0), limbs 0..N-2 in (-2^BITS, 2^BITS): a lazy operand for the Montgomery product that
// follows.  Returns the number of batches taken (tests).
template <int N, int BITS, int MAX_BATCHES>
DR_DIVSTEP_FN int inv_divsteps(const uint32_t (&P)[N], uint32_t n0, const int32_t (&x)[N], int32_t (&out)[N]) {
    constexpr int64_t M = ((int64_t)1 << BITS) - 1;
    constexpr int32_t HALF = (int32_t)1 << (BITS - 1);
    int32_t f[N], g[N], d[N], e[N];
#pragma unroll
    for (int i = 0; i < N; i++) { f[i] = (int32_t)P[i]; g[i] = x[i]; d[i] = 0; e[i] = 0; }
    e[0] = 1;
    int32_t delta = 1;
    int batches = 0;
#pragma unroll 1
    for (; batches < MAX_BATCHES; batches++) {
        int32_t any = 0;
#pragma unroll
        for (int i = 0; i < N; i++) any |= g[i];
        if (any == 0) break;
        // ---- BITS steps on the low words
        uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << BITS), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << BITS);
        int32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 4
        for (int i = 0; i < BITS; i++) {
            const bool swap = delta > 0 && (g0 & 1u);
            const uint32_t f1 = swap ? g0 : f0, g1 = swap ? 0u - f0 : g0;
            const int32_t u1 = swap ? q : u, v1 = swap ? r : v, q1 = swap ? -u : q, r1 = swap ? -v : r;
            delta = swap ? -delta : delta;
            const bool odd = g1 & 1u;
            f0 = f1;
            g0 = (g1 + (odd ? f1 : 0u)) >> 1;
            q = q1 + (odd ? u1 : 0);
            r = r1 + (odd ? v1 : 0);
            u = u1 << 1;
            v = v1 << 1;
            delta++;
        }
        // ---- (f, g) <- t (f, g) / 2^BITS (exact)
        {
            int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
            int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
            cf >>= BITS;
            cg >>= BITS;
#pragma unroll
            for (int i = 1; i < N; i++) {
                cf += (int64_t)u * f[i] + (int64_t)v * g[i];
                cg += (int64_t)q * f[i] + (int64_t)r * g[i];
                f[i - 1] = (int32_t)(cf & M);
                g[i - 1] = (int32_t)(cg & M);
                cf >>= BITS;
                cg >>= BITS;
            }
            f[N - 1] = (int32_t)cf;
            g[N - 1] = (int32_t)cg;
        }
        // ---- (d, e) <- t (d, e) / 2^BITS mod p
        {
            int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
            int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
            int32_t md = (int32_t)(((uint32_t)cd * n0) & (uint32_t)M), me = (int32_t)(((uint32_t)ce * n0) & (uint32_t)M);
            md = (md ^ HALF) - HALF;                                    // centred: [-2^(BITS-1), 2^(BITS-1))
            me = (me ^ HALF) - HALF;
            cd += (int64_t)md * (int32_t)P[0];
            ce += (int64_t)me * (int32_t)P[0];
            cd >>= BITS;
            ce >>= BITS;
#pragma unroll
            for (int i = 1; i < N; i++) {
                cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)md * (int32_t)P[i];
                ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)me * (int32_t)P[i];
                d[i - 1] = (int32_t)(cd & M);
                e[i - 1] = (int32_t)(ce & M);
                cd >>= BITS;
                ce >>= BITS;
            }
            d[N - 1] = (int32_t)cd;
            e[N - 1] = (int32_t)ce;
        }
    }
    // g = 0, f = +-1 (x != 0): x^-1 = f d.  f = -1 shows in the signed top limb.
    const bool negate = f[N - 1] < 0;
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = negate ? -d[i] : d[i];
    return batches;
}

// Fq: 14 x 28 bits, 40 batches cover the 1101 steps of a 381-bit modulus
constexpr int DIVSTEP_MAX_BATCHES = 40;
DR_DIVSTEP_FN int inv_divsteps28(const uint32_t (&P)[14], uint32_t n0, const int32_t (&x)[14], int32_t (&out)[14]) {
    return inv_divsteps<14, 28, DIVSTEP_MAX_BATCHES>(P, n0, x, out);
}

}  // namespace dr
