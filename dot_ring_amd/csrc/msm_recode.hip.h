// How a scalar becomes bucket entries of the G1 / twisted Edwards Pippenger kernels: signed window digits (for_each_digit) and,
// over tables with a row per bit, the width-w non-adjacent form (for_each_wnaf_digit).  Plain C++ over uint32_t words: compiles for
// the device (hipcc, kernels_g1.hip.h) and for the host (g++, tests/native/recode_check.cpp runs these very functions against big integers).
#pragma once
#include <cstdint>

#include "dev_types.hpp"

#if defined(__HIPCC__)
#define DR_RECODE_FN __device__ __forceinline__
#else
#define DR_RECODE_FN static inline
#endif

namespace dr {

// visit the signed digits of scalar k: f(window, digit) for windows [w_lo, w_hi) (the carry chain always starts at 0).
// The scalar words are indexed only by the unrolled outer loop: a run-time index (k[start >> 5]) would put the array in
// scratch memory and cost one memory round trip per digit (measured: 5.4 -> 1.x ms for the prover's sort kernel).
template <bool WITH_ZEROS = false, class F>
DR_RECODE_FN void for_each_digit(const uint32_t (&k)[9], const WindowTable& wt, int w_lo, int w_hi, F&& f) {
    uint32_t carry = 0;
    int w = 0;
#pragma unroll
    for (int li = 0; li < 8; li++) {
        const uint64_t two = (uint64_t)k[li] | ((uint64_t)k[li + 1] << 32);
        while (w < w_hi && (wt.start[w] >> 5) == li) {
            const int c = wt.width[w], sh = wt.start[w] & 31;
            const uint32_t half = 1u << (c - 1);
            uint32_t raw = ((uint32_t)(two >> sh) & ((1u << c) - 1)) + carry;
            int32_t d;
            if (raw > half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
            else { d = (int32_t)raw; carry = 0; }
            if (w >= w_lo && (WITH_ZEROS || d != 0)) f(w, d);
            w++;
        }
    }
}

// Width-w non-adjacent form (WindowTable::odd == 2, w = wt.cmax) of a scalar k < 2^255, for tables with a row per bit: odd digits
// |d| < 2^(w-1) — 256 / (w + 1) non-zero digits on average where w-bit windows have 256 / w —, every one of them an odd multiple,
// i.e. a bucket of the set as it is.  Scanning up from bit 0 with a carry c: the next digit starts at the first position whose bit
// differs from c, takes a bits from there (+ c) as v, d = v or v - 2^a (then c = 1), and the scan resumes a positions on.
// The width a is w until the top comes near.  Left at that, the LAST digit would be whatever bits remain — one to w of them with equal
// probability, so a quarter of all scalars would end in +-1 and bucket 0 of every set would hold 25 times the average list (measured:
// 1400 entries against 57; 1 ms per dense launch in the long-list kernels).  So within WNAF_ZONE * w bits of the top the remaining R bits
// are shared evenly by the ceil(R / w) digits still to come: a = ceil(R / ceil(R / w)) — the last two or three digits have 9 .. 13 bits
// each (+0.7 % digits, lowest buckets at 4x the average instead of 25x).
// Positions are counted on K = k << (W w - 256), whose length is exactly W slots of w positions: a digit at K-position P has R = W w - P
// bits above it and ceil(R / w) = W - floor(P / w) digits to go including itself, and since a >= R / (W - j) the next digit starts in
// a later slot: slot j = K-positions [w j, w j + w) starts at most one digit.  f(j, o, d): the digit d at table row wt.row[j] + o
// (wt.row[0] = 0, wt.row[j] = w j - shift).  The words of K are indexed by the unrolled outer loop only (for_each_digit).
// WITH_ZEROS: f(j, 0, 0) for slots that start none.
constexpr uint32_t WNAF_EMPTY16 = 0x7800u;                // u16 digit rows: sign << 15 | offset << 11 | bucket; offset 15 = no digit
constexpr uint32_t WNAF_ZONE = 3;
template <bool WITH_ZEROS = false, class F>
DR_RECODE_FN void for_each_wnaf_digit(const uint32_t (&k)[9], const WindowTable& wt, F&& f) {
    const uint32_t w = (uint32_t)wt.cmax, wmask = (1u << w) - 1u, L = (uint32_t)wt.W * w, shift = L - 256u;      // shift < w <= 13
    uint32_t K[10];
    K[0] = k[0] << shift;
#pragma unroll
    for (int i = 1; i < 9; i++) K[i] = shift ? (k[i] << shift) | (k[i - 1] >> (32u - shift)) : k[i];
    K[9] = 0;
    uint32_t c = 0, r = 0;                                // carry; first offset of the slot at which a digit may start
    int j = 0;
#pragma unroll
    for (int li = 0; li < 9; li++) {
        const uint64_t two = (uint64_t)K[li] | ((uint64_t)K[li + 1] << 32);
        while (j < wt.W && (wt.start[j] >> 5) == li) {
            const uint32_t chunk = (uint32_t)(two >> (wt.start[j] & 31));          // 32 valid bits; offset + width <= 2 w - 1 <= 25 are used
            const uint32_t m = (((c ? ~chunk : chunk) & wmask) >> r) << r;
            if (m) {
                const uint32_t o = (uint32_t)__builtin_ctz(m);
                const uint32_t R = L - ((uint32_t)wt.start[j] + o);                  // bits from this position to the top
                uint32_t a = w;
                if (R <= WNAF_ZONE * w) {
                    const uint32_t q = (R + w - 1u) / w;                             // digits still to come (1 .. WNAF_ZONE)
                    a = (R + q - 1u) / q;
                }
                const uint32_t v = ((chunk >> o) & ((1u << a) - 1u)) + c;           // odd
                int32_t d;
                if (v > (1u << (a - 1))) { d = (int32_t)v - (int32_t)(1u << a); c = 1; }
                else { d = (int32_t)v; c = 0; }
                f(j, j == 0 ? o - shift : o, d);
                r = o + a - w;                                                       // >= 0: the next digit starts in a later slot
            } else {
                if (WITH_ZEROS) f(j, 0u, 0);
                r = 0;
            }
            j++;
        }
    }
}

}  // namespace dr
