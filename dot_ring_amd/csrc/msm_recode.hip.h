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
// |d| < 2^(w-1) at least w positions apart — 256 / (w + 1) non-zero digits on average where w-bit windows have 256 / w —, every one
// of them an odd multiple, i.e. a bucket of the set as it is.  Scanning up from bit 0 with a carry c: the next digit starts at the
// first position whose bit differs from c, takes the w bits from there (+ c) as v, d = v or v - 2^w (then c = 1), and the scan resumes w
// positions on.  Slot j = positions [w j, w j + w) therefore starts at most one digit: f(j, offset in the slot, d); the words of k
// are indexed by the unrolled outer loop only (for_each_digit).  WITH_ZEROS: f(j, 0, 0) for slots that start none.
constexpr uint32_t WNAF_EMPTY16 = 0x7800u;                // u16 digit rows: sign << 15 | offset << 11 | bucket; offset 15 = no digit
template <bool WITH_ZEROS = false, class F>
DR_RECODE_FN void for_each_wnaf_digit(const uint32_t (&k)[9], const WindowTable& wt, F&& f) {
    const uint32_t w = (uint32_t)wt.cmax, wmask = (1u << w) - 1u, half = 1u << (w - 1);
    uint32_t c = 0, r = 0;                                // carry; first offset of the slot at which a digit may start
    int j = 0;
#pragma unroll
    for (int li = 0; li < 8; li++) {
        const uint64_t two = (uint64_t)k[li] | ((uint64_t)k[li + 1] << 32);
        while (j < wt.W && (wt.start[j] >> 5) == li) {
            const uint32_t chunk = (uint32_t)(two >> (wt.start[j] & 31));          // >= 33 valid bits; 2 w - 1 <= 27 are used
            const uint32_t m = (((c ? ~chunk : chunk) & wmask) >> r) << r;
            if (m) {
                const uint32_t o = (uint32_t)__builtin_ctz(m);
                const uint32_t v = ((chunk >> o) & wmask) + c;                      // odd
                int32_t d;
                if (v > half) { d = (int32_t)v - (int32_t)(1u << w); c = 1; }
                else { d = (int32_t)v; c = 0; }
                f(j, o, d);
                r = o;
            } else {
                if (WITH_ZEROS) f(j, 0u, 0);
                r = 0;
            }
            j++;
        }
    }
}

}  // namespace dr
