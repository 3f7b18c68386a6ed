// Plain types shared by the host orchestration (capi_*.hip) and the kernel headers.
#pragma once
#include <cstdint>
#include <vector>

#include "hostmath.hpp"

namespace dr {

// Window table: the 256 scalar bits are tiled by W windows of width cmax or cmax-1 (wider ones on top), so every
// window has about the same number of live buckets.  A narrow top window would otherwise hold only a few bits
// and funnel n/2^t points into each of its few buckets — one lane then walks a chain thousands of points long.
// Fixed-base tables come in two shapes: one row per window (row[w] = w: table[w][i] = 2^(start_w) * base[i]) or one row per BIT
// (row[w] = start[w]: table[s][i] = 2^s * base[i] for every s < 256; any tiling of the 256 bits can then be used per call).
// `odd` == 2 (bit rows only): only odd multiples have buckets — 2^(cmax-2) per set, a set's value is sum_j (2j + 1) B_j — and the scalar
// is recoded in width-cmax non-adjacent form (msm_recode.hip.h: for_each_wnaf_digit): odd digits |d| < 2^(cmax-1), about
// 256 / (cmax + 1) of them instead of 256 / cmax window digits; a digit at bit position p takes its point from row p.  The W
// "windows" are then slots of k << (W cmax - 256): slot j = positions [cmax j, cmax j + cmax) starts at most one digit
// (start[j] = cmax j, row[j] = the table row of the slot's first position, width[j] = cmax).
struct WindowTable {
    int W, cmax;
    uint8_t start[40];   // first bit of window w   (W <= 40: widths >= 7 ... see make_plan)
    uint8_t width[40];
    uint8_t row[40];     // table row of window w (fixed-base table mode)
    int odd;
};
constexpr int MAX_WINDOWS = 40;

// NTT twiddle tables cached per (n, omega) in a context (kernels_ntt.hip.h: ntt_run)
struct TwiddleCache {
    struct Entry {
        unsigned log2n;
        drh::Fr omega;
        uint32_t* d_tw;
    };
    std::vector<Entry> entries;
};

}  // namespace dr
