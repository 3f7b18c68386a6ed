// Plain types shared by the host orchestration (capi_*.hip) and the kernel headers.
#pragma once
#include <cstdint>
#include <vector>

#include "hostmath.hpp"

namespace dr {

// Window table: the 256 scalar bits are tiled by W windows of width cmax or cmax-1 (wider ones on top), so every
// window has about the same number of live buckets.  A narrow top window would otherwise hold only a few bits
// and funnel n/2^t points into each of its few buckets — one lane then walks a chain thousands of points long.
struct WindowTable {
    int W, cmax;
    uint8_t start[40];   // first bit of window w   (W <= 40: widths >= 7 ... see make_plan)
    uint8_t width[40];
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
