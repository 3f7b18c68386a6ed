// BLS12-381 G1 Pippenger MSM (seam B, kernel K5) — the KZG commit behind RingRoot / ring proofs.
// Replaces blst.P1_Affines.mult_pippenger as called from dot_ring/ring_proof/pcs/kzg.py:152-175.
//
// Data layout in HBM
//   bases    : G1Affine[n]            96 B AoS, Montgomery form, (0,0) = infinity. One gather of a base is
//                                     six 16-B loads from one or two 128-B lines.
//   scalars  : u32[batch][n][8]       little-endian limbs as uploaded (any value < 2^256)
//   digits   : i32[batch*W][n]        signed window digits in (-2^(c-1), 2^(c-1)], coalesced along n
//   counts / offsets : u32[batch*W*H (+1)]   H = 2^(c-1) buckets per window
//   sorted   : u32[nnz]               base index | sign<<31, grouped by bucket (counting sort)
//   buckets  : G1Xyzz[batch*W*H]      192 B AoS
//   partial  : G1Xyzz[batch*W*T]      per-chunk running-sum results, T = H / L
//   winsum   : G1Xyzz[batch*W]        sum_j (j+1)*bucket_j for each window
// "batch" independent scalar vectors over the same bases are handled as extra windows (window id =
// b*W + w): the bases are read once per window from L2/HBM and nothing else changes.
//
// Pipeline: k_g1_digits -> scan (3 small kernels) -> k_g1_scatter -> k_g1_accumulate (dominant)
//           -> k_g1_reduce_chunks -> k_g1_reduce_windows -> [host or k_g1_horner] combine windows.
#pragma once
#include "dev_types.hpp"
#include "g1.hip.h"
#include "msm_recode.hip.h"

namespace dr {

// standard-form little-endian limbs -> Montgomery, in place (SRS load). (0,0) stays (0,0).
__global__ __launch_bounds__(256) void k_g1_bases_to_mont(uint32_t* bases, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* p = bases + (size_t)i * 24;
    uint32_t wx[12], wy[12];
    load_words12(p, wx);
    load_words12(p + 12, wy);
    store_fq28(p, to_mont28(wx));
    store_fq28(p + 12, to_mont28(wy));
}

// synthetic bases for full-size measurements: bases[i] = (first + i) * seed   (seed affine, Montgomery)
__global__ __launch_bounds__(256) void k_g1_synth_bases(uint32_t* bases, uint32_t n, uint32_t first, const uint32_t* seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine s = load_affine(seed, 0);
    uint32_t k = first + i;
    G1Xyzz acc = g1_inf();
#pragma unroll 1
    for (int bit = 31 - __clz(k | 1); bit >= 0; bit--) {
        acc = g1_dbl(acc);
        if ((k >> bit) & 1) acc = g1_madd(acc, s);
    }
    store_affine(bases, i, g1_to_affine_dev(acc));
}

// known-tau SRS for domains the shipped file does not cover: bases[i] = scalars[i] * seed with
// scalars[i] = tau^i (standard form, < r) prepared by the host; one lane per base
__global__ __launch_bounds__(256) void k_g1_scalar_bases(uint32_t* bases, uint32_t n, const uint32_t* __restrict__ scalars, const uint32_t* seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = scalars[(size_t)i * 8 + j];
    G1Affine s = load_affine(seed, 0);
    G1Xyzz acc = g1_inf();
#pragma unroll 1
    for (int bit = 254; bit >= 0; bit--) {
        acc = g1_dbl(acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) acc = g1_madd(acc, s);
    }
    store_affine(bases, i, g1_to_affine_dev(acc));
}

// zcash-compressed G1 (48 bytes, big-endian; flag bits: 7 compressed, 6 infinity, 5 "y is the larger root") ->
// Montgomery affine, the layout the MSM kernels read; (0,0) = infinity.  On-curve by construction, no subgroup
// check — as blst's P1_Affine(bytes) behind KZG.decompress_g1 (dot_ring/ring_proof/pcs/kzg.py:137-144).
// ok[i] = 0 for malformed encodings (the point is then written as infinity).
__global__ __launch_bounds__(256) void k_g1_decompress(const uint8_t* __restrict__ enc /* n*48 */, uint32_t* __restrict__ bases /* n*24 */, uint32_t* __restrict__ ok, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = enc + (size_t)i * 48;
    uint32_t xs[12];
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const uint8_t* q = p + 44 - 4 * j;
        xs[j] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    const uint32_t flags = xs[11] >> 29;
    xs[11] &= 0x1fffffffu;
    bool valid = (flags & 4u) != 0;
    G1Affine out;
    out.x = Fq28::zero(); out.y = Fq28::zero(); out.inf = 1;
    uint32_t xnz = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) xnz |= xs[j];
    if (flags & 2u) {
        if (xnz != 0 || (flags & 1u)) valid = false;
    } else {
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) (void)subb(xs[j], FqParams::P[j], borrow);
        if (!borrow) {
            valid = false;
#pragma unroll
            for (int j = 0; j < 12; j++) xs[j] = 0;
        }
        Fq28 X = to_mont28(xs);
        Fq28 rhs = add(mul(sqr(X), X), Fq28::constant<Fq28Params::FOUR>());      // limbs < 2^29
        // p = 3 mod 4: y = rhs^((p+1)/4)
        constexpr uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                    0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
        // (p+1)/4 has 379 bits, 256 of them set: sliding windows of up to three bits over rhs, rhs^3, rhs^5, rhs^7 — ~100 products
        // instead of 256 (the exponent is the same in every lane: the control flow is scalar)
        Fq28 y;
        {
            const Fq28 r1 = carry(rhs), r2 = sqr(r1), r3 = mul(r1, r2), r5 = mul(r3, r2), r7 = mul(r5, r2);
            auto bit = [&](int i) -> uint32_t { return (E[i >> 5] >> (i & 31)) & 1u; };
            int i = 378;
            bool started = false;
            y = Fq28::one();
#pragma unroll 1
            while (i >= 0) {
                if (!bit(i)) { y = sqr(y); i--; continue; }
                int l = i >= 2 ? 3 : i + 1;
                while (!bit(i - l + 1)) l--;
                uint32_t v = 0;
                for (int k = 0; k < l; k++) v = (v << 1) | bit(i - k);
                if (started) {
#pragma unroll 1
                    for (int k = 0; k < l; k++) y = sqr(y);
                }
                Fq28 m;
#pragma unroll
                for (int t = 0; t < L28; t++) m.l[t] = v == 1 ? r1.l[t] : v == 3 ? r3.l[t] : v == 5 ? r5.l[t] : r7.l[t];
                y = started ? mul(y, m) : m;
                started = true;
                i -= l;
            }
        }
        if (!is_zero_mod_p(sub(sqr(y), rhs))) valid = false;
        // sign: the flag says whether y is the larger of (y, p - y) as standard-form integers
        uint32_t ys[12], nys[12];
        from_mont28(y, ys);
        uint32_t ynz = 0;
        borrow = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) { nys[j] = subb(FqParams::P[j], ys[j], borrow); ynz |= ys[j]; }
        bool y_larger = false;
#pragma unroll
        for (int j = 11; j >= 0; j--) {
            if (ys[j] != nys[j]) { y_larger = ys[j] > nys[j]; break; }
        }
        if (ynz == 0) y_larger = false;                  // y = 0: -y = 0 as well (not on the curve anyway)
        out.x = X;
        out.y = cneg(y, y_larger != ((flags & 1u) != 0));
        out.inf = valid ? 0u : 1u;
    }
    store_affine(bases, i, out);
    ok[i] = valid ? 1u : 0u;
}

// Montgomery affine -> standard-form little-endian limbs (SRS download)
__global__ __launch_bounds__(256) void k_g1_bases_from_mont(const uint32_t* bases, uint32_t* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = bases + (size_t)i * 24;
    uint32_t* q = out + (size_t)i * 24;
    uint32_t w[12];
    from_mont28(load_fq28(p), w);
    store_words12(q, w);
    from_mont28(load_fq28(p + 12), w);
    store_words12(q + 12, w);
}

// (WindowTable — how the 256 scalar bits are tiled by windows — lives in dev_types.hpp: the host plans it.)

// ---- 1. signed window digits + bucket histogram.  One lane per scalar.
// `single` = the bases are a fixed-base window table (k_g1_window_table): all windows of an MSM share ONE bucket set.
// In that mode the points of one MSM may be split into `groups` index ranges, each with its own bucket set, so that
// a single huge MSM still fills the chip (bucket set id = b*groups + i*groups/n).
__global__ void k_g1_digits(const uint32_t* __restrict__ scalars, uint32_t n, uint32_t batch, WindowTable wt, int single, uint32_t groups,
                            int32_t* __restrict__ digits, uint32_t* __restrict__ counts) {
    const int W = wt.W;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * batch) return;
    uint32_t b = (uint32_t)(gid / n), i = (uint32_t)(gid % n);
    uint32_t k[9];
    {
        const uint4* q = reinterpret_cast<const uint4*>(scalars + gid * 8);
        uint4 lo = q[0], hi = q[1];
        k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
        k[8] = 0;
    }
    // scalars are taken mod r (the group order): 2^256 < 3r, so two conditional subtractions.  k < r < 2^255 and
    // the windows tile 256 bits, so the top window has a spare bit and the signed recoding never carries out.
    {
        constexpr uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
#pragma unroll 1
        for (int it = 0; it < 2; it++) {
            uint32_t d[8], borrow = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) d[j] = subb(k[j], R[j], borrow);
#pragma unroll
            for (int j = 0; j < 8; j++) k[j] = borrow ? k[j] : d[j];
        }
    }
    const uint32_t H = 1u << (wt.cmax - 1);     // bucket stride per window
    // the scalar words are indexed by the unrolled outer loop only (a run-time word index would spill k[] to scratch)
    uint32_t carry = 0;
    int w = 0;
#pragma unroll
    for (int li = 0; li < 8; li++) {
        const uint64_t two = (uint64_t)k[li] | ((uint64_t)k[li + 1] << 32);
        while (w < W && (wt.start[w] >> 5) == li) {
            const int c = wt.width[w], sh = wt.start[w] & 31;
            const uint32_t half = 1u << (c - 1);
            uint32_t raw = ((uint32_t)(two >> sh) & ((1u << c) - 1)) + carry;
            int32_t d;
            if (raw > half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
            else { d = (int32_t)raw; carry = 0; }
            size_t win = (size_t)b * W + w;
            digits[win * n + i] = d;
            if (d != 0) {
                uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
                size_t bset = single ? (size_t)b * groups + (size_t)(((uint64_t)i * groups) / n) : win;
                atomicAdd(&counts[bset * H + (mag - 1)], 1u);
            }
            w++;
        }
    }
}

// ---- 2. exclusive scan of the histogram (three passes; the array is at most a few million entries)
constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

template <int BLOCK = SCAN_BLOCK>
DR_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* smem, uint32_t& total) {
    // wave scan with shuffles, then scan of the BLOCK / 64 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t y = __shfl_up(x, s, 64);
        if (lane >= s) x += y;
    }
    if (lane == 63) smem[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < BLOCK / 64; w++) {
        uint32_t t = smem[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                           uint32_t* __restrict__ tile_sums, size_t n) {
    __shared__ uint32_t smem[SCAN_BLOCK / 64];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = base + j < n ? in[base + j] : 0;
        sum += v[j];
    }
    uint32_t total;
    uint32_t excl = block_exclusive_scan(sum, smem, total);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) out[base + j] = excl;
        excl += v[j];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
// single block: exclusive scan of the tile sums (serial over tiles of 256; tile count is small)
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tile_sums(uint32_t* tile_sums, uint32_t ntiles, uint32_t* grand_total) {
    __shared__ uint32_t smem[SCAN_BLOCK / 64];
    uint32_t running = 0;
    for (uint32_t base = 0; base < ntiles; base += SCAN_BLOCK) {
        uint32_t idx = base + threadIdx.x;
        uint32_t v = idx < ntiles ? tile_sums[idx] : 0;
        uint32_t total;
        uint32_t excl = block_exclusive_scan(v, smem, total);
        if (idx < ntiles) tile_sums[idx] = running + excl;
        running += total;
    }
    if (threadIdx.x == 0) *grand_total = running;
}
__global__ void k_scan_add(uint32_t* out, const uint32_t* tile_sums, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_sums[i / SCAN_TILE];
}

// ---- 3. counting-sort scatter: group (index|sign) by bucket
// W > 0 selects the fixed-base-table mode: the entry indexes table[w][tbl_offset + i] and the bucket set is per MSM.
__global__ void k_g1_scatter(const int32_t* __restrict__ digits, uint32_t n, size_t windows, uint32_t H, int W, WindowTable wt, uint32_t tbl_stride,
                             uint32_t tbl_offset, uint32_t groups, const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                             uint32_t* __restrict__ sorted) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= windows * n) return;
    int32_t d = digits[gid];
    if (d == 0) return;
    size_t win = gid / n;
    uint32_t i = (uint32_t)(gid % n);
    uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    size_t bset = win;
    uint32_t entry = i;
    if (W > 0) {
        bset = (win / W) * groups + (size_t)(((uint64_t)i * groups) / n);
        entry = (uint32_t)wt.row[win % W] * tbl_stride + tbl_offset + i;
    }
    size_t bucket = bset * H + (mag - 1);
    uint32_t pos = offsets[bucket] + atomicAdd(&cursor[bucket], 1u);
    sorted[pos] = entry | (d < 0 ? 0x80000000u : 0u);
}

// fixed-base window table: table[w][i] = 2^(start_w) * base[i]  (affine, Montgomery), one lane per base
__global__ __launch_bounds__(256) void k_g1_window_table(const uint32_t* __restrict__ bases, uint32_t n, WindowTable wt, uint32_t pt_words, uint32_t* __restrict__ table) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = load_affine(bases, i);
    G1Xyzz cur = g1_from_affine(a);
#pragma unroll 1
    for (int w = 0; w < wt.W; w++) {
        if (w > 0) {
#pragma unroll 1
            for (int j = 0; j < wt.width[w - 1]; j++) cur = g1_dbl(cur);
            a = g1_to_affine_dev(cur);
        }
        store_affine(table + ((size_t)w * n + i) * (pt_words - 24), (size_t)w * n + i, a);      // record of pt_words words
    }
}

// fixed-base table with one row per bit: table[s][i] = 2^s * base[i], s < rows (affine, Montgomery), one lane per base
__global__ __launch_bounds__(128) void k_g1_bit_table(const uint32_t* __restrict__ bases, uint32_t n, uint32_t rows, uint32_t pt_words, uint32_t* __restrict__ table) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = load_affine(bases, i);
#pragma unroll 1
    for (uint32_t s = 0; s < rows; s++) {
        if (s > 0 && !a.inf) a = g1_to_affine_dev(g1_dbl_affine(a));
        store_affine(table + ((size_t)s * n + i) * (pt_words - 24), (size_t)s * n + i, a);      // record of pt_words words
    }
}

// a digit of magnitude mag in window w -> its bin within the set; `row` = the table row its point comes from (windows: one bucket per
// magnitude; the non-adjacent form of bit-row tables has its own visitor, msm_recode.hip.h: for_each_wnaf_digit)
DR_DEV uint32_t digit_bin(const WindowTable& wt, int w, uint32_t mag, uint32_t& row) {
    row = wt.row[w];
    return mag - 1u;
}

// ---- 1'-3'. LDS counting sort: ONE workgroup owns one bucket set.  When a bucket set is small (H <= 8192 counters =
// 32 KiB of LDS) and fed by a bounded number of digits (the batched prover: 7k MSMs x 2048 buckets), the histogram,
// its exclusive scan and the placement all happen in LDS: no digit array in HBM, no global atomics, no global scan.
// Each set gets a fixed-capacity segment of `sorted` (capacity = the most digits it can receive), so segment bases
// need no cross-set scan.  Pass 1 counts, pass 2 recomputes the digits and places them.
constexpr int SORT_BLOCK = 256;
constexpr uint32_t SORT_MAX_H = 8192;

struct SortSetParams {
    uint32_t n, batch, H, groups;      // groups: table-mode index groups per MSM (1 otherwise)
    int single;                         // 1: window-table mode (set = MSM x group, all windows), 0: set = (MSM, window)
    uint32_t tbl_stride, tbl_offset;
    uint32_t capacity;                  // entries reserved per set in `sorted`
    uint32_t short_from, n_short;       // scalar vectors b >= short_from are zero beyond n_short entries: not even read
    uint32_t n_pad, digits_per_set;     // staged variant: row length (a multiple of 8) and u16 digits reserved per set
    uint32_t sets;
    int fold;                           // table mode: scalars above r / 2 are replaced by their negatives (scalar_fold_sign)
};

DR_DEV void load_scalar_mod_r(const uint32_t* __restrict__ scalars, size_t idx, uint32_t (&k)[9]) {
    const uint4* q = reinterpret_cast<const uint4*>(scalars + idx * 8);
    uint4 lo = q[0], hi = q[1];
    k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
    k[8] = 0;
    constexpr uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
        uint32_t d[8], borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) d[j] = subb(k[j], R[j], borrow);
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = borrow ? k[j] : d[j];
    }
}

// k (already < r) -> min(k, r - k); true when it was replaced: k P = -(r - k) P for a point of the order-r group.  A scalar just below r —
// the "-1" of a difference column — then has ONE non-zero digit instead of one in almost every window (asked for by the caller of a fixed-base
// table MSM whose scalars are differences — MsmTable::fold_sign, the summation-by-parts commitments; the bases are SRS-derived points of G1).
DR_DEV bool scalar_fold_sign(uint32_t (&k)[9]) {
    constexpr uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    constexpr uint32_t HALF[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};   // (r - 1) / 2
    uint32_t borrow = 0, d[8];
#pragma unroll
    for (int j = 0; j < 8; j++) (void)subb(HALF[j], k[j], borrow);       // borrow <=> k > (r - 1) / 2
    const bool neg = borrow != 0;
    borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = subb(R[j], k[j], borrow);
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = neg ? d[j] : k[j];
    return neg;
}

// (for_each_digit / for_each_wnaf_digit — how a scalar becomes bucket entries — live in msm_recode.hip.h: the host tests run them too)

__global__ __launch_bounds__(SORT_BLOCK) void k_g1_sort_sets(const uint32_t* __restrict__ scalars, WindowTable wt, SortSetParams sp,
                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ sorted) {
    __shared__ uint32_t bins[SORT_MAX_H];
    __shared__ uint32_t smem[SORT_BLOCK / 64];
    const uint32_t H = sp.H, HB = sp.H, set = blockIdx.x;
    // which scalars and windows feed this set
    uint32_t b, i_lo, i_hi;
    int w_lo, w_hi;
    if (sp.single) {
        b = set / sp.groups;
        uint32_t g = set % sp.groups;
        const uint32_t n_eff = b >= sp.short_from ? sp.n_short : sp.n;
        i_lo = (uint32_t)(((uint64_t)g * n_eff + sp.groups - 1) / sp.groups);
        i_hi = (uint32_t)(((uint64_t)(g + 1) * n_eff + sp.groups - 1) / sp.groups);
        w_lo = 0; w_hi = wt.W;
    } else {
        // set = ((b * W) + w) * groups + g: window w of scalar vector b, index group g (groups > 1: the variable-base
        // Bandersnatch Pippenger, where one bucket set per window would leave the chip empty)
        const uint32_t sw = set / sp.groups, g = set % sp.groups;
        b = sw / wt.W;
        w_lo = (int)(sw % wt.W); w_hi = w_lo + 1;
        i_lo = (uint32_t)(((uint64_t)g * sp.n + sp.groups - 1) / sp.groups);
        i_hi = (uint32_t)(((uint64_t)(g + 1) * sp.n + sp.groups - 1) / sp.groups);
    }
    for (uint32_t j = threadIdx.x; j < HB; j += SORT_BLOCK) bins[j] = 0;
    __syncthreads();
    // pass 1: histogram
    for (uint32_t i = i_lo + threadIdx.x; i < i_hi; i += SORT_BLOCK) {
        uint32_t k[9];
        load_scalar_mod_r(scalars, (size_t)b * sp.n + i, k);
        if (sp.fold) (void)scalar_fold_sign(k);
        if (wt.odd == 2) {
            for_each_wnaf_digit(k, wt, [&](int, uint32_t, int32_t d) { atomicAdd(&bins[((uint32_t)(d < 0 ? -d : d) - 1u) >> 1], 1u); });
            continue;
        }
        for_each_digit(k, wt, w_lo, w_hi, [&](int w, int32_t d) {
            uint32_t row;
            atomicAdd(&bins[digit_bin(wt, w, (uint32_t)(d < 0 ? -d : d), row)], 1u);
        });
    }
    __syncthreads();
    // counts out; in-place exclusive scan of the H bins (each lane owns H/SORT_BLOCK consecutive bins)
    const uint32_t per = (HB + SORT_BLOCK - 1) / SORT_BLOCK, lo = threadIdx.x * per < HB ? threadIdx.x * per : HB, hi = lo + per < HB ? lo + per : HB;
    uint32_t local = 0;
    for (uint32_t j = lo; j < hi; j++) {
        uint32_t c = bins[j];
        counts[(size_t)set * H + j] = c;
        local += c;
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(local, smem, total);
    const uint32_t base = set * sp.capacity;
    for (uint32_t j = lo; j < hi; j++) {
        uint32_t c = bins[j];
        bins[j] = run;                                  // becomes the placement cursor
        offsets[(size_t)set * H + j] = base + run;
        run += c;
    }
    __syncthreads();
    // pass 2: placement
    for (uint32_t i = i_lo + threadIdx.x; i < i_hi; i += SORT_BLOCK) {
        uint32_t k[9];
        load_scalar_mod_r(scalars, (size_t)b * sp.n + i, k);
        const bool neg = sp.fold ? scalar_fold_sign(k) : false;
        if (wt.odd == 2) {
            for_each_wnaf_digit(k, wt, [&](int j, uint32_t o, int32_t d) {
                const uint32_t pos = atomicAdd(&bins[((uint32_t)(d < 0 ? -d : d) - 1u) >> 1], 1u);
                const uint32_t entry = ((uint32_t)wt.row[j] + o) * sp.tbl_stride + sp.tbl_offset + i;
                sorted[base + pos] = entry | ((d < 0) != neg ? 0x80000000u : 0u);
            });
            continue;
        }
        for_each_digit(k, wt, w_lo, w_hi, [&](int w, int32_t d) {
            uint32_t row;
            uint32_t pos = atomicAdd(&bins[digit_bin(wt, w, (uint32_t)(d < 0 ? -d : d), row)], 1u);
            uint32_t entry = sp.single ? row * sp.tbl_stride + sp.tbl_offset + i : i;
            sorted[base + pos] = entry | ((d < 0) != neg ? 0x80000000u : 0u);
        });
    }
}

// ---- 3a'. the same sort with COALESCED stores (round 2).  The kernel above writes every (index, sign) entry as a scattered
// 4-byte store into its set's 0.5 MB segment: 139 M stores per dense prover launch, 32-byte sectors, 3 of the kernel's 5 ms.
// Here the sorted segment is assembled in LDS, one chunk of ~26 k entries at a time, and copied out in whole lines:
//   pass 1   histogram in LDS as before, and every digit is written once as a u16 (sign << 15 | magnitude), row-major by
//            window so that lanes write consecutive addresses (2 bytes per entry instead of recomputing the digits later);
//   scan     bucket offsets; bucket ranges [jb_k, jb_k+1) whose segments start inside chunk k (positions [k C, (k+1) C));
//   pass 2.k re-read the u16 digits (16-byte loads, L2 / Infinity-Cache hits), take the entries whose bucket lies in chunk k,
//            place them in the LDS stage at (position - k C), copy the stage out.  A bucket that runs past the stage (skewed
//            scalars) stores its overflow directly.
// One workgroup of 1024 lanes per set (144 KB of LDS: one workgroup per CU, four waves per SIMD).
// Two instances: up to 2048 buckets per set (the prover's 12-bit SRS windows: 8 KB of bins, 36864 staged entries — four
// chunks for a dense 135 k-entry set) and up to 8192 (32 KB of bins, 28672 staged entries).
constexpr int SORT2_BLOCK = 1024;
constexpr uint32_t SORT2_SLACK = 2048, SORT2_MAX_CHUNKS = 64;
constexpr uint32_t SORT2_CAP_SMALL_H = 36864, SORT2_CAP_LARGE_H = 28672, SORT2_SMALL_H = 2048;

template <uint32_t MAX_H, uint32_t SORT2_CAP>
__global__ __launch_bounds__(SORT2_BLOCK) void k_g1_sort_sets_staged(const uint32_t* __restrict__ scalars, WindowTable wt, SortSetParams sp,
                                                                    uint16_t* __restrict__ digits16, uint32_t* __restrict__ counts,
                                                                    uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
    constexpr uint32_t SORT2_CHUNK = SORT2_CAP - SORT2_SLACK;
    __shared__ uint32_t bins[MAX_H];
    __shared__ uint32_t stage[SORT2_CAP];
    __shared__ uint32_t cs[SORT2_MAX_CHUNKS + 1], jb[SORT2_MAX_CHUNKS + 1];
    __shared__ uint32_t smem[SORT2_BLOCK / 64];
    const uint32_t H = sp.H, set = blockIdx.x, tid = threadIdx.x;
    uint32_t b, i_lo, i_hi;
    int w_lo, w_hi;
    if (sp.single) {
        b = set / sp.groups;
        uint32_t g = set % sp.groups;
        const uint32_t n_eff = b >= sp.short_from ? sp.n_short : sp.n;
        i_lo = (uint32_t)(((uint64_t)g * n_eff + sp.groups - 1) / sp.groups);
        i_hi = (uint32_t)(((uint64_t)(g + 1) * n_eff + sp.groups - 1) / sp.groups);
        w_lo = 0; w_hi = wt.W;
    } else {
        const uint32_t sw = set / sp.groups, g = set % sp.groups;
        b = sw / wt.W;
        w_lo = (int)(sw % wt.W); w_hi = w_lo + 1;
        i_lo = (uint32_t)(((uint64_t)g * sp.n + sp.groups - 1) / sp.groups);
        i_hi = (uint32_t)(((uint64_t)(g + 1) * sp.n + sp.groups - 1) / sp.groups);
    }
    const uint32_t n_set = i_hi - i_lo, n_pad = (n_set + 7u) & ~7u, rows = (uint32_t)(w_hi - w_lo);
    uint16_t* dg = digits16 + (size_t)set * sp.digits_per_set;
    for (uint32_t j = tid; j < H; j += SORT2_BLOCK) bins[j] = 0;
    for (uint32_t k = tid; k <= SORT2_MAX_CHUNKS; k += SORT2_BLOCK) { cs[k] = 0xffffffffu; jb[k] = H; }
    __syncthreads();
    // pass 1: histogram + the digits as u16 rows [window][scalar], zero-padded to n_pad
    for (uint32_t ii = tid; ii < n_pad; ii += SORT2_BLOCK) {
        if (ii < n_set) {
            uint32_t k[9];
            load_scalar_mod_r(scalars, (size_t)b * sp.n + i_lo + ii, k);
            const bool neg = sp.fold ? scalar_fold_sign(k) : false;
            if (wt.odd == 2) {
                for_each_wnaf_digit<true>(k, wt, [&](int j, uint32_t o, int32_t d) {
                    uint32_t enc = WNAF_EMPTY16;
                    if (d != 0) {
                        const uint32_t bin = ((uint32_t)(d < 0 ? -d : d) - 1u) >> 1;
                        atomicAdd(&bins[bin], 1u);
                        enc = bin | (o << 11) | ((d < 0) != neg ? 0x8000u : 0u);
                    }
                    dg[(size_t)j * n_pad + ii] = (uint16_t)enc;
                });
            } else
            for_each_digit<true>(k, wt, w_lo, w_hi, [&](int w, int32_t d) {
                const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                uint32_t row, enc = mag;
                if (d != 0) {
                    atomicAdd(&bins[digit_bin(wt, w, mag, row)], 1u);
                }
                dg[(size_t)(w - w_lo) * n_pad + ii] = (uint16_t)(enc | ((d < 0) != neg && d != 0 ? 0x8000u : 0u));
            });
        } else {
            for (uint32_t r = 0; r < rows; r++) dg[(size_t)r * n_pad + ii] = wt.odd == 2 ? (uint16_t)WNAF_EMPTY16 : (uint16_t)0;
        }
    }
    __syncthreads();
    // counts out; exclusive scan of the bins; bucket range and first position of every chunk
    const uint32_t per = (H + SORT2_BLOCK - 1) / SORT2_BLOCK, lo = tid * per < H ? tid * per : H, hi = lo + per < H ? lo + per : H;
    uint32_t local = 0;
    for (uint32_t j = lo; j < hi; j++) {
        uint32_t c = bins[j];
        counts[(size_t)set * H + j] = c;
        local += c;
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<SORT2_BLOCK>(local, smem, total);
    const uint32_t base = set * sp.capacity;
    const uint32_t K = (total + SORT2_CHUNK - 1) / SORT2_CHUNK;            // <= SORT2_MAX_CHUNKS (host: capacity / SORT2_CHUNK)
    for (uint32_t j = lo; j < hi; j++) {
        uint32_t c = bins[j];
        bins[j] = run;                                  // becomes the placement cursor
        offsets[(size_t)set * H + j] = base + run;
        const uint32_t k = run / SORT2_CHUNK;           // chunk in which this bucket's segment starts (k <= K; k == K only for
        atomicMin(&jb[k], j);                           //  empty buckets at the very end)
        atomicMin(&cs[k], run);
        run += c;
    }
    __syncthreads();
    if (tid == 0) {
        // chunks in which no bucket starts (a bucket longer than a chunk) inherit the next chunk's bounds
        cs[K] = total; jb[K] = H;
        for (int k = (int)K - 1; k >= 0; k--) {
            if (cs[k] == 0xffffffffu) cs[k] = cs[k + 1];
            if (jb[k] > jb[k + 1]) jb[k] = jb[k + 1];
        }
    }
    __syncthreads();
    // pass 2: chunk by chunk
    const uint32_t tbl = sp.single ? sp.tbl_offset : 0u;
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t j_lo = jb[k], j_hi = jb[k + 1], p_lo = cs[k], p_hi = cs[k + 1], c0 = k * SORT2_CHUNK;
        if (j_lo < j_hi) {
            // the rows x (n_pad / 8) sixteen-byte vectors of the set as one index space (no idle lanes at row ends)
            const uint32_t nv = n_pad / 8, nvec = rows * nv;
            const uint4* vecs = reinterpret_cast<const uint4*>(dg);
            uint32_t r = 0, v = tid;
            for (uint32_t idx = tid; idx < nvec; idx += SORT2_BLOCK, v += SORT2_BLOCK) {
                while (v >= nv) { v -= nv; r++; }
                const uint32_t row_entry = (sp.single ? (uint32_t)wt.row[w_lo + (int)r] * sp.tbl_stride : 0u) + tbl + i_lo;
                {
                    const uint4 q = vecs[idx];
                    const uint32_t words[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const uint32_t dd = (words[t >> 1] >> (16 * (t & 1))) & 0xffffu;
                        // dd == 0 (windows) / offset field 15 (non-adjacent form) -> 0xffffffff: outside every range
                        const uint32_t j = wt.odd == 2 ? (((dd >> 11) & 15u) == 15u ? 0xffffffffu : (dd & 0x7ffu)) : (dd & 0x7fffu) - 1u;
                        if (j >= j_lo && j < j_hi) {
                            uint32_t up = 0;
                            if (wt.odd == 2) {
                                up = ((dd >> 11) & 15u) * sp.tbl_stride;      // the digit's offset in its slot: row[slot] + offset
                            }
                            const uint32_t pos = atomicAdd(&bins[j], 1u);
                            const uint32_t entry = (row_entry + up + v * 8 + (uint32_t)t) | ((dd & 0x8000u) << 16);
                            const uint32_t rel = pos - c0;
                            if (rel < SORT2_CAP) stage[rel] = entry;
                            else sorted[base + pos] = entry;
                        }
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t r_lo = p_lo - c0, r_hi = (p_hi - c0) < SORT2_CAP ? (p_hi - c0) : SORT2_CAP;
        for (uint32_t rel = r_lo + tid; rel < r_hi; rel += SORT2_BLOCK) sorted[base + c0 + rel] = stage[rel];
        __syncthreads();
    }
}

// ---- 3a''. two-pass sort for a FEW HUGE bucket sets (one 2^16 .. 2^22-point MSM over a window table: at 2^20 one set of 2^19
// buckets and 13.6 M entries with 20-bit windows — too many for one workgroup's LDS, and the digits -> global-atomic histogram ->
// scan -> global-atomic scatter chain costs 1.5 ms of such an MSM).
//   pass A (k_g1_part_scatter): a workgroup takes a tile of scalars of one set (2048 at <= 16 windows), computes their digits and
//     splits the entries by the top bits of the bucket index into P <= 1024 partitions: counters in LDS (per wave while P <= 64: a
//     handful of partitions would be 1024 lanes on a few addresses), the tile's entries grouped by partition in an LDS stage, ONE
//     global atomic per (tile, partition) to reserve room in that partition's stream, then the stage is copied out in runs.
//     A stream record is 64 bits: the finished `sorted` entry (table index | sign << 31) and the bucket within the partition.
//   pass B (k_g1_part_sort): one workgroup per (set, partition) sorts its ~30 k records by bucket entirely in LDS (histogram,
//     scan, placement into a stage, coalesced copy-out — k_g1_sort_sets_staged's second half on a stream instead of digit
//     rows), and writes counts / offsets / sorted in the layout the accumulate kernel reads.
// Room per stream: first try, cap_part records each (4x an even share) at p * cap_part; pass A counts every entry in part_fill
// whether or not it fitted, so when a stream was overfilled (few distinct scalars) the host knows the exact sizes and runs pass A
// again with exact stream offsets (exact_base) — no scalar distribution falls back to global atomics unless one partition holds
// more than 64 stage chunks (2.2 M entries).  Wave-uniform bins (all-equal scalars) cost one LDS atomic per wave, not 64.
constexpr int PART_BLOCK = 1024;
constexpr uint32_t PART_TILE_ENTRIES = 32768, PART_MAX_P = 1024, PART_MAX_HP = 1024, PART_STAGE = 36864, PART_SLACK = 2048, PART_MAX_CHUNKS = 64;

struct PartParams {
    uint32_t n, H, groups, tiles_per_set, P, pshift;      // pshift = log2(H / P): bucket >> pshift = partition
    uint32_t tile;                                        // scalars per pass-A workgroup: tile * W <= PART_TILE_ENTRIES, tile <= 2048
    uint32_t cap_part;                                    // records reserved per partition stream (first try)
    uint32_t capacity;                                    // entries reserved per set in `sorted`
    uint32_t tbl_stride, tbl_offset;
    uint8_t row[32];                                      // table row of window w (WindowTable::row; W <= 32 here)
};

DR_DEV void part_set_range(const PartParams& pp, uint32_t set, uint32_t& b, uint32_t& i_lo, uint32_t& i_hi) {
    b = set / pp.groups;
    const uint32_t g = set % pp.groups;
    i_lo = (uint32_t)(((uint64_t)g * pp.n + pp.groups - 1) / pp.groups);
    i_hi = (uint32_t)(((uint64_t)(g + 1) * pp.n + pp.groups - 1) / pp.groups);
}

// one more entry in bins[j] for every active lane; returns the lane's position.  Lanes of a wave that name the same bin are peeled off
// together, up to four bins per call (few distinct scalars: a million entries in a handful of buckets): one lane adds the group's
// count and the others take their rank — 64 serialised LDS atomics on one address become one.  Whatever is left (random scalars: the
// first peel finds its lane alone and the loop stops) takes a plain atomic.
DR_DEV uint32_t lds_count_aggregated(uint32_t* bins, uint32_t j) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long todo = __ballot(1);
    uint32_t pos = 0;
    bool done = false;
#pragma unroll 1
    for (int round = 0; round < 4 && todo; round++) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t j0 = (uint32_t)__shfl((int)j, leader, 64);
        const unsigned long long same = __ballot(!done && j == j0);
        if (__popcll(same) < 2 && round == 0) break;              // spread-out bins: no point in peeling
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&bins[j0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (!done && j == j0) { pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); done = true; }
        todo &= ~same;
    }
    if (!done) pos = atomicAdd(&bins[j], 1u);
    return pos;
}

__global__ __launch_bounds__(PART_BLOCK) void k_g1_part_scatter(const uint32_t* __restrict__ scalars, WindowTable wt, PartParams pp,
                                                                uint32_t* __restrict__ part_fill, const uint32_t* __restrict__ exact_base,
                                                                uint2* __restrict__ streams) {
    constexpr int WAVES = PART_BLOCK / 64;
    __shared__ uint32_t stage[PART_TILE_ENTRIES];
    __shared__ uint32_t cnt[PART_MAX_P];                          // P <= 64: [wave][P]; else one counter per partition
    __shared__ uint32_t pbase[PART_MAX_P + 1], gbase[PART_MAX_P];
    __shared__ uint32_t smem[PART_BLOCK / 64];
    const uint32_t set = blockIdx.x / pp.tiles_per_set, tile = blockIdx.x % pp.tiles_per_set, tid = threadIdx.x, wave = tid >> 6;
    const bool per_wave = pp.P * WAVES <= PART_MAX_P;
    uint32_t b, i_lo, i_hi;
    part_set_range(pp, set, b, i_lo, i_hi);
    const uint32_t t_lo = i_lo + tile * pp.tile < i_hi ? i_lo + tile * pp.tile : i_hi, t_hi = t_lo + pp.tile < i_hi ? t_lo + pp.tile : i_hi;
    for (uint32_t k = tid; k < PART_MAX_P; k += PART_BLOCK) cnt[k] = 0;
    __syncthreads();
    const uint32_t lowmask = (1u << pp.pshift) - 1u, woff = per_wave ? wave * pp.P : 0u;
    // count per partition
    for (uint32_t i = t_lo + tid; i < t_hi; i += PART_BLOCK) {
        uint32_t k[9];
        load_scalar_mod_r(scalars, (size_t)b * pp.n + i, k);
        for_each_digit(k, wt, 0, wt.W, [&](int, int32_t d) {
            const uint32_t j = (uint32_t)(d < 0 ? -d : d) - 1u;
            atomicAdd(&cnt[woff + (j >> pp.pshift)], 1u);
        });
    }
    __syncthreads();
    // stage layout: partition-major (wave-minor); cnt becomes the cursor of each run
    {
        uint32_t total = 0;
        if (tid < pp.P) {
            if (per_wave) for (int w = 0; w < WAVES; w++) total += cnt[w * pp.P + tid];
            else total = cnt[tid];
        }
        uint32_t grand;
        const uint32_t run0 = block_exclusive_scan<PART_BLOCK>(total, smem, grand);
        if (tid < pp.P) {
            pbase[tid] = run0;
            gbase[tid] = total ? atomicAdd(&part_fill[(size_t)set * pp.P + tid], total) : 0u;
            if (per_wave) {
                uint32_t run = run0;
                for (int w = 0; w < WAVES; w++) { const uint32_t c = cnt[w * pp.P + tid]; cnt[w * pp.P + tid] = run; run += c; }
            } else {
                cnt[tid] = run0;
            }
        }
        if (tid == 0) pbase[pp.P] = grand;
    }
    __syncthreads();
    for (uint32_t i = t_lo + tid; i < t_hi; i += PART_BLOCK) {
        uint32_t k[9];
        load_scalar_mod_r(scalars, (size_t)b * pp.n + i, k);
        const uint32_t local = i - t_lo;                                        // < 2048
        for_each_digit(k, wt, 0, wt.W, [&](int w, int32_t d) {
            const uint32_t j = (uint32_t)(d < 0 ? -d : d) - 1u;
            const uint32_t pos = atomicAdd(&cnt[woff + (j >> pp.pshift)], 1u);
            stage[pos] = (d < 0 ? 0x80000000u : 0u) | ((j & lowmask) << 16) | ((uint32_t)w << 11) | local;
        });
    }
    __syncthreads();
    // copy out: record idx of the stage belongs to the partition whose [pbase[p], pbase[p+1]) holds it
    const uint32_t total = pbase[pp.P], entry0 = pp.tbl_offset + t_lo;
    for (uint32_t idx = tid; idx < total; idx += PART_BLOCK) {
        uint32_t lo = 0, hi = pp.P;                                             // largest p with pbase[p] <= idx
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pbase[mid] <= idx) lo = mid; else hi = mid; }
        const uint32_t at = gbase[lo] + (idx - pbase[lo]), e = stage[idx];
        const size_t blk = (size_t)set * pp.P + lo;
        const uint32_t entry = ((uint32_t)pp.row[(e >> 11) & 31u] * pp.tbl_stride + entry0 + (e & 0x7ffu)) | (e & 0x80000000u);
        if (exact_base) streams[(size_t)exact_base[blk] + at] = make_uint2(entry, (e >> 16) & 0x3ffu);
        else if (at < pp.cap_part) streams[blk * pp.cap_part + at] = make_uint2(entry, (e >> 16) & 0x3ffu);   // an overfull stream: the host sees its fill count
    }
}

__global__ __launch_bounds__(PART_BLOCK) void k_g1_part_sort(const uint2* __restrict__ streams, const uint32_t* __restrict__ part_fill,
                                                             const uint32_t* __restrict__ exact_base, PartParams pp, uint32_t* __restrict__ overflow,
                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
    constexpr uint32_t CHUNK = PART_STAGE - PART_SLACK, MAX_CHUNKS = PART_MAX_CHUNKS;
    __shared__ uint32_t bins[PART_MAX_HP];
    __shared__ uint32_t stage[PART_STAGE];
    __shared__ uint32_t cs[MAX_CHUNKS + 1], jb[MAX_CHUNKS + 1];
    __shared__ uint32_t smem[PART_BLOCK / 64];
    const uint32_t blk = blockIdx.x, set = blk / pp.P, part = blk % pp.P, tid = threadIdx.x, HP = 1u << pp.pshift;
    const uint32_t n_ent = part_fill[blk];
    const uint2* src = streams + (exact_base ? (size_t)exact_base[blk] : (size_t)blk * pp.cap_part);
    if ((!exact_base && n_ent > pp.cap_part) || n_ent > MAX_CHUNKS * CHUNK) {
        // the first try's stream was overfilled (or no stage plan covers this partition): its buckets read as empty, the flag tells the
        // host to run the sort again with exact stream offsets — the host does not wait for the fill counters before launching this kernel
        const size_t bucket0 = (size_t)set * pp.H + (size_t)part * HP;
        for (uint32_t j = tid; j < HP; j += PART_BLOCK) { counts[bucket0 + j] = 0; offsets[bucket0 + j] = 0; }
        if (tid == 0) atomicOr(overflow, 1u);
        return;
    }
    for (uint32_t j = tid; j < HP; j += PART_BLOCK) bins[j] = 0;
    for (uint32_t k = tid; k <= MAX_CHUNKS; k += PART_BLOCK) { cs[k] = 0xffffffffu; jb[k] = HP; }
    // entries of the set's earlier partitions: where this partition's segment of `sorted` begins
    uint32_t part_base;
    {
        uint32_t v = 0;
        for (uint32_t q = tid; q < part; q += PART_BLOCK) v += part_fill[(size_t)set * pp.P + q];
        (void)block_exclusive_scan<PART_BLOCK>(v, smem, part_base);          // (ends with a barrier: bins / cs / jb are initialised)
    }
    // (four records in flight per lane: a partition of a million records — equal scalars — is one workgroup's work)
    for (uint32_t i0 = tid; i0 < n_ent; i0 += 4 * PART_BLOCK) {
        uint32_t jj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) jj[u] = i0 + u * PART_BLOCK < n_ent ? src[i0 + u * PART_BLOCK].y : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (jj[u] != 0xffffffffu) (void)lds_count_aggregated(bins, jj[u]);
    }
    __syncthreads();
    const uint32_t per = (HP + PART_BLOCK - 1) / PART_BLOCK, lo = tid * per < HP ? tid * per : HP, hi = lo + per < HP ? lo + per : HP;
    uint32_t local = 0;
    const size_t bucket0 = (size_t)set * pp.H + (size_t)part * HP;
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t c = bins[j];
        counts[bucket0 + j] = c;
        local += c;
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<PART_BLOCK>(local, smem, total);
    const uint32_t base = set * pp.capacity + part_base;
    const uint32_t K = (total + CHUNK - 1) / CHUNK;                 // 1 for evenly spread scalars; <= MAX_CHUNKS (host)
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t c = bins[j];
        bins[j] = run;
        offsets[bucket0 + j] = base + run;
        const uint32_t k = run / CHUNK;
        atomicMin(&jb[k], j);
        atomicMin(&cs[k], run);
        run += c;
    }
    __syncthreads();
    if (tid == 0) {
        cs[K] = total; jb[K] = HP;
        for (int k = (int)K - 1; k >= 0; k--) {
            if (cs[k] == 0xffffffffu) cs[k] = cs[k + 1];
            if (jb[k] > jb[k + 1]) jb[k] = jb[k + 1];
        }
    }
    __syncthreads();
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t j_lo = jb[k], j_hi = jb[k + 1], p_lo = cs[k], p_hi = cs[k + 1], c0 = k * CHUNK;
        if (j_lo < j_hi) {
            for (uint32_t i0 = tid; i0 < n_ent; i0 += 4 * PART_BLOCK) {
                uint2 e[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = i0 + u * PART_BLOCK < n_ent ? src[i0 + u * PART_BLOCK] : make_uint2(0u, 0xffffffffu);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (e[u].y >= j_lo && e[u].y < j_hi) {
                        const uint32_t pos = lds_count_aggregated(bins, e[u].y);
                        const uint32_t rel = pos - c0;
                        if (rel < PART_STAGE) stage[rel] = e[u].x;
                        else sorted[base + pos] = e[u].x;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t r_lo = p_lo - c0, r_hi = (p_hi - c0) < PART_STAGE ? (p_hi - c0) : PART_STAGE;
        for (uint32_t rel = r_lo + tid; rel < r_hi; rel += PART_BLOCK) sorted[base + c0 + rel] = stage[rel];
        __syncthreads();
    }
}

// ---- 3b. order buckets by size (descending) so that the 64 lanes of a wave walk chains of nearly equal length.
// A wave otherwise waits for its longest bucket: with ~20 points per bucket (Poisson) a third of the lane-cycles idle.
// Counting sort over 256 size classes (sizes >= 696 share the first class, see size_class): per-workgroup histograms in LDS, a scan
// over (class-major, workgroup-minor) cells, then each workgroup places its buckets.  Order inside a class is free.
constexpr int SZ_BLOCK = 256, SZ_ITEMS = 8, SZ_TILE = SZ_BLOCK * SZ_ITEMS, SZ_CLASSES = 256;
// class 0 = largest.  One class per size up to 191 entries, then steps of 8 (lanes of a wave differ by < 4 % there) up to 695: lists
// of G1_LONG_BUCKET entries or more are shared by 16 or 64 lanes (k_g1_accumulate_long), similar lengths side by side.
DR_DEV uint32_t size_class(uint32_t count) {
    const uint32_t idx = count < 192u ? count : 192u + (((count - 192u) >> 3) > 63u ? 63u : ((count - 192u) >> 3));
    return 255u - idx;
}

__global__ __launch_bounds__(SZ_BLOCK) void k_size_hist(const uint32_t* __restrict__ counts, size_t nbuckets, uint32_t nblocks,
                                                        uint32_t* __restrict__ cells /* [SZ_CLASSES][nblocks] */) {
    __shared__ uint32_t h[SZ_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * SZ_TILE;
#pragma unroll
    for (int j = 0; j < SZ_ITEMS; j++) {
        size_t b = base + (size_t)j * SZ_BLOCK + threadIdx.x;
        if (b < nbuckets) atomicAdd(&h[size_class(counts[b])], 1u);
    }
    __syncthreads();
    cells[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(SZ_BLOCK) void k_size_place(const uint32_t* __restrict__ counts, size_t nbuckets, uint32_t nblocks,
                                                         const uint32_t* __restrict__ cell_offsets, uint32_t* __restrict__ perm) {
    __shared__ uint32_t cur[SZ_CLASSES];
    cur[threadIdx.x] = cell_offsets[(size_t)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    size_t base = (size_t)blockIdx.x * SZ_TILE;
#pragma unroll
    for (int j = 0; j < SZ_ITEMS; j++) {
        size_t b = base + (size_t)j * SZ_BLOCK + threadIdx.x;
        if (b < nbuckets) perm[atomicAdd(&cur[size_class(counts[b])], 1u)] = (uint32_t)b;
    }
}

// ---- 4. bucket accumulation: one lane per bucket walks its segment with mixed additions.  DOMINANT KERNEL.
// Algorithmic traffic: 96 B base + 32 B scalar per (base,scalar) pair; the gather of bases is the
// only large stream, the segment lists are 4 B per entry.
// Long lists are not walked by one lane: skewed scalars — a 0/1 column, many equal values — put thousands of points into one bucket,
// and one lane adding them one after the other would be the kernel's whole run time (6145 equal scalars: 28 ms in one lane, 0.5 ms
// in a wave); odd-multiple buckets put a few hundred into the lowest buckets of EVERY set.  Lists of G1_LONG_BUCKET (256) entries
// or more — the size ordering puts them first in `perm` — are left to k_g1_accumulate_long: 16 lanes per bucket (four buckets per
// wave) up to G1_HEAVY_BUCKET entries; longer ones are cut into segments, a wave each (k_g1_accumulate_heavy).  The prover's dense MSMs over window rows (~66 points per bucket,
// Poisson) never get there.
constexpr uint32_t G1_LONG_BUCKET = 512;                  // (upper bound of the per-launch limit, k_size_pick)
// The list length from which a launch hands its lists to k_g1_accumulate_long.  One lane adds an entry per ~17 us (two waves share a
// SIMD); the whole launch needs total / 65536 lanes x 8.4 us when every lane is busy.  A list may take half of that: limit = total
// entries / 265 k, within [64, G1_LONG_BUCKET], at a size-class boundary.  The dense launches (126 - 168 M entries) get 256 and, with
// the twins, send nothing to the long-list kernel; the summation-by-parts launch (22 M entries, but ~130 "+-1" entries in one bucket of
// every bit column) gets ~85 — it used to last as long as one lane needs for its longest list.  From the size-class histogram (cells
// after their exclusive scan), one block.
__global__ __launch_bounds__(256) void k_size_pick(const uint32_t* __restrict__ cell_offsets, uint32_t nblocks, size_t nbuckets, uint32_t* __restrict__ pick) {
    __shared__ unsigned long long tot[256];
    const uint32_t c = threadIdx.x, idx = 255u - c;
    const uint32_t lo = cell_offsets[(size_t)c * nblocks], hi = c == 255u ? (uint32_t)nbuckets : cell_offsets[(size_t)(c + 1) * nblocks];
    const uint32_t n_c = hi - lo, size = idx < 192u ? idx : 192u + 8u * (idx - 192u) + 4u;
    tot[c] = (unsigned long long)n_c * size;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (c < s) tot[c] += tot[c + s];
        __syncthreads();
    }
    if (c == 0) {
        // ... and never below twice the launch's mean list (+ 16): a launch of ~262 k buckets — the plan's target for a batch of 64 .. 512
        // proofs over window rows — has total / 265 k = its MEAN length, and half of its lists went through the 16-lane kernel with its
        // four-addition shuffle tree (batch 128: the walk at 3.3 G additions/s)
        const unsigned long long mean2 = nbuckets ? 2ull * tot[0] / (unsigned long long)nbuckets + 16ull : 0ull;
        const unsigned long long share = tot[0] / 265000ull, want = share > mean2 ? share : mean2;
        // (never below 64: the 16-lane kernel ends with four dependent full additions, ~0.1 ms — as long as one lane needs for ~6 entries more —
        //  and a small launch, a single 2^16-point MSM, is a latency chain that would only get longer)
        const uint32_t L = want < 64ull ? 64u : want > (unsigned long long)G1_LONG_BUCKET ? G1_LONG_BUCKET : (uint32_t)want;
        const uint32_t idx_l = L < 192u ? L : 192u + (L - 192u) / 8u;
        pick[0] = idx_l < 192u ? idx_l : 192u + 8u * (idx_l - 192u);       // first length of that class
        pick[1] = idx_l;
    }
}
constexpr uint32_t G1_HEAVY_BUCKET = 1024;

DR_DEV G1Xyzz g1_walk(const uint32_t* __restrict__ bases, uint32_t pt_words, const uint32_t* __restrict__ sorted, uint32_t beg, uint32_t len, uint32_t first,
                      uint32_t stride) {
    G1Xyzz acc = g1_inf();
#pragma unroll 1
    for (uint32_t p = first; p < len; p += stride) {
        uint32_t e = sorted[beg + p];
        G1Affine q = load_affine_at(bases, e & 0x7fffffffu, pt_words);
        q = g1_neg_affine(q, (e >> 31) != 0);
        acc = g1_madd(acc, q);
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_g1_accumulate(const uint32_t* __restrict__ bases, uint32_t pt_words /* 24, or 32: a table with one point per line */,
                                                       const uint32_t* __restrict__ sorted,
                                                       const uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ perm /* size-ordered bucket ids */,
                                                       const uint32_t* __restrict__ pick /* k_size_pick */,
                                                       uint32_t* __restrict__ buckets, size_t nbuckets) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbuckets) return;
    const size_t b = perm[t];
    const uint32_t len = counts[b];
    if (len >= pick[0]) return;
    store_xyzz(buckets, b, g1_walk(bases, pt_words, sorted, offsets[b], len, 0, 1));
}

template <int WIDTH>
DR_DEV G1Xyzz xyzz_shfl_down(const G1Xyzz& p, unsigned delta) {
    G1Xyzz o;
#pragma unroll
    for (int t = 0; t < L28; t++) {
        o.x.l[t] = __shfl_down(p.x.l[t], delta, WIDTH);
        o.y.l[t] = __shfl_down(p.y.l[t], delta, WIDTH);
        o.zz.l[t] = __shfl_down(p.zz.l[t], delta, WIDTH);
        o.zzz.l[t] = __shfl_down(p.zzz.l[t], delta, WIDTH);
    }
    o.inf = __shfl_down(p.inf, delta, WIDTH);
    return o;
}
// LANES (16 or 64) lanes per long bucket: perm[0 .. n_long), n_long = where the size classes below G1_LONG_BUCKET begin in the
// exclusive scan of the [class][block] cells.  The lanes stride over the list, a shuffle tree folds their partial sums.  The
// Lists of [the launch's limit, G1_HEAVY_BUCKET) entries, four per wave; longer ones: k_g1_accumulate_heavy.
// Grid-stride, so a fixed small grid serves any number of them; returns at once when there are none.
template <int LANES>
__global__ __launch_bounds__(64) void k_g1_accumulate_long(const uint32_t* __restrict__ bases, uint32_t pt_words, const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cell_offsets,
                                                           uint32_t nblocks, const uint32_t* __restrict__ pick /* k_size_pick */, uint32_t* __restrict__ buckets) {
    constexpr uint32_t PER_WAVE = 64 / LANES;
    // class index >= pick[1]  <=>  class <= 255 - pick[1]; the next class starts at cell (256 - pick[1]) * nblocks
    const uint32_t long_from = pick[0];
    const uint32_t n_long = cell_offsets[(size_t)(256u - pick[1]) * nblocks];
    const uint32_t sub = threadIdx.x / LANES, lane = threadIdx.x % LANES;
#pragma unroll 1
    for (uint32_t t0 = blockIdx.x * PER_WAVE; t0 < n_long; t0 += gridDim.x * PER_WAVE) {
        const uint32_t t = t0 + sub;
        const bool have = t < n_long;
        const size_t b = have ? perm[t] : 0;
        const uint32_t len = have ? counts[b] : 0;
        const bool mine = have && len >= long_from && len < G1_HEAVY_BUCKET;
        if (__ballot(mine) == 0) continue;
        G1Xyzz acc = g1_walk(bases, pt_words, sorted, have ? offsets[b] : 0u, mine ? len : 0u, lane, LANES);
#pragma unroll 1
        for (unsigned d = LANES / 2; d >= 1; d >>= 1) acc = g1_add(acc, xyzz_shfl_down<LANES>(acc, d));
        if (mine && lane == 0) store_xyzz(buckets, b, acc);
    }
}

// Lists of G1_HEAVY_BUCKET entries or more — equal scalars, 0/1 columns, r - 1 everywhere: a window's whole digit row in one bucket,
// a million entries at 2^20 — are cut into segments (heavy_segment_size), one wave per segment
// (k_g1_accumulate_heavy), and the segment sums of a bucket are folded by one wave (k_g1_heavy_fold): no wave serialises more than
// 64 additions per lane, whatever the scalars.  Both kernels find their work by the same scan: the buckets of the largest size class
// (>= 696 entries) lead `perm`; 64 of them at a time, a wave prefix sum over their segment counts numbers the segments, and segment s
// belongs to block s mod gridDim.  Fixed small grids; both return at once when the class is empty.
constexpr uint32_t G1_HEAVY_SLOTS = 2048;                 // grid of k_g1_accumulate_heavy: two waves per SIMD
// Segment length of a launch: the heavy lists' entries spread over the walk's 2048 waves — total / (2048 - heavy buckets), every
// bucket's last segment being partial —, a multiple of 64, at least 1024 (a segment ends with a six-addition shuffle tree); 4096 when
// there are too many heavy buckets for that.  Both kernels compute it the same way (one pass over the leading size class).
DR_DEV uint32_t heavy_segment_size(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ perm, uint32_t n_class0) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long total = 0;
    uint32_t n_heavy = 0;
#pragma unroll 1
    for (uint32_t t0 = 0; t0 < n_class0; t0 += 64) {
        const uint32_t t = t0 + lane;
        const uint32_t len = t < n_class0 ? counts[perm[t]] : 0u;
        const bool heavy = len >= G1_HEAVY_BUCKET;
        unsigned long long v = heavy ? len : 0u;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += (unsigned long long)__shfl_xor((long long)v, sft, 64);
        total += v;
        n_heavy += (uint32_t)__popcll(__ballot(heavy));
    }
    if (n_heavy == 0 || n_heavy > G1_HEAVY_SLOTS / 2) return 4096u;
    const unsigned long long per = (total + (G1_HEAVY_SLOTS - n_heavy) - 1) / (G1_HEAVY_SLOTS - n_heavy);
    const unsigned long long seg = (per + 63ull) & ~63ull;
    return seg < 1024ull ? 1024u : seg > 0x40000000ull ? 0x40000000u : (uint32_t)seg;
}
template <class F>
DR_DEV void for_each_heavy_bucket(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ perm, uint32_t n_class0, uint32_t seg, F&& f) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t seg_base = 0;
#pragma unroll 1
    for (uint32_t t0 = 0; t0 < n_class0; t0 += 64) {
        const uint32_t t = t0 + lane;
        const uint32_t b = t < n_class0 ? perm[t] : 0u;
        const uint32_t len = t < n_class0 ? counts[b] : 0u;
        const uint32_t nseg = len >= G1_HEAVY_BUCKET ? (len + seg - 1) / seg : 0u;
        uint32_t incl = nseg;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, sft, 64);
            if ((int)lane >= sft) incl += y;
        }
        unsigned long long heavy = __ballot(nseg != 0);
        while (heavy) {
            const int src = __builtin_ctzll(heavy);
            heavy &= heavy - 1;
            f((uint32_t)__shfl((int)b, src, 64), (uint32_t)__shfl((int)len, src, 64), (uint32_t)__shfl((int)nseg, src, 64),
              seg_base + (uint32_t)__shfl((int)(incl - nseg), src, 64));
        }
        seg_base += (uint32_t)__shfl((int)incl, 63, 64);
    }
}

__global__ __launch_bounds__(64) void k_g1_accumulate_heavy(
    const uint32_t* __restrict__ bases, uint32_t pt_words, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cell_offsets, uint32_t nblocks,
    uint32_t* __restrict__ buckets, uint32_t* __restrict__ seg_sums) {
    const uint32_t n_class0 = cell_offsets[nblocks], lane = threadIdx.x;
    if (n_class0 == 0) return;
    const uint32_t seg = heavy_segment_size(counts, perm, n_class0);
    for_each_heavy_bucket(counts, perm, n_class0, seg, [&](uint32_t b, uint32_t len, uint32_t nseg, uint32_t first_seg) {
        // segments first_seg .. first_seg + nseg - 1; this block takes those congruent to its index
        uint32_t s = (blockIdx.x + gridDim.x - first_seg % gridDim.x) % gridDim.x;
#pragma unroll 1
        for (; s < nseg; s += gridDim.x) {
            const uint32_t lo = s * seg, cnt = len - lo < seg ? len - lo : seg;
            G1Xyzz acc = g1_walk(bases, pt_words, sorted, offsets[b] + lo, cnt, lane, 64);
#pragma unroll 1
            for (unsigned d = 32; d >= 1; d >>= 1) acc = g1_add(acc, xyzz_shfl_down<64>(acc, d));
            if (lane == 0) {
                if (nseg == 1) store_xyzz(buckets, b, acc);
                else store_xyzz(seg_sums, first_seg + s, acc);
            }
        }
    });
}

__global__ __launch_bounds__(64) void k_g1_heavy_fold(
    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cell_offsets, uint32_t nblocks,
    const uint32_t* __restrict__ seg_sums, uint32_t* __restrict__ buckets) {
    const uint32_t n_class0 = cell_offsets[nblocks], lane = threadIdx.x;
    if (n_class0 == 0) return;
    const uint32_t seg = heavy_segment_size(counts, perm, n_class0);
    for_each_heavy_bucket(counts, perm, n_class0, seg, [&](uint32_t b, uint32_t, uint32_t nseg, uint32_t first_seg) {
        if (nseg < 2 || first_seg % gridDim.x != blockIdx.x) return;
        G1Xyzz acc = g1_inf();
#pragma unroll 1
        for (uint32_t s = lane; s < nseg; s += 64) acc = g1_add(acc, load_xyzz(seg_sums, first_seg + s));
#pragma unroll 1
        for (unsigned d = 32; d >= 1; d >>= 1) acc = g1_add(acc, xyzz_shfl_down<64>(acc, d));
        if (lane == 0) store_xyzz(buckets, b, acc);
    });
}

// ---- 5. bucket reduction.  Window value = sum_j (j+1) * B_j.  Chunk [s, s+L): running sums give
//        sum_j (j-s+1) B_j and A = sum_j B_j; the chunk contributes that plus s*A (double-and-add, s < H).
// Register budget: a full XYZZ addition with both operands and the result live is ~185 VGPRs (k_g1_reduce_windows); a
// second accumulator on top of that spilled 97 dwords per lane to scratch.  So only ONE accumulator lives in registers
// (`run`, later `t`); the other one (`sum`, later A) is parked in LDS, word-major so that the 64 lanes of a wave hit 64
// different banks, and comes in as the second operand of the one inlined addition of each loop.
constexpr int RC_BLOCK = 128;
DR_DEV void park_put(uint32_t* park, const G1Xyzz& v) { put_raw(park + threadIdx.x, RC_BLOCK, v); }
DR_DEV G1Xyzz park_get(const uint32_t* park) { return get_raw(park + threadIdx.x, RC_BLOCK); }
__global__ __launch_bounds__(RC_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_reduce_chunks(const uint32_t* __restrict__ buckets, size_t windows,
                                                               uint32_t H, uint32_t L, uint32_t* __restrict__ partial) {
    __shared__ uint32_t park[XYZZ_RAW_WORDS * RC_BLOCK];         // 28.5 KB: one parked XYZZ value per lane, raw limbs (private slots: no barriers)
    const uint32_t T = H / L;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= windows * T) return;
    // few sets (one huge MSM in 16 index groups): neighbouring lanes take the SAME chunk of different sets, so a wave's chunk
    // offsets s share all but two bits and the double-and-add below executes an addition only where that shared bit is set
    // (~19 instead of ~25 operations for a 15-bit s); many sets: chunk-minor as before
    const bool set_minor = windows <= 32;
    size_t win = set_minor ? gid % windows : gid / T;
    uint32_t ch = (uint32_t)(set_minor ? gid / windows : gid % T), s = ch * L;
    gid = win * T + ch;                                          // index of this lane's result
    G1Xyzz run = g1_inf();
    park_put(park, run);                                         // sum = O
#pragma unroll 1
    for (uint32_t step = 0; step < 2 * L; step++) {              // run += B_j ; sum += run, j = L-1 .. 0
        const bool first = (step & 1) == 0;
        G1Xyzz b = first ? load_xyzz(buckets, win * H + s + (L - 1 - (step >> 1))) : park_get(park);
        G1Xyzz r = g1_add(run, b);
        if (first) run = r; else park_put(park, r);
    }
    if (s == 0 || run.inf) {
        store_xyzz(partial, gid, park_get(park));
        return;
    }
    // result = sum + s * run: double-and-add over the bits of s with A = run parked, then the running-sum result
    store_xyzz(partial, gid, park_get(park));
    park_put(park, run);
    G1Xyzz t = g1_inf();
    const int top = 31 - __clz(s);
#pragma unroll 1
    for (int step = 2 * top + 1; step >= -1; step--) {
        if (step >= 0 && (step & 1)) { t = g1_dbl(t); continue; }
        const bool last = step < 0;
        if (!last && !((s >> (step >> 1)) & 1)) continue;
        G1Xyzz b = last ? load_xyzz(partial, gid) : park_get(park);
        t = g1_add(t, b);
    }
    store_xyzz(partial, gid, t);
}

// one workgroup per window: lanes stride over the T chunk results, then an LDS tree folds the workgroup.
constexpr int RW_BLOCK = 128;
__global__ __launch_bounds__(RW_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_reduce_windows(const uint32_t* __restrict__ partial, uint32_t T,
                                                                uint32_t* __restrict__ winsum) {
    // scratch-free like the other reduction kernels: ONE inlined addition serves the strided pass over the T chunk results and
    // the log2(RW_BLOCK) tree levels (operand picked per step; an infinite operand leaves acc as it is)
    __shared__ uint32_t sm[RW_BLOCK * XYZZ_RAW_WORDS];
    size_t win = blockIdx.x;
    G1Xyzz acc = g1_inf();
    const int pre = (int)((T + RW_BLOCK - 1) / RW_BLOCK);
    int s = RW_BLOCK;
#pragma unroll 1
    for (int step = 0; step < pre + 7; step++) {
        G1Xyzz b = g1_inf();
        if (step < pre) {
            const uint32_t t = threadIdx.x + (uint32_t)step * RW_BLOCK;
            if (t < T) b = load_xyzz(partial, win * T + t);
        } else {
            put_raw(sm + threadIdx.x, RW_BLOCK, acc);
            __syncthreads();
            s >>= 1;
            if ((int)threadIdx.x < s) b = get_raw(sm + threadIdx.x + s, RW_BLOCK);
            __syncthreads();                    // partners are read before anyone overwrites its own slot in the next level
        }
        acc = g1_add(acc, b);
    }
    static_assert(RW_BLOCK == 128, "seven tree levels");
    if (threadIdx.x == 0) store_xyzz(winsum, win, acc);
}

// ---- 5b. bucket reduction without scalar multiplications, for MANY bucket sets (the batched prover: 10^7 buckets).
// With j = t L + i:  sum_j (j+1) B_j = L * sum_t (t+1) S_t - sum_t Q_t,  S_t = sum_i B_{t,i},  Q_t = sum_i (L-1-i) B_{t,i}:
// the same problem on the T = H/L chunk sums, minus a correction.  Each level costs 2 additions per entry (against
// 3.25 per bucket for the chunk + double-and-add form above) and shrinks the set by L; corrections are carried along as
// C (scaled by L^(level-1)), so the value of a set is  L^K * W_K - sum C_K  with W_K the direct running sum over the
// last <= 16 entries.  Latency is a few serial levels: only worth it when the first level alone fills the chip.
// Both kernels are scratch-free: ONE inlined addition per loop (operands muxed), the scaling doublings in a second
// loop with one inlined doubling.
// level 1 (94 % of the work): two accumulators only — q += run; run += B_i; q waits in LDS (see k_g1_reduce_chunks)
__global__ __launch_bounds__(RC_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_reduce_level1(const uint32_t* __restrict__ in_s, size_t sets,
                                                                                                        uint32_t n_in, uint32_t L,
                                                                                                        uint32_t* __restrict__ out_s,
                                                                                                        uint32_t* __restrict__ out_c) {
    __shared__ uint32_t park[XYZZ_RAW_WORDS * RC_BLOCK];
    const uint32_t T = n_in / L;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= sets * T) return;
    size_t set = gid / T;
    uint32_t s = (uint32_t)(gid % T) * L;
    G1Xyzz run = g1_inf();
    park_put(park, run);
#pragma unroll 1
    for (uint32_t step = 0; step < 2 * L; step++) {
        const bool first = (step & 1) == 0;
        G1Xyzz b = first ? park_get(park) : load_xyzz(in_s, set * n_in + s + (step >> 1));
        G1Xyzz r = g1_add(run, b);
        if (first) park_put(park, r); else run = r;
    }
    store_xyzz(out_s, gid, run);
    store_xyzz(out_c, gid, park_get(park));
}
// levels >= 2 (only when a set has more than 2048 buckets and there are many sets): entries S_i with corrections C_i
__global__ __launch_bounds__(128) void k_g1_reduce_level(const uint32_t* __restrict__ in_s, const uint32_t* __restrict__ in_c,
                                                         size_t sets, uint32_t n_in, uint32_t L, int level,
                                                         uint32_t* __restrict__ out_s, uint32_t* __restrict__ out_c) {
    const uint32_t T = n_in / L;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= sets * T) return;
    size_t set = gid / T;
    uint32_t s = (uint32_t)(gid % T) * L;
    G1Xyzz run = g1_inf(), q = g1_inf(), c = g1_inf();
    // per entry: q += run; run += S_i; c += C_i; after the loop one more step: c += L^(level-1) q
    const uint32_t steps = L * 3;
#pragma unroll 1
    for (uint32_t step = 0; step <= steps; step++) {
        if (step == steps) {
#pragma unroll 1
            for (int k = 0; k < 4 * (level - 1); k++) q = g1_dbl(q);
        }
        const uint32_t i = step / 3, ph = step == steps ? 3u : step % 3;
        const size_t idx = set * n_in + s + (i < L ? i : 0);
        G1Xyzz a, b;
        if (ph == 0) { a = q; b = run; }
        else if (ph == 1) { a = run; b = load_xyzz(in_s, idx); }
        else if (ph == 2) { a = c; b = load_xyzz(in_c, idx); }
        else { a = c; b = q; }
        G1Xyzz r = g1_add(a, b);
        if (ph == 0) q = r; else if (ph == 1) run = r; else c = r;
    }
    store_xyzz(out_s, gid, run);
    store_xyzz(out_c, gid, c);
}
// ---- 5c. a few lanes per bucket set finish the first level (the batched prover's MSMs: thousands of sets of T = H/16 <= 128
// first-level chunks).  With S_t the chunk sums and Q_t the chunk corrections of k_g1_reduce_level1,
//     value = 16 * sum_t (t+1) S_t - sum_t Q_t.
// Lane l of a set takes 4 consecutive chunks: a running sum gives R_l = sum S_t and w_l = sum (t - 4l + 1) S_t over its block, so
//     sum_t (t+1) S_t = sum_l w_l + 4 sum_l l R_l,    sum_l l R_l = sum_{k >= 1} X_k,   X_k = sum_{l >= k} R_l:
// a suffix scan of the R_l over the set's T/4 lanes (log2 steps), Y_l = 16 (w_l + 4 X_l [l >= 1]) - sum Q_t per lane, then a tree
// sum of the Y_l.  Per set: T/4 lanes x (8 + 2 log2(T/4) + 11) operations — 930 for T = 128, against the 128 x 17 of the
// chunk kernel's double-and-add tail plus the fold, and 2 additions per bucket at the first level instead of 3.
// Sets are packed into 64-lane workgroups; one inlined addition and one inlined doubling, operands muxed per step.
constexpr int RS_BLOCK = 64, RS_GROUP = 4;
// (no occupancy cap: three live accumulators need ~390 registers, and with T/4 lanes per set the launch is at most one wave
// per SIMD anyway)
// `odd`: bucket j holds the odd multiple 2j + 1 (WindowTable::odd): value = 2 * [sum_j (j + 1) B_j] - sum_j B_j — one more step, on
// the tree's result and the set total that the suffix scan left in lane 0.
__global__ __launch_bounds__(RS_BLOCK) void k_g1_reduce_set_scan(const uint32_t* __restrict__ in_s, const uint32_t* __restrict__ in_c, size_t sets,
                                                                 uint32_t T /* power of two, 8 .. 256 */, int odd, uint32_t* __restrict__ winsum) {
    __shared__ uint32_t sm[RS_BLOCK * XYZZ_RAW_WORDS];
    const uint32_t LS = T / RS_GROUP;                       // lanes per set (a power of two <= 64)
    const uint32_t per_block = RS_BLOCK / LS;
    const uint32_t l = threadIdx.x % LS;
    const size_t set = (size_t)blockIdx.x * per_block + threadIdx.x / LS;
    const bool live = set < sets;
    const size_t c0 = (live ? set : 0) * T + (size_t)l * RS_GROUP;      // first chunk of this lane
    int lg = 0;
    while ((1u << lg) < LS) lg++;
    // steps: 0..7 local running sums (run += S_t; w += run, t descending), then lg scan steps on R, one combine step
    // (v = 4 X [l >= 1] + w), one scaling step (v = 16 v), 4 correction steps (v -= Q_t), lg tree steps
    G1Xyzz run = g1_inf(), w = g1_inf(), v = g1_inf();
    const int n_local = 2 * RS_GROUP, s_scan = n_local, s_comb = s_scan + lg, s_corr = s_comb + 1, s_tree = s_corr + RS_GROUP, s_end = s_tree + lg;
#pragma unroll 1
    for (int step = 0; step < s_end + (odd ? 1 : 0); step++) {
        G1Xyzz a, b = g1_inf();
        int dst;                                            // 0: run, 1: w, 2: v
        if (step == s_end) {                                // odd multiples: 2 v - (set total)
            a = g1_dbl(v); b = run; b.y = neg(b.y); dst = 2;
        } else if (step < n_local) {
            if ((step & 1) == 0) { a = run; if (live) b = load_xyzz(in_s, c0 + (RS_GROUP - 1 - (step >> 1))); dst = 0; }
            else { a = w; b = run; dst = 1; }
        } else if (step < s_comb) {                         // suffix scan of the block sums: run_l += run_{l + 2^k}
            const uint32_t dist = 1u << (step - s_scan);
            put_raw(sm + threadIdx.x, RS_BLOCK, run);
            __syncthreads();
            if (l + dist < LS) b = get_raw(sm + threadIdx.x + dist, RS_BLOCK);
            __syncthreads();
            a = run; dst = 0;
        } else if (step == s_comb) {                        // v = 4 X_l (lanes >= 1) + w_l, then x 16
            if (l >= 1) { v = g1_dbl(run); v = g1_dbl(v); }
            a = v; b = w; dst = 2;
        } else if (step < s_tree) {                         // corrections; the x 16 comes first
            if (step == s_corr) {
#pragma unroll 1
                for (int k = 0; k < 4; k++) v = g1_dbl(v);
            }
            if (live) { b = load_xyzz(in_c, c0 + (step - s_corr)); b.y = neg(b.y); }
            a = v; dst = 2;
        } else {                                            // tree over the set's lanes
            const uint32_t dist = LS >> (step - s_tree + 1);
            put_raw(sm + threadIdx.x, RS_BLOCK, v);
            __syncthreads();
            if (l < dist) b = get_raw(sm + threadIdx.x + dist, RS_BLOCK);
            __syncthreads();
            a = v; dst = 2;
        }
        const G1Xyzz r = g1_add(a, b);
        if (dst == 0) run = r; else if (dst == 1) w = r; else v = r;
    }
    if (live && l == 0) store_xyzz(winsum, set, v);
}

// ---- 5d. one huge bucket set per index group (a single 2^18 .. 2^22-point MSM over a 16-bit window table: 16 sets of 32768
// buckets).  The chunk kernel's per-lane chain there is 32 running-sum additions + a 15-bit double-and-add (~22 operations), then
// a fold of 2048 chunk results per set: 0.80 + 0.22 ms at 2^20, pure latency on half a wave per SIMD.  Here a workgroup of 256 lanes
// takes 2048 consecutive buckets of one set, 8 per lane:
//     lane l:   R_l = sum_i B_i,  w_l = sum_i (i + 1) B_i          (16 additions, buckets descending; 8 buckets per lane here)
//     bucket j = 2048 g + 8 l + i of the set has weight j + 1:  sum = sum_l w_l + 8 sum_l l R_l + 2048 g sum_l R_l
//     sum_l l R_l = sum_{k >= 1} X_k,  X_k = sum_{l >= k} R_l:  a suffix scan of R over the 256 lanes (8 log steps through LDS),
//     Y_l = w_l + 8 X_l [l >= 1] (three doublings, one addition), a tree sum of Y (8 steps).
// Per workgroup out: V = sum_l Y_l and S = X_0 = sum_l R_l; the host adds V + 2048 g S over the H / 2048 workgroups of a set (a
// running sum of a few dozen points on the worker threads) — 33 additions + 3 doublings deep where the chunk kernel and its fold
// are ~66.  One inlined addition, operands muxed per step, like k_g1_reduce_set_scan.
// Buckets per lane: 8 fills one wave per SIMD at 2^19 buckets (the kernel's three accumulators take ~390 registers: one resident wave);
// a smaller MSM has fewer buckets than the chip has such lanes and takes 4, 2 or 1 per lane — the chain is 2 PER_LANE + 17 additions
// deep (33 at 8, 19 at 1), and the chain is all this launch costs (2^16 pairs: 0.54 -> ~0.3 ms).
constexpr int WS_BLOCK = 256;
template <int WS_PER_LANE>
__global__ __launch_bounds__(WS_BLOCK) void k_g1_reduce_wg_scan(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ out /* [workgroup][2]: V, S */) {
    __shared__ uint32_t sm[WS_BLOCK * XYZZ_RAW_WORDS];
    const uint32_t l = threadIdx.x;
    const size_t b0 = ((size_t)blockIdx.x * WS_BLOCK + l) * WS_PER_LANE;      // first bucket of this lane (sets are multiples of the span)
    G1Xyzz run = g1_inf(), w = g1_inf(), v = g1_inf();
    constexpr int n_local = 2 * WS_PER_LANE, lg = 8, s_scan = n_local, s_comb = s_scan + lg, s_tree = s_comb + 1, s_end = s_tree + lg;
#pragma unroll 1
    for (int step = 0; step < s_end; step++) {
        G1Xyzz a, b = g1_inf();
        int dst;                                            // 0: run, 1: w, 2: v
        if (step < n_local) {
            if ((step & 1) == 0) { a = run; b = load_xyzz(buckets, b0 + (WS_PER_LANE - 1 - (step >> 1))); dst = 0; }
            else { a = w; b = run; dst = 1; }
        } else if (step < s_comb) {                         // suffix scan: run_l += run_{l + 2^k}
            const uint32_t dist = 1u << (step - s_scan);
            put_raw(sm + l, WS_BLOCK, run);
            __syncthreads();
            if (l + dist < WS_BLOCK) b = get_raw(sm + l + dist, WS_BLOCK);
            __syncthreads();
            a = run; dst = 0;
        } else if (step == s_comb) {                        // v = PER_LANE X_l (lanes >= 1) + w_l
            if (l >= 1) {
                v = run;
#pragma unroll 1
                for (int k = 1; k < WS_PER_LANE; k <<= 1) v = g1_dbl(v);
            }
            a = v; b = w; dst = 2;
        } else {                                            // tree over the workgroup
            const uint32_t dist = WS_BLOCK >> (step - s_tree + 1);
            put_raw(sm + l, WS_BLOCK, v);
            __syncthreads();
            if (l < dist) b = get_raw(sm + l + dist, WS_BLOCK);
            __syncthreads();
            a = v; dst = 2;
        }
        const G1Xyzz r = g1_add(a, b);
        if (dst == 0) run = r; else if (dst == 1) w = r; else v = r;
    }
    if (l == 0) {
        store_xyzz(out, (size_t)blockIdx.x * 2, v);
        store_xyzz(out, (size_t)blockIdx.x * 2 + 1, run);   // after the scan lane 0 holds the sum of all R_l
    }
}

// one lane per set: direct running sum over the last n <= 16 entries, then value = L^levels * W - sum C
__global__ __launch_bounds__(64) void k_g1_reduce_final(const uint32_t* __restrict__ in_s, const uint32_t* __restrict__ in_c, size_t sets, uint32_t n, int levels,
                                  uint32_t* __restrict__ winsum) {
    size_t set = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (set >= sets) return;
    G1Xyzz run = g1_inf(), w = g1_inf(), c = g1_inf();
    const uint32_t steps = 3 * n;
#pragma unroll 1
    for (uint32_t step = 0; step <= steps; step++) {
        if (step == steps) {
#pragma unroll 1
            for (int k = 0; k < 4 * levels; k++) w = g1_dbl(w);
            c.y = neg(c.y);
        }
        const uint32_t ph = step == steps ? 3u : step % 3;
        const size_t idx = set * n + (n - 1 - (step < steps ? step / 3 : 0));
        G1Xyzz a, b;
        if (ph == 0) { a = run; b = load_xyzz(in_s, idx); }
        else if (ph == 1) { a = w; b = run; }
        else if (ph == 2) { a = c; b = load_xyzz(in_c, idx); }
        else { a = w; b = c; }
        G1Xyzz r = g1_add(a, b);
        if (ph == 0) run = r; else if (ph == 1 || ph == 3) w = r; else c = r;
    }
    store_xyzz(winsum, set, w);
}

// ---- 6. (batch > 1) combine the W window sums of each MSM on the device: Horner over windows, width[w] doublings each.
__global__ __launch_bounds__(256) void k_g1_horner(const uint32_t* __restrict__ winsum, uint32_t batch, WindowTable wt, uint32_t* __restrict__ out) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    G1Xyzz acc = load_xyzz(winsum, (size_t)b * wt.W + wt.W - 1);
#pragma unroll 1
    for (int w = wt.W - 2; w >= 0; w--) {
#pragma unroll 1
        for (int j = 0; j < wt.width[w]; j++) acc = g1_dbl(acc);
        acc = g1_add(acc, load_xyzz(winsum, (size_t)b * wt.W + w));
    }
    store_xyzz(out, b, acc);
}

// batched results: XYZZ (Montgomery) -> affine standard-form little-endian limbs, (0,0) for infinity
__global__ __launch_bounds__(256) void k_g1_results_affine(const uint32_t* __restrict__ xyzz, uint32_t count, uint32_t* __restrict__ out /* count*24 */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    G1Affine a = g1_to_affine_dev(load_xyzz(xyzz, i));
    uint32_t w[12] = {0};
    if (!a.inf) from_mont28(a.x, w);
    store_words12(out + (size_t)i * 24, w);
    if (!a.inf) from_mont28(a.y, w);
    store_words12(out + (size_t)i * 24 + 12, w);
}

}  // namespace dr
