// libdotring_hip.so — C ABI implementation (include/dotring_hip.h): host orchestration of the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/dotring_hip.h"
#include "hostmath.hpp"
#include "hostpairing.hpp"
#include "hostproto.hpp"
#include "kernels_bsn.hip.h"
#include "kernels_g1.hip.h"
#include "kernels_ntt.hip.h"
#include "kernels_ring.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE,                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

struct ProfEntry {
    double ms = 0;
    int launches = 0;
};

// grow-only device scratch buffer
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return DR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e != hipSuccess) return fail(DR_ERR_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
        cap = want;
        return DR_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() { return reinterpret_cast<T*>(p); }
};

}  // namespace

struct dr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, ProfEntry> prof_data;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    // MSM workspaces
    Scratch scalars, digits, counts, offsets, cursor, tiles, sorted, buckets, partial, winsum, result, io_a, io_b, io_c, perm, cells, cell_off;
    Scratch vfy_bases, vfy_in, vfy_std;      // dr_ringvrf_verify_batch: decompressed G1 points stay resident between its steps
    dr_ctx* aux = nullptr;                   // second stream for the latency-bound Bandersnatch side of the batch verifier
    dr_ctx* aux2 = nullptr;                  // third stream: the verifier's two G1 MSMs run side by side
    std::vector<dr_ctx*> helpers;            // further streams working for this context (a prover's Pedersen stream): profiling only
    dr::TwiddleCache twiddles;
};

struct dr_srs {
    int device = 0;
    size_t count = 0;
    uint32_t* d_bases = nullptr;   // G1Affine[count], Montgomery
    // optional fixed-base window table: table[w][i] = 2^(start_w) * base[i]; all windows share one bucket set
    uint32_t* d_table = nullptr;
    dr::WindowTable table_wt{};
    // optional comb table over the window table: comb[j][w][d-1] = d * table[w][j], every digit magnitude precomputed
    uint32_t* d_comb = nullptr;
    uint32_t comb_h = 0;
    // derived bases for summation-by-parts commitments, keyed by log2(domain size): PS_j = sum_{i<=j} L_i(tau) G
    std::map<unsigned, dr_srs*> lagrange_prefix;
    std::mutex derive_mutex;             // provers for the same SRS may be created from different threads
};

namespace {

int use_ctx(dr_ctx* ctx) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return DR_OK;
}

// kernel launch wrapper with optional hipEvent timing on the ctx stream
template <class F>
int launch(dr_ctx* ctx, const char* name, F&& f) {
    if (ctx->prof) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, ctx->stream));
        f();
        HIP_TRY(hipEventRecord(b, ctx->stream));
        ctx->prof_pending.push_back({name, {a, b}});
    } else {
        f();
    }
    HIP_TRY(hipGetLastError());
    return DR_OK;
}

int prof_collect(dr_ctx* ctx) {
    for (auto& it : ctx->prof_pending) {
        HIP_TRY(hipEventSynchronize(it.second.second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, it.second.first, it.second.second));
        auto& e = ctx->prof_data[it.first];
        e.ms += ms;
        e.launches += 1;
        (void)hipEventDestroy(it.second.first);
        (void)hipEventDestroy(it.second.second);
    }
    ctx->prof_pending.clear();
    return DR_OK;
}

#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != DR_OK) return rc_; \
    } while (0)

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// The device keeps Fq in Montgomery form with R = 2^392 (14 x 28-bit limbs, fq28.hip.h), the host with R = 2^384
// (6 x 64-bit limbs, hostmath.hpp); both store canonical little-endian words, so crossing the boundary is one Montgomery
// product per coordinate: device -> host multiplies by 2^-8 (host-Montgomery constant 2^376), host -> device by 2^8 (2^392).
inline drh::Fq fq_dev_to_host(const drh::Fq& v) {
    static const drh::Fq k = [] { drh::Fq c = drh::Fq::zero(); c.l[5] = 0x0100000000000000ULL; return c; }();
    return v * k;
}
inline drh::Fq fq_host_to_dev(const drh::Fq& v) {
    static const drh::Fq k = [] {
        drh::Fq c;
        const uint64_t w[6] = {0x19d800000347fcb8ULL, 0x12e00cde6d2002b1ULL, 0x37669f83a2090c72ULL, 0x09b09b42da0f73e0ULL, 0xa7c515d98f1297bbULL, 0x0577a659fcfa012cULL};
        std::memcpy(c.l, w, sizeof w);
        return c;
    }();
    return v * k;
}
inline void g1_dev_to_host(drh::G1* pts, size_t n) {
    for (size_t i = 0; i < n; i++) {
        pts[i].x = fq_dev_to_host(pts[i].x); pts[i].y = fq_dev_to_host(pts[i].y);
        pts[i].zz = fq_dev_to_host(pts[i].zz); pts[i].zzz = fq_dev_to_host(pts[i].zzz);
    }
}
inline void g1_host_to_dev(drh::G1* pts, size_t n) {
    for (size_t i = 0; i < n; i++) {
        pts[i].x = fq_host_to_dev(pts[i].x); pts[i].y = fq_host_to_dev(pts[i].y);
        pts[i].zz = fq_host_to_dev(pts[i].zz); pts[i].zzz = fq_host_to_dev(pts[i].zzz);
    }
}

// ---- window plan for the GPU Pippenger.  Scalars are reduced mod r (< 2^255) on the device and the 256 bits
// are tiled by W = ceil(256/c) windows of width cmax or cmax-1 (see WindowTable).  Work ~ W*n mixed adds +
// W*2^(c-1)*(2 full adds) + per-chunk scalar multiplications; a full add costs ~1.4 mixed adds; pick the c
// minimising that, within [7,16] (W <= 37 fits the table).
bool window_ok(int c) { return c >= 7 && c <= 16; }
bool table_window_ok(int c) { return c >= 7 && c <= 22; }     // one bucket set per MSM: wider windows stay cheap
int pick_window(size_t n) {
    int best = 7;
    double best_cost = 1e300;
    for (int c = 7; c <= 16; c++) {
        int W = (256 + c - 1) / c;
        double cost = (double)W * ((double)n + 2.8 * (double)(1u << (c - 1)) + 40.0 * (double)((1u << (c - 1)) / 16 + 1));
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

uint32_t g_chunk_len = 16;   // buckets per lane in k_g1_reduce_chunks (DOTRING_MSM_CHUNK)
// launch kernel template K<CV> for the curve id cv (dr::CV_BANDERSNATCH / dr::CV_JUBJUB)
#define LAUNCH_CV(cv, K, ...)                                                                  \
    do {                                                                                        \
        if ((cv) == dr::CV_JUBJUB) hipLaunchKernelGGL((K<dr::CV_JUBJUB>), __VA_ARGS__);         \
        else hipLaunchKernelGGL((K<dr::CV_BANDERSNATCH>), __VA_ARGS__);                         \
    } while (0)
bool g_bsn_glv = true;       // GLV lane-pair kernels for latency-bound Bandersnatch launches (DOTRING_BSN_GLV=0: plain 64-window kernels)
bool g_use_comb = true;      // use comb tables when an SRS has one (DOTRING_MSM_COMB=0: bucket method)
bool g_chain_wave = true;    // one wave per proof for the witness accumulator chain (DOTRING_CHAIN_WAVE=0: one lane per proof)
size_t g_level_threshold = (size_t)1 << 18;   // chunk lanes from which the level-wise reduction is used (DOTRING_MSM_LEVEL_LANES)
bool g_reduce_levels = true; // level-wise bucket reduction for many bucket sets (DOTRING_MSM_LEVELS=0 disables)

struct MsmPlan {
    dr::WindowTable wt;
    int W;
    uint32_t H, L, T;
};
dr::WindowTable make_window_table(int c) {
    dr::WindowTable wt;
    wt.W = (256 + c - 1) / c;
    int base = 256 / wt.W, rem = 256 % wt.W;
    wt.cmax = base + (rem ? 1 : 0);
    int bit = 0;
    for (int w = 0; w < wt.W; w++) {
        int width = base + (w >= wt.W - rem ? 1 : 0);
        wt.start[w] = (uint8_t)bit;
        wt.width[w] = (uint8_t)width;
        bit += width;
    }
    return wt;
}

MsmPlan make_plan(size_t n, int force_c) {
    MsmPlan p;
    int c = window_ok(force_c) ? force_c : pick_window(n);
    p.W = (256 + c - 1) / c;
    int base = 256 / p.W, rem = 256 % p.W;       // `rem` windows of width base+1 (placed on top), the rest base
    p.wt.W = p.W;
    p.wt.cmax = base + (rem ? 1 : 0);
    int bit = 0;
    for (int w = 0; w < p.W; w++) {
        int width = base + (w >= p.W - rem ? 1 : 0);
        p.wt.start[w] = (uint8_t)bit;
        p.wt.width[w] = (uint8_t)width;
        bit += width;
    }
    p.H = 1u << (p.wt.cmax - 1);
    p.L = std::min<uint32_t>(p.H, g_chunk_len);
    p.T = p.H / p.L;
    return p;
}

int g_force_c = 0;   // test hook: DOTRING_MSM_WINDOW

// core: bases/scalars on the device; writes batch results (XYZZ, Montgomery) into host vector
// Fixed-base table descriptor for msm_device (table == nullptr: plain bases, one bucket set per window).
struct MsmTable {
    const uint32_t* table = nullptr;
    dr::WindowTable wt{};
    uint32_t stride = 0, offset = 0;
    const uint32_t* comb = nullptr;      // comb[j][w][d-1], see k_g1_comb_msm
    uint32_t comb_h = 0;
    uint32_t short_from = 0xffffffffu, n_short = 0;   // batched MSM: vectors from this index on are zero beyond n_short (sort hint)
};

int msm_device(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch,
               std::vector<drh::G1>& results, const MsmTable* tbl = nullptr) {
    results.assign(batch, drh::G1::inf());
    if (n == 0 || batch == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "MSM size must be below 2^31");
    const bool single = tbl != nullptr && tbl->table != nullptr;
    // comb table + many MSMs: a plain sum of looked-up points per MSM, nothing to sort or reduce
    if (single && tbl->comb && batch >= 32 && g_use_comb) {
        TRY(ctx->result.reserve(batch * 192));
        // threads per MSM: enough waves to fill 1024 SIMDs x 3 resident waves, at most 4 waves (one block)
        unsigned waves = (unsigned)std::max<size_t>(1, std::min<size_t>(4, (3072 + batch / 2) / batch));
        while (waves > 1 && (size_t)waves * 64 > n) waves--;
        const unsigned threads = waves * 64;
        const size_t lds = (size_t)tbl->wt.W * threads * 2;
        TRY(ctx->partial.reserve(batch * threads * 192));
        TRY(launch(ctx, "k_g1_comb_msm", [&] {
            hipLaunchKernelGGL(dr::k_g1_comb_msm, dim3((unsigned)batch), dim3(threads), lds, ctx->stream, d_scalars, (uint32_t)n, tbl->wt, tbl->comb,
                               tbl->comb_h, tbl->offset, ctx->partial.as<uint32_t>());
        }));
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)batch), dim3(dr::RW_BLOCK), 0, ctx->stream, ctx->partial.as<uint32_t>(), threads,
                               ctx->result.as<uint32_t>());
        }));
        return DR_OK;
    }
    MsmPlan pl = make_plan(n, g_force_c);
    if (single) {
        pl.wt = tbl->wt;
        pl.W = tbl->wt.W;
        pl.H = 1u << (tbl->wt.cmax - 1);
        pl.L = std::min<uint32_t>(pl.H, g_chunk_len);
        pl.T = pl.H / pl.L;
        d_bases = tbl->table;
        if ((uint64_t)pl.W * tbl->stride >= (1ull << 31)) return fail(DR_ERR_INVALID, "window table too large");
    }
    // table mode: split the points of each MSM into index groups when one bucket set per MSM would leave lanes idle
    uint32_t groups = 1;
    if (single) {
        const size_t target_lanes = 524288;       // 8 waves per SIMD: finer slices balance better than 4 (2^20 bases: accumulate 3.98 -> 3.6 ms)
        while (groups < 64 && batch * groups * (size_t)pl.H < target_lanes && (size_t)n / (groups * 2) >= 64) groups *= 2;
        static const int force_groups = std::getenv("DOTRING_MSM_GROUPS") ? std::atoi(std::getenv("DOTRING_MSM_GROUPS")) : 0;
        if (force_groups > 0 && batch == 1 && (size_t)n / (size_t)force_groups >= 64) groups = (uint32_t)force_groups;
    }
    const size_t windows = batch * (size_t)pl.W;                   // digit rows
    const size_t bsets = single ? batch * groups : windows;        // bucket sets
    // few bucket sets of moderate size (a single MSM over a window table): the reduction is a latency chain of
    // 2L additions + a log2(H)-bit double-and-add + the fold of H/L partial sums; L = 4 makes it ~40 % shorter
    if (single && pl.L == 16 && pl.H >= 256 && pl.H <= 4096 && bsets * (size_t)(pl.H / 16) < ((size_t)1 << 17)) {
        pl.L = 4;
        pl.T = pl.H / 4;
    }
    // the same for a small MSM over plain bases (the verifier's 2- and 11-point folds): 41 -> 16 dependent additions
    if (!single && pl.L == 16 && pl.H >= 16 && bsets * (size_t)(pl.H / 16) < ((size_t)1 << 12)) {
        pl.L = 4;
        pl.T = pl.H / 4;
    }
    const size_t nbuckets = bsets * pl.H;
    const size_t ndigits = windows * n;
    if (nbuckets >= (1ull << 32) || ndigits >= (1ull << 32))
        return fail(DR_ERR_INVALID, "MSM batch too large for one launch (split the batch)");
    TRY(ctx->counts.reserve(nbuckets * 4));
    TRY(ctx->offsets.reserve((nbuckets + 1) * 4));
    const unsigned szblocks = div_up(nbuckets, dr::SZ_TILE);
    const size_t ncells = (size_t)dr::SZ_CLASSES * szblocks;
    TRY(ctx->tiles.reserve((size_t)(div_up(std::max(ncells, nbuckets), dr::SCAN_TILE) + 1) * 4));
    TRY(ctx->perm.reserve(nbuckets * 4));
    TRY(ctx->cells.reserve(ncells * 4));
    TRY(ctx->cell_off.reserve(ncells * 4));
    TRY(ctx->buckets.reserve(nbuckets * 192));
    TRY(ctx->partial.reserve(bsets * pl.T * 192));
    TRY(ctx->winsum.reserve(bsets * 192));
    hipStream_t st = ctx->stream;
    auto exclusive_scan = [&](const uint32_t* in, uint32_t* out, size_t count) {
        const unsigned nt = div_up(count, dr::SCAN_TILE);
        hipLaunchKernelGGL(dr::k_scan_tiles, dim3(nt), dim3(dr::SCAN_BLOCK), 0, st, in, out, ctx->tiles.as<uint32_t>(), count);
        hipLaunchKernelGGL(dr::k_scan_tile_sums, dim3(1), dim3(dr::SCAN_BLOCK), 0, st, ctx->tiles.as<uint32_t>(), nt, ctx->tiles.as<uint32_t>() + nt);
        hipLaunchKernelGGL(dr::k_scan_add, dim3(div_up(count, 256)), dim3(256), 0, st, out, ctx->tiles.as<uint32_t>(), count);
    };
    // Sorting the digits by bucket.  Small bucket sets fed by a bounded number of digits (the batched prover) are
    // sorted by one workgroup each, entirely in LDS; a few huge sets (one 2^20-point MSM) use global atomics.
    const size_t per_set_scalars = single ? (n + groups - 1) / groups : n;
    const size_t per_set_digits = single ? per_set_scalars * (size_t)pl.W : n;
    const bool lds_sort = pl.H <= dr::SORT_MAX_H && bsets >= 64 && per_set_digits <= (1u << 20) && bsets * per_set_digits < (1ull << 32);
    if (lds_sort) {
        dr::SortSetParams sp;
        sp.n = (uint32_t)n; sp.batch = (uint32_t)batch; sp.H = pl.H; sp.groups = groups; sp.single = single ? 1 : 0;
        sp.tbl_stride = single ? tbl->stride : 0; sp.tbl_offset = single ? tbl->offset : 0;
        sp.capacity = (uint32_t)per_set_digits;
        sp.short_from = single ? tbl->short_from : 0xffffffffu;
        sp.n_short = single ? std::min<uint32_t>(tbl->n_short, (uint32_t)n) : 0;
        TRY(ctx->sorted.reserve(bsets * per_set_digits * 4));
        TRY(launch(ctx, "k_g1_sort_sets", [&] {
            hipLaunchKernelGGL(dr::k_g1_sort_sets, dim3((unsigned)bsets), dim3(dr::SORT_BLOCK), 0, st, d_scalars, pl.wt, sp,
                               ctx->counts.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->sorted.as<uint32_t>());
        }));
    } else {
        TRY(ctx->digits.reserve(ndigits * 4));
        TRY(ctx->cursor.reserve(nbuckets * 4));
        TRY(ctx->sorted.reserve(ndigits * 4));
        HIP_TRY(hipMemsetAsync(ctx->counts.p, 0, nbuckets * 4, st));
        HIP_TRY(hipMemsetAsync(ctx->cursor.p, 0, nbuckets * 4, st));
        TRY(launch(ctx, "k_g1_digits", [&] {
            hipLaunchKernelGGL(dr::k_g1_digits, dim3(div_up(n * batch, 256)), dim3(256), 0, st, d_scalars, (uint32_t)n,
                               (uint32_t)batch, pl.wt, single ? 1 : 0, groups, ctx->digits.as<int32_t>(), ctx->counts.as<uint32_t>());
        }));
        TRY(launch(ctx, "k_scan", [&] { exclusive_scan(ctx->counts.as<uint32_t>(), ctx->offsets.as<uint32_t>(), nbuckets); }));
        TRY(launch(ctx, "k_g1_scatter", [&] {
            hipLaunchKernelGGL(dr::k_g1_scatter, dim3(div_up(ndigits, 256)), dim3(256), 0, st, ctx->digits.as<int32_t>(),
                               (uint32_t)n, windows, pl.H, single ? pl.W : 0, single ? tbl->stride : 0u, single ? tbl->offset : 0u, groups,
                               ctx->offsets.as<uint32_t>(), ctx->cursor.as<uint32_t>(),
                               ctx->sorted.as<uint32_t>());
        }));
    }
    // size-ordered bucket permutation for the accumulate kernel
    TRY(launch(ctx, "k_size_sort", [&] {
        hipLaunchKernelGGL(dr::k_size_hist, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cells.as<uint32_t>());
        exclusive_scan(ctx->cells.as<uint32_t>(), ctx->cell_off.as<uint32_t>(), ncells);
        hipLaunchKernelGGL(dr::k_size_place, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cell_off.as<uint32_t>(), ctx->perm.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_g1_accumulate", [&] {
        hipLaunchKernelGGL(dr::k_g1_accumulate, dim3(div_up(nbuckets, 256)), dim3(256), 0, st, d_bases,
                           ctx->sorted.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(),
                           ctx->buckets.as<uint32_t>(), nbuckets);
    }));
    // many bucket sets (batched prover): level-wise reduction, 2 additions per entry and no scalar multiplications;
    // few sets (single MSMs): chunk sums + double-and-add, whose latency is one short chain
    const bool leveled = g_reduce_levels && pl.L == 16 && pl.H >= 256 && bsets * (size_t)(pl.H / 16) >= g_level_threshold;
    if (leveled) {
        // level outputs live in ctx->partial: [S | C] per level, sizes sets * H/16, sets * H/256, ...
        size_t total = 0;
        for (uint32_t n = pl.H; n > 16; n /= 16) total += 2 * bsets * (n / 16);
        TRY(ctx->partial.reserve(total * 192));
        uint32_t* base = ctx->partial.as<uint32_t>();
        const uint32_t* in_s = ctx->buckets.as<uint32_t>();
        const uint32_t* in_c = nullptr;
        uint32_t n = pl.H;
        int level = 0;
        size_t off = 0;
        while (n > 16) {
            level++;
            const size_t cnt = bsets * (n / 16);
            uint32_t* out_s = base + off * 48;
            uint32_t* out_c = base + (off + cnt) * 48;
            TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
                if (in_c)
                    hipLaunchKernelGGL(dr::k_g1_reduce_level<true>, dim3(div_up(cnt, 128)), dim3(128), 0, st, in_s, in_c, bsets, n, 16u, level, out_s, out_c);
                else
                    hipLaunchKernelGGL(dr::k_g1_reduce_level<false>, dim3(div_up(cnt, 128)), dim3(128), 0, st, in_s, in_c, bsets, n, 16u, level, out_s, out_c);
            }));
            in_s = out_s; in_c = out_c;
            off += 2 * cnt;
            n /= 16;
        }
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_final, dim3(div_up(bsets, 64)), dim3(64), 0, st, in_s, in_c, bsets, n, level, ctx->winsum.as<uint32_t>());
        }));
    } else {
        TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_chunks, dim3(div_up(bsets * pl.T, 128)), dim3(128), 0, st,
                               ctx->buckets.as<uint32_t>(), bsets, pl.H, pl.L, ctx->partial.as<uint32_t>());
        }));
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)bsets), dim3(dr::RW_BLOCK), 0, st,
                               ctx->partial.as<uint32_t>(), pl.T, ctx->winsum.as<uint32_t>());
        }));
    }

    static_assert(sizeof(drh::G1) == 192, "XYZZ layout");
    if (single) {
        // the bucket-set sum IS the MSM value: no window combination
        if (groups > 1 || batch == 1) {
            // few MSMs: fetch the per-group sums and add them on the host (<= 64 additions per MSM)
            std::vector<drh::G1> parts(bsets);
            HIP_TRY(hipMemcpyAsync(parts.data(), ctx->winsum.p, bsets * 192, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            g1_dev_to_host(parts.data(), parts.size());
            for (size_t b = 0; b < batch; b++) {
                drh::G1 acc = drh::G1::inf();
                for (uint32_t g = 0; g < groups; g++) acc = drh::g1_add(acc, parts[b * groups + g]);
                results[b] = acc;
            }
            if (batch > 1) {      // keep the batched contract: results in ctx->result for the device-side affine pass
                TRY(ctx->result.reserve(batch * 192));
                std::vector<drh::G1> up(results);
                g1_host_to_dev(up.data(), up.size());
                HIP_TRY(hipMemcpyAsync(ctx->result.p, up.data(), batch * 192, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
        } else {
            TRY(ctx->result.reserve(batch * 192));
            HIP_TRY(hipMemcpyAsync(ctx->result.p, ctx->winsum.p, batch * 192, hipMemcpyDeviceToDevice, st));
        }
    } else if (batch == 1) {
        // window combination on the host: a 255-doubling serial chain is ~50x faster on one CPU core
        std::vector<drh::G1> ws(pl.W);
        HIP_TRY(hipMemcpyAsync(ws.data(), ctx->winsum.p, (size_t)pl.W * 192, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        g1_dev_to_host(ws.data(), ws.size());
        drh::G1 acc = ws[pl.W - 1];
        for (int w = pl.W - 2; w >= 0; w--) {
            for (int j = 0; j < pl.wt.width[w]; j++) acc = drh::g1_dbl(acc);
            acc = drh::g1_add(acc, ws[w]);
        }
        results[0] = acc;
    } else {
        TRY(ctx->result.reserve(batch * 192));
        TRY(launch(ctx, "k_g1_horner", [&] {
            hipLaunchKernelGGL(dr::k_g1_horner, dim3(div_up(batch, 64)), dim3(64), 0, st, ctx->winsum.as<uint32_t>(),
                               (uint32_t)batch, pl.wt, ctx->result.as<uint32_t>());
        }));
        // results stay in ctx->result; msm_batch_results_to_bytes() finishes them on the device
    }
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

MsmTable srs_table(const dr_srs* srs, size_t offset) {
    MsmTable t;
    if (srs->d_table) {
        t.table = srs->d_table;
        t.wt = srs->table_wt;
        t.stride = (uint32_t)srs->count;
        t.offset = (uint32_t)offset;
        t.comb = srs->d_comb;
        t.comb_h = srs->comb_h;
    }
    return t;
}

void g1_result_to_bytes(const drh::G1& r, uint8_t* out96, int* is_inf);
int msm_batch_results_to_bytes(dr_ctx* ctx, size_t batch, uint8_t* out_be_xy, int* is_inf);

// MSM(s) with results written as BE affine records
int msm_to_bytes(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf,
                 const MsmTable* tbl = nullptr) {
    std::vector<drh::G1> res;
    TRY(msm_device(ctx, d_bases, d_scalars, n, batch, res, tbl));
    if (batch == 1 || n == 0) {
        for (size_t b = 0; b < batch; b++) g1_result_to_bytes(res[b], out_be_xy + 96 * b, is_inf ? is_inf + b : nullptr);
        return DR_OK;
    }
    return msm_batch_results_to_bytes(ctx, batch, out_be_xy, is_inf);
}

// batch > 1: results were left in ctx->result (XYZZ).  The affine conversion is one 381-bit field inversion per
// result: on the GPU a branch-free binary-Euclid chain (≈ 0.5 ms of pure latency per call, whatever the batch; 0.85 ms with
// the Fermat power it replaced).  Measured alternative for whole batches (DOTRING_AFFINE_ON_HOST=1): download XYZZ and invert
// on the worker threads — less GPU time but more wall time per 1024 proofs, so the kernel stays the default there.
int msm_batch_results_to_bytes(dr_ctx* ctx, size_t batch, uint8_t* out_be_xy, int* is_inf) {
    static const bool on_host = std::getenv("DOTRING_AFFINE_ON_HOST") && std::atoi(std::getenv("DOTRING_AFFINE_ON_HOST")) != 0;
    // a handful of results (a single proof's 4 witness commitments, 2 openings): the kernel's one inversion chain is 0.6 ms of
    // latency whatever the count, the host inverts in ~15 us each
    if (on_host || batch <= 16) {
        static_assert(sizeof(drh::G1) == 192, "XYZZ layout");
        std::vector<drh::G1> res(batch);
        HIP_TRY(hipMemcpyAsync(res.data(), ctx->result.p, batch * 192, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
        g1_dev_to_host(res.data(), res.size());
        drh::parallel_for(batch, [&](size_t b) { g1_result_to_bytes(res[b], out_be_xy + 96 * b, is_inf ? is_inf + b : nullptr); });
        return DR_OK;
    }
    TRY(ctx->io_c.reserve(batch * 96));
    TRY(launch(ctx, "k_g1_results_affine", [&] {
        hipLaunchKernelGGL(dr::k_g1_results_affine, dim3(div_up(batch, 64)), dim3(64), 0, ctx->stream, ctx->result.as<uint32_t>(),
                           (uint32_t)batch, ctx->io_c.as<uint32_t>());
    }));
    std::vector<uint8_t> le(batch * 96);
    HIP_TRY(hipMemcpyAsync(le.data(), ctx->io_c.p, batch * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    for (size_t b = 0; b < batch; b++) {
        bool allz = true;
        for (int j = 0; j < 96; j++) if (le[96 * b + j]) { allz = false; break; }
        if (is_inf) is_inf[b] = allz ? 1 : 0;
        for (int j = 0; j < 48; j++) {
            out_be_xy[96 * b + j] = le[96 * b + 47 - j];
            out_be_xy[96 * b + 48 + j] = le[96 * b + 95 - j];
        }
    }
    return DR_OK;
}

void g1_result_to_bytes(const drh::G1& r, uint8_t* out96, int* is_inf) {
    drh::Fq ax, ay;
    if (!drh::g1_to_affine(r, ax, ay)) {
        std::memset(out96, 0, 96);
        if (is_inf) *is_inf = 1;
        return;
    }
    ax.store_be(out96);
    ay.store_be(out96 + 48);
    if (is_inf) *is_inf = 0;
}

// BE x||y records -> LE standard-form limbs (device converts to Montgomery). Validates range; infinity -> zeros.
int g1_be_to_le_limbs(const uint8_t* be, size_t m, std::vector<uint8_t>& le, bool check_curve) {
    le.resize(m * 96);
    for (size_t i = 0; i < m; i++) {
        const uint8_t* rec = be + 96 * i;
        uint8_t* dst = le.data() + 96 * i;
        bool inf = (rec[0] & 0x40) != 0;
        if (!inf) {
            bool allz = true;
            for (int j = 0; j < 96; j++) if (rec[j]) { allz = false; break; }
            inf = allz;
        }
        if (inf) { std::memset(dst, 0, 96); continue; }
        if (rec[0] & 0xe0) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
        for (int j = 0; j < 48; j++) { dst[j] = rec[47 - j]; dst[48 + j] = rec[95 - j]; }
        drh::Fq x, y;
        if (!drh::Fq::load_le(x, dst) || !drh::Fq::load_le(y, dst + 48))
            return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
        if (check_curve && !drh::g1_on_curve(x, y)) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
    }
    return DR_OK;
}

}  // namespace

// =================================================================================== C ABI
extern "C" {

const char* dr_version(void) { return "dotring_hip 0.1 (gfx950)"; }
const char* dr_last_error(void) { return g_err.c_str(); }

int dr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// live contexts (a prover unregisters its helper stream from its context only if that context still exists)
static std::mutex g_live_mutex;
static std::set<dr_ctx*> g_live_ctx;
static bool ctx_alive(dr_ctx* c) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return g_live_ctx.count(c) != 0;
}

namespace {
// the Elligator / Tonelli-Shanks constants of kernels_bsn.hip.h, computed with the host field and copied to the
// device's constant block once per context
int bsn_consts_init(hipStream_t st) {
    using drh::Fr;
    static const uint8_t D_LE[32] = {0xe7, 0x58, 0x8d, 0x18, 0xf5, 0xf2, 0x69, 0xb3, 0x92, 0x4f, 0xe5, 0x77, 0x71, 0x67, 0x66, 0xcb,
                                     0xd8, 0xb6, 0xe3, 0x6b, 0xf8, 0x3b, 0x6e, 0xc6, 0xcb, 0x67, 0xc2, 0x33, 0x26, 0xc1, 0x89, 0x63};
    Fr d;
    if (!Fr::load_le(d, D_LE)) return fail(DR_ERR_DEVICE, "bad curve constant");
    Fr five = Fr::from_u64(5), a = five.neg();
    Fr inv_den = (a - d).inv();
    Fr mont_a = (a + d).dbl() * inv_den, mont_b = Fr::from_u64(4) * inv_den;
    Fr aob = mont_a * mont_b.inv(), inv_b2 = mont_b.sqr().inv();
    static const uint64_t Q[4] = {0xfffe5bfeffffffffULL, 0x09a1d80553bda402ULL, 0x299d7d483339d808ULL, 0x0000000073eda753ULL};   // (p-1) / 2^32
    dr::BsnConsts h;
    auto put = [](uint32_t (&w)[8], const Fr& v) { std::memcpy(w, v.l, 32); };     // Montgomery limbs, same R on host and device
    put(h.mont_b, mont_b); put(h.a_over_b, aob); put(h.inv_b2, inv_b2);
    static const uint8_t GLV_B_LE[32] = {0xb4, 0x10, 0x25, 0x17, 0x4d, 0x01, 0x0f, 0xee, 0xd6, 0xf4, 0x9a, 0x0d, 0x77, 0x12, 0xa7, 0x2e,
                                         0x88, 0x1a, 0x51, 0x63, 0x3a, 0x0d, 0xf0, 0x61, 0xa5, 0x26, 0x84, 0x82, 0x8b, 0xf2, 0xc9, 0x52};
    static const uint8_t GLV_C_LE[32] = {0x3d, 0x0b, 0x65, 0xdf, 0x6c, 0x80, 0x5c, 0x51, 0xe9, 0xf4, 0x36, 0xff, 0xcf, 0xab, 0x56, 0x84,
                                         0x07, 0xd1, 0x17, 0x6c, 0xfd, 0x6e, 0x7c, 0xa9, 0xc3, 0x57, 0x54, 0x86, 0xcf, 0x24, 0xc6, 0x6c};
    Fr gb, gc;
    if (!Fr::load_le(gb, GLV_B_LE) || !Fr::load_le(gc, GLV_C_LE)) return fail(DR_ERR_DEVICE, "bad curve constant");
    put(h.glv_b, gb); put(h.glv_c, gc);
    Fr c = five.pow(Q, 4);
    for (int j = 0; j < 32; j++) { put(h.c_pow[j], c); c = c.sqr(); }
    if (!(c == Fr::one())) return fail(DR_ERR_DEVICE, "bad Tonelli-Shanks constants");
    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(dr::g_bsn_consts), &h, sizeof h, 0, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DR_OK;
}
}  // namespace

int dr_ctx_create(int device_id, dr_ctx** out) {
    if (!out) return fail(DR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = dr_device_count();
    if (n <= 0) return fail(DR_ERR_DEVICE, "no HIP device available (libdotring_hip needs an MI355X / gfx950 GPU)");
    if (device_id < 0 || device_id >= n) return fail(DR_ERR_INVALID, "device id out of range");
    HIP_TRY(hipSetDevice(device_id));
    dr_ctx* ctx = new (std::nothrow) dr_ctx();
    if (!ctx) return fail(DR_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(DR_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    const char* fc = std::getenv("DOTRING_MSM_WINDOW");
    g_force_c = fc ? std::atoi(fc) : 0;
    if (!window_ok(g_force_c)) g_force_c = 0;
    const char* cl = std::getenv("DOTRING_MSM_CHUNK");
    if (cl) {
        int v = std::atoi(cl);
        if (v == 8 || v == 16 || v == 32 || v == 64 || v == 128) g_chunk_len = (uint32_t)v;
    }
    if (const char* lv = std::getenv("DOTRING_MSM_LEVELS")) g_reduce_levels = std::atoi(lv) != 0;
    if (const char* ll = std::getenv("DOTRING_MSM_LEVEL_LANES")) g_level_threshold = (size_t)std::max(1L, std::atol(ll));
    if (const char* cw = std::getenv("DOTRING_CHAIN_WAVE")) g_chain_wave = std::atoi(cw) != 0;
    if (const char* cb = std::getenv("DOTRING_MSM_COMB")) g_use_comb = std::atoi(cb) != 0;
    if (const char* gl = std::getenv("DOTRING_BSN_GLV")) g_bsn_glv = std::atoi(gl) != 0;
    int rc = bsn_consts_init(ctx->stream);
    if (rc != DR_OK) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return rc;
    }
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_ctx.insert(ctx);
    }
    *out = ctx;
    return DR_OK;
}

void dr_ctx_destroy(dr_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_ctx.erase(ctx);
    }
    if (ctx->aux) { dr_ctx_destroy(ctx->aux); ctx->aux = nullptr; }
    if (ctx->aux2) { dr_ctx_destroy(ctx->aux2); ctx->aux2 = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (Scratch* s : {&ctx->scalars, &ctx->digits, &ctx->counts, &ctx->offsets, &ctx->cursor, &ctx->tiles, &ctx->sorted,
                       &ctx->buckets, &ctx->partial, &ctx->winsum, &ctx->result, &ctx->io_a, &ctx->io_b, &ctx->io_c, &ctx->perm, &ctx->cells,
                       &ctx->cell_off, &ctx->vfy_bases, &ctx->vfy_in, &ctx->vfy_std})
        s->release();
    for (auto& it : ctx->prof_pending) {
        (void)hipEventDestroy(it.second.first);
        (void)hipEventDestroy(it.second.second);
    }
    for (auto& e : ctx->twiddles.entries) (void)hipFree(e.d_tw);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dr_ctx_sync(dr_ctx* ctx) {
    TRY(use_ctx(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

int dr_dev_alloc(dr_ctx* ctx, size_t bytes, void** dptr) {
    TRY(use_ctx(ctx));
    if (!dptr) return fail(DR_ERR_INVALID, "null out pointer");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return DR_OK;
}
int dr_dev_free(dr_ctx* ctx, void* dptr) {
    TRY(use_ctx(ctx));
    if (dptr) HIP_TRY(hipFree(dptr));
    return DR_OK;
}
int dr_dev_upload(dr_ctx* ctx, void* dptr, const void* host, size_t bytes) {
    TRY(use_ctx(ctx));
    HIP_TRY(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DR_OK;
}
int dr_dev_download(dr_ctx* ctx, void* host, const void* dptr, size_t bytes) {
    TRY(use_ctx(ctx));
    HIP_TRY(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DR_OK;
}

// profiling covers the helper streams of a context too: their kernels belong to the same calls
static void for_each_stream(dr_ctx* ctx, const std::function<void(dr_ctx*)>& f) {
    f(ctx);
    if (ctx->aux) f(ctx->aux);
    if (ctx->aux2) f(ctx->aux2);
    for (dr_ctx* h : ctx->helpers) f(h);
}
int dr_prof_enable(dr_ctx* ctx, int on) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    for_each_stream(ctx, [&](dr_ctx* c) { c->prof = on != 0; });
    return DR_OK;
}
int dr_prof_reset(dr_ctx* ctx) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    for_each_stream(ctx, [&](dr_ctx* c) { c->prof_data.clear(); });
    return DR_OK;
}
int dr_prof_get(dr_ctx* ctx, const char* name, double* total_ms, int* launches) {
    if (!ctx || !name) return fail(DR_ERR_INVALID, "null argument");
    double ms = 0.0;
    int n = 0;
    for_each_stream(ctx, [&](dr_ctx* c) {
        auto it = c->prof_data.find(name);
        if (it != c->prof_data.end()) { ms += it->second.ms; n += it->second.launches; }
    });
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return DR_OK;
}

// ------------------------------------------------------------------------------- seam A
static int te_scalar_mul_batch_dev(dr_ctx* ctx, int cv, const void* d_pts, const void* d_scalars, size_t n, void* d_out) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    // 4-bit windows (64 KiB of LDS per wave, 2 waves per CU) while the launch is latency-bound; 2-bit windows
    // (16 KiB, 10 waves per CU) once there are more waves than the 4-bit kernel can keep resident
    static const long w2_from = std::getenv("DOTRING_BSN_W2_FROM") ? std::atol(std::getenv("DOTRING_BSN_W2_FROM")) : 32768;
    TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
        if (w2_from > 0 && n >= (size_t)w2_from)
            LAUNCH_CV(cv, dr::k_bsn_scalar_mul_w2, dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                      (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t*)d_out, (uint32_t)n);
        else
            LAUNCH_CV(cv, dr::k_bsn_scalar_mul, dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                      (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t*)d_out, (uint32_t)n);
    }));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}
static int check_curve(int cv) { return drh::te_curve(cv) ? DR_OK : fail(DR_ERR_INVALID, "unknown curve id"); }
int dr_bsn_scalar_mul_batch_dev(dr_ctx* ctx, const void* d_pts, const void* d_scalars, size_t n, void* d_out) {
    return te_scalar_mul_batch_dev(ctx, dr::CV_BANDERSNATCH, d_pts, d_scalars, n, d_out);
}

static int check_fr_elems(const uint8_t* p, size_t count, const char* what) {
    // canonical = below p as a little-endian integer; a plain limb comparison, on worker threads for large batches
    std::atomic<bool> bad{false};
    auto check = [&](size_t i) {
        uint64_t v[4];
        drh::load_le32(p + 32 * i, v);
        if (drh::Fr::geq_p(v)) bad.store(true, std::memory_order_relaxed);
    };
    if (count >= 65536) drh::parallel_for(count, check);        // below that, starting threads costs more than the loop
    else for (size_t i = 0; i < count; i++) check(i);
    if (bad.load()) return fail(DR_ERR_INVALID, std::string(what) + " coordinate is not a canonical field element");
    return DR_OK;
}

// scalars -> GLV halves for the lane-pair kernels: 12 words per term (|k1|, |k2|, two sign words, padding)
static int glv_split_scalars(const uint8_t* scalars, size_t n, std::vector<uint32_t>& out) {
    out.assign(n * 12, 0);
    std::atomic<bool> bad{false};
    auto one = [&](size_t i) {
        uint64_t k[4];
        drh::mod_n().reduce_bytes(scalars + 32 * i, 32, false, k);
        drh::GlvSplit s;
        if (!drh::glv_decompose(k, s)) { bad.store(true); return; }
        uint32_t* o = out.data() + 12 * i;
        o[0] = (uint32_t)s.k1[0]; o[1] = (uint32_t)(s.k1[0] >> 32); o[2] = (uint32_t)s.k1[1]; o[3] = (uint32_t)(s.k1[1] >> 32);
        o[4] = (uint32_t)s.k2[0]; o[5] = (uint32_t)(s.k2[0] >> 32); o[6] = (uint32_t)s.k2[1]; o[7] = (uint32_t)(s.k2[1] >> 32);
        o[8] = (uint32_t)s.neg1; o[9] = (uint32_t)s.neg2;
    };
    if (n >= 4096) drh::parallel_for(n, one);
    else for (size_t i = 0; i < n; i++) one(i);
    if (bad.load()) return fail(DR_ERR_DEVICE, "GLV decomposition out of range");
    return DR_OK;
}

static int te_scalar_mul_batch(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (n == 0) return DR_OK;
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    if (g_bsn_glv && drh::te_curve(cv)->glv && n < 16384) {       // latency-bound launch: halve the chain with GLV on lane pairs
        std::vector<uint32_t> split;
        TRY(glv_split_scalars(scalars, n, split));
        TRY(ctx->io_a.reserve(n * 64));
        TRY(ctx->io_b.reserve(n * 48));
        TRY(ctx->io_c.reserve(n * 64));
        HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->io_b.p, split.data(), n * 48, hipMemcpyHostToDevice, ctx->stream));
        TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
            hipLaunchKernelGGL(dr::k_bsn_scalar_mul_glv, dim3(div_up(2 * n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                               ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>(), (uint32_t)n);
        }));
        HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
        return DR_OK;
    }
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 32));
    TRY(ctx->io_c.reserve(n * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    TRY(te_scalar_mul_batch_dev(ctx, cv, ctx->io_a.p, ctx->io_b.p, n, ctx->io_c.p));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DR_OK;
}
int dr_bsn_scalar_mul_batch(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy) {
    return te_scalar_mul_batch(ctx, dr::CV_BANDERSNATCH, pts_xy, scalars, n, out_xy);
}
int dr_te_scalar_mul_batch(dr_ctx* ctx, int curve, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy) {
    return te_scalar_mul_batch(ctx, curve, pts_xy, scalars, n, out_xy);
}

static int te_msm_groups(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (groups == 0) return DR_OK;
    if (m == 0 || m > 64) return fail(DR_ERR_INVALID, "group size must be in 1..64");
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    size_t n = groups * m;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    if (g_bsn_glv && drh::te_curve(cv)->glv && m <= 32 && n < 16384) {
        std::vector<uint32_t> split;
        TRY(glv_split_scalars(scalars, n, split));
        uint32_t mpad2 = 2;
        while (mpad2 < 2 * m) mpad2 <<= 1;
        TRY(ctx->io_a.reserve(n * 64));
        TRY(ctx->io_b.reserve(n * 48));
        TRY(ctx->io_c.reserve(groups * 64));
        HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->io_b.p, split.data(), n * 48, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t per_block2 = dr::BSN_BLOCK / mpad2;
        TRY(launch(ctx, "k_bsn_msm_groups", [&] {
            hipLaunchKernelGGL(dr::k_bsn_msm_groups_glv, dim3(div_up(groups, per_block2)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                               ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>(), (uint32_t)groups, (uint32_t)m, mpad2);
        }));
        HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, groups * 64, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
        return DR_OK;
    }
    uint32_t mpad = 1;
    while (mpad < m) mpad <<= 1;
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 32));
    TRY(ctx->io_c.reserve(groups * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t per_block = dr::BSN_BLOCK / mpad;
    TRY(launch(ctx, "k_bsn_msm_groups", [&] {
        LAUNCH_CV(cv, dr::k_bsn_msm_groups, dim3(div_up(groups, per_block)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                  ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>(), (uint32_t)groups, (uint32_t)m, mpad);
    }));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, groups * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}
int dr_bsn_msm_groups(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    return te_msm_groups(ctx, dr::CV_BANDERSNATCH, pts_xy, scalars, groups, m, out_xy);
}
int dr_te_msm_groups(dr_ctx* ctx, int curve, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    return te_msm_groups(ctx, curve, pts_xy, scalars, groups, m, out_xy);
}

static int te_msm(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (!out_xy) return fail(DR_ERR_INVALID, "null buffer");
    if (n == 0) {
        std::memset(out_xy, 0, 64);
        out_xy[32] = 1;
        return DR_OK;
    }
    // fold 64 terms at a time on the device (one launch); the n/64 partial sums are then added on the host in extended
    // coordinates — a second device pass would pay a full scalar-multiplication latency for scalars that are all 1
    if (n <= 64) return te_msm_groups(ctx, cv, pts_xy, scalars, 1, n, out_xy);
    const size_t parts = (n + 63) / 64;
    std::vector<uint8_t> part(parts * 64);
    if (n % 64 == 0) {
        TRY(te_msm_groups(ctx, cv, pts_xy, scalars, parts, 64, part.data()));
    } else {        // pad the last group with 0 * (0, 1) so that everything is ONE launch
        std::vector<uint8_t> pp(parts * 64 * 64, 0), kk(parts * 64 * 32, 0);
        std::memcpy(pp.data(), pts_xy, n * 64);
        std::memcpy(kk.data(), scalars, n * 32);
        for (size_t i = n; i < parts * 64; i++) pp[64 * i + 32] = 1;
        TRY(te_msm_groups(ctx, cv, pp.data(), kk.data(), parts, 64, part.data()));
    }
    using drh::Fr;
    uint8_t D_LE[32];
    drh::store_le32(drh::te_curve(cv)->d, D_LE);
    Fr d, five = Fr::from_u64(drh::te_curve(cv)->neg_a[0]);        // -a
    if (!Fr::load_le(d, D_LE)) return fail(DR_ERR_DEVICE, "bad curve constant");
    Fr X = Fr::zero(), Y = Fr::one(), Z = Fr::one(), T = Fr::zero();          // identity
    for (size_t i = 0; i < parts; i++) {                                      // add-2008-hwcd with Z2 = 1
        Fr x2, y2;
        if (!Fr::load_le(x2, part.data() + 64 * i) || !Fr::load_le(y2, part.data() + 64 * i + 32)) return fail(DR_ERR_DEVICE, "kernel result out of range");
        Fr A = X * x2, B = Y * y2, C = T * d * (x2 * y2), D = Z;
        Fr E = (X + Y) * (x2 + y2) - A - B, F = D - C, G = D + C, H = B + A * five;      // H = B - a*A
        X = E * F; Y = G * H; T = E * H; Z = F * G;
    }
    Fr zi = Z.inv();
    (X * zi).store_le(out_xy);
    (Y * zi).store_le(out_xy + 32);
    return DR_OK;
}
int dr_bsn_msm(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    return te_msm(ctx, dr::CV_BANDERSNATCH, pts_xy, scalars, n, out_xy);
}
int dr_te_msm(dr_ctx* ctx, int curve, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    return te_msm(ctx, curve, pts_xy, scalars, n, out_xy);
}

// launch the point decoder for `n` encodings already at d_enc: Bandersnatch = the GLV lane-pair kernel, JubJub = the
// generic one; tai = candidates of try-and-increment (output hP, no subgroup test)
static void launch_decode_points(dr_ctx* ctx, hipStream_t st, int cv, bool tai, const uint32_t* d_enc, uint32_t* d_xy, uint32_t* d_ok, size_t n) {
    if (tai) {
        if (cv == dr::CV_JUBJUB)
            hipLaunchKernelGGL((dr::k_te_decode_points<dr::CV_JUBJUB, true>), dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, st, d_enc, d_xy, d_ok, (uint32_t)n);
        else
            hipLaunchKernelGGL((dr::k_te_decode_points<dr::CV_BANDERSNATCH, true>), dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, st, d_enc, d_xy, d_ok, (uint32_t)n);
    } else if (cv == dr::CV_JUBJUB) {
        hipLaunchKernelGGL((dr::k_te_decode_points<dr::CV_JUBJUB, false>), dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, st, d_enc, d_xy, d_ok, (uint32_t)n);
    } else {
        hipLaunchKernelGGL(dr::k_bsn_decode_points, dim3(div_up(2 * n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, st, d_enc, d_xy, d_ok, (uint32_t)n);
    }
    (void)ctx;
}

static int te_decode_points(dr_ctx* ctx, int cv, bool tai, const uint8_t* enc, size_t n, uint8_t* out_xy, uint8_t* ok) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (n == 0) return DR_OK;
    if (!enc || !out_xy || !ok) return fail(DR_ERR_INVALID, "null buffer");
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(ctx->io_a.reserve(n * 32));
    TRY(ctx->io_b.reserve(n * 64));
    TRY(ctx->io_c.reserve(n * 4));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, enc, n * 32, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch(ctx, "k_bsn_decode_points", [&] {
        launch_decode_points(ctx, ctx->stream, cv, tai, ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>(), n);
    }));
    std::vector<uint32_t> flags(n);
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_b.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(flags.data(), ctx->io_c.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    for (size_t i = 0; i < n; i++) ok[i] = flags[i] ? 1 : 0;
    return DR_OK;
}
int dr_bsn_decode_points(dr_ctx* ctx, const uint8_t* enc, size_t n, uint8_t* out_xy, uint8_t* ok) {
    return te_decode_points(ctx, dr::CV_BANDERSNATCH, false, enc, n, out_xy, ok);
}
int dr_te_decode_points(dr_ctx* ctx, int curve, const uint8_t* enc, size_t n, uint8_t* out_xy, uint8_t* ok) {
    return te_decode_points(ctx, curve, false, enc, n, out_xy, ok);
}

int dr_bsn_encode_to_curve_batch(dr_ctx* ctx, const uint8_t* u_pairs, size_t n, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (!u_pairs || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(check_fr_elems(u_pairs, 2 * n, "field element"));
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_c.reserve(n * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, u_pairs, n * 64, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch(ctx, "k_bsn_encode_to_curve", [&] {
        hipLaunchKernelGGL(dr::k_bsn_encode_to_curve, dim3(div_up(2 * n, 64)), dim3(64), 0, ctx->stream, ctx->io_a.as<uint32_t>(),
                           ctx->io_c.as<uint32_t>(), (uint32_t)n);
    }));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

int dr_fr_sqrt(const uint8_t in[32], uint8_t out[32]) {
    if (!in || !out) return fail(DR_ERR_INVALID, "null buffer");
    drh::Fr x, r;
    if (!drh::Fr::load_le(x, in)) return fail(DR_ERR_INVALID, "input is not a canonical field element");
    if (!drh::fr_sqrt(r, x)) return fail(DR_ERR_NOTSQUARE, "No square root exists");
    r.store_le(out);
    return DR_OK;
}

// ------------------------------------------------------------------------------- seam B
int dr_srs_load(dr_ctx* ctx, const uint8_t* g1_be_xy, size_t m, dr_srs** out) {
    TRY(use_ctx(ctx));
    if (!out) return fail(DR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!g1_be_xy || m == 0) return fail(DR_ERR_INVALID, "empty SRS");
    if (m >= (1ull << 31)) return fail(DR_ERR_INVALID, "SRS too large");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(g1_be_xy, m, le, /*check_curve=*/m <= 65536));
    dr_srs* s = new (std::nothrow) dr_srs();
    if (!s) return fail(DR_ERR_NOMEM, "out of host memory");
    s->device = ctx->device;
    s->count = m;
    hipError_t e = hipMalloc((void**)&s->d_bases, m * 96);
    if (e != hipSuccess) {
        delete s;
        return fail(DR_ERR_NOMEM, "hipMalloc for the SRS failed");
    }
    e = hipMemcpyAsync(s->d_bases, le.data(), m * 96, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(div_up(m, 256)), dim3(256), 0, ctx->stream, s->d_bases, (uint32_t)m);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) {
        (void)hipFree(s->d_bases);
        delete s;
        return fail(DR_ERR_DEVICE, std::string("SRS upload: ") + hipGetErrorString(e));
    }
    *out = s;
    return DR_OK;
}

int dr_srs_synthetic(dr_ctx* ctx, const uint8_t seed_be_xy[96], uint32_t first, size_t count, dr_srs** out) {
    TRY(use_ctx(ctx));
    if (!out || !seed_be_xy) return fail(DR_ERR_INVALID, "null argument");
    *out = nullptr;
    if (count == 0 || count >= (1ull << 31) || (uint64_t)first + count >= (1ull << 32) || first == 0)
        return fail(DR_ERR_INVALID, "bad synthetic SRS range");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(seed_be_xy, 1, le, true));
    dr_srs* s = new (std::nothrow) dr_srs();
    if (!s) return fail(DR_ERR_NOMEM, "out of host memory");
    s->device = ctx->device;
    s->count = count;
    uint32_t* d_seed = nullptr;
    hipError_t e = hipMalloc((void**)&s->d_bases, count * 96);
    if (e == hipSuccess) e = hipMalloc((void**)&d_seed, 96);
    if (e == hipSuccess) e = hipMemcpyAsync(d_seed, le.data(), 96, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(1), dim3(64), 0, ctx->stream, d_seed, 1u);
        hipLaunchKernelGGL(dr::k_g1_synth_bases, dim3(div_up(count, 128)), dim3(128), 0, ctx->stream, s->d_bases, (uint32_t)count, first, d_seed);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (d_seed) (void)hipFree(d_seed);
    if (e != hipSuccess) {
        if (s->d_bases) (void)hipFree(s->d_bases);
        delete s;
        return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, std::string("synthetic SRS: ") + hipGetErrorString(e));
    }
    *out = s;
    return DR_OK;
}

int dr_srs_powers(dr_ctx* ctx, const uint8_t base_be_xy[96], const uint8_t tau_le[32], size_t count, dr_srs** out) {
    TRY(use_ctx(ctx));
    if (!out || !base_be_xy || !tau_le) return fail(DR_ERR_INVALID, "null argument");
    *out = nullptr;
    if (count == 0 || count >= (1ull << 28)) return fail(DR_ERR_INVALID, "bad SRS size");
    drh::Fr tau;
    if (!drh::Fr::load_le(tau, tau_le)) return fail(DR_ERR_INVALID, "tau is not a canonical scalar");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(base_be_xy, 1, le, true));
    std::vector<uint8_t> pw(count * 32);
    drh::Fr t = drh::Fr::one();
    for (size_t i = 0; i < count; i++) {
        t.store_le(pw.data() + 32 * i);
        t = t * tau;
    }
    dr_srs* s = new (std::nothrow) dr_srs();
    if (!s) return fail(DR_ERR_NOMEM, "out of host memory");
    s->device = ctx->device;
    s->count = count;
    uint32_t *d_seed = nullptr, *d_pw = nullptr;
    hipError_t e = hipMalloc((void**)&s->d_bases, count * 96);
    if (e == hipSuccess) e = hipMalloc((void**)&d_seed, 96);
    if (e == hipSuccess) e = hipMalloc((void**)&d_pw, count * 32);
    if (e == hipSuccess) e = hipMemcpyAsync(d_seed, le.data(), 96, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pw, pw.data(), count * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(1), dim3(64), 0, ctx->stream, d_seed, 1u);
        hipLaunchKernelGGL(dr::k_g1_scalar_bases, dim3(div_up(count, 64)), dim3(64), 0, ctx->stream, s->d_bases, (uint32_t)count, d_pw, d_seed);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (d_seed) (void)hipFree(d_seed);
    if (d_pw) (void)hipFree(d_pw);
    if (e != hipSuccess) {
        if (s->d_bases) (void)hipFree(s->d_bases);
        delete s;
        return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, std::string("SRS powers: ") + hipGetErrorString(e));
    }
    *out = s;
    return DR_OK;
}

int dr_g2_mul(const uint8_t g2_be[192], const uint8_t scalar_le[32], uint8_t out_be[192]) {
    if (!g2_be || !scalar_le || !out_be) return fail(DR_ERR_INVALID, "null buffer");
    drh::G2Affine Q;
    Q.inf = false;
    if (!drh::Fq::load_be(Q.x.c1, g2_be) || !drh::Fq::load_be(Q.x.c0, g2_be + 48) || !drh::Fq::load_be(Q.y.c1, g2_be + 96) ||
        !drh::Fq::load_be(Q.y.c0, g2_be + 144) || !drh::g2_on_curve(Q))
        return fail(DR_ERR_INVALID, "invalid BLS12-381 G2 encoding");
    drh::G2Affine R = drh::g2_mul(Q, scalar_le);
    std::memset(out_be, 0, 192);
    if (R.inf) { out_be[0] = 0x40; return DR_OK; }
    R.x.c1.store_be(out_be);
    R.x.c0.store_be(out_be + 48);
    R.y.c1.store_be(out_be + 96);
    R.y.c0.store_be(out_be + 144);
    return DR_OK;
}

int dr_srs_download(dr_ctx* ctx, const dr_srs* srs, size_t offset, size_t count, uint8_t* out_be_xy) {
    TRY(use_ctx(ctx));
    if (!srs || !out_be_xy) return fail(DR_ERR_INVALID, "null argument");
    if (offset > srs->count || count > srs->count - offset) return fail(DR_ERR_INVALID, "range exceeds SRS size");
    if (count == 0) return DR_OK;
    TRY(ctx->io_a.reserve(count * 96));
    hipLaunchKernelGGL(dr::k_g1_bases_from_mont, dim3(div_up(count, 256)), dim3(256), 0, ctx->stream,
                       srs->d_bases + offset * 24, ctx->io_a.as<uint32_t>(), (uint32_t)count);
    std::vector<uint8_t> le(count * 96);
    HIP_TRY(hipMemcpyAsync(le.data(), ctx->io_a.p, count * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < count; i++)
        for (int j = 0; j < 48; j++) {
            out_be_xy[96 * i + j] = le[96 * i + 47 - j];
            out_be_xy[96 * i + 48 + j] = le[96 * i + 95 - j];
        }
    return DR_OK;
}

int dr_srs_precompute(dr_ctx* ctx, dr_srs* srs, int window_bits) {
    TRY(use_ctx(ctx));
    if (!srs) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
    if (srs->d_comb) { (void)hipFree(srs->d_comb); srs->d_comb = nullptr; srs->comb_h = 0; }     // derived from the window table
    if (window_bits == 0) {
        if (srs->d_table) (void)hipFree(srs->d_table);
        srs->d_table = nullptr;
        return DR_OK;
    }
    if (!table_window_ok(window_bits)) return fail(DR_ERR_INVALID, "window_bits must be in 7..22 (0 drops the table)");
    dr::WindowTable wt = make_window_table(window_bits);
    if ((uint64_t)wt.W * srs->count >= (1ull << 31)) return fail(DR_ERR_INVALID, "window table too large");
    if (srs->d_table) (void)hipFree(srs->d_table);
    srs->d_table = nullptr;
    HIP_TRY(hipMalloc((void**)&srs->d_table, (size_t)wt.W * srs->count * 96));
    hipLaunchKernelGGL(dr::k_g1_window_table, dim3(div_up(srs->count, 128)), dim3(128), 0, ctx->stream, srs->d_bases, (uint32_t)srs->count, wt,
                       srs->d_table);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(srs->d_table);
        srs->d_table = nullptr;
        return fail(DR_ERR_DEVICE, std::string("window table: ") + hipGetErrorString(e));
    }
    srs->table_wt = wt;
    return DR_OK;
}

int dr_srs_precompute_comb(dr_ctx* ctx, dr_srs* srs) {
    TRY(use_ctx(ctx));
    if (!srs) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
    if (!srs->d_table) return fail(DR_ERR_INVALID, "dr_srs_precompute must come first");
    if (srs->d_comb) return DR_OK;
    const dr::WindowTable& wt = srs->table_wt;
    if (wt.cmax > 14) return fail(DR_ERR_INVALID, "comb tables need window_bits <= 14");
    const uint32_t Hc = 1u << (wt.cmax - 1);
    const size_t rows = (size_t)srs->count * wt.W;
    const size_t bytes = rows * Hc * (size_t)dr::COMB_STRIDE * 4;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (bytes + ((size_t)8 << 30) > free_b) return fail(DR_ERR_NOMEM, "comb table of " + std::to_string(bytes >> 20) + " MiB does not fit");
    uint32_t* comb = nullptr;
    if (hipMalloc((void**)&comb, bytes) != hipSuccess) return fail(DR_ERR_NOMEM, "comb table allocation failed");
    // rows per launch bounded by 4 GiB of staging (XYZZ + prefix product per entry)
    const size_t per_row = (size_t)Hc * (192 + 48);
    const size_t chunk = std::max<size_t>(128, std::min<size_t>(rows, ((size_t)4 << 30) / per_row) / 128 * 128);
    uint32_t *tx = nullptr, *tp = nullptr;
    hipError_t e = hipMalloc((void**)&tx, chunk * Hc * 192);
    if (e == hipSuccess) e = hipMalloc((void**)&tp, chunk * Hc * 48);
    for (size_t lo = 0; e == hipSuccess && lo < rows; lo += chunk) {
        const uint32_t cnt = (uint32_t)std::min(chunk, rows - lo);
        hipLaunchKernelGGL(dr::k_g1_comb_build, dim3(div_up(cnt, 128)), dim3(128), 0, ctx->stream, srs->d_table, (uint32_t)srs->count, wt, Hc, lo, cnt,
                           comb, tx, tp);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (tx) (void)hipFree(tx);
    if (tp) (void)hipFree(tp);
    if (e != hipSuccess) {
        (void)hipFree(comb);
        return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, std::string("comb table: ") + hipGetErrorString(e));
    }
    srs->d_comb = comb;
    srs->comb_h = Hc;
    return DR_OK;
}

void dr_srs_destroy(dr_srs* srs) {
    if (!srs) return;
    (void)hipSetDevice(srs->device);
    if (srs->d_comb) (void)hipFree(srs->d_comb);
    for (auto& it : srs->lagrange_prefix) dr_srs_destroy(it.second);
    srs->lagrange_prefix.clear();
    if (srs->d_table) (void)hipFree(srs->d_table);
    if (srs->d_bases) (void)hipFree(srs->d_bases);
    delete srs;
}

size_t dr_srs_size(const dr_srs* srs) { return srs ? srs->count : 0; }

int dr_g1_msm_batch_dev(dr_ctx* ctx, const dr_srs* srs, const void* d_scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf) {
    TRY(use_ctx(ctx));
    if (!srs || !out_be_xy) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
    if (n > srs->count) return fail(DR_ERR_INVALID, "polynomial degree exceeds SRS size");
    MsmTable t = srs_table(srs, 0);
    return msm_to_bytes(ctx, srs->d_bases, (const uint32_t*)d_scalars, n, batch, out_be_xy, is_inf, &t);
}

int dr_g1_msm_batch(dr_ctx* ctx, const dr_srs* srs, const uint8_t* scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf) {
    TRY(use_ctx(ctx));
    if (n && batch && !scalars) return fail(DR_ERR_INVALID, "null buffer");
    TRY(ctx->scalars.reserve(n * batch * 32));
    if (n && batch) HIP_TRY(hipMemcpyAsync(ctx->scalars.p, scalars, n * batch * 32, hipMemcpyHostToDevice, ctx->stream));
    return dr_g1_msm_batch_dev(ctx, srs, ctx->scalars.p, n, batch, out_be_xy, is_inf);
}

int dr_g1_msm_dev(dr_ctx* ctx, const dr_srs* srs, size_t offset, const void* d_scalars, size_t n, uint8_t out_be_xy[96], int* is_inf) {
    TRY(use_ctx(ctx));
    if (!srs || !out_be_xy) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
    if (offset > srs->count || n > srs->count - offset) return fail(DR_ERR_INVALID, "polynomial degree exceeds SRS size");
    std::vector<drh::G1> res;
    MsmTable t = srs_table(srs, offset);
    TRY(msm_device(ctx, srs->d_bases + offset * 24, (const uint32_t*)d_scalars, n, 1, res, &t));
    g1_result_to_bytes(res[0], out_be_xy, is_inf);
    return DR_OK;
}

int dr_g1_msm(dr_ctx* ctx, const dr_srs* srs, size_t offset, const uint8_t* scalars, size_t n, uint8_t out_be_xy[96], int* is_inf) {
    TRY(use_ctx(ctx));
    if (n && !scalars) return fail(DR_ERR_INVALID, "null buffer");
    TRY(ctx->scalars.reserve(n * 32));
    if (n) HIP_TRY(hipMemcpyAsync(ctx->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return dr_g1_msm_dev(ctx, srs, offset, ctx->scalars.p, n, out_be_xy, is_inf);
}

int dr_g1_msm_points(dr_ctx* ctx, const uint8_t* pts_be_xy, const uint8_t* scalars, size_t n, uint8_t out_be_xy[96], int* is_inf) {
    TRY(use_ctx(ctx));
    if (!out_be_xy) return fail(DR_ERR_INVALID, "null argument");
    if (n == 0) {
        std::memset(out_be_xy, 0, 96);
        if (is_inf) *is_inf = 1;
        return DR_OK;
    }
    if (!pts_be_xy || !scalars) return fail(DR_ERR_INVALID, "null buffer");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(pts_be_xy, n, le, true));
    TRY(ctx->io_a.reserve(n * 96));
    TRY(ctx->scalars.reserve(n * 32));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, le.data(), n * 96, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, ctx->io_a.as<uint32_t>(), (uint32_t)n);
    std::vector<drh::G1> res;
    TRY(msm_device(ctx, ctx->io_a.as<uint32_t>(), ctx->scalars.as<uint32_t>(), n, 1, res));
    g1_result_to_bytes(res[0], out_be_xy, is_inf);
    return DR_OK;
}

int dr_g1_sum(const uint8_t* pts_be_xy, size_t n, uint8_t out_be_xy[96], int* is_inf) {
    if (!out_be_xy || (n && !pts_be_xy)) return fail(DR_ERR_INVALID, "null buffer");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(pts_be_xy, n, le, true));
    drh::G1 acc = drh::G1::inf();
    for (size_t i = 0; i < n; i++) {
        drh::G1 p;
        bool allz = true;
        for (int j = 0; j < 96; j++) if (le[96 * i + j]) { allz = false; break; }
        if (allz) continue;
        drh::Fq::load_le(p.x, le.data() + 96 * i);
        drh::Fq::load_le(p.y, le.data() + 96 * i + 48);
        p.zz = drh::Fq::one();
        p.zzz = drh::Fq::one();
        acc = drh::g1_add(acc, p);
    }
    g1_result_to_bytes(acc, out_be_xy, is_inf);
    return DR_OK;
}

namespace {
int miller_product(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, drh::Fq12& f) {
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(g1_be_xy, n, le, true));
    std::vector<drh::Fq> px, py;
    std::vector<drh::G2Affine> qs;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* q = g2_be + 192 * i;
        drh::G2Affine Q;
        bool allz = true;
        for (int j = 0; j < 192; j++) if (q[j]) { allz = false; break; }
        Q.inf = allz || (q[0] & 0x40);
        bool p_inf = true;
        for (int j = 0; j < 96; j++) if (le[96 * i + j]) { p_inf = false; break; }
        if (Q.inf || p_inf) continue;                       // e(O, Q) = e(P, O) = 1
        // zcash layout: x.c1 || x.c0 || y.c1 || y.c0, 48-byte big-endian each (pcs/srs.py:78-88)
        if (!drh::Fq::load_be(Q.x.c1, q) || !drh::Fq::load_be(Q.x.c0, q + 48) || !drh::Fq::load_be(Q.y.c1, q + 96) ||
            !drh::Fq::load_be(Q.y.c0, q + 144) || !drh::g2_on_curve(Q))
            return fail(DR_ERR_INVALID, "invalid BLS12-381 G2 encoding");
        drh::Fq x, y;
        drh::Fq::load_le(x, le.data() + 96 * i);
        drh::Fq::load_le(y, le.data() + 96 * i + 48);
        px.push_back(x); py.push_back(y); qs.push_back(Q);
    }
    f = drh::multi_miller_loop(px.data(), py.data(), qs.data(), qs.size());
    return DR_OK;
}
}  // namespace

int dr_pairing_check(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, int* ok) {
    if (!ok || (n && (!g1_be_xy || !g2_be))) return fail(DR_ERR_INVALID, "null buffer");
    drh::Fq12 f;
    TRY(miller_product(g1_be_xy, g2_be, n, f));
    *ok = drh::final_exponentiation_check(f) == drh::Fq12::one() ? 1 : 0;
    return DR_OK;
}

// diagnostic: the fast final exponentiation (Frobenius maps + x-chain, exponent 3(p^12-1)/r) against the plain
// square-and-multiply one; *consistent = 1 iff fast == reference^3 for this Miller-loop product
int dr_pairing_selfcheck(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, int* consistent) {
    if (!consistent || (n && (!g1_be_xy || !g2_be))) return fail(DR_ERR_INVALID, "null buffer");
    drh::Fq12 f;
    TRY(miller_product(g1_be_xy, g2_be, n, f));
    drh::Fq12 ref = drh::final_exponentiation(f);
    *consistent = drh::final_exponentiation_check(f) == ref * ref * ref ? 1 : 0;
    return DR_OK;
}

int dr_g1_compress(const uint8_t xy[96], int is_inf, uint8_t out[48]) {
    if (!xy || !out) return fail(DR_ERR_INVALID, "null buffer");
    bool inf = is_inf != 0 || (xy[0] & 0x40);
    if (!inf) {
        bool allz = true;
        for (int j = 0; j < 96; j++) if (xy[j]) { allz = false; break; }
        inf = allz;
    }
    if (inf) {
        std::memset(out, 0, 48);
        out[0] = 0xc0;
        return DR_OK;
    }
    drh::Fq x, y;
    if (!drh::Fq::load_be(x, xy) || !drh::Fq::load_be(y, xy + 48)) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
    std::memcpy(out, xy, 48);
    out[0] |= 0x80;
    drh::Fq ys = y.from_mont(), nys = y.neg().from_mont();
    if (drh::Fq::gt_std(ys, nys)) out[0] |= 0x20;
    return DR_OK;
}

int dr_g1_decompress(const uint8_t in[48], uint8_t out_xy[96], int* is_inf) {
    if (!in || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    uint8_t flags = in[0] >> 5;
    if (!(flags & 4)) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
    uint8_t xb[48];
    std::memcpy(xb, in, 48);
    xb[0] &= 0x1f;
    if (flags & 2) {
        bool allz = true;
        for (int j = 0; j < 48; j++) if (xb[j]) { allz = false; break; }
        if (!allz || (flags & 1)) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
        std::memset(out_xy, 0, 96);
        if (is_inf) *is_inf = 1;
        return DR_OK;
    }
    drh::Fq x;
    if (!drh::Fq::load_be(x, xb)) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
    drh::Fq rhs = x.sqr() * x + drh::Fq::from_u64(4);
    // p = 3 mod 4: y = rhs^((p+1)/4)
    static const uint64_t E[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                                  0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
    drh::Fq y = rhs.pow(E, 6);
    if (y.sqr() != rhs) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
    drh::Fq ny = y.neg();
    bool y_larger = drh::Fq::gt_std(y.from_mont(), ny.from_mont());
    if (y_larger != ((flags & 1) != 0)) y = ny;
    std::memcpy(out_xy, xb, 48);
    y.store_be(out_xy + 48);
    if (is_inf) *is_inf = 0;
    return DR_OK;
}

// KZG.decompress_g1 for n points in one launch (zcash 48-byte encodings -> BE x||y records; ok[i] = 0 for malformed
// encodings, infinity decodes to an all-zero record with ok = 1)
int dr_g1_decompress_batch(dr_ctx* ctx, const uint8_t* enc, size_t n, uint8_t* out_be_xy, uint8_t* ok) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (!enc || !out_be_xy || !ok) return fail(DR_ERR_INVALID, "null buffer");
    if (n >= (1ull << 28)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(ctx->vfy_in.reserve(n * 48));
    TRY(ctx->vfy_bases.reserve(n * 96));
    TRY(ctx->vfy_std.reserve(n * 96));
    TRY(ctx->io_c.reserve(n * 4));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->vfy_in.p, enc, n * 48, hipMemcpyHostToDevice, st));
    TRY(launch(ctx, "k_g1_decompress", [&] {
        hipLaunchKernelGGL(dr::k_g1_decompress, dim3(div_up(n, 64)), dim3(64), 0, st, ctx->vfy_in.as<uint8_t>(), ctx->vfy_bases.as<uint32_t>(),
                           ctx->io_c.as<uint32_t>(), (uint32_t)n);
        hipLaunchKernelGGL(dr::k_g1_bases_from_mont, dim3(div_up(n, 256)), dim3(256), 0, st, ctx->vfy_bases.as<uint32_t>(), ctx->vfy_std.as<uint32_t>(),
                           (uint32_t)n);
    }));
    std::vector<uint8_t> le(n * 96);
    std::vector<uint32_t> flags(n);
    HIP_TRY(hipMemcpyAsync(le.data(), ctx->vfy_std.p, n * 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(flags.data(), ctx->io_c.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->prof) TRY(prof_collect(ctx));
    for (size_t i = 0; i < n; i++) {
        ok[i] = flags[i] ? 1 : 0;
        for (int j = 0; j < 48; j++) {
            out_be_xy[96 * i + j] = le[96 * i + 47 - j];
            out_be_xy[96 * i + 48 + j] = le[96 * i + 95 - j];
        }
    }
    return DR_OK;
}

int dr_g1_serialize_check(const uint8_t xy[96]) {
    std::vector<uint8_t> le;
    return g1_be_to_le_limbs(xy, 1, le, true);
}

// ------------------------------------------------------------------------------- seam C
int dr_ntt_dev(dr_ctx* ctx, void* d_data, unsigned log2n, size_t batch, const uint8_t omega[32], const uint8_t* scale) {
    TRY(use_ctx(ctx));
    if (!omega) return fail(DR_ERR_INVALID, "null omega");
    if (log2n < 1 || log2n > 24) return fail(DR_ERR_INVALID, "native NTT plan size must be a power of two >= 2");
    if (batch == 0) return DR_OK;
    drh::Fr w, sc;
    if (!drh::Fr::load_le(w, omega)) return fail(DR_ERR_INVALID, "omega is not a canonical field element");
    if (scale && !drh::Fr::load_le(sc, scale)) return fail(DR_ERR_INVALID, "scale is not a canonical field element");
    return dr::ntt_run(ctx->stream, [&](const char* name, auto&& f) { return launch(ctx, name, f); }, ctx->twiddles,
                       ctx->io_b, (uint32_t*)d_data, log2n, batch, w, scale ? &sc : nullptr,
                       [&]() -> int { HIP_TRY(hipStreamSynchronize(ctx->stream)); if (ctx->prof) TRY(prof_collect(ctx)); return DR_OK; });
}

int dr_ntt(dr_ctx* ctx, uint8_t* data, unsigned log2n, size_t batch, const uint8_t omega[32], const uint8_t* scale) {
    TRY(use_ctx(ctx));
    if (!data) return fail(DR_ERR_INVALID, "null buffer");
    if (log2n < 1 || log2n > 24) return fail(DR_ERR_INVALID, "native NTT plan size must be a power of two >= 2");
    size_t bytes = ((size_t)32 << log2n) * batch;
    if (bytes == 0) return DR_OK;
    TRY(check_fr_elems(data, bytes / 32, "NTT input"));
    TRY(ctx->io_a.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    TRY(dr_ntt_dev(ctx, ctx->io_a.p, log2n, batch, omega, scale));
    HIP_TRY(hipMemcpyAsync(data, ctx->io_a.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------- batched ring prover
struct dr_ring_prover {
    dr_ctx* ctx = nullptr;
    const dr_srs* srs = nullptr;
    int curve = dr::CV_BANDERSNATCH;     // which twisted Edwards curve the ring's keys live on
    int device = 0;                      // copy of ctx->device: destroy may run after the context is gone (finalizers)
    dr::RingConsts rc{};
    drh::Fr omega_n, omega_4n;          // Montgomery
    const dr_srs* ps_srs = nullptr;      // prefix-summed Lagrange bases of this domain (owned by srs->lagrange_prefix)
    dr_ctx* aux_ctx = nullptr;           // second stream of the same GPU: dr_ringvrf_prove_batch runs the Pedersen tail on it
    // per-ring tables
    Scratch ring_pts_mont;              // [N][16]
    Scratch fixed_coef;                 // [3][N][8] std (px, py, s coefficients)
    Scratch fixed4, lag4, not_last;     // Montgomery tables on the 4N domain
    uint8_t root[3 * 96];
    int root_inf[3];
    // per-batch state
    size_t batch = 0;
    Scratch idx, blind, zk, chain_ext, prefix, chain_aff, cnt, relation, rps, cols, wit4, alphas, agg, q, zetas, evals, ks, lin,
        nus, aggo, chunkv, quot1, quot2, diffs;
};

namespace {

int ring_ntt(dr_ring_prover* p, uint32_t* d_data, unsigned log2n, size_t batch, bool inverse, bool in_mont = false, bool out_mont = false,
             const uint32_t* d_src = nullptr, int pad = 0) {
    dr_ctx* ctx = p->ctx;
    const drh::Fr& w = log2n == p->rc.log2n ? p->omega_n : p->omega_4n;
    drh::Fr wi = inverse ? w.inv() : w;
    drh::Fr scale;
    if (inverse) scale = drh::Fr::from_u64((uint64_t)1 << log2n).inv();
    // dr_ntt limits one launch to 65535 transforms (grid.y): split larger batches
    for (size_t done = 0; done < batch;) {
        size_t take = std::min<size_t>(batch - done, 65535);
        int rc = dr::ntt_run(ctx->stream, [&](const char* name, auto&& f) { return launch(ctx, name, f); }, ctx->twiddles, ctx->io_b,
                             d_data + done * ((size_t)8 << log2n), log2n, take, wi, inverse ? &scale : nullptr,
                             [&]() -> int { return DR_OK; }, in_mont, out_mont,
                             d_src ? d_src + done * ((size_t)8 << (log2n - pad)) : nullptr, pad);
        if (rc != DR_OK) return rc == DR_ERR_NOMEM ? fail(rc, "out of device memory in NTT") : fail(rc, "NTT launch failed");
        done += take;
    }
    return DR_OK;
}

dr::FrArg arg_of(const drh::Fr& v) { return dr::to_arg(v); }

// PS_j = sum_{i<=j} L_i(tau) G for the size-2^log2n domain, as a derived dr_srs with its own window table.
// One batched MSM (N MSMs of N points) over the monomial SRS; cached in srs->lagrange_prefix.
int lagrange_prefix_srs(dr_ctx* ctx, const dr_srs* srs_c, unsigned log2n, const drh::Fr& omega_n, const dr_srs** out) {
    dr_srs* srs = const_cast<dr_srs*>(srs_c);
    std::lock_guard<std::mutex> lock(srs->derive_mutex);
    auto hit = srs->lagrange_prefix.find(log2n);
    if (hit != srs->lagrange_prefix.end()) { *out = hit->second; return DR_OK; }
    const uint32_t n = 1u << log2n;
    if (srs->count < n) return fail(DR_ERR_INVALID, "polynomial degree exceeds SRS size");
    hipStream_t st = ctx->stream;
    Scratch mat;
    TRY(mat.reserve((size_t)n * n * 32));
    hipLaunchKernelGGL(dr::k_ring_ps_scalars, dim3(div_up(n, 64)), dim3(64), 0, st, mat.as<uint32_t>(), n, arg_of(omega_n.inv()),
                       arg_of(drh::Fr::from_u64(n).inv()));
    std::vector<uint8_t> be((size_t)n * 96);
    std::vector<int> inf(n);
    MsmTable t = srs_table(srs, 0);
    int rc = DR_OK;
    // keep each launch within the 32-bit digit / bucket index limits
    const size_t step = std::max<size_t>(1, std::min<size_t>(n, (size_t)1 << (26 - log2n)));
    for (size_t done = 0; done < n && rc == DR_OK; done += step) {
        size_t take = std::min<size_t>(step, n - done);
        rc = msm_to_bytes(ctx, srs->d_bases, mat.as<uint32_t>() + done * n * 8, n, take, be.data() + done * 96, inf.data() + done, &t);
    }
    mat.release();
    if (rc != DR_OK) return rc;
    dr_srs* ps = nullptr;
    TRY(dr_srs_load(ctx, be.data(), n, &ps));
    // the by-parts scalars are sparse (~1.4k non-zero of N per column): a narrower window keeps the bucket sets, and with
    // them the bucket reduction, small (DOTRING_PS_WINDOW, default 10)
    int ps_bits = 10;
    if (const char* e = std::getenv("DOTRING_PS_WINDOW")) { int v = std::atoi(e); if (v >= 7 && v <= 16) ps_bits = v; }
    rc = dr_srs_precompute(ctx, ps, ps_bits);
    if (rc != DR_OK) { dr_srs_destroy(ps); return rc; }
    // the by-parts scalars are sparse: in the comb kernel a wave skips a slot only when all 64 lanes have a zero digit,
    // while the bucket method never sees zero digits at all — DOTRING_PS_COMB=1 builds the comb table anyway
    if (srs->d_comb && ps_bits <= 14 && std::getenv("DOTRING_PS_COMB") && std::atoi(std::getenv("DOTRING_PS_COMB")) != 0)
        (void)dr_srs_precompute_comb(ctx, ps);
    srs->lagrange_prefix[log2n] = ps;
    *out = ps;
    return DR_OK;
}

}  // namespace

extern "C" {

int dr_ring_prover_create(dr_ctx* ctx, const dr_srs* srs, unsigned log2n, uint32_t max_ring, const uint8_t omega_n[32],
                          const uint8_t omega_4n[32], const uint8_t* nm_points_xy, const uint8_t seed_xy[64], dr_ring_prover** out) {
    return dr_ring_prover_create_te(ctx, dr::CV_BANDERSNATCH, srs, log2n, max_ring, omega_n, omega_4n, nm_points_xy, seed_xy, out);
}

int dr_ring_prover_create_te(dr_ctx* ctx, int curve, const dr_srs* srs, unsigned log2n, uint32_t max_ring, const uint8_t omega_n[32],
                             const uint8_t omega_4n[32], const uint8_t* nm_points_xy, const uint8_t seed_xy[64], dr_ring_prover** out) {
    TRY(use_ctx(ctx));
    if (!out || !srs || !omega_n || !omega_4n || !nm_points_xy || !seed_xy) return fail(DR_ERR_INVALID, "null argument");
    *out = nullptr;
    TRY(check_curve(curve));
    if (log2n < 9 || log2n > 12) return fail(DR_ERR_INVALID, "domain_size must be between 512 and 4096");
    const uint32_t n = 1u << log2n, m = 4 * n;
    if (max_ring + drh::te_curve(curve)->scalar_bits + 4 > n) return fail(DR_ERR_INVALID, "max_ring_size exceeds supported size");
    if (srs->count < 3 * (size_t)n + 1) return fail(DR_ERR_INVALID, "polynomial degree exceeds SRS size");
    TRY(check_fr_elems(nm_points_xy, 2 * (size_t)n, "ring point"));
    TRY(check_fr_elems(seed_xy, 2, "seed point"));
    dr_ring_prover* p = new (std::nothrow) dr_ring_prover();
    if (!p) return fail(DR_ERR_NOMEM, "out of host memory");
    std::unique_ptr<dr_ring_prover, void (*)(dr_ring_prover*)> guard(p, [](dr_ring_prover* q) { dr_ring_prover_destroy(q); });
    p->ctx = ctx;
    p->device = ctx->device;
    p->srs = srs;
    p->curve = curve;
    if (!drh::Fr::load_le(p->omega_n, omega_n) || !drh::Fr::load_le(p->omega_4n, omega_4n))
        return fail(DR_ERR_INVALID, "omega is not a canonical field element");
    if (std::getenv("DOTRING_WITNESS_BY_PARTS") == nullptr || std::atoi(std::getenv("DOTRING_WITNESS_BY_PARTS")) != 0)
        TRY(lagrange_prefix_srs(ctx, srs, log2n, p->omega_n, &p->ps_srs));
    dr::RingConsts& rc = p->rc;
    rc.log2n = log2n; rc.n = n; rc.max_ring = max_ring; rc.rows = n - 4;
    drh::Fr sx, sy;
    drh::Fr::load_le(sx, seed_xy);
    drh::Fr::load_le(sy, seed_xy + 32);
    rc.seed_x = arg_of(sx); rc.seed_y = arg_of(sy);
    rc.omega = arg_of(p->omega_n);
    // domain[-k] = w^-k ; last_x = w^(N-4) = w^-4
    drh::Fr winv = p->omega_n.inv();
    drh::Fr w1 = winv, w2 = winv * winv, w3 = w2 * winv, w4 = w2 * w2;
    rc.last_x = arg_of(w4);
    // tail(X) = (X - w^-1)(X - w^-2)(X - w^-3)
    drh::Fr e1 = w1 + w2 + w3, e2 = w1 * w2 + w1 * w3 + w2 * w3, e3 = w1 * w2 * w3;
    rc.tail[0] = arg_of(e3.neg()); rc.tail[1] = arg_of(e2); rc.tail[2] = arg_of(e1.neg()); rc.tail[3] = arg_of(drh::Fr::one());
    hipStream_t st = ctx->stream;
    // ring points -> Montgomery table ; fixed evaluation columns
    TRY(p->ring_pts_mont.reserve((size_t)n * 64));
    HIP_TRY(hipMemcpyAsync(p->ring_pts_mont.p, nm_points_xy, (size_t)n * 64, hipMemcpyHostToDevice, st));
    TRY(p->fixed_coef.reserve((size_t)3 * n * 32));
    hipLaunchKernelGGL(dr::k_ring_fixed_evals, dim3(div_up(n, 256)), dim3(256), 0, st, p->ring_pts_mont.as<uint32_t>(), n, max_ring,
                       p->fixed_coef.as<uint32_t>());
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up((size_t)2 * n, 256)), dim3(256), 0, st, p->ring_pts_mont.as<uint32_t>(), (size_t)2 * n);
    TRY(ring_ntt(p, p->fixed_coef.as<uint32_t>(), log2n, 3, true));            // interpolate px, py, s
    {
        MsmTable t = srs_table(srs, 0);
        TRY(msm_to_bytes(ctx, srs->d_bases, p->fixed_coef.as<uint32_t>(), n, 3, p->root, p->root_inf, &t));
    }
    // 4N-domain tables (Montgomery)
    TRY(p->fixed4.reserve((size_t)3 * m * 32));
    hipLaunchKernelGGL(dr::k_ring_pad, dim3(div_up((size_t)3 * m, 256)), dim3(256), 0, st, p->fixed_coef.as<uint32_t>(), n,
                       p->fixed4.as<uint32_t>(), m, (size_t)3);
    TRY(ring_ntt(p, p->fixed4.as<uint32_t>(), log2n + 2, 3, false));
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up((size_t)3 * m, 256)), dim3(256), 0, st, p->fixed4.as<uint32_t>(), (size_t)3 * m);
    Scratch lagc;
    TRY(lagc.reserve((size_t)2 * n * 32));
    hipLaunchKernelGGL(dr::k_ring_lagrange, dim3(div_up(n, 256)), dim3(256), 0, st, lagc.as<uint32_t>(), n,
                       arg_of(drh::Fr::from_u64(n).inv()), arg_of(w4.inv()));
    TRY(p->lag4.reserve((size_t)2 * m * 32));
    hipLaunchKernelGGL(dr::k_ring_pad, dim3(div_up((size_t)2 * m, 256)), dim3(256), 0, st, lagc.as<uint32_t>(), n, p->lag4.as<uint32_t>(), m, (size_t)2);
    TRY(ring_ntt(p, p->lag4.as<uint32_t>(), log2n + 2, 2, false));
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up((size_t)2 * m, 256)), dim3(256), 0, st, p->lag4.as<uint32_t>(), (size_t)2 * m);
    TRY(p->not_last.reserve((size_t)m * 32));
    hipLaunchKernelGGL(dr::k_ring_not_last, dim3(div_up(m, 256)), dim3(256), 0, st, p->not_last.as<uint32_t>(), m, arg_of(p->omega_4n), arg_of(w4));
    HIP_TRY(hipStreamSynchronize(st));
    lagc.release();
    HIP_TRY(hipGetLastError());
    guard.release();
    *out = p;
    return DR_OK;
}

void dr_ring_prover_destroy(dr_ring_prover* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (Scratch* s : {&p->ring_pts_mont, &p->fixed_coef, &p->fixed4, &p->lag4, &p->not_last, &p->idx, &p->blind, &p->zk, &p->chain_ext,
                       &p->prefix, &p->chain_aff, &p->cnt, &p->relation, &p->rps, &p->cols, &p->wit4, &p->alphas, &p->agg, &p->q, &p->zetas,
                       &p->evals, &p->ks, &p->lin, &p->nus, &p->aggo, &p->chunkv, &p->quot1, &p->quot2, &p->diffs})
        s->release();
    if (p->aux_ctx) {
        if (p->ctx && ctx_alive(p->ctx)) {
            auto& hs = p->ctx->helpers;
            hs.erase(std::remove(hs.begin(), hs.end(), p->aux_ctx), hs.end());
        }
        dr_ctx_destroy(p->aux_ctx);
    }
    delete p;
}

int dr_ring_prover_root(const dr_ring_prover* p, uint8_t out_commitments[3 * 96], int is_inf[3]) {
    if (!p || !out_commitments) return fail(DR_ERR_INVALID, "null argument");
    std::memcpy(out_commitments, p->root, sizeof p->root);
    if (is_inf) std::memcpy(is_inf, p->root_inf, sizeof p->root_inf);
    return DR_OK;
}

int dr_ring_prover_fixed_coeffs(dr_ring_prover* p, uint8_t* out /* 3*N*32: px, py, s */) {
    if (!p || !out) return fail(DR_ERR_INVALID, "null argument");
    TRY(use_ctx(p->ctx));
    HIP_TRY(hipMemcpyAsync(out, p->fixed_coef.p, (size_t)3 * p->rc.n * 32, hipMemcpyDeviceToHost, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    return DR_OK;
}

// phase A: witness columns, interpolation, 4 commitments per proof (order b, accip, accx, accy)
int dr_ring_prove_witness(dr_ring_prover* p, size_t batch, const uint32_t* producer_index, const uint8_t* blinding, const uint8_t* zk_rows,
                          uint8_t* out_relation_xy, uint8_t* out_commitments, int* is_inf) {
    if (!p || !producer_index || !blinding || !out_relation_xy || !out_commitments) return fail(DR_ERR_INVALID, "null argument");
    dr_ctx* ctx = p->ctx;
    TRY(use_ctx(ctx));
    if (batch == 0 || batch > 16383) return fail(DR_ERR_INVALID, "batch must be in 1..16383");
    const dr::RingConsts& rc = p->rc;
    const uint32_t n = rc.n;
    for (size_t i = 0; i < batch; i++)
        if (producer_index[i] >= rc.max_ring) return fail(DR_ERR_INVALID, "producer key is not in ring");
    {   // blinding factors are scalars of the ring's curve: their bits select rows max_ring .. max_ring + bits(n) - 1
        const drh::Mod256& order = drh::te_curve(p->curve)->n;
        for (size_t i = 0; i < batch; i++) {
            uint64_t v[4];
            drh::load_le32(blinding + 32 * i, v);
            if (drh::Mod256::geq(v, order.m)) return fail(DR_ERR_INVALID, "blinding factor is not a canonical scalar of the curve");
        }
    }
    if (zk_rows) TRY(check_fr_elems(zk_rows, batch * 12, "hidden row"));
    p->batch = batch;
    hipStream_t st = ctx->stream;
    TRY(p->idx.reserve(batch * 4));
    TRY(p->blind.reserve(batch * 32));
    TRY(p->chain_ext.reserve(batch * dr::RING_CHAIN * 128));
    TRY(p->prefix.reserve(batch * dr::RING_CHAIN * 32));
    TRY(p->chain_aff.reserve(batch * dr::RING_CHAIN * 64));
    TRY(p->cnt.reserve(batch * 4));
    TRY(p->relation.reserve(batch * 64));
    TRY(p->rps.reserve(batch * 64));
    TRY(p->cols.reserve(batch * 4 * (size_t)n * 32));
    // the columns are built in evaluation form in the wit4 buffer (unused until the quotient phase, which needs 4x this size anyway)
    // and interpolated from there into `cols`: the inverse NTT's first pass cannot run in place, a separate source saves its
    // temporary and the copy back
    TRY(p->wit4.reserve(batch * 4 * (size_t)n * 4 * 32));
    uint32_t* col_evals = p->wit4.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(p->idx.p, producer_index, batch * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->blind.p, blinding, batch * 32, hipMemcpyHostToDevice, st));
    if (zk_rows) {
        TRY(p->zk.reserve(batch * 12 * 32));
        HIP_TRY(hipMemcpyAsync(p->zk.p, zk_rows, batch * 12 * 32, hipMemcpyHostToDevice, st));
    }
    TRY(launch(ctx, "k_ring_chain", [&] {
        if (g_chain_wave)
            LAUNCH_CV(p->curve, dr::k_ring_chain_wave, dim3((unsigned)batch), dim3(64), 0, st, p->ring_pts_mont.as<uint32_t>(), p->idx.as<uint32_t>(),
                               p->blind.as<uint32_t>(), rc, (uint32_t)batch, p->chain_ext.as<uint32_t>(), p->chain_aff.as<uint32_t>(),
                               p->cnt.as<uint32_t>());
        else
            LAUNCH_CV(p->curve, dr::k_ring_chain, dim3(div_up(batch, 64)), dim3(64), 0, st, p->ring_pts_mont.as<uint32_t>(), p->idx.as<uint32_t>(),
                               p->blind.as<uint32_t>(), rc, (uint32_t)batch, p->chain_ext.as<uint32_t>(), p->prefix.as<uint32_t>(),
                               p->chain_aff.as<uint32_t>(), p->cnt.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_ring_columns", [&] {
        hipLaunchKernelGGL(dr::k_ring_relations, dim3(div_up(batch, 64)), dim3(64), 0, st, p->chain_aff.as<uint32_t>(), p->cnt.as<uint32_t>(),
                           (uint32_t)batch, p->relation.as<uint32_t>(), p->rps.as<uint32_t>());
        hipLaunchKernelGGL(dr::k_ring_columns, dim3(div_up(batch * n, 256)), dim3(256), 0, st, p->idx.as<uint32_t>(), p->blind.as<uint32_t>(),
                           p->chain_aff.as<uint32_t>(), zk_rows ? p->zk.as<uint32_t>() : nullptr, rc, (uint32_t)batch, col_evals);
    }));
    HIP_TRY(hipMemcpyAsync(out_relation_xy, p->relation.p, batch * 64, hipMemcpyDeviceToHost, st));
    if (p->ps_srs) {
        // commit in evaluation form by summation by parts (sparse scalars), then interpolate for the later phases
        TRY(p->diffs.reserve(batch * 4 * (size_t)n * 32));
        TRY(launch(ctx, "k_ring_diff", [&] {
            hipLaunchKernelGGL(dr::k_ring_diff, dim3(div_up(batch * 4 * n, 256)), dim3(256), 0, st, col_evals, n, batch * 4,
                               p->diffs.as<uint32_t>());
        }));
        TRY(ring_ntt(p, p->cols.as<uint32_t>(), rc.log2n, batch * 4, true, false, false, col_evals, 0));
        MsmTable t = srs_table(p->ps_srs, 0);
        return msm_to_bytes(ctx, p->ps_srs->d_bases, p->diffs.as<uint32_t>(), n, batch * 4, out_commitments, is_inf, &t);
    }
    TRY(ring_ntt(p, p->cols.as<uint32_t>(), rc.log2n, batch * 4, true, false, false, col_evals, 0));
    MsmTable t = srs_table(p->srs, 0);
    return msm_to_bytes(ctx, p->srs->d_bases, p->cols.as<uint32_t>(), n, batch * 4, out_commitments, is_inf, &t);
}

// phase B: constraints on the 4N domain, aggregation with the alphas, quotient polynomial and its commitment
int dr_ring_prove_quotient(dr_ring_prover* p, size_t batch, const uint8_t* alphas, uint8_t* out_cq, int* is_inf) {
    if (!p || !alphas || !out_cq) return fail(DR_ERR_INVALID, "null argument");
    dr_ctx* ctx = p->ctx;
    TRY(use_ctx(ctx));
    if (batch != p->batch || batch == 0) return fail(DR_ERR_INVALID, "phase called with a different batch size");
    TRY(check_fr_elems(alphas, batch * 7, "alpha"));
    const dr::RingConsts& rc = p->rc;
    const uint32_t n = rc.n, m = 4 * n, qn = 3 * n + 1;
    hipStream_t st = ctx->stream;
    TRY(p->alphas.reserve(batch * 7 * 32));
    TRY(p->wit4.reserve(batch * 4 * (size_t)m * 32));
    TRY(p->agg.reserve(batch * (size_t)m * 32));
    TRY(p->q.reserve(batch * (size_t)qn * 32));
    HIP_TRY(hipMemcpyAsync(p->alphas.p, alphas, batch * 7 * 32, hipMemcpyHostToDevice, st));
    // 7 alphas per proof are read by every point of the 4N domain: convert them to Montgomery form once
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up(batch * 7, 256)), dim3(256), 0, st, p->alphas.as<uint32_t>(), batch * 7);
    // N coefficients per column -> evaluations on the 4N domain: the NTT reads the columns directly (zero padding implied) and
    // leaves the evaluations in Montgomery form
    TRY(ring_ntt(p, p->wit4.as<uint32_t>(), rc.log2n + 2, batch * 4, false, false, true, p->cols.as<uint32_t>(), 2));
    TRY(launch(ctx, "k_ring_constraints", [&] {
        LAUNCH_CV(p->curve, dr::k_ring_constraints, dim3(div_up(batch * m, 256)), dim3(256), 0, st, p->wit4.as<uint32_t>(), p->fixed4.as<uint32_t>(),
                           p->lag4.as<uint32_t>(), p->not_last.as<uint32_t>(), p->alphas.as<uint32_t>(), p->rps.as<uint32_t>(), rc,
                           (uint32_t)batch, p->agg.as<uint32_t>());
    }));
    // the constraint kernel wrote Montgomery form; the coefficients land in the (now free) wit4 buffer: the first pass cannot
    // run in place, so a separate output saves the temporary and the copy back
    TRY(ring_ntt(p, p->wit4.as<uint32_t>(), rc.log2n + 2, batch, true, true, false, p->agg.as<uint32_t>(), 0));
    TRY(launch(ctx, "k_ring_quotient", [&] {
        hipLaunchKernelGGL(dr::k_ring_quotient, dim3(div_up(batch * qn, 256)), dim3(256), 0, st, p->wit4.as<uint32_t>(), rc, (uint32_t)batch,
                           p->q.as<uint32_t>());
    }));
    MsmTable t = srs_table(p->srs, 0);
    return msm_to_bytes(ctx, p->srs->d_bases, p->q.as<uint32_t>(), qn, batch, out_cq, is_inf, &t);
}

// phase C1: register evaluations at zeta, linearisation polynomial and its value at zeta*omega
int dr_ring_prove_evals(dr_ring_prover* p, size_t batch, const uint8_t* zetas, uint8_t* out_evals /* B*8*32 */) {
    if (!p || !zetas || !out_evals) return fail(DR_ERR_INVALID, "null argument");
    dr_ctx* ctx = p->ctx;
    TRY(use_ctx(ctx));
    if (batch != p->batch || batch == 0) return fail(DR_ERR_INVALID, "phase called with a different batch size");
    TRY(check_fr_elems(zetas, batch, "zeta"));
    const dr::RingConsts& rc = p->rc;
    const uint32_t n = rc.n;
    hipStream_t st = ctx->stream;
    TRY(p->zetas.reserve(batch * 32));
    TRY(p->evals.reserve(batch * 8 * 32));
    TRY(p->ks.reserve(batch * 3 * 32));
    TRY(p->lin.reserve(batch * (size_t)n * 32));
    HIP_TRY(hipMemcpyAsync(p->zetas.p, zetas, batch * 32, hipMemcpyHostToDevice, st));
    TRY(launch(ctx, "k_ring_eval", [&] {
        hipLaunchKernelGGL(dr::k_ring_eval, dim3(7, (unsigned)batch), dim3(dr::EV_BLOCK), 0, st, p->fixed_coef.as<uint32_t>(), 3u,
                           p->cols.as<uint32_t>(), 4u, n, p->zetas.as<uint32_t>(), 0, rc, p->evals.as<uint32_t>(), 8u, 0u);
    }));
    TRY(launch(ctx, "k_ring_linpoly", [&] {
        LAUNCH_CV(p->curve, dr::k_ring_lin_scalars, dim3(div_up(batch, 64)), dim3(64), 0, st, p->evals.as<uint32_t>(), p->alphas.as<uint32_t>(),
                           p->zetas.as<uint32_t>(), rc, (uint32_t)batch, p->ks.as<uint32_t>());
        hipLaunchKernelGGL(dr::k_ring_linpoly, dim3(div_up(batch * n, 256)), dim3(256), 0, st, p->cols.as<uint32_t>(), p->ks.as<uint32_t>(), n,
                           (uint32_t)batch, p->lin.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_ring_eval", [&] {
        hipLaunchKernelGGL(dr::k_ring_eval, dim3(1, (unsigned)batch), dim3(dr::EV_BLOCK), 0, st, (const uint32_t*)nullptr, 0u, p->lin.as<uint32_t>(),
                           1u, n, p->zetas.as<uint32_t>(), 1, rc, p->evals.as<uint32_t>(), 8u, 7u);
    }));
    HIP_TRY(hipMemcpyAsync(out_evals, p->evals.p, batch * 8 * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

// phase C2: nu-aggregated polynomial, the two opening quotients and their commitments (agg at zeta, lin at zeta*omega)
int dr_ring_prove_openings(dr_ring_prover* p, size_t batch, const uint8_t* nus, uint8_t* out_openings /* B*2*96 */, int* is_inf /* B*2 */) {
    if (!p || !nus || !out_openings) return fail(DR_ERR_INVALID, "null argument");
    dr_ctx* ctx = p->ctx;
    TRY(use_ctx(ctx));
    if (batch != p->batch || batch == 0) return fail(DR_ERR_INVALID, "phase called with a different batch size");
    TRY(check_fr_elems(nus, batch * 8, "nu"));
    const dr::RingConsts& rc = p->rc;
    const uint32_t n = rc.n, qn = 3 * n + 1;
    hipStream_t st = ctx->stream;
    TRY(p->nus.reserve(batch * 8 * 32));
    TRY(p->aggo.reserve(batch * (size_t)qn * 32));
    const uint32_t nch1 = (qn + dr::SD_CHUNK - 1) / dr::SD_CHUNK, nch2 = (n + dr::SD_CHUNK - 1) / dr::SD_CHUNK;
    TRY(p->chunkv.reserve(batch * (size_t)nch1 * 32));
    // both quotients of all proofs share ONE batched MSM: [2*batch][3N] scalar vectors, the short second quotient
    // zero-padded (zero scalars produce no digits) — one sort / accumulate / reduce / affine pipeline instead of two
    TRY(p->quot1.reserve(2 * batch * (size_t)(qn - 1) * 32));
    TRY(p->quot2.reserve(batch * (size_t)(n - 1) * 32));
    HIP_TRY(hipMemcpyAsync(p->nus.p, nus, batch * 8 * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up(batch * 8, 256)), dim3(256), 0, st, p->nus.as<uint32_t>(), batch * 8);     // multipliers: Montgomery form
    TRY(launch(ctx, "k_ring_aggpoly", [&] {
        hipLaunchKernelGGL(dr::k_ring_aggpoly, dim3(div_up(batch * qn, 256)), dim3(256), 0, st, p->fixed_coef.as<uint32_t>(), p->cols.as<uint32_t>(),
                           p->q.as<uint32_t>(), p->nus.as<uint32_t>(), n, (uint32_t)batch, p->aggo.as<uint32_t>());
    }));
    auto syndiv = [&](const uint32_t* poly, uint32_t len, int mul_omega, uint32_t* quot, uint32_t nch) -> int {
        return launch(ctx, "k_syndiv", [&] {
            hipLaunchKernelGGL(dr::k_syndiv_local, dim3(div_up(batch * nch, 128)), dim3(128), 0, st, poly, len, p->zetas.as<uint32_t>(), mul_omega, rc,
                               (uint32_t)batch, p->chunkv.as<uint32_t>());
            hipLaunchKernelGGL(dr::k_syndiv_link, dim3(div_up(batch, 64)), dim3(64), 0, st, p->chunkv.as<uint32_t>(), len, p->zetas.as<uint32_t>(),
                               mul_omega, rc, (uint32_t)batch);
            hipLaunchKernelGGL(dr::k_syndiv_write, dim3(div_up(batch * nch, 128)), dim3(128), 0, st, poly, len, p->zetas.as<uint32_t>(), mul_omega, rc,
                               (uint32_t)batch, p->chunkv.as<uint32_t>(), quot);
        });
    };
    TRY(syndiv(p->aggo.as<uint32_t>(), qn, 0, p->quot1.as<uint32_t>(), nch1));
    TRY(syndiv(p->lin.as<uint32_t>(), n, 1, p->quot2.as<uint32_t>(), nch2));
    TRY(launch(ctx, "k_ring_pad", [&] {
        hipLaunchKernelGGL(dr::k_ring_pad, dim3(div_up(batch * (size_t)(qn - 1), 256)), dim3(256), 0, st, p->quot2.as<uint32_t>(), n - 1,
                           p->quot1.as<uint32_t>() + batch * (size_t)(qn - 1) * 8, qn - 1, batch);
    }));
    std::vector<uint8_t> o(2 * batch * 96);
    std::vector<int> inf(2 * batch);
    MsmTable t = srs_table(p->srs, 0);
    t.short_from = (uint32_t)batch;          // the second half of the vectors (quot2) has N - 1 coefficients, the rest is padding
    t.n_short = n - 1;
    TRY(msm_to_bytes(ctx, p->srs->d_bases, p->quot1.as<uint32_t>(), qn - 1, 2 * batch, o.data(), inf.data(), &t));
    for (size_t b = 0; b < batch; b++) {
        std::memcpy(out_openings + 192 * b, o.data() + 96 * b, 96);
        std::memcpy(out_openings + 192 * b + 96, o.data() + 96 * (batch + b), 96);
        if (is_inf) { is_inf[2 * b] = inf[b]; is_inf[2 * b + 1] = inf[batch + b]; }
    }
    return DR_OK;
}

// ---- host hashing exposed for tests and callers that batch their own transcripts ----------------------------------
int dr_host_hash(int kind, const uint8_t* data, size_t len, uint8_t* out, size_t out_len) {
    if ((len && !data) || !out) return fail(DR_ERR_INVALID, "null buffer");
    switch (kind) {
        case DR_HASH_SHA512:
            if (out_len != 64) return fail(DR_ERR_INVALID, "SHA-512 digests are 64 bytes");
            drh::Sha512::hash(data, len, out);
            return DR_OK;
        case DR_HASH_SHAKE128: { drh::Shake128 s; s.update(data, len); s.digest(out, out_len); return DR_OK; }
        case DR_HASH_SHAKE256: { drh::Shake256 s; s.update(data, len); s.digest(out, out_len); return DR_OK; }
    }
    return fail(DR_ERR_INVALID, "unknown hash kind");
}

namespace {
// DOTRING_TRACE=1: wall-clock phase breakdown of the native batch calls on stderr
struct PhaseTrace {
    bool on;
    const char* what;
    std::chrono::steady_clock::time_point t0, last;
    std::string line;
    explicit PhaseTrace(const char* w) : on(std::getenv("DOTRING_TRACE") != nullptr), what(w) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char* name) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        char buf[64];
        std::snprintf(buf, sizeof buf, " %s=%.2f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~PhaseTrace() {
        if (!on) return;
        std::fprintf(stderr, "[dotring] %s total=%.2f ms |%s\n", what,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), line.c_str());
    }
};
int load_suite(const dr_vrf_suite* s, drh::VrfSuite& out) {
    if (!s || !s->suite_id || s->suite_id_len == 0 || s->suite_id_len > 200) return fail(DR_ERR_INVALID, "bad VRF suite");
    out.suite_id.assign(s->suite_id, s->suite_id + s->suite_id_len);
    out.xof = s->xof != 0;
    std::memcpy(out.generator, s->generator_xy, 64);
    std::memcpy(out.blinding_base, s->blinding_base_xy, 64);
    out.cv = drh::te_curve(s->curve);
    if (!out.cv) return fail(DR_ERR_INVALID, "unknown curve id in VRF suite");
    return DR_OK;
}

// encode_to_curve of B messages salt_i || data_i (salts nullable) into affine points.  Elligator 2 suites: hash_to_field
// on worker threads + one launch.  Try-and-increment suites (dot_ring/curve/point.py:252-296): the candidates of counters
// [0,4) of every message go through ONE decode launch, the (1/16 of the) messages none of whose candidates decompressed
// continue with counters [4,12), and so on — the first counter that works is the one the sequential loop would stop at.
int encode_to_curve_msgs(dr_ctx* ctx, const drh::VrfSuite& su, size_t B, const uint8_t* data, const uint64_t* off, const uint8_t* salts,
                         const uint64_t* salt_off, uint8_t* out_xy) {
    if (B == 0) return DR_OK;
    std::vector<drh::Bytes> msgs(B);
    auto build = [&](size_t i) {
        drh::Bytes& m = msgs[i];
        if (salt_off) drh::put(m, salts + salt_off[i], salt_off[i + 1] - salt_off[i]);
        drh::put(m, data + off[i], off[i + 1] - off[i]);
    };
    if (!su.cv->tai) {
        std::vector<uint8_t> us(B * 64);
        drh::parallel_for(B, [&](size_t i) {
            build(i);
            drh::hash_to_field2(su, msgs[i].data(), msgs[i].size(), us.data() + 64 * i);
        });
        return dr_bsn_encode_to_curve_batch(ctx, us.data(), B, out_xy);
    }
    drh::parallel_for(B, build);
    std::vector<size_t> pending(B);
    for (size_t i = 0; i < B; i++) pending[i] = i;
    std::vector<uint8_t> cand, xy, ok;
    for (unsigned base = 0; !pending.empty();) {
        if (base >= 256) return fail(DR_ERR_INVALID, "hash_to_curve_tai failed");
        const unsigned K = std::min<unsigned>(base == 0 ? 4 : 8, 256 - base);
        const size_t n = pending.size() * K;
        cand.resize(n * 32); xy.resize(n * 64); ok.resize(n);
        drh::parallel_for(n, [&](size_t j) {
            const drh::Bytes& m = msgs[pending[j / K]];
            drh::tai_candidate(su, m.data(), m.size(), base + (unsigned)(j % K), cand.data() + 32 * j);
        });
        TRY(te_decode_points(ctx, su.cv->id, true, cand.data(), n, xy.data(), ok.data()));
        std::vector<size_t> still;
        for (size_t q = 0; q < pending.size(); q++) {
            unsigned k = 0;
            while (k < K && !ok[q * K + k]) k++;
            if (k == K) still.push_back(pending[q]);
            else std::memcpy(out_xy + 64 * pending[q], xy.data() + 64 * (q * K + k), 64);
        }
        pending.swap(still);
        base += K;
    }
    return DR_OK;
}
}  // namespace

int dr_hash_to_field_batch(const dr_vrf_suite* suite, const uint8_t* msgs, const uint64_t* off, size_t count, uint8_t* out_u_pairs) {
    drh::VrfSuite su;
    TRY(load_suite(suite, su));
    if (count && (!off || !out_u_pairs || (off[count] && !msgs))) return fail(DR_ERR_INVALID, "null buffer");
    for (size_t i = 0; i < count; i++)
        if (off[i + 1] < off[i]) return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
    drh::parallel_for(count, [&](size_t i) { drh::hash_to_field2(su, msgs + off[i], off[i + 1] - off[i], out_u_pairs + 64 * i); });
    return DR_OK;
}

int dr_encode_to_curve_batch(dr_ctx* ctx, const dr_vrf_suite* suite, const uint8_t* msgs, const uint64_t* off, const uint8_t* salts,
                             const uint64_t* salt_off, size_t count, uint8_t* out_xy) {
    try {
        TRY(use_ctx(ctx));
        drh::VrfSuite su;
        TRY(load_suite(suite, su));
        if (count && (!off || !out_xy || (off[count] && !msgs))) return fail(DR_ERR_INVALID, "null buffer");
        for (size_t i = 0; i < count; i++)
            if (off[i + 1] < off[i] || (salt_off && salt_off[i + 1] < salt_off[i])) return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
        return encode_to_curve_msgs(ctx, su, count, msgs, off, salts, salt_off, out_xy);
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("encode_to_curve: ") + e.what());
    }
}

// I_i = encode_to_curve(salt_i || alpha_i) and O_i = x_i * I_i for an Elligator suite without a host round trip in between:
// the Elligator kernel writes the affine inputs to device memory, the GLV lane-pair kernel (scalars split on the host while
// the first kernel runs) reads them from there; ONE synchronisation and download for both.  Other cases (try-and-increment
// suites, batches beyond the GLV kernel's range) take the two separate calls.
int encode_and_mul(dr_ctx* ctx, const drh::VrfSuite& su, size_t B, const uint8_t* data, const uint64_t* off, const uint8_t* salts,
                   const uint64_t* salt_off, const uint8_t* xs, uint8_t* inputs_xy, uint8_t* outs_xy) {
    if (su.cv->tai || !su.cv->glv || !g_bsn_glv || B == 0 || B >= 16384) {
        TRY(encode_to_curve_msgs(ctx, su, B, data, off, salts, salt_off, inputs_xy));
        return te_scalar_mul_batch(ctx, su.cv->id, inputs_xy, xs, B, outs_xy);
    }
    TRY(use_ctx(ctx));
    std::vector<uint8_t> us(B * 64);
    drh::parallel_for(B, [&](size_t i) {
        drh::Bytes m;
        if (salt_off) drh::put(m, salts + salt_off[i], salt_off[i + 1] - salt_off[i]);
        drh::put(m, data + off[i], off[i + 1] - off[i]);
        drh::hash_to_field2(su, m.data(), m.size(), us.data() + 64 * i);
    });
    TRY(ctx->io_a.reserve(B * 64));
    TRY(ctx->io_b.reserve(B * 48));
    TRY(ctx->io_c.reserve(2 * B * 64));
    uint32_t* d_in = ctx->io_c.as<uint32_t>();
    uint32_t* d_out = d_in + B * 16;
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, us.data(), B * 64, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch(ctx, "k_bsn_encode_to_curve", [&] {
        hipLaunchKernelGGL(dr::k_bsn_encode_to_curve, dim3(div_up(2 * B, 64)), dim3(64), 0, ctx->stream, ctx->io_a.as<uint32_t>(), d_in, (uint32_t)B);
    }));
    std::vector<uint32_t> split;
    TRY(glv_split_scalars(xs, B, split));                  // on the host, while the Elligator kernel runs
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, split.data(), B * 48, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
        hipLaunchKernelGGL(dr::k_bsn_scalar_mul_glv, dim3(div_up(2 * B, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream, d_in,
                           ctx->io_b.as<uint32_t>(), d_out, (uint32_t)B);
    }));
    HIP_TRY(hipMemcpyAsync(inputs_xy, d_in, B * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(outs_xy, d_out, B * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

// The whole batch in one call: Pedersen VRF part (pedersen/vrf.py:86-126) then the ring proof
// (proof_builder.py:38-315) — GPU phases through the entry points above, the hashing between them on worker threads.
// Pedersen VRF prover for a batch (pedersen/vrf.py:86-126): head() = hash-to-curve, outputs, transcripts and blinding
// factors (what the ring proof needs); tail() = blinded keys, nonces, R / O_k, challenge, responses and the 192 encoded
// bytes.  The two halves may run on different contexts (streams) of the same GPU.
struct PedersenBatch {
    const drh::VrfSuite& su;
    size_t B;
    std::vector<uint8_t> us, xs, inputs, outs, blind, gb_pts, sc, ybar, ks, kbs, pts3, sc3, third;
    std::vector<drh::Bytes> tr;
    PedersenBatch(const drh::VrfSuite& s, size_t b) : su(s), B(b) {}

    int head(dr_ctx* ctx, const uint8_t* alphas, const uint64_t* alpha_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
             const uint64_t* salt_off, const uint8_t* secret_scalars, PhaseTrace& tr_) {
        const drh::Mod256& mn = su.cv->n;
        const int cv = su.cv->id;
        // 1. secrets mod n
        xs.resize(B * 32);
        for (size_t i = 0; i < B; i++) {
            uint64_t x[4];
            mn.reduce_bytes(secret_scalars + 32 * i, 32, false, x);
            drh::store_le32(x, xs.data() + 32 * i);
        }
        // 2. I_i = encode_to_curve(salt || alpha), O_i = x_i * I_i
        inputs.resize(B * 64); outs.resize(B * 64);
        TRY(encode_and_mul(ctx, su, B, alphas, alpha_off, salts, salt_off, xs.data(), inputs.data(), outs.data()));
        tr_.mark("encode+x*I");
        // 3. transcripts, blinding factors
        tr.assign(B, drh::Bytes());
        blind.resize(B * 32); gb_pts.resize(B * 128); sc.resize(B * 64);
        std::vector<int> bad(B, 0);
        drh::parallel_for(B, [&](size_t i) {
            drh::Bytes& t = tr[i];
            t = su.suite_id;
            drh::put8(t, 0x02);                                    // PEDERSEN_VRF
            drh::put_le64(t, 1);                                   // one (input, output) pair
            uint8_t enc[32];
            drh::enc_te_point(inputs.data() + 64 * i, enc); drh::put(t, enc, 32);
            drh::enc_te_point(outs.data() + 64 * i, enc); drh::put(t, enc, 32);
            size_t adl = ad_off[i + 1] - ad_off[i];
            drh::put_le64(t, adl);
            drh::put(t, ads + ad_off[i], adl);
            drh::Bytes tb = t;
            drh::put8(tb, 0x12);                                   // PEDERSEN_BLINDING
            uint64_t x[4], b[4];
            drh::load_le32(xs.data() + 32 * i, x);
            if (!drh::vrf_nonce(su, tb, x, b)) bad[i] = 1;
            drh::store_le32(b, blind.data() + 32 * i);
            std::memcpy(gb_pts.data() + 128 * i, su.generator, 64);
            std::memcpy(gb_pts.data() + 128 * i + 64, su.blinding_base, 64);
            std::memcpy(sc.data() + 64 * i, xs.data() + 32 * i, 32);
            std::memcpy(sc.data() + 64 * i + 32, blind.data() + 32 * i, 32);
        });
        for (size_t i = 0; i < B; i++) if (bad[i]) return fail(DR_ERR_INVALID, "nonce scalar is zero");
        tr_.mark("blinding");
        return DR_OK;
    }

    // out_proofs: 192 bytes per proof at `stride`; out_aux (nullable): O, Y_bar, R, O_k affine (4*64) + blinding (32) at aux_stride
    int tail(dr_ctx* actx, uint8_t* out_proofs, size_t stride, uint8_t* out_aux, size_t aux_stride) {
        const drh::Mod256& mn = su.cv->n;
        const int cv = su.cv->id;
        ybar.resize(B * 64); ks.resize(B * 32); kbs.resize(B * 32); pts3.resize(2 * B * 128); sc3.resize(2 * B * 64); third.resize(2 * B * 64);
        // 4. blinded public keys  Y_bar_i = x_i*G + b_i*B
        TRY(te_msm_groups(actx, cv, gb_pts.data(), sc.data(), B, 2, ybar.data()));
        // 5. nonces
        std::vector<int> bad2(B, 0);
        drh::parallel_for(B, [&](size_t i) {
            uint8_t enc[32];
            drh::enc_te_point(ybar.data() + 64 * i, enc);
            drh::put(tr[i], enc, 32);
            uint64_t x[4], b[4], k[4], kb[4];
            drh::load_le32(xs.data() + 32 * i, x);
            drh::load_le32(blind.data() + 32 * i, b);
            if (!drh::vrf_nonce(su, tr[i], x, k) || !drh::vrf_nonce(su, tr[i], b, kb)) bad2[i] = 1;
            drh::store_le32(k, ks.data() + 32 * i);
            drh::store_le32(kb, kbs.data() + 32 * i);
            // group i: k*G + kb*B ; group B+i: k*I + 0*I
            std::memcpy(pts3.data() + 128 * i, su.generator, 64);
            std::memcpy(pts3.data() + 128 * i + 64, su.blinding_base, 64);
            std::memcpy(sc3.data() + 64 * i, ks.data() + 32 * i, 32);
            std::memcpy(sc3.data() + 64 * i + 32, kbs.data() + 32 * i, 32);
            std::memcpy(pts3.data() + 128 * (B + i), inputs.data() + 64 * i, 64);
            std::memcpy(pts3.data() + 128 * (B + i) + 64, inputs.data() + 64 * i, 64);
            std::memcpy(sc3.data() + 64 * (B + i), ks.data() + 32 * i, 32);
            std::memset(sc3.data() + 64 * (B + i) + 32, 0, 32);
        });
        for (size_t i = 0; i < B; i++) if (bad2[i]) return fail(DR_ERR_INVALID, "nonce scalar is zero");
        TRY(te_msm_groups(actx, cv, pts3.data(), sc3.data(), 2 * B, 2, third.data()));
        // 6. challenge, responses, the 192 encoded bytes
        drh::parallel_for(B, [&](size_t i) {
            uint8_t* out = out_proofs + stride * i;
            drh::enc_te_point(outs.data() + 64 * i, out);
            drh::enc_te_point(ybar.data() + 64 * i, out + 32);
            drh::enc_te_point(third.data() + 64 * i, out + 64);
            drh::enc_te_point(third.data() + 64 * (B + i), out + 96);
            uint64_t c[4], x[4], b[4], k[4], kb[4], s[4], sb[4];
            drh::vrf_challenge(su, tr[i], out + 64, 2, c);
            drh::load_le32(xs.data() + 32 * i, x);
            drh::load_le32(blind.data() + 32 * i, b);
            drh::load_le32(ks.data() + 32 * i, k);
            drh::load_le32(kbs.data() + 32 * i, kb);
            mn.mul(c, x, s);  mn.add(s, k, s);
            mn.mul(c, b, sb); mn.add(sb, kb, sb);
            drh::store_le32(s, out + 128);
            drh::store_le32(sb, out + 160);
            if (out_aux) {
                uint8_t* a = out_aux + aux_stride * i;
                std::memcpy(a, outs.data() + 64 * i, 64);
                std::memcpy(a + 64, ybar.data() + 64 * i, 64);
                std::memcpy(a + 128, third.data() + 64 * i, 64);
                std::memcpy(a + 192, third.data() + 64 * (B + i), 64);
                std::memcpy(a + 256, blind.data() + 32 * i, 32);
            }
        });
        return DR_OK;
    }
};

static int ringvrf_prove_batch_impl(dr_ring_prover* p, const dr_vrf_suite* suite, size_t batch, const uint8_t* alphas, const uint64_t* alpha_off,
                                    const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                                    const uint8_t* secret_scalars, const uint32_t* producer_index, const uint8_t* fs_prefix, size_t fs_prefix_len,
                                    const uint8_t* zk_random48, uint8_t* out_proofs, uint8_t* out_aux) {
    if (!p || !alpha_off || !ad_off || !secret_scalars || !producer_index || !fs_prefix || !out_proofs) return fail(DR_ERR_INVALID, "null argument");
    if (batch == 0) return DR_OK;
    if (batch > 4096) return fail(DR_ERR_INVALID, "batch must be at most 4096 per call");
    drh::VrfSuite su;
    TRY(load_suite(suite, su));
    if (su.cv->id != p->curve) return fail(DR_ERR_INVALID, "VRF suite and ring prover are on different curves");
    for (size_t i = 0; i < batch; i++)
        if (alpha_off[i + 1] < alpha_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
            return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
    dr_ctx* ctx = p->ctx;
    const size_t B = batch;
    PhaseTrace tr_("prove_batch");

    PedersenBatch ped(su, B);
    TRY(ped.head(ctx, alphas, alpha_off, ads, ad_off, salts, salt_off, secret_scalars, tr_));
    std::vector<uint8_t>& blind = ped.blind;
    // 4.-6. the rest of the Pedersen part needs nothing from the ring proof and the ring proof needs only the blinding
    // factors: it runs on a second stream (own context: scratch + stream) from a helper thread while this thread drives
    // the ring phases.  Its kernels are latency-bound (16..64 waves) and hide under the chip-filling MSMs.
    if (!p->aux_ctx) {
        TRY(dr_ctx_create(ctx->device, &p->aux_ctx));
        ctx->helpers.push_back(p->aux_ctx);
    }
    dr_ctx* actx = p->aux_ctx;
    actx->prof = ctx->prof;
    int ped_rc = DR_OK;
    std::string ped_err;
    const bool overlap = std::getenv("DOTRING_PROVE_OVERLAP") == nullptr || std::atoi(std::getenv("DOTRING_PROVE_OVERLAP")) != 0;
    std::thread ped_thread;
    if (overlap) {
        ped_thread = std::thread([&] {
            ped_rc = ped.tail(actx, out_proofs, 784, out_aux, DR_RINGVRF_AUX_BYTES);
            if (ped_rc != DR_OK) ped_err = dr_last_error();
        });
    } else {
        TRY(ped.tail(ctx, out_proofs, 784, out_aux, DR_RINGVRF_AUX_BYTES));
    }
    struct Joiner {          // every exit path below must wait for the helper before the buffers it uses go away
        std::thread& t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{ped_thread};

    // 7. ring proof: witness columns
    std::vector<uint8_t> zk;
    if (zk_random48) {
        zk.resize(B * 12 * 32);
        drh::parallel_for(B * 12, [&](size_t j) {
            uint64_t v[4];
            drh::mod_p().reduce_bytes(zk_random48 + 48 * j, 48, false, v);
            drh::store_le32(v, zk.data() + 32 * j);
        });
    }
    std::vector<uint8_t> relation(B * 64), wit(B * 4 * 96), cq(B * 96), evals(B * 256), opens(B * 192);
    std::vector<int> wit_inf(B * 4), cq_inf(B), open_inf(B * 2);
    tr_.mark("spawn+zk");
    TRY(dr_ring_prove_witness(p, B, producer_index, blind.data(), zk_random48 ? zk.data() : nullptr, relation.data(), wit.data(), wit_inf.data()));
    tr_.mark("witness");
    drh::FsTranscript base;
    base.sh.update(fs_prefix, fs_prefix_len);
    std::vector<drh::FsTranscript> fs(B, base);
    std::vector<uint8_t> alphas7(B * 7 * 32), zetas(B * 32), nus(B * 8 * 32);
    drh::parallel_for(B, [&](size_t i) {
        fs[i].absorb_labeled("instance", relation.data() + 64 * i, 64);
        uint8_t ser[4 * 96];
        for (int c = 0; c < 4; c++) drh::g1_serialized(wit.data() + 96 * (4 * i + c), wit_inf[4 * i + c], ser + 96 * c);
        fs[i].absorb_labeled("committed_cols", ser, sizeof ser);
        fs[i].challenges("constraints_aggregation", 7, alphas7.data() + 224 * i);
    });
    tr_.mark("fs1");
    TRY(dr_ring_prove_quotient(p, B, alphas7.data(), cq.data(), cq_inf.data()));
    tr_.mark("quotient");
    drh::parallel_for(B, [&](size_t i) {
        uint8_t ser[96];
        drh::g1_serialized(cq.data() + 96 * i, cq_inf[i], ser);
        fs[i].absorb_labeled("quotient", ser, 96);
        fs[i].challenges("evaluation_point", 1, zetas.data() + 32 * i);
    });
    tr_.mark("fs2");
    TRY(dr_ring_prove_evals(p, B, zetas.data(), evals.data()));
    tr_.mark("evals");
    drh::parallel_for(B, [&](size_t i) {
        fs[i].absorb_labeled("register_evaluations", evals.data() + 256 * i, 224);
        fs[i].absorb_labeled("shifted_linearization_evaluation", evals.data() + 256 * i + 224, 32);
        fs[i].challenges("kzg_aggregation", 8, nus.data() + 256 * i);
    });
    tr_.mark("fs3");
    TRY(dr_ring_prove_openings(p, B, nus.data(), opens.data(), open_inf.data()));
    tr_.mark("openings");
    // 8. payload: 4 compressed commitments, 7 evaluations, C_q, l(zeta*omega), 2 opening proofs  (proof_payload.py:68-117)
    std::vector<int> rc(B, DR_OK);
    drh::parallel_for(B, [&](size_t i) {
        uint8_t* out = out_proofs + 784 * i + 192;
        int r = DR_OK;
        for (int c = 0; c < 4 && r == DR_OK; c++) r = dr_g1_compress(wit.data() + 96 * (4 * i + c), wit_inf[4 * i + c], out + 48 * c);
        std::memcpy(out + 192, evals.data() + 256 * i, 224);
        if (r == DR_OK) r = dr_g1_compress(cq.data() + 96 * i, cq_inf[i], out + 416);
        std::memcpy(out + 464, evals.data() + 256 * i + 224, 32);
        if (r == DR_OK) r = dr_g1_compress(opens.data() + 192 * i, open_inf[2 * i], out + 496);
        if (r == DR_OK) r = dr_g1_compress(opens.data() + 192 * i + 96, open_inf[2 * i + 1], out + 544);
        rc[i] = r;
        if (out_aux) {
            uint8_t* a = out_aux + DR_RINGVRF_AUX_BYTES * i + 288;
            for (int c = 0; c < 4; c++) drh::g1_serialized(wit.data() + 96 * (4 * i + c), wit_inf[4 * i + c], a + 96 * c);
            drh::g1_serialized(cq.data() + 96 * i, cq_inf[i], a + 384);
            drh::g1_serialized(opens.data() + 192 * i, open_inf[2 * i], a + 480);
            drh::g1_serialized(opens.data() + 192 * i + 96, open_inf[2 * i + 1], a + 576);
        }
    });
    for (size_t i = 0; i < B; i++) if (rc[i] != DR_OK) return rc[i];
    tr_.mark("payload");
    if (ped_thread.joinable()) ped_thread.join();
    tr_.mark("join");
    if (ped_rc != DR_OK) return fail(ped_rc, ped_err.empty() ? "Pedersen part failed" : ped_err);
    return DR_OK;
}

// RingVRF.batch_verify over encoded proofs (vrf/ring/vrf.py:239-283, pedersen/vrf.py:171-242, ring_proof/verify.py:51-324,
// pcs/kzg.py:304-338): decode + validate every point on the GPU, replay the transcripts on worker threads, fold all
// claims into one Bandersnatch MSM (5B+2 points, must be the identity) and two G1 MSMs + one pairing equation.
// C++ exceptions (allocation failures of the host-side staging vectors, thread creation) must not cross the C ABI
int dr_ringvrf_prove_batch(dr_ring_prover* p, const dr_vrf_suite* suite, size_t batch, const uint8_t* alphas, const uint64_t* alpha_off,
                           const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                           const uint8_t* secret_scalars, const uint32_t* producer_index, const uint8_t* fs_prefix, size_t fs_prefix_len,
                           const uint8_t* zk_random48, uint8_t* out_proofs, uint8_t* out_aux) {
    try {
        return ringvrf_prove_batch_impl(p, suite, batch, alphas, alpha_off, ads, ad_off, salts, salt_off, secret_scalars, producer_index, fs_prefix,
                                        fs_prefix_len, zk_random48, out_proofs, out_aux);
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native prover: ") + e.what());
    }
}

// Pedersen VRF batch verification core (pedersen/vrf.py:171-242) on decoded points: challenges, weights from one
// transcript over all (c, s, s_b), then ONE (5B+2)-point MSM that must be the identity.  proofs: 192 bytes per proof at
// `stride`; te_xy: the four decoded points of each proof (O, Y_bar, R, O_k affine); in_pts: encode_to_curve of the inputs.
static int pedersen_verify_core(dr_ctx* actx, const drh::VrfSuite& su, size_t B, const uint8_t* proofs, size_t stride,
                                const std::vector<uint8_t>& te_xy, const std::vector<uint8_t>& in_pts, const uint8_t* ads,
                                const uint64_t* ad_off, int& ped_ok) {
    const drh::Mod256& mn = su.cv->n;
    std::vector<uint8_t> cs(B * 32);
    drh::parallel_for(B, [&](size_t i) {
        const uint8_t* pr = proofs + stride * i;
        drh::Bytes t = su.suite_id;
        drh::put8(t, 0x02);
        drh::put_le64(t, 1);
        uint8_t enc[32];
        drh::enc_te_point(in_pts.data() + 64 * i, enc);
        drh::put(t, enc, 32);
        drh::put(t, pr, 32);                                   // output point, as encoded in the proof
        size_t adl = ad_off[i + 1] - ad_off[i];
        drh::put_le64(t, adl);
        drh::put(t, ads + ad_off[i], adl);
        drh::put(t, pr + 32, 32);                              // blinded public key
        uint64_t c[4];
        drh::vrf_challenge(su, t, pr + 64, 2, c);              // R, O_k
        drh::store_le32(c, cs.data() + 32 * i);
    });
    {
        drh::Bytes absorbed = su.suite_id;
        drh::put8(absorbed, 0x50);                             // BATCH_VERIFY
        for (size_t i = 0; i < B; i++) {
            drh::put(absorbed, cs.data() + 32 * i, 32);
            drh::put(absorbed, proofs + stride * i + 128, 64);    // s, s_b
        }
        std::vector<uint8_t> weights(32 * B);
        drh::vrf_squeeze(su.xof, absorbed.data(), absorbed.size(), weights.data(), weights.size());
        std::vector<uint8_t> pts((5 * B + 2) * 64), sc((5 * B + 2) * 32);
        std::vector<uint64_t> gen_part(B * 4), blind_part(B * 4);
        drh::parallel_for(B, [&](size_t i) {
            const uint8_t* pr = proofs + stride * i;
            uint64_t w_io[4], w_cm[4], c[4], s[4], sb[4], t[4];
            mn.reduce_bytes(weights.data() + 32 * i, 16, false, w_io);
            mn.reduce_bytes(weights.data() + 32 * i + 16, 16, false, w_cm);
            drh::load_le32(cs.data() + 32 * i, c);
            drh::load_le32(pr + 128, s);
            drh::load_le32(pr + 160, sb);
            uint8_t* p = pts.data() + 320 * i;
            uint8_t* k = sc.data() + 160 * i;
            std::memcpy(p, te_xy.data() + 64 * (4 * i + 3), 64);       drh::store_le32(w_io, k);                       // O_k
            std::memcpy(p + 64, te_xy.data() + 64 * (4 * i), 64);      mn.mul(w_io, c, t); drh::store_le32(t, k + 32);  // output
            std::memcpy(p + 128, in_pts.data() + 64 * i, 64);          mn.mul(w_io, s, t); mn.neg(t, t); drh::store_le32(t, k + 64);   // input
            std::memcpy(p + 192, te_xy.data() + 64 * (4 * i + 2), 64); drh::store_le32(w_cm, k + 96);                  // R
            std::memcpy(p + 256, te_xy.data() + 64 * (4 * i + 1), 64); mn.mul(w_cm, c, t); drh::store_le32(t, k + 128); // Y_bar
            mn.mul(w_cm, s, &gen_part[4 * i]);
            mn.mul(w_cm, sb, &blind_part[4 * i]);
        });
        uint64_t gs[4] = {0, 0, 0, 0}, bs[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < B; i++) { mn.sub(gs, &gen_part[4 * i], gs); mn.sub(bs, &blind_part[4 * i], bs); }
        std::memcpy(pts.data() + 320 * B, su.generator, 64);           drh::store_le32(gs, sc.data() + 160 * B);
        std::memcpy(pts.data() + 320 * B + 64, su.blinding_base, 64);  drh::store_le32(bs, sc.data() + 160 * B + 32);
        uint8_t sum[64];
        TRY(te_msm(actx, su.cv->id, pts.data(), sc.data(), 5 * B + 2, sum));
        uint8_t ident[64] = {0};
        ident[32] = 1;
        ped_ok = std::memcmp(sum, ident, 64) == 0 ? 1 : 0;
    }
    return DR_OK;
}

static int ringvrf_verify_batch_impl(dr_ctx* ctx, const dr_vrf_suite* suite, const dr_ring_verifier_key* vk, size_t batch, const uint8_t* proofs,
                                     const uint8_t* inputs, const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
                                     const uint64_t* salt_off, const uint8_t seed32[32], int* ok) {
    TRY(use_ctx(ctx));
    if (!vk || !proofs || !in_off || !ad_off || !seed32 || !ok || !vk->fs_prefix) return fail(DR_ERR_INVALID, "null argument");
    *ok = 0;
    if (batch == 0) { *ok = 1; return DR_OK; }
    if (batch > 4096) return fail(DR_ERR_INVALID, "batch must be at most 4096 per call");
    if (vk->log2n < 9 || vk->log2n > 16) return fail(DR_ERR_INVALID, "bad domain size");
    drh::VrfSuite su;
    TRY(load_suite(suite, su));
    for (size_t i = 0; i < batch; i++)
        if (in_off[i + 1] < in_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
            return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
    const size_t B = batch;
    const drh::Mod256& mn = su.cv->n;
    const drh::Mod256& mp = drh::mod_p();
    hipStream_t st = ctx->stream;

    PhaseTrace tr_("verify_batch");
    // ---- 1. canonical scalars; gather encoded points
    std::vector<uint8_t> te_enc(B * 4 * 32), g1_enc(B * 7 * 48);
    bool canonical = true;
    for (size_t i = 0; i < B; i++) {
        const uint8_t* pr = proofs + 784 * i;
        std::memcpy(te_enc.data() + 128 * i, pr, 128);
        uint64_t v[4];
        for (int k = 0; k < 2; k++) { drh::load_le32(pr + 128 + 32 * k, v); if (drh::Mod256::geq(v, mn.m)) canonical = false; }      // dec_scalar
        const uint8_t* pl = pr + 192;
        for (int k = 0; k < 7; k++) { drh::load_le32(pl + 192 + 32 * k, v); if (drh::Mod256::geq(v, mp.m)) canonical = false; }
        drh::load_le32(pl + 464, v); if (drh::Mod256::geq(v, mp.m)) canonical = false;
        uint8_t* g = g1_enc.data() + 336 * i;
        std::memcpy(g, pl, 192);                   // C_b, C_accip, C_accx, C_accy
        std::memcpy(g + 192, pl + 416, 48);        // C_q
        std::memcpy(g + 240, pl + 496, 96);        // Phi_zeta, Phi_zeta_omega
    }
    if (!canonical) return DR_OK;

    // ---- 2. GPU: decode + validate the 4B Bandersnatch points, decompress the 7B G1 points; meanwhile a helper thread
    // hashes the inputs to the curve on a second stream (all three kernels are latency-bound: a few dozen waves)
    if (!ctx->aux) TRY(dr_ctx_create(ctx->device, &ctx->aux));
    dr_ctx* actx = ctx->aux;
    actx->prof = ctx->prof;
    // One helper thread for the whole Pedersen side (second stream): hash the inputs to the curve right away, then wait at a
    // gate until this thread has decoded and validated the proof points, then the challenges and the (5B+2)-point MSM.  The main
    // thread never waits for the Elligator kernels (they took the decode phase from 2.2 to 3.1 ms when it joined them there).
    const size_t n_te = 4 * B, n_g1 = 7 * B + 4;
    std::vector<uint8_t> in_pts(B * 64), te_xy(n_te * 64);
    int side_rc = DR_OK, ped_ok = 0;
    std::string side_err;
    std::mutex gate_m;
    std::condition_variable gate_cv;
    int gate = -1;                                   // -1 closed, 0 give up, 1 go on
    auto open_gate = [&](int v) {
        { std::lock_guard<std::mutex> lk(gate_m); if (gate < 0) gate = v; }
        gate_cv.notify_all();
    };
    std::thread side([&] {
        side_rc = encode_to_curve_msgs(actx, su, B, inputs, in_off, salts, salt_off, in_pts.data());
        if (side_rc != DR_OK) { side_err = dr_last_error(); return; }
        {
            std::unique_lock<std::mutex> lk(gate_m);
            gate_cv.wait(lk, [&] { return gate >= 0; });
            if (gate == 0) return;
        }
        side_rc = pedersen_verify_core(actx, su, B, proofs, 784, te_xy, in_pts, ads, ad_off, ped_ok);
        if (side_rc != DR_OK) side_err = dr_last_error();
    });
    struct Joiner {
        std::thread& t;
        std::function<void()> give_up;
        ~Joiner() { give_up(); if (t.joinable()) t.join(); }
    } joiner{side, [&] { open_gate(0); }};
    TRY(ctx->io_a.reserve(n_te * 32));
    TRY(ctx->io_b.reserve(n_te * 64));
    TRY(ctx->io_c.reserve(n_te * 4 + n_g1 * 4));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, te_enc.data(), n_te * 32, hipMemcpyHostToDevice, st));
    uint32_t* d_ok = ctx->io_c.as<uint32_t>();
    TRY(launch(ctx, "k_bsn_decode_points", [&] {
        launch_decode_points(ctx, st, su.cv->id, false, ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), d_ok, n_te);
    }));
    std::vector<uint32_t> flags(n_te + n_g1);
    HIP_TRY(hipMemcpyAsync(te_xy.data(), ctx->io_b.p, n_te * 64, hipMemcpyDeviceToHost, st));
    // G1: bases buffer = 7B decompressed points followed by C_px, C_py, C_s and G1[0]
    Scratch &g1_bases = ctx->vfy_bases, &g1_in = ctx->vfy_in, &g1_std = ctx->vfy_std;
    TRY(g1_bases.reserve(n_g1 * 96));
    TRY(g1_in.reserve(7 * B * 48));
    TRY(g1_std.reserve(n_g1 * 96));
    HIP_TRY(hipMemcpyAsync(g1_in.p, g1_enc.data(), 7 * B * 48, hipMemcpyHostToDevice, st));
    TRY(launch(ctx, "k_g1_decompress", [&] {
        hipLaunchKernelGGL(dr::k_g1_decompress, dim3(div_up(7 * B, 64)), dim3(64), 0, st, g1_in.as<uint8_t>(), g1_bases.as<uint32_t>(), d_ok + n_te,
                           (uint32_t)(7 * B));
    }));
    {
        uint8_t tail_be[4 * 96];
        std::memcpy(tail_be, vk->fixed_commitments, 3 * 96);
        std::memcpy(tail_be + 288, vk->g1_generator, 96);
        for (int k = 0; k < 3; k++) if (tail_be[96 * k] & 0x40) std::memset(tail_be + 96 * k, 0, 96);       // serialised infinity
        std::vector<uint8_t> le;
        TRY(g1_be_to_le_limbs(tail_be, 4, le, true));
        HIP_TRY(hipMemcpyAsync(g1_bases.as<uint32_t>() + 7 * B * 24, le.data(), 4 * 96, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));          // `le` is a stack-lifetime staging buffer
        hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(1), dim3(64), 0, st, g1_bases.as<uint32_t>() + 7 * B * 24, 4u);
    }
    hipLaunchKernelGGL(dr::k_g1_bases_from_mont, dim3(div_up(7 * B, 256)), dim3(256), 0, st, g1_bases.as<uint32_t>(), g1_std.as<uint32_t>(), (uint32_t)(7 * B));
    std::vector<uint8_t> g1_le(7 * B * 96);
    HIP_TRY(hipMemcpyAsync(g1_le.data(), g1_std.p, 7 * B * 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(flags.data(), d_ok, (n_te + 7 * B) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < n_te + 7 * B; i++) if (!flags[i]) return DR_OK;                  // malformed / invalid point: ok = 0
    tr_.mark("decode");

    // ---- 3. Pedersen part: the helper thread may go on (te_xy is complete)
    open_gate(1);

    // ---- 4. ring proofs: transcript replay + verifier scalar pass per proof, random linear combination of all claims
    drh::RingVerifierDomain dm;
    dm.init(vk->log2n, vk->omega_n, vk->seed_xy);
    drh::FsTranscript base;
    base.sh.update(vk->fs_prefix, vk->fs_prefix_len);
    std::vector<uint8_t> lhs_sc(n_g1 * 32), rhs_sc(2 * B * 32);
    std::vector<uint64_t> fixed_part(B * 16);           // per proof: r1*nu0, r1*nu1, r1*nu2, r1*agg + r2*l_zw
    std::vector<int> bad(B, 0);
    auto be_rec = [&](size_t idx, uint8_t out[96]) {     // device LE limbs -> serialize() form
        const uint8_t* s = g1_le.data() + 96 * idx;
        bool inf = true;
        for (int j = 0; j < 96; j++) if (s[j]) { inf = false; break; }
        if (inf) { std::memset(out, 0, 96); out[0] = 0x40; return; }
        for (int j = 0; j < 48; j++) { out[j] = s[47 - j]; out[48 + j] = s[95 - j]; }
    };
    drh::parallel_for(B, [&](size_t i) {
        const uint8_t* pr = proofs + 784 * i;
        const uint8_t* pl = pr + 192;
        drh::FsTranscript t = base;
        uint8_t result_seed[64], ser[4 * 96], al[7 * 32], zeta[32], nus[8 * 32];
        const uint8_t* relation = te_xy.data() + 64 * (4 * i + 1);            // blinded public key
        t.absorb_labeled("instance", relation, 64);
        for (int k = 0; k < 4; k++) be_rec(7 * i + k, ser + 96 * k);
        t.absorb_labeled("committed_cols", ser, sizeof ser);
        t.challenges("constraints_aggregation", 7, al);
        be_rec(7 * i + 4, ser);
        t.absorb_labeled("quotient", ser, 96);
        t.challenges("evaluation_point", 1, zeta);
        t.absorb_labeled("register_evaluations", pl + 192, 224);
        t.absorb_labeled("shifted_linearization_evaluation", pl + 464, 32);
        t.challenges("kzg_aggregation", 8, nus);
        drh::te_add_affine(*su.cv, vk->seed_xy, relation, result_seed);
        drh::RingClaimScalars cl;
        if (!drh::ring_verifier_terms(*su.cv, dm, al, nus, zeta, pl + 192, pl + 464, result_seed, cl)) { bad[i] = 1; return; }
        // verifier randomness: two non-zero coefficients per proof
        uint64_t r[2][4];
        for (int k = 0; k < 2; k++) {
            drh::Shake256 sh;
            sh.update(seed32, 32);
            uint8_t ctr[9] = {0};
            for (int j = 0; j < 8; j++) ctr[j] = (uint8_t)((uint64_t)(2 * i + k) >> (8 * j));
            sh.update(ctr, 8);
            uint8_t raw[48];
            sh.digest(raw, 48);
            mp.reduce_bytes(raw, 48, true, r[k]);
            if (mp.is_zero(r[k])) mp.set_u64(1, r[k]);
        }
        uint64_t v[4], w[4];
        uint8_t* L = lhs_sc.data() + 224 * i;
        mp.mul(r[0], cl.nus[3], v); drh::store_le32(v, L);                                                      // C_b
        mp.mul(r[0], cl.nus[4], v); mp.mul(r[1], cl.k_ip, w); mp.add(v, w, v); drh::store_le32(v, L + 32);       // C_accip
        mp.mul(r[0], cl.nus[5], v); mp.mul(r[1], cl.k_x, w); mp.add(v, w, v); drh::store_le32(v, L + 64);        // C_accx
        mp.mul(r[0], cl.nus[6], v); mp.mul(r[1], cl.k_y, w); mp.add(v, w, v); drh::store_le32(v, L + 96);        // C_accy
        mp.mul(r[0], cl.nus[7], v); drh::store_le32(v, L + 128);                                                 // C_q
        mp.mul(r[0], cl.zeta, v); drh::store_le32(v, L + 160);                                                   // Phi_zeta
        mp.mul(r[1], cl.zeta_omega, v); drh::store_le32(v, L + 192);                                             // Phi_zeta_omega
        drh::store_le32(r[0], rhs_sc.data() + 64 * i);
        drh::store_le32(r[1], rhs_sc.data() + 64 * i + 32);
        uint64_t* fp = &fixed_part[16 * i];
        for (int k = 0; k < 3; k++) mp.mul(r[0], cl.nus[k], fp + 4 * k);
        mp.mul(r[0], cl.agg_zeta, v); mp.mul(r[1], cl.l_zw, w); mp.add(v, w, fp + 12);
    });
    for (size_t i = 0; i < B; i++) if (bad[i]) return DR_OK;
    tr_.mark("transcripts");
    {
        uint64_t acc[4][4] = {{0}};
        for (size_t i = 0; i < B; i++)
            for (int k = 0; k < 4; k++) mp.add(acc[k], &fixed_part[16 * i + 4 * k], acc[k]);
        mp.neg(acc[3], acc[3]);                                                   // - sum_v on G1[0]
        for (int k = 0; k < 4; k++) drh::store_le32(acc[k], lhs_sc.data() + 224 * B + 32 * k);
    }
    // two MSMs over the decompressed bases (already resident): lhs over all 7B+4 points, rhs with zero scalars on
    // everything but the 2B opening proofs (zero digits cost nothing).  Two single MSMs rather than a batch of two:
    // the final 255-doubling window combination of a single MSM runs on the host (0.2 ms), a batch leaves it to one
    // GPU lane per MSM (4 ms).
    std::vector<uint8_t> rhs_full(n_g1 * 32, 0);
    for (size_t i = 0; i < B; i++) std::memcpy(rhs_full.data() + 224 * i + 160, rhs_sc.data() + 64 * i, 64);
    TRY(ctx->scalars.reserve(n_g1 * 32));
    uint8_t pair_g1[2 * 96];
    int pair_inf[2] = {0, 0};
    // the two MSMs are independent and each is a short latency chain (sort, accumulate, reduce, fold): the rhs runs on
    // a third stream from a helper thread while this thread does the lhs
    if (!ctx->aux2) TRY(dr_ctx_create(ctx->device, &ctx->aux2));
    dr_ctx* bctx = ctx->aux2;
    bctx->prof = ctx->prof;
    int rhs_rc = DR_OK;
    std::string rhs_err;
    std::thread rhs_thread([&] {
        rhs_rc = [&]() -> int {
            TRY(use_ctx(bctx));
            TRY(bctx->scalars.reserve(n_g1 * 32));
            HIP_TRY(hipMemcpyAsync(bctx->scalars.p, rhs_full.data(), n_g1 * 32, hipMemcpyHostToDevice, bctx->stream));
            return msm_to_bytes(bctx, g1_bases.as<uint32_t>(), bctx->scalars.as<uint32_t>(), n_g1, 1, pair_g1 + 96, pair_inf + 1);
        }();
        if (rhs_rc != DR_OK) rhs_err = dr_last_error();
    });
    struct RhsJoiner {
        std::thread& t;
        ~RhsJoiner() { if (t.joinable()) t.join(); }
    } rhs_joiner{rhs_thread};
    HIP_TRY(hipMemcpyAsync(ctx->scalars.p, lhs_sc.data(), n_g1 * 32, hipMemcpyHostToDevice, st));
    TRY(msm_to_bytes(ctx, g1_bases.as<uint32_t>(), ctx->scalars.as<uint32_t>(), n_g1, 1, pair_g1, pair_inf));
    rhs_thread.join();
    tr_.mark("g1 msms");
    if (rhs_rc != DR_OK) return fail(rhs_rc, rhs_err.empty() ? "rhs MSM failed" : rhs_err);
    const int inf_r = pair_inf[1];
    // (a vanishing rhs can only come from r1 = r2 = 0 or infinity openings: the pairing equation then demands lhs = O)
    if (!inf_r) {                                                                 // e(lhs, G2[0]) * e(-rhs, G2[1]) == 1
        drh::Fq y;
        if (!drh::Fq::load_be(y, pair_g1 + 144)) return fail(DR_ERR_DEVICE, "MSM result out of range");
        y.neg().store_be(pair_g1 + 144);
    }
    int pok = 0;
    TRY(dr_pairing_check(pair_g1, vk->g2, 2, &pok));
    tr_.mark("pairing");
    side.join();
    tr_.mark("pedersen join");
    if (side_rc != DR_OK) return fail(side_rc, side_err.empty() ? "Pedersen part failed" : side_err);
    *ok = pok && ped_ok;
    return DR_OK;
}

int dr_ringvrf_verify_batch(dr_ctx* ctx, const dr_vrf_suite* suite, const dr_ring_verifier_key* vk, size_t batch, const uint8_t* proofs,
                            const uint8_t* inputs, const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
                            const uint64_t* salt_off, const uint8_t seed32[32], int* ok) {
    try {
        return ringvrf_verify_batch_impl(ctx, suite, vk, batch, proofs, inputs, in_off, ads, ad_off, salts, salt_off, seed32, ok);
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native verifier: ") + e.what());
    }
}

// PedersenVRF.prove for a batch (pedersen/vrf.py:86-126): 192 bytes per proof; same code as the Pedersen part of
// dr_ringvrf_prove_batch.  out_aux (nullable): per proof O, Y_bar, R, O_k affine (4*64) and the blinding factor (32).
int dr_pedersen_prove_batch(dr_ctx* ctx, const dr_vrf_suite* suite, size_t batch, const uint8_t* alphas, const uint64_t* alpha_off,
                            const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                            const uint8_t* secret_scalars, uint8_t* out_proofs, uint8_t* out_aux) {
    try {
        TRY(use_ctx(ctx));
        if (!alpha_off || !ad_off || !secret_scalars || !out_proofs) return fail(DR_ERR_INVALID, "null argument");
        if (batch == 0) return DR_OK;
        if (batch > 65536) return fail(DR_ERR_INVALID, "batch must be at most 65536 per call");
        drh::VrfSuite su;
        TRY(load_suite(suite, su));
        for (size_t i = 0; i < batch; i++)
            if (alpha_off[i + 1] < alpha_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
                return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
        PhaseTrace tr_("pedersen_prove_batch");
        PedersenBatch ped(su, batch);
        TRY(ped.head(ctx, alphas, alpha_off, ads, ad_off, salts, salt_off, secret_scalars, tr_));
        TRY(ped.tail(ctx, out_proofs, 192, out_aux, DR_PEDERSEN_AUX_BYTES));
        tr_.mark("tail");
        return DR_OK;
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native prover: ") + e.what());
    }
}

// PedersenVRF.batch_verify (pedersen/vrf.py:171-242) over ENCODED proofs (192 bytes each): point decoding + subgroup
// checks and hash-to-curve on the GPU, challenges on worker threads, one (5B+2)-point MSM.  *ok = 1 iff all verify.
int dr_pedersen_verify_batch(dr_ctx* ctx, const dr_vrf_suite* suite, size_t batch, const uint8_t* proofs, const uint8_t* inputs,
                             const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                             int* ok) {
    try {
        TRY(use_ctx(ctx));
        if (!proofs || !in_off || !ad_off || !ok) return fail(DR_ERR_INVALID, "null argument");
        *ok = 0;
        if (batch == 0) { *ok = 1; return DR_OK; }
        if (batch > 65536) return fail(DR_ERR_INVALID, "batch must be at most 65536 per call");
        drh::VrfSuite su;
        TRY(load_suite(suite, su));
        const size_t B = batch;
        const drh::Mod256& mn = su.cv->n;
        for (size_t i = 0; i < B; i++)
            if (in_off[i + 1] < in_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
                return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
        std::vector<uint8_t> te_enc(B * 128), te_xy(B * 256), flags(B * 4), in_pts(B * 64);
        for (size_t i = 0; i < B; i++) {
            std::memcpy(te_enc.data() + 128 * i, proofs + 192 * i, 128);
            uint64_t v[4];
            for (int k = 0; k < 2; k++) { drh::load_le32(proofs + 192 * i + 128 + 32 * k, v); if (drh::Mod256::geq(v, mn.m)) return DR_OK; }   // dec_scalar
        }
        TRY(te_decode_points(ctx, su.cv->id, false, te_enc.data(), 4 * B, te_xy.data(), flags.data()));
        for (size_t i = 0; i < 4 * B; i++) if (!flags[i]) return DR_OK;
        TRY(encode_to_curve_msgs(ctx, su, B, inputs, in_off, salts, salt_off, in_pts.data()));
        int ped_ok = 0;
        TRY(pedersen_verify_core(ctx, su, B, proofs, 192, te_xy, in_pts, ads, ad_off, ped_ok));
        *ok = ped_ok;
        return DR_OK;
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native verifier: ") + e.what());
    }
}

// TinyVRF.prove / ThinVRF.prove for a batch (vrf/ietf/tiny.py:53-70, thin.py): I = encode_to_curve, pk = x G, O = x I;
// transcript over the two (input, output) pairs (G, pk), (I, O); delinearised input M = G + z I; k = nonce; R = k M;
// c = challenge(R); s = k + c x.  Tiny proof = O || c (16) || s (80 bytes), Thin proof = O || R || s (96 bytes).
int dr_ietf_prove_batch(dr_ctx* ctx, const dr_vrf_suite* suite, int thin, size_t batch, const uint8_t* alphas, const uint64_t* alpha_off,
                        const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                        const uint8_t* secret_scalars, uint8_t* out_proofs, uint8_t* out_aux) {
    try {
        TRY(use_ctx(ctx));
        if (!alpha_off || !ad_off || !secret_scalars || !out_proofs) return fail(DR_ERR_INVALID, "null argument");
        if (batch == 0) return DR_OK;
        if (batch > 65536) return fail(DR_ERR_INVALID, "batch must be at most 65536 per call");
        drh::VrfSuite su;
        TRY(load_suite(suite, su));
        const size_t B = batch, plen = thin ? 96 : 80;
        const drh::Mod256& mn = su.cv->n;
        const int cv = su.cv->id;
        for (size_t i = 0; i < B; i++)
            if (alpha_off[i + 1] < alpha_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
                return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
        std::vector<uint8_t> xs(B * 32), inputs(B * 64);
        for (size_t i = 0; i < B; i++) {
            uint64_t x[4];
            mn.reduce_bytes(secret_scalars + 32 * i, 32, false, x);
            drh::store_le32(x, xs.data() + 32 * i);
        }
        TRY(encode_to_curve_msgs(ctx, su, B, alphas, alpha_off, salts, salt_off, inputs.data()));
        // pk_i = x_i G and O_i = x_i I_i in one launch
        std::vector<uint8_t> pts(2 * B * 64), sc(2 * B * 32), firsts(2 * B * 64);
        for (size_t i = 0; i < B; i++) {
            std::memcpy(pts.data() + 64 * i, su.generator, 64);
            std::memcpy(pts.data() + 64 * (B + i), inputs.data() + 64 * i, 64);
            std::memcpy(sc.data() + 32 * i, xs.data() + 32 * i, 32);
            std::memcpy(sc.data() + 32 * (B + i), xs.data() + 32 * i, 32);
        }
        TRY(te_scalar_mul_batch(ctx, cv, pts.data(), sc.data(), 2 * B, firsts.data()));
        const uint8_t* pks = firsts.data();
        const uint8_t* outs = firsts.data() + 64 * B;
        // transcripts, delinearisation scalar z, nonces
        std::vector<drh::Bytes> tr(B);
        std::vector<uint8_t> gpts(B * 128), gsc(B * 64), ks(B * 32);
        std::vector<int> bad(B, 0);
        uint8_t enc_g[32];
        drh::enc_te_point(su.generator, enc_g);
        drh::parallel_for(B, [&](size_t i) {
            drh::Bytes& t = tr[i];
            t = su.suite_id;
            drh::put8(t, thin ? 0x01 : 0x00);                      // THIN_VRF / TINY_VRF
            drh::put_le64(t, 2);
            uint8_t enc[32];
            drh::put(t, enc_g, 32);
            drh::enc_te_point(pks + 64 * i, enc); drh::put(t, enc, 32);
            drh::enc_te_point(inputs.data() + 64 * i, enc); drh::put(t, enc, 32);
            drh::enc_te_point(outs + 64 * i, enc); drh::put(t, enc, 32);
            size_t adl = ad_off[i + 1] - ad_off[i];
            drh::put_le64(t, adl);
            drh::put(t, ads + ad_off[i], adl);
            drh::Bytes d = t;
            drh::put8(d, 0x30);                                    // DELINEARIZE
            uint8_t raw[16];
            drh::vrf_squeeze(su.xof, d.data(), d.size(), raw, 16);
            uint64_t z[4], x[4], k[4], one[4] = {1, 0, 0, 0};
            mn.reduce_bytes(raw, 16, false, z);
            std::memcpy(gpts.data() + 128 * i, su.generator, 64);
            std::memcpy(gpts.data() + 128 * i + 64, inputs.data() + 64 * i, 64);
            drh::store_le32(one, gsc.data() + 64 * i);
            drh::store_le32(z, gsc.data() + 64 * i + 32);
            drh::load_le32(xs.data() + 32 * i, x);
            if (!drh::vrf_nonce(su, t, x, k)) bad[i] = 1;
            drh::store_le32(k, ks.data() + 32 * i);
        });
        for (size_t i = 0; i < B; i++) if (bad[i]) return fail(DR_ERR_INVALID, "nonce scalar is zero");
        std::vector<uint8_t> merged(B * 64), rs(B * 64);
        TRY(te_msm_groups(ctx, cv, gpts.data(), gsc.data(), B, 2, merged.data()));
        TRY(te_scalar_mul_batch(ctx, cv, merged.data(), ks.data(), B, rs.data()));
        drh::parallel_for(B, [&](size_t i) {
            uint8_t* out = out_proofs + plen * i;
            uint8_t enc_r[32];
            drh::enc_te_point(outs + 64 * i, out);
            drh::enc_te_point(rs.data() + 64 * i, enc_r);
            uint64_t c[4], x[4], k[4], s[4];
            drh::vrf_challenge(su, tr[i], enc_r, 1, c);
            drh::load_le32(xs.data() + 32 * i, x);
            drh::load_le32(ks.data() + 32 * i, k);
            mn.mul(c, x, s);
            mn.add(s, k, s);
            if (out_aux) {
                std::memcpy(out_aux + 128 * i, outs + 64 * i, 64);
                std::memcpy(out_aux + 128 * i + 64, rs.data() + 64 * i, 64);
            }
            if (thin) {
                std::memcpy(out + 32, enc_r, 32);
                drh::store_le32(s, out + 64);
            } else {
                uint8_t cb[32];
                drh::store_le32(c, cb);
                std::memcpy(out + 32, cb, 16);
                drh::store_le32(s, out + 48);
            }
        });
        return DR_OK;
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native prover: ") + e.what());
    }
}

}  // extern "C"
