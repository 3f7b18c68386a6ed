// libdotring_hip.so — C ABI, part 2 of 5: the G1 Pippenger pipeline over kernels_g1.hip.h, seam B (SRS handles, MSM entry
// points, G1 codecs) and the pairing entry points.
#include "capi_internal.hpp"
#include "kernels_g1.hip.h"
#include "kernels_te_msm.hip.h"

using namespace dri;

namespace {
// ---- window plan for the GPU Pippenger.  Scalars are reduced mod r (< 2^255) on the device and the 256 bits
// are tiled by W = ceil(256/c) windows of width cmax or cmax-1 (see WindowTable).  Work ~ W*n mixed adds +
// W*2^(c-1)*(2 full adds) + per-chunk scalar multiplications; a full add costs ~1.4 mixed adds; pick the c
// minimising that, within [7,16] (W <= 37 fits the table).
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

// One small MSM over plain bases (the verifier's two folds of ~7 k and ~2 k points, KZG commits of a few thousand coefficients) is a
// latency chain, not a throughput problem: the lanes of a launch are far fewer than the chip holds, so what counts is the longest
// dependent chain — the bucket walk's ~(m + 3 sqrt(m)) mixed additions for m points per bucket, then the 4-bucket chunks' running sums
// and (c - 3)-bit double-and-add, then the fold.  In units of one dependent addition (~11 us mixed, ~15 us full on a lone wave):
int pick_window_latency(size_t n) {
    int best = 7;
    double best_t = 1e300;
    for (int c = 7; c <= 13; c++) {
        const int W = (256 + c - 1) / c;
        const double H = (double)(1u << (c - 1));
        if ((double)W * H > 131072.0) continue;
        const double m = (double)n / H;
        // (from 256 buckets per window on the reduction is the workgroup scan: 2 x buckets per lane + 17 additions, msm_device)
        const double reduce = H >= 256.0 ? 15.0 * (2.0 * std::min(8.0, std::max(1.0, H / 1024.0)) + 17.0)
                                         : 15.0 * (8.0 + 1.5 * (c - 3)) + 15.0 * (std::log2(std::max(H / 4.0, 2.0)) + 4.0);
        const double t = 11.0 * (m + 3.0 * std::sqrt(m) + 1.0) + reduce;
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

struct MsmPlan {
    dr::WindowTable wt;
    int W;
    uint32_t H, L, T;
};
dr::WindowTable make_window_table(int c, int bits = 256) {      // `bits` scalar bits tiled by ceil(bits / c) windows of near-equal width
    dr::WindowTable wt;
    wt.W = (bits + c - 1) / c;
    int base = bits / wt.W, rem = bits % wt.W;
    wt.cmax = base + (rem ? 1 : 0);
    int bit = 0;
    for (int w = 0; w < wt.W; w++) {
        int width = base + (w >= wt.W - rem ? 1 : 0);
        wt.start[w] = (uint8_t)bit;
        wt.width[w] = (uint8_t)width;
        wt.row[w] = (uint8_t)w;
        bit += width;
    }
    wt.odd = 0;
    return wt;
}

MsmPlan make_plan(size_t n, int force_c, bool latency_bound = false) {
    MsmPlan p;
    int c = window_ok(force_c) ? force_c : (latency_bound ? pick_window_latency(n) : pick_window(n));
    p.W = (256 + c - 1) / c;
    int base = 256 / p.W, rem = 256 % p.W;       // `rem` windows of width base+1 (placed on top), the rest base
    p.wt.W = p.W;
    p.wt.cmax = base + (rem ? 1 : 0);
    int bit = 0;
    for (int w = 0; w < p.W; w++) {
        int width = base + (w >= p.W - rem ? 1 : 0);
        p.wt.start[w] = (uint8_t)bit;
        p.wt.width[w] = (uint8_t)width;
        p.wt.row[w] = (uint8_t)w;
        bit += width;
    }
    p.wt.odd = 0;
    p.H = 1u << (p.wt.cmax - 1);
    p.L = std::min<uint32_t>(p.H, 16u);
    p.T = p.H / p.L;
    return p;
}
}  // namespace


// fewer first-level chunks than this over all sets of a table MSM: chunks of 4 buckets instead of 16 (shorter dependent chains for
// launches that do not fill the chip)
// (2^15 since the end of round 4 — 256 sets of 2048 buckets: same-box sweeps of prove_batch at 256 / 384 / 512 proofs gave 20.7 / 28.4 / 32.9 ms with
//  2^17, 20.2 / 27.1 / 31.8 with 2^16, 19.6 / 26.6 / 31.6 with 2^15; 1024 proofs the same)
static constexpr size_t l4_below() { return (size_t)1 << 15; }

// DOTRING_SRS_TILING=rows: batched MSMs keep the window rows of a bit-row table (default: the non-adjacent form below)
static bool naf_tiling_on() {
    static const bool v = [] {
        const char* e = std::getenv("DOTRING_SRS_TILING");
        return !(e && std::strcmp(e, "rows") == 0);
    }();
    return v;
}

// A table with a row per bit and hundreds of MSMs (the batched prover): a digit may sit at ANY bit position, so every scalar is recoded
// in width-w non-adjacent form (msm_recode.hip.h: for_each_wnaf_digit): 256 / (w + 1) + ~0.55 odd digits on average — 18.8 for w = 13
// where 13-bit windows have 20 — into 2^(w-2) odd-multiple buckets per set (value of a set: sum_j (2j + 1) B_j).  Round 3 reached the
// same bucket count with 13-bit windows whose digits 2^k u went to bucket (u - 1) / 2 with the point of row start + k, which put every
// power of two of a window into bucket 0 and needed twin buckets and a merge kernel; the non-adjacent form has odd digits only and
// shares the bits at the top evenly, so the fullest bucket holds ~4x the average list and stays in the one-lane walk.
// Needs the per-set LDS sort and the set-scan reduction (hundreds of sets; with fewer the L = 4 latency reduction of msm_device
// applies); w <= 13: the staged sort's u16 digit rows hold 11 bucket bits + 4 offset bits + sign.  Among the widths that qualify the
// cheapest wins: n x digits bucket additions + ~1.5 addition-equivalents per bucket of the reduction (measured: level 1 + set scan
// per bucket against the walk's time per entry): w = 13 for the 3N = 6144-point vectors of domain 2048 and the 12288 of domain 4096.
Tiling tiling_for(const MsmTable& t, size_t n, size_t batch) {
    Tiling none{0, 0, 0, 0.0};
    if (!t.table || !t.bit_rows || t.naf_delta == -1 || batch < 256 || n == 0 || !naf_tiling_on()) return none;
    const int cn = t.wt.cmax;
    const int lo = t.naf_delta >= 0 ? cn + t.naf_delta : cn - 1, hi = t.naf_delta >= 0 ? cn + t.naf_delta : cn + 2;
    Tiling best = none;
    double best_cost = 0;
    for (int w = lo; w <= hi; w++) {
        if (w < 9 || w > 13) continue;
        const size_t H = (size_t)1 << (w - 2), slots = (256 + w - 1) / w;
        if (batch * (H / 16) < l4_below()) continue;
        if ((n + 64) * slots > ((size_t)1 << 20) || batch * (n + 64) * slots >= (1ull << 32)) continue;
        const double digits = 256.0 / (w + 1) + 0.55;        // (+ the evenly shared digits at the top and the end effects: 18.8 measured at w = 13)
        if (t.naf_delta < 0 && (double)n * digits / (double)H > 160.0) continue;      // the fullest lists (~4x) stay near the one-lane limit
        // (1.5 addition-equivalents per bucket: with the widths 12 and 13 both admitted at ring 256 — 3N = 3072 terms, 1024 proofs — the
        //  narrower one saved 0.15 ms of reduction per step and cost 1.15 ms of walk; the 5.2 of the first fit priced the reduction at its
        //  issue rate, which launches of this size do not reach)
        const double cost = (double)n * digits + 1.5 * (double)H;
        if (!best.mode || cost < best_cost) { best = Tiling{2, w, (int)slots, digits}; best_cost = cost; }
    }
    return best;
}

int msm_device(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch,
               std::vector<drh::G1>& results, const MsmTable* tbl, bool exact_streams) {
    results.assign(batch, drh::G1::inf());
    const uint32_t* const d_bases_in = d_bases;          // (d_bases is redirected to the table below; a second run starts from the caller's)
    if (n == 0 || batch == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "MSM size must be below 2^31");
    const bool single = tbl != nullptr && tbl->table != nullptr;
    PhaseTrace tr_("msm_device");                 // DOTRING_TRACE=1: where the wall time of a call goes
    // The plan — tiling of the scalars, index groups, reduction chunks — and the two paths every later step depends on: the per-set
    // LDS sort and the set-scan reduction.  The non-adjacent form over a bit-row table needs both; they are decided HERE, from one set
    // of predicates, and a plan that would not get them falls back to the table's window rows before anything is launched.
    MsmPlan pl;
    uint32_t groups = 1;
    size_t windows = 0, bsets = 0, per_set_scalars = 0, per_set_digits = 0;
    bool lds_sort = false, setscan = false;
    const auto make = [&](bool allow_naf) {
        pl = make_plan(n, g_force_c, !single && batch == 1 && n <= 32768);
        if (single) {
            // a table with a row per bit and hundreds of MSMs (the batched prover): non-adjacent form, buckets for odd multiples (tiling_for)
            const Tiling tl = allow_naf ? tiling_for(*tbl, n, batch) : Tiling{0, 0, 0, 0.0};
            const bool odd = tl.mode != 0;
            dr::WindowTable wo{};
            if (odd) {                                       // slots of the non-adjacent form: positions [c j, c j + c) of k << shift
                wo.W = tl.slots;
                wo.cmax = tl.c;
                const int shift = wo.W * tl.c - 256;          // (msm_recode.hip.h: for_each_wnaf_digit)
                for (int j = 0; j < wo.W; j++) {
                    wo.start[j] = (uint8_t)(tl.c * j);
                    wo.row[j] = (uint8_t)(j ? tl.c * j - shift : 0);
                    wo.width[j] = (uint8_t)tl.c;
                }
                wo.odd = 2;
            }
            pl.wt = odd ? wo : tbl->wt;
            pl.W = pl.wt.W;
            pl.H = odd ? 1u << (pl.wt.cmax - 2) : 1u << (pl.wt.cmax - 1);
            pl.L = std::min<uint32_t>(pl.H, 16u);
            pl.T = pl.H / pl.L;
        }
        // table mode: split the points of each MSM into index groups when one bucket set per MSM would leave lanes idle
        groups = 1;
        if (single) {
            // 8 waves per SIMD: finer slices balance better than 4 (2^20 bases: accumulate 3.98 -> 3.6 ms); below 2^19 points half of that —
            // every bucket is another lane-step of the reduction's chain, and the walk is short anyway (2^16 pairs over 16-bit windows:
            // 8 groups 0.875 ms, 16 groups 1.03, 2 groups 0.96)
            const size_t target_lanes = n >= ((size_t)1 << 19) ? 524288 : 262144;
            while (groups < 64 && batch * groups * (size_t)pl.H < target_lanes && (size_t)n / (groups * 2) >= 64) groups *= 2;
            static const int force_groups = std::getenv("DOTRING_MSM_GROUPS") ? std::atoi(std::getenv("DOTRING_MSM_GROUPS")) : 0;
            if (force_groups > 0 && batch == 1 && (size_t)n / (size_t)force_groups >= 64) groups = (uint32_t)force_groups;
        }
        windows = batch * (size_t)pl.W;                   // digit rows
        bsets = single ? batch * groups : windows;        // bucket sets
        // few bucket sets of moderate size (a single MSM over a window table): the reduction is a latency chain of
        // 2L additions + a log2(H)-bit double-and-add + the fold of H/L partial sums; L = 4 makes it ~40 % shorter
        if (single && pl.L == 16 && pl.H >= 256 && pl.H <= 4096 && bsets * (size_t)(pl.H / 16) < l4_below()) {
            pl.L = 4;
            pl.T = pl.H / 4;
        }
        // (one huge MSM, 16 groups x 32768 buckets: L stays 16 — measured 0.90 ms for the chunk kernel against 1.15 at L = 8 and 1.00
        //  at L = 4: every chunk pays a 15-bit double-and-add whatever its length)
        // the same for a small MSM over plain bases (the verifier's 2- and 11-point folds): 41 -> 16 dependent additions
        if (!single && pl.L == 16 && pl.H >= 16 && bsets * (size_t)(pl.H / 16) < ((size_t)1 << 12)) {
            pl.L = 4;
            pl.T = pl.H / 4;
        }
        per_set_scalars = single ? (n + groups - 1) / groups : n;
        per_set_digits = single ? per_set_scalars * (size_t)pl.W : n;
        // small bucket sets fed by a bounded number of digits (the batched prover): one workgroup sorts a set entirely in LDS
        lds_sort = pl.H <= dr::SORT_MAX_H && bsets >= 64 && per_set_digits <= (1u << 20) && bsets * per_set_digits < (1ull << 32);
        // many sets of <= 4096 buckets (every batched MSM of the prover): first level with 2 additions per bucket, then one workgroup
        // per set scans and folds its <= 256 chunk results
        setscan = pl.L == 16 && pl.T >= 8 && pl.T <= 256 && bsets >= 256;
    };
    make(true);
    if (pl.wt.odd && !(lds_sort && setscan)) make(false);
    if (single) {
        d_bases = tbl->table;
        if (((uint64_t)pl.wt.row[pl.W - 1] + pl.wt.cmax + 1) * tbl->stride >= (1ull << 31)) return fail(DR_ERR_INVALID, "window table too large");
    }
    const size_t nbuckets = bsets * (size_t)pl.H;
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
    TRY(ctx->cell_off.reserve((ncells + 2) * 4));
    TRY(ctx->buckets.reserve(nbuckets * 192));
    TRY(ctx->partial.reserve((bsets * pl.T + bsets * (pl.T / 256 + 1)) * 192));
    TRY(ctx->winsum.reserve(bsets * 192));
    // segment sums of lists >= 1024 entries: sum_i ceil(len_i / seg) <= total / seg + n_heavy.  Up to 1024 heavy lists: seg >= 1024, so
    // total / 1024 + 1024; more of them: seg = 4096 and n_heavy <= total / 1024, so total / 4096 + total / 1024 (heavy_segment_size)
    TRY(ctx->heavy.reserve((ndigits / 1024 + ndigits / 4096 + dr::G1_HEAVY_SLOTS / 2 + 64) * 192));
    hipStream_t st = ctx->stream;
    auto exclusive_scan = [&](const uint32_t* in, uint32_t* out, size_t count) {
        const unsigned nt = div_up(count, dr::SCAN_TILE);
        hipLaunchKernelGGL(dr::k_scan_tiles, dim3(nt), dim3(dr::SCAN_BLOCK), 0, st, in, out, ctx->tiles.as<uint32_t>(), count);
        hipLaunchKernelGGL(dr::k_scan_tile_sums, dim3(1), dim3(dr::SCAN_BLOCK), 0, st, ctx->tiles.as<uint32_t>(), nt, ctx->tiles.as<uint32_t>() + nt);
        hipLaunchKernelGGL(dr::k_scan_add, dim3(div_up(count, 256)), dim3(256), 0, st, out, ctx->tiles.as<uint32_t>(), count);
    };
    // Sorting the digits by bucket.  Small bucket sets fed by a bounded number of digits (the batched prover) are
    // sorted by one workgroup each, entirely in LDS; a few huge sets (one 2^20-point MSM) use global atomics.
    // a few huge sets over a window table (one 2^20-point MSM): two-pass partition sort (k_g1_part_scatter / k_g1_part_sort)
    uint32_t part_p = 1, part_shift = 0;
    {
        const size_t chunk = dr::PART_STAGE - dr::PART_SLACK;
        while (part_p < dr::PART_MAX_P && (pl.H / part_p > dr::PART_MAX_HP || per_set_digits / part_p > chunk - chunk / 16)) part_p *= 2;
        while ((pl.H >> part_shift) > part_p) part_shift++;
    }
    const bool part_sort = !lds_sort && single && batch == 1 && pl.W <= 32 && pl.H >= part_p &&
                           pl.H / part_p <= dr::PART_MAX_HP && per_set_digits / part_p <= 48 * (size_t)(dr::PART_STAGE - dr::PART_SLACK) &&
                           bsets * per_set_digits < (1ull << 32) && per_set_digits >= 65536;
    if (lds_sort) {
        dr::SortSetParams sp;
        sp.n = (uint32_t)n; sp.batch = (uint32_t)batch; sp.H = pl.H; sp.groups = groups; sp.single = single ? 1 : 0;
        sp.tbl_stride = single ? tbl->stride : 0; sp.tbl_offset = single ? tbl->offset : 0;
        sp.capacity = (uint32_t)per_set_digits;
        sp.short_from = single ? tbl->short_from : 0xffffffffu;
        sp.n_short = single ? std::min<uint32_t>(tbl->n_short, (uint32_t)n) : 0;
        sp.sets = (uint32_t)bsets;
        sp.fold = single && tbl->fold_sign ? 1 : 0;
        TRY(ctx->sorted.reserve(bsets * per_set_digits * 4));
        // sets of more than a few thousand entries: the sorted segment is assembled in LDS and written in whole lines
        // (k_g1_sort_sets_staged)
        const bool small_h = pl.H <= dr::SORT2_SMALL_H;
        const uint32_t stage_chunk = (small_h ? dr::SORT2_CAP_SMALL_H : dr::SORT2_CAP_LARGE_H) - dr::SORT2_SLACK;
        const bool staged = per_set_digits >= 4096 && per_set_digits / stage_chunk + 1 <= dr::SORT2_MAX_CHUNKS;
        if (staged) {
            sp.n_pad = (uint32_t)((per_set_scalars + 7) & ~(size_t)7);
            sp.digits_per_set = sp.n_pad * (uint32_t)(single ? pl.W : 1);
            TRY(ctx->digits.reserve(bsets * (size_t)sp.digits_per_set * 2));
        }
        TRY(launch(ctx, "k_g1_sort_sets", [&] {
            if (staged && small_h)
                hipLaunchKernelGGL((dr::k_g1_sort_sets_staged<dr::SORT2_SMALL_H, dr::SORT2_CAP_SMALL_H>), dim3((unsigned)bsets), dim3(dr::SORT2_BLOCK),
                                   0, st, d_scalars, pl.wt, sp, ctx->digits.as<uint16_t>(), ctx->counts.as<uint32_t>(),
                                   ctx->offsets.as<uint32_t>(), ctx->sorted.as<uint32_t>());
            else if (staged)
                hipLaunchKernelGGL((dr::k_g1_sort_sets_staged<dr::SORT_MAX_H, dr::SORT2_CAP_LARGE_H>), dim3((unsigned)bsets), dim3(dr::SORT2_BLOCK),
                                   0, st, d_scalars, pl.wt, sp, ctx->digits.as<uint16_t>(), ctx->counts.as<uint32_t>(),
                                   ctx->offsets.as<uint32_t>(), ctx->sorted.as<uint32_t>());
            else
                hipLaunchKernelGGL(dr::k_g1_sort_sets, dim3((unsigned)bsets), dim3(dr::SORT_BLOCK), 0, st, d_scalars, pl.wt, sp,
                                   ctx->counts.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->sorted.as<uint32_t>());
        }));
    }
    // partition sort: every partition stream gets room for 4x its share of the set's entries (at least 64 k records of 8 bytes: 0.44 GB
    // of streams at 2^20 pairs and 512 partitions).  Pass A counts every entry whether or not it fitted; a distribution that overfills a
    // stream (few distinct scalars) shows in the fill counters, and pass A then runs again with the streams packed at their exact
    // offsets.  Only a partition of more than 64 stage chunks (2.2 M entries) takes the global-atomic path below.
    bool sorted_done = lds_sort;
    const uint32_t* part_flag = nullptr;
    uint32_t part_overflow = 0;
    if (!sorted_done && part_sort) {
        dr::PartParams pp{};
        pp.n = (uint32_t)n; pp.H = pl.H; pp.groups = groups; pp.P = part_p; pp.pshift = part_shift;
        pp.tile = std::min<uint32_t>(2048, dr::PART_TILE_ENTRIES / (uint32_t)pl.W);
        pp.tiles_per_set = (uint32_t)((per_set_scalars + pp.tile - 1) / pp.tile);
        pp.cap_part = (uint32_t)std::min<size_t>(per_set_digits, std::max<size_t>(4 * per_set_digits / part_p, 65536));
        pp.capacity = (uint32_t)per_set_digits;
        pp.tbl_stride = tbl->stride; pp.tbl_offset = tbl->offset;
        for (int w = 0; w < pl.W; w++) pp.row[w] = pl.wt.row[w];
        const size_t nparts = bsets * (size_t)part_p;
        TRY(ctx->digits.reserve(nparts * pp.cap_part * 8));
        TRY(ctx->cursor.reserve((nparts + 1) * 4));               // fill counters + the overflow flag
        TRY(ctx->part_base.reserve(nparts * 4));
        TRY(ctx->sorted.reserve(bsets * per_set_digits * 4));
        uint32_t* d_flag = ctx->cursor.as<uint32_t>() + nparts;
        const uint32_t* d_exact = nullptr;
        bool fits = true;
        if (exact_streams) {
            // second run of this call: the first try overfilled a stream.  Its fill counters — still in place — are exact whether or not
            // a record fitted: pack the streams at their exact offsets (they take at most the first try's room) and scatter again.
            std::vector<uint32_t> fill(nparts), exact(nparts);
            HIP_TRY(hipMemcpyAsync(fill.data(), ctx->cursor.p, nparts * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            uint64_t run = 0;
            uint32_t most = 0;
            for (size_t q = 0; q < nparts; q++) { exact[q] = (uint32_t)run; run += fill[q]; most = std::max(most, fill[q]); }
            fits = most <= dr::PART_MAX_CHUNKS * (dr::PART_STAGE - dr::PART_SLACK) && run <= nparts * (uint64_t)pp.cap_part;
            if (fits) {
                HIP_TRY(hipMemcpyAsync(ctx->part_base.p, exact.data(), nparts * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));               // `exact` leaves scope
                d_exact = ctx->part_base.as<uint32_t>();
            }
        }
        if (fits) {
            HIP_TRY(hipMemsetAsync(ctx->cursor.p, 0, (nparts + 1) * 4, st));
            TRY(launch(ctx, "k_g1_part_scatter", [&] {
                hipLaunchKernelGGL(dr::k_g1_part_scatter, dim3((unsigned)(bsets * pp.tiles_per_set)), dim3(dr::PART_BLOCK), 0, st, d_scalars, pl.wt, pp,
                                   ctx->cursor.as<uint32_t>(), d_exact, ctx->digits.as<uint2>());
            }));
        }
        if (!exact_streams) part_flag = d_flag;                   // read back with the results: set => this call runs again, exact
        if (fits) {
            TRY(launch(ctx, "k_g1_part_sort", [&] {
                hipLaunchKernelGGL(dr::k_g1_part_sort, dim3((unsigned)nparts), dim3(dr::PART_BLOCK), 0, st, ctx->digits.as<uint2>(),
                                   ctx->cursor.as<uint32_t>(), d_exact, pp, d_flag, ctx->counts.as<uint32_t>(), ctx->offsets.as<uint32_t>(),
                                   ctx->sorted.as<uint32_t>());
            }));
            sorted_done = true;
        }
    }
    if (!sorted_done) {
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
                               (uint32_t)n, windows, pl.H, single ? pl.W : 0, pl.wt, single ? tbl->stride : 0u, single ? tbl->offset : 0u, groups,
                               ctx->offsets.as<uint32_t>(), ctx->cursor.as<uint32_t>(),
                               ctx->sorted.as<uint32_t>());
        }));
    }
    tr_.mark("sort");
    // size-ordered bucket permutation for the accumulate kernel
    TRY(launch(ctx, "k_size_sort", [&] {
        hipLaunchKernelGGL(dr::k_size_hist, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cells.as<uint32_t>());
        exclusive_scan(ctx->cells.as<uint32_t>(), ctx->cell_off.as<uint32_t>(), ncells);
        hipLaunchKernelGGL(dr::k_size_place, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cell_off.as<uint32_t>(), ctx->perm.as<uint32_t>());
        // the launch's limit between the one-lane walk and the 16-lane walk, from the histogram (two words behind the cell offsets)
        hipLaunchKernelGGL(dr::k_size_pick, dim3(1), dim3(256), 0, st, ctx->cell_off.as<uint32_t>(), szblocks, nbuckets, ctx->cell_off.as<uint32_t>() + ncells);
    }));
    const uint32_t* d_pick = ctx->cell_off.as<uint32_t>() + ncells;
    TRY(launch(ctx, "k_g1_accumulate", [&] {
        const uint32_t pt_words = single ? tbl->pt_words : 24u;
        hipLaunchKernelGGL(dr::k_g1_accumulate, dim3(div_up(nbuckets, 256)), dim3(256), 0, st, d_bases, pt_words,
                           ctx->sorted.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(), d_pick,
                           ctx->buckets.as<uint32_t>(), nbuckets);
        // lists of 256 entries or more (the lowest odd-multiple buckets of every set; skewed scalars): 16 lanes or a wave each; both
        // launches return at once when there are none
        hipLaunchKernelGGL(dr::k_g1_accumulate_long<16>, dim3(2048), dim3(64), 0, st, d_bases, pt_words, ctx->sorted.as<uint32_t>(),
                           ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(), ctx->cell_off.as<uint32_t>(),
                           szblocks, d_pick, ctx->buckets.as<uint32_t>());
        // lists of 1024 entries or more (equal scalars, 0 / 1 columns): segments spread over 2048 waves, then one wave per bucket folds
        // its segment sums
        hipLaunchKernelGGL(dr::k_g1_accumulate_heavy, dim3(dr::G1_HEAVY_SLOTS), dim3(64), 0, st, d_bases, pt_words, ctx->sorted.as<uint32_t>(),
                           ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(), ctx->cell_off.as<uint32_t>(),
                           szblocks, ctx->buckets.as<uint32_t>(), ctx->heavy.as<uint32_t>());
        hipLaunchKernelGGL(dr::k_g1_heavy_fold, dim3(256), dim3(64), 0, st, ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(),
                           ctx->cell_off.as<uint32_t>(), szblocks, ctx->heavy.as<uint32_t>(), ctx->buckets.as<uint32_t>());
    }));
    // many bucket sets (batched prover): level-wise reduction, 2 additions per entry and no scalar multiplications;
    // few sets (single MSMs): chunk sums + double-and-add, whose latency is one short chain
    const bool leveled = pl.L == 16 && pl.H >= 256 && bsets * (size_t)(pl.H / 16) >= ((size_t)1 << 18);
    if (pl.wt.odd && !(setscan && lds_sort)) return fail(DR_ERR_DEVICE, "internal: non-adjacent form on a path that does not support it");   // (cannot happen: see the plan)
    // a single MSM over a wide window table (H >= 8192 buckets per index group): workgroup scan, (V, S) pairs to the host
    // buckets per lane of that scan: as few as keep the launch within one wave per SIMD (65536 lanes), at most 8
    // One plain MSM of a few thousand points (the batch verifier's folds, a single KZG.commit; ~22 windows of 256 .. 4096 buckets): the same
    // scan, four workgroups per window — 2 x buckets-per-lane + 17 additions deep where the chunk kernel (8 running-sum additions, a
    // double-and-add over the chunk index) and its fold were ~37: reduction 0.49 -> 0.3 ms of a 0.85 ms call.
    const bool plain_one = !single && batch == 1 && !setscan && !leveled && pl.H >= 256 && !pl.wt.odd;
    uint32_t ws_per_lane = 1;
    if (plain_one) ws_per_lane = std::min<uint32_t>(8u, std::max<uint32_t>(1u, pl.H / 1024u));
    else while (ws_per_lane < 8 && bsets * (size_t)pl.H > (size_t)65536 * ws_per_lane) ws_per_lane *= 2;
    const uint32_t ws_span = dr::WS_BLOCK * ws_per_lane;
    // ... and up to 32 MSMs over a window table whose launch does not fill the chip (RingVRF.prove of ONE proof: 1, 2 and 4 commitments,
    // up to 32 index groups of 512 .. 2048 buckets each; prove_batch of 8 / 16 / 32 proofs: 6.7 -> 6.5, 6.95 -> 6.4, 8.1 -> 7.6 ms): the chunk kernel's chain there was 8 additions + an 11-bit double-and-add +
    // the fold, ~0.65 ms per call; the scan is 19 additions deep
    const bool few_table = single && batch <= 32 && !setscan && !leveled && pl.H >= 256 && !pl.wt.odd && bsets * (size_t)pl.H <= ((size_t)1 << 19);
    const bool wgscan = (plain_one || few_table || (!setscan && !leveled && single && batch == 1 && pl.H >= 8192)) && pl.H % ws_span == 0;
    const size_t wg_per_set = pl.H / ws_span, wg_count = bsets * wg_per_set;
    if (setscan) {
        const size_t cnt = bsets * pl.T;
        TRY(ctx->partial.reserve(2 * cnt * 192));
        uint32_t* out_s = ctx->partial.as<uint32_t>();
        uint32_t* out_c = out_s + cnt * 48;
        TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_level1, dim3(div_up(cnt, 128)), dim3(128), 0, st, ctx->buckets.as<uint32_t>(), bsets, pl.H, 16u,
                               out_s, out_c);
        }));
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            const uint32_t per_block = dr::RS_BLOCK / (pl.T / dr::RS_GROUP);
            hipLaunchKernelGGL(dr::k_g1_reduce_set_scan, dim3(div_up(bsets, per_block)), dim3(dr::RS_BLOCK), 0, st, out_s, out_c, bsets, pl.T,
                               pl.wt.odd, ctx->winsum.as<uint32_t>());
        }));
    } else if (leveled) {
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
                    hipLaunchKernelGGL(dr::k_g1_reduce_level, dim3(div_up(cnt, 128)), dim3(128), 0, st, in_s, in_c, bsets, n, 16u, level, out_s, out_c);
                else
                    hipLaunchKernelGGL(dr::k_g1_reduce_level1, dim3(div_up(cnt, 128)), dim3(128), 0, st, in_s, bsets, n, 16u, out_s, out_c);
            }));
            in_s = out_s; in_c = out_c;
            off += 2 * cnt;
            n /= 16;
        }
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_final, dim3(div_up(bsets, 64)), dim3(64), 0, st, in_s, in_c, bsets, n, level, ctx->winsum.as<uint32_t>());
        }));
    } else if (wgscan) {
        // one huge bucket set per index group: workgroups of 2048 buckets scan and fold themselves (k_g1_reduce_wg_scan); their
        // (V, S) pairs are combined on the host below
        TRY(ctx->partial.reserve(2 * wg_count * 192));
        TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
            uint32_t* in = ctx->buckets.as<uint32_t>();
            uint32_t* out = ctx->partial.as<uint32_t>();
            const dim3 grid((unsigned)wg_count), block(dr::WS_BLOCK);
            if (ws_per_lane == 8) hipLaunchKernelGGL(dr::k_g1_reduce_wg_scan<8>, grid, block, 0, st, in, out);
            else if (ws_per_lane == 4) hipLaunchKernelGGL(dr::k_g1_reduce_wg_scan<4>, grid, block, 0, st, in, out);
            else if (ws_per_lane == 2) hipLaunchKernelGGL(dr::k_g1_reduce_wg_scan<2>, grid, block, 0, st, in, out);
            else hipLaunchKernelGGL(dr::k_g1_reduce_wg_scan<1>, grid, block, 0, st, in, out);
        }));
    } else {
        TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
            hipLaunchKernelGGL(dr::k_g1_reduce_chunks, dim3(div_up(bsets * pl.T, 128)), dim3(128), 0, st,
                               ctx->buckets.as<uint32_t>(), bsets, pl.H, pl.L, ctx->partial.as<uint32_t>());
        }));
        TRY(launch(ctx, "k_g1_reduce_windows", [&] {
            if (pl.T > 512 && pl.T % 256 == 0) {
                // thousands of chunk results per set (one huge MSM): fold 256 at a time first — 2 + 7 additions deep, then
                // T / 256 values per set — instead of T / 128 + 7 in one workgroup per set
                uint32_t* mid = ctx->partial.as<uint32_t>() + bsets * pl.T * 48;
                hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)(bsets * (pl.T / 256))), dim3(dr::RW_BLOCK), 0, st,
                                   ctx->partial.as<uint32_t>(), 256u, mid);
                hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)bsets), dim3(dr::RW_BLOCK), 0, st, mid, pl.T / 256, ctx->winsum.as<uint32_t>());
            } else {
                hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)bsets), dim3(dr::RW_BLOCK), 0, st,
                                   ctx->partial.as<uint32_t>(), pl.T, ctx->winsum.as<uint32_t>());
            }
        }));
    }

    static_assert(sizeof(drh::G1) == 192, "XYZZ layout");
    tr_.mark("enqueue");
    if (wgscan) {
        // set value = sum_g (V_g + span g S_g) over the set's workgroups (span = 256 x buckets per lane).  Segments of 16 workgroups are folded side by side on the
        // worker threads — (v, r, w) = (sum V_g, sum S_g, sum (g - g0) S_g) by a running sum —, then sum_g g S_g = sum_s w_s + 16 sum_s s r_s
        // is a second running sum over the segments (one 2^19-bucket set: 16 segments of ~50 group operations, then ~40).
        std::vector<drh::G1> vs(2 * wg_count);
        HIP_TRY(hipMemcpyAsync(vs.data(), ctx->partial.p, vs.size() * 192, hipMemcpyDeviceToHost, st));
        if (part_flag) HIP_TRY(hipMemcpyAsync(&part_overflow, part_flag, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        tr_.mark("gpu");
        if (part_overflow) return msm_device(ctx, d_bases_in, d_scalars, n, batch, results, tbl, true);
        constexpr size_t SEG = 16;
        const size_t segs_per_set = (wg_per_set + SEG - 1) / SEG, nseg = bsets * segs_per_set;
        std::vector<drh::G1> seg_v(nseg), seg_r(nseg), seg_w(nseg);
        const std::function<void(size_t)> one_seg = [&](size_t t) {
            const size_t set = t / segs_per_set, g0 = (t % segs_per_set) * SEG, g1 = std::min(wg_per_set, g0 + SEG);
            drh::G1* p = vs.data() + 2 * (set * wg_per_set + g0);
            g1_dev_to_host(p, 2 * (g1 - g0));
            drh::G1 run = drh::G1::inf(), w = drh::G1::inf(), v = drh::G1::inf();
            for (size_t g = g1 - g0; g-- > 0;) {
                v = drh::g1_add(v, p[2 * g]);
                run = drh::g1_add(run, p[2 * g + 1]);
                if (g >= 1) w = drh::g1_add(w, run);                                               // w = sum_g (g - g0) S_g
            }
            seg_v[t] = v; seg_r[t] = run; seg_w[t] = w;
        };
        // a task is ~50 group operations (~40 us): one per worker thread (parallel_for would keep so few items on one thread)
        if (drh::WorkerPool* pool = drh::worker_pool()) pool->run(nseg, (unsigned)std::min<size_t>(nseg, drh::host_threads()), one_seg);
        else for (size_t t = 0; t < nseg; t++) one_seg(t);
        std::vector<drh::G1> set_sum(bsets);
        const std::function<void(size_t)> one_set = [&](size_t set) {
            drh::G1 v = drh::G1::inf(), w = drh::G1::inf(), run = drh::G1::inf(), sr = drh::G1::inf();
            for (size_t sg = segs_per_set; sg-- > 0;) {
                const size_t t = set * segs_per_set + sg;
                v = drh::g1_add(v, seg_v[t]);
                w = drh::g1_add(w, seg_w[t]);
                if (sg >= 1) { run = drh::g1_add(run, seg_r[t]); sr = drh::g1_add(sr, run); }       // sr = sum_s s r_s
            }
            for (int k = 0; k < 4; k++) sr = drh::g1_dbl(sr);                                       // x 16
            w = drh::g1_add(w, sr);
            for (uint32_t k = 1; k < ws_span; k <<= 1) w = drh::g1_dbl(w);                          // x span
            set_sum[set] = drh::g1_add(v, w);
        };
        if (drh::WorkerPool* pool = bsets > 1 ? drh::worker_pool() : nullptr) pool->run(bsets, (unsigned)std::min<size_t>(bsets, drh::host_threads()), one_set);
        else for (size_t set = 0; set < bsets; set++) one_set(set);
        drh::G1 acc = drh::G1::inf();
        if (single && batch > 1) {
            // a few MSMs over the table: each is the sum of its index groups' sets; the batched contract wants them in ctx->result
            for (size_t b = 0; b < batch; b++) {
                drh::G1 v = drh::G1::inf();
                for (uint32_t g = 0; g < groups; g++) v = drh::g1_add(v, set_sum[b * groups + g]);
                results[b] = v;
            }
            TRY(ctx->result.reserve(batch * 192));
            std::vector<drh::G1> up(results);
            g1_host_to_dev(up.data(), up.size());
            HIP_TRY(hipMemcpyAsync(ctx->result.p, up.data(), batch * 192, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            tr_.mark("host_fold");
            if (ctx->prof) TRY(prof_collect(ctx));
            return DR_OK;
        }
        if (single) {
            for (size_t set = 0; set < bsets; set++) acc = drh::g1_add(acc, set_sum[set]);      // the bucket-set sums ARE the MSM value
        } else {
            // plain bases: one set per window; Horner over the windows (255 doublings: ~50x faster on one CPU core than on one GPU lane)
            acc = set_sum[pl.W - 1];
            for (int w = pl.W - 2; w >= 0; w--) {
                for (int j = 0; j < pl.wt.width[w]; j++) acc = drh::g1_dbl(acc);
                acc = drh::g1_add(acc, set_sum[w]);
            }
        }
        results[0] = acc;
        tr_.mark("host_fold");
    } else if (single) {
        // the bucket-set sum IS the MSM value: no window combination
        if (groups > 1 || batch == 1) {
            // few MSMs: fetch the per-group sums and add them on the host (<= 64 additions per MSM)
            std::vector<drh::G1> parts(bsets);
            HIP_TRY(hipMemcpyAsync(parts.data(), ctx->winsum.p, bsets * 192, hipMemcpyDeviceToHost, st));
            if (part_flag) HIP_TRY(hipMemcpyAsync(&part_overflow, part_flag, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (part_overflow) return msm_device(ctx, d_bases_in, d_scalars, n, batch, results, tbl, true);
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
            hipLaunchKernelGGL(dr::k_g1_horner, dim3(div_up(batch, 64)), dim3(64), 0, ctx->stream, ctx->winsum.as<uint32_t>(),
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
        t.pt_words = srs->table_pt_words;
        t.bit_rows = srs->table_bit_rows;
        t.naf_delta = srs->table_naf_delta;
        t.stride = (uint32_t)srs->count;
        t.offset = (uint32_t)offset;
    }
    return t;
}

// MSM(s) with results written as BE affine records
int msm_to_bytes(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf,
                 const MsmTable* tbl) {
    std::vector<drh::G1> res;
    TRY(msm_device(ctx, d_bases, d_scalars, n, batch, res, tbl));
    if (batch == 1 || n == 0) {
        for (size_t b = 0; b < batch; b++) g1_result_to_bytes(res[b], out_be_xy + 96 * b, is_inf ? is_inf + b : nullptr);
        return DR_OK;
    }
    return msm_batch_results_to_bytes(ctx, batch, out_be_xy, is_inf);
}

// batch > 1: results were left in ctx->result (XYZZ).  The affine conversion is one 381-bit field inversion per
// result: on the GPU a division-step chain (divstep28.hip.h: ≈ 0.08 ms of pure latency per call, whatever the batch; 0.5 ms with
// the binary Euclid, 0.85 ms with the Fermat power before that).  (Downloading XYZZ and inverting on the worker threads was measured for whole
// batches: less GPU time but more wall time per 1024 proofs.)
int msm_batch_results_to_bytes(dr_ctx* ctx, size_t batch, uint8_t* out_be_xy, int* is_inf) {
    // a handful of results (a single proof's 4 witness commitments, 2 openings): the kernel's one inversion chain is 0.6 ms of
    // latency whatever the count, the host inverts in ~15 us each
    if (batch <= 16) {
        static_assert(sizeof(drh::G1) == 192, "XYZZ layout");
        std::vector<drh::G1> res(batch);
        HIP_TRY(hipMemcpyAsync(res.data(), ctx->result.p, batch * 192, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
        g1_dev_to_host(res.data(), res.size());
        drh::parallel_for(batch, [&](size_t b) { g1_result_to_bytes(res[b], out_be_xy + 96 * b, is_inf ? is_inf + b : nullptr); }, 1);    // an inversion each
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


// ------------------------------------------------------------------------------- K4: twisted Edwards Pippenger
// One variable-base MSM on Bandersnatch / JubJub by the bucket method (kernels_te_msm.hip.h).  Bucket sets are
// (window, index group) pairs so that a few thousand terms still make tens of thousands of bucket lanes; digits, counting
// sort and size ordering are the G1 pipeline's kernels.  The W x G set sums come back to the host for the final combination.
namespace {
struct TeHost {                 // extended coordinates over the host field (same Montgomery form as the device's Fr)
    drh::Fr x, y, z, t;
};
TeHost te_host_identity() { return {drh::Fr::zero(), drh::Fr::one(), drh::Fr::one(), drh::Fr::zero()}; }
TeHost te_host_add(const TeHost& p, const TeHost& q, const drh::Fr& d, const drh::Fr& neg_a) {      // add-2008-hwcd, unified
    drh::Fr A = p.x * q.x, B = p.y * q.y, C = p.t * d * q.t, D = p.z * q.z;
    drh::Fr E = (p.x + p.y) * (q.x + q.y) - A - B, F = D - C, G = D + C, H = B + A * neg_a;
    return {E * F, G * H, F * G, E * H};
}
}  // namespace

int te_msm_pippenger(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    TRY(use_ctx(ctx));
    const drh::TeCurveHost* cu = drh::te_curve(cv);
    if (!cu) return fail(DR_ERR_INVALID, "unknown curve id");
    if (n == 0 || n >= (1ull << 26)) return fail(DR_ERR_INVALID, "bad MSM size");
    // scalars mod n (253 / 252 bits: the signed recoding over 256-bit windows never carries out of the top)
    std::vector<uint8_t> ks(n * 32);
    auto red = [&](size_t i) {
        uint64_t k[4];
        cu->n.reduce_bytes(scalars + 32 * i, 32, false, k);
        drh::store_le32(k, ks.data() + 32 * i);
    };
    if (n >= 4096) drh::parallel_for(n, red);
    else for (size_t i = 0; i < n; i++) red(i);
    // window width by size (the reference's rule grows the same way, bandersnatch.py:23-36); index groups until a bucket
    // holds ~8 points or 64 groups
    const int c = n < 4096 ? 7 : n < 16384 ? 8 : n < 65536 ? 9 : 10;
    // tile scalar_bits + 1 bits, not 256: a top window holding one or two live bits would put half of all points into one bucket
    const dr::WindowTable wt = make_window_table(c, (int)cu->scalar_bits + 1);
    const uint32_t H = 1u << (wt.cmax - 1), L = 8, T = H / L;
    uint32_t groups = 1;
    while (groups < 64 && n / ((size_t)groups * 2 * H) >= 8) groups *= 2;
    const size_t sets = (size_t)wt.W * groups, nbuckets = sets * H;
    const size_t per_set = (n + groups - 1) / groups;
    hipStream_t st = ctx->stream;
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 96));
    TRY(ctx->scalars.reserve(n * 32));
    TRY(ctx->counts.reserve(nbuckets * 4));
    TRY(ctx->offsets.reserve((nbuckets + 1) * 4));
    TRY(ctx->sorted.reserve(sets * per_set * 4));
    const unsigned szblocks = div_up(nbuckets, dr::SZ_TILE);
    const size_t ncells = (size_t)dr::SZ_CLASSES * szblocks;
    TRY(ctx->tiles.reserve((size_t)(div_up(ncells, dr::SCAN_TILE) + 1) * 4));
    TRY(ctx->perm.reserve(nbuckets * 4));
    TRY(ctx->cells.reserve(ncells * 4));
    TRY(ctx->cell_off.reserve(ncells * 4));
    TRY(ctx->buckets.reserve(nbuckets * 128));
    TRY(ctx->partial.reserve(sets * T * 128));
    TRY(ctx->winsum.reserve(sets * 128));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->scalars.p, ks.data(), n * 32, hipMemcpyHostToDevice, st));
    dr::SortSetParams sp{};
    sp.n = (uint32_t)n; sp.batch = 1; sp.H = H; sp.groups = groups; sp.single = 0;
    sp.capacity = (uint32_t)per_set; sp.short_from = 0xffffffffu; sp.n_short = 0;
    TRY(launch(ctx, "k_te_msm_prepare", [&] {
        LAUNCH_CV(cv, dr::k_te_msm_prepare, dim3(div_up(n, 256)), dim3(256), 0, st, ctx->io_a.as<uint32_t>(), (uint32_t)n, ctx->io_b.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_g1_sort_sets", [&] {
        hipLaunchKernelGGL(dr::k_g1_sort_sets, dim3((unsigned)sets), dim3(dr::SORT_BLOCK), 0, st, ctx->scalars.as<uint32_t>(), wt, sp,
                           ctx->counts.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->sorted.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_size_sort", [&] {
        hipLaunchKernelGGL(dr::k_size_hist, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cells.as<uint32_t>());
        const unsigned nt = div_up(ncells, dr::SCAN_TILE);
        hipLaunchKernelGGL(dr::k_scan_tiles, dim3(nt), dim3(dr::SCAN_BLOCK), 0, st, ctx->cells.as<uint32_t>(), ctx->cell_off.as<uint32_t>(),
                           ctx->tiles.as<uint32_t>(), ncells);
        hipLaunchKernelGGL(dr::k_scan_tile_sums, dim3(1), dim3(dr::SCAN_BLOCK), 0, st, ctx->tiles.as<uint32_t>(), nt, ctx->tiles.as<uint32_t>() + nt);
        hipLaunchKernelGGL(dr::k_scan_add, dim3(div_up(ncells, 256)), dim3(256), 0, st, ctx->cell_off.as<uint32_t>(), ctx->tiles.as<uint32_t>(), ncells);
        hipLaunchKernelGGL(dr::k_size_place, dim3(szblocks), dim3(dr::SZ_BLOCK), 0, st, ctx->counts.as<uint32_t>(), nbuckets, szblocks,
                           ctx->cell_off.as<uint32_t>(), ctx->perm.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_te_msm_accumulate", [&] {
        LAUNCH_CV(cv, dr::k_te_msm_accumulate, dim3(div_up(nbuckets, 256)), dim3(256), 0, st, ctx->io_b.as<uint32_t>(), ctx->sorted.as<uint32_t>(),
                  ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->perm.as<uint32_t>(), ctx->buckets.as<uint32_t>(), nbuckets);
        LAUNCH_CV(cv, dr::k_te_msm_accumulate_heavy, dim3((unsigned)nbuckets), dim3(64), 0, st, ctx->io_b.as<uint32_t>(), ctx->sorted.as<uint32_t>(),
                  ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(), ctx->buckets.as<uint32_t>(), nbuckets);
    }));
    TRY(launch(ctx, "k_te_msm_reduce", [&] {
        LAUNCH_CV(cv, dr::k_te_msm_reduce, dim3(div_up(sets * T, 128)), dim3(128), 0, st, ctx->buckets.as<uint32_t>(), sets, H, L,
                  ctx->partial.as<uint32_t>());
        LAUNCH_CV(cv, dr::k_te_msm_fold, dim3(div_up(sets, 64)), dim3(64), 0, st, ctx->partial.as<uint32_t>(), sets, T, ctx->winsum.as<uint32_t>());
    }));
    static_assert(sizeof(TeHost) == 128, "extended point layout");
    std::vector<TeHost> sums(sets);
    HIP_TRY(hipMemcpyAsync(sums.data(), ctx->winsum.p, sets * 128, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->prof) TRY(prof_collect(ctx));
    // device layout is X, Y, Z, T; TeHost is x, y, z, t in that order
    uint8_t D_LE[32];
    drh::store_le32(cu->d, D_LE);
    drh::Fr d, neg_a = drh::Fr::from_u64(cu->neg_a[0]);
    if (!drh::Fr::load_le(d, D_LE)) return fail(DR_ERR_DEVICE, "bad curve constant");
    TeHost acc = te_host_identity();
    for (int w = wt.W - 1; w >= 0; w--) {
        if (w != wt.W - 1)
            for (int j = 0; j < wt.width[w]; j++) acc = te_host_add(acc, acc, d, neg_a);
        for (uint32_t g = 0; g < groups; g++) acc = te_host_add(acc, sums[(size_t)w * groups + g], d, neg_a);
    }
    drh::Fr zi = acc.z.inv();
    (acc.x * zi).store_le(out_xy);
    (acc.y * zi).store_le(out_xy + 32);
    return DR_OK;
}

void g1_launch_decompress(hipStream_t st, const uint8_t* d_enc, uint32_t* d_bases, uint32_t* d_ok, size_t n) {
    hipLaunchKernelGGL(dr::k_g1_decompress, dim3(div_up(n, 64)), dim3(64), 0, st, d_enc, d_bases, d_ok, (uint32_t)n);
}
void g1_launch_bases_to_mont(hipStream_t st, uint32_t* d_bases, size_t n) {
    hipLaunchKernelGGL(dr::k_g1_bases_to_mont, dim3(div_up(n, 256)), dim3(256), 0, st, d_bases, (uint32_t)n);
}
void g1_launch_bases_from_mont(hipStream_t st, const uint32_t* d_bases, uint32_t* d_out, size_t n) {
    hipLaunchKernelGGL(dr::k_g1_bases_from_mont, dim3(div_up(n, 256)), dim3(256), 0, st, d_bases, d_out, (uint32_t)n);
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

int dr_srs_precompute(dr_ctx* ctx, dr_srs* srs, int window_bits) { return srs_precompute(ctx, srs, window_bits, true); }

// bit_rows = false: window rows only (the prover's summation-by-parts bases: a few hundred scalars per vector, most of them +-1 — short,
// uneven lists that gain nothing from a window less and measured 0.7 ms per batch slower with odd-multiple buckets)
int srs_precompute(dr_ctx* ctx, dr_srs* srs, int window_bits, bool allow_bit_rows) {
    TRY(use_ctx(ctx));
    if (!srs) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
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
    srs->table_bit_rows = false;
    // A small SRS gets a row for every bit (32 KB per base: 201 MB for the 6145 points of a 2048-point domain) — the window rows are a
    // subset of it, and batched MSMs may then recode the scalars in non-adjacent form (tiling_for).
    // DOTRING_SRS_BIT_ROWS_MB (default 512, 0 = never) bounds the table; the width of that form is chosen per call (tiling_for).
    static const size_t bit_rows_mb = std::getenv("DOTRING_SRS_BIT_ROWS_MB") ? (size_t)std::atol(std::getenv("DOTRING_SRS_BIT_ROWS_MB")) : 512;
    // One table point per 128-byte line (24 of 32 words used): a packed 96-byte record straddles two lines five times out of eight, and
    // the bucket walk — one random table point per addition — measured 1.1 % faster with a third fewer lines to fetch although the table is
    // a third larger (A/B on one box, three alternations: 40.50 - 40.63 against 40.85 - 41.17 ms per 1024 proofs).
    const uint32_t pt_words = 32;
    bool bit_rows = allow_bit_rows && window_bits <= 16 && (size_t)256 * srs->count * 4 * pt_words <= (bit_rows_mb << 20);
    if (bit_rows && hipMalloc((void**)&srs->d_table, (size_t)256 * srs->count * 4 * pt_words) != hipSuccess) {
        (void)hipGetLastError();                   // a device short of memory keeps the window rows (W rows instead of 256)
        srs->d_table = nullptr;
        bit_rows = false;
    }
    if (!bit_rows) HIP_TRY(hipMalloc((void**)&srs->d_table, (size_t)wt.W * srs->count * 4 * pt_words));
    srs->table_pt_words = pt_words;
    if (bit_rows) {
        for (int w = 0; w < wt.W; w++) wt.row[w] = wt.start[w];
        hipLaunchKernelGGL(dr::k_g1_bit_table, dim3(div_up(srs->count, 128)), dim3(128), 0, ctx->stream, srs->d_bases, (uint32_t)srs->count, 256u,
                           pt_words, srs->d_table);
        srs->table_bit_rows = true;
    } else {
        hipLaunchKernelGGL(dr::k_g1_window_table, dim3(div_up(srs->count, 128)), dim3(128), 0, ctx->stream, srs->d_bases, (uint32_t)srs->count, wt,
                           pt_words, srs->d_table);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(srs->d_table);
        srs->d_table = nullptr;
        return fail(DR_ERR_DEVICE, std::string("window table: ") + hipGetErrorString(e));
    }
    srs->table_wt = wt;
    return DR_OK;
}

void dr_srs_destroy(dr_srs* srs) {
    if (!srs) return;
    (void)hipSetDevice(srs->device);
    for (auto& it : srs->lagrange_prefix) dr_srs_destroy(it.second);
    srs->lagrange_prefix.clear();
    if (srs->d_table) (void)hipFree(srs->d_table);
    if (srs->d_bases) (void)hipFree(srs->d_bases);
    delete srs;
}

size_t dr_srs_size(const dr_srs* srs) { return srs ? srs->count : 0; }

int dr_srs_table_info(const dr_srs* srs, size_t n, size_t batch, int info[6]) {
    if (!srs || !info) return fail(DR_ERR_INVALID, "null argument");
    for (int i = 0; i < 6; i++) info[i] = 0;
    if (!srs->d_table) return DR_OK;
    const MsmTable t = srs_table(srs, 0);
    const Tiling tl = tiling_for(t, n, batch);
    info[0] = srs->table_wt.cmax;
    info[1] = srs->table_bit_rows ? 256 : srs->table_wt.W;
    info[2] = tl.mode ? tl.slots : srs->table_wt.W;
    info[3] = tl.c;
    info[4] = tl.mode;
    info[5] = (int)std::lround(1000.0 * (tl.mode ? tl.digits : (double)srs->table_wt.W));
    return DR_OK;
}

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
// prepared G2 points by their 192-byte encoding (a verifier key has two; a handful of SRS files per process)
// (shared ownership: verifiers run their Miller loops concurrently — two per verify, several verifying threads — and an entry evicted by one
//  thread must outlive the loop of another that is still reading it)
std::shared_ptr<const drh::G2Prepared> g2_prepared_cached(const uint8_t enc[192], const drh::G2Affine& q) {
    static std::mutex m;
    static std::vector<std::pair<std::array<uint8_t, 192>, std::shared_ptr<const drh::G2Prepared>>> cache;
    std::array<uint8_t, 192> key;
    std::memcpy(key.data(), enc, 192);
    std::lock_guard<std::mutex> lock(m);
    for (auto& e : cache) if (e.first == key) return e.second;
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.emplace_back(key, std::make_shared<const drh::G2Prepared>(drh::g2_prepare(q)));
    return cache.back().second;
}
int miller_product(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, drh::Fq12& f, bool prepared = true) {
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(g1_be_xy, n, le, true));
    std::vector<drh::Fq> px, py;
    std::vector<drh::G2Affine> qs;
    std::vector<const drh::G2Prepared*> preps;
    std::vector<std::shared_ptr<const drh::G2Prepared>> held;           // keeps the cached entries alive for the duration of the loop
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
        if (prepared) {
            held.push_back(g2_prepared_cached(q, Q));
            preps.push_back(held.back().get());
        }
    }
    f = prepared ? drh::multi_miller_loop_prepared(px.data(), py.data(), preps.data(), preps.size())
                 : drh::multi_miller_loop(px.data(), py.data(), qs.data(), qs.size());
    return DR_OK;
}
}  // namespace

int dri::pairing_miller(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, drh::Fq12& f) { return miller_product(g1_be_xy, g2_be, n, f); }
bool dri::pairing_product_is_one(const drh::Fq12& f) { return drh::final_exponentiation_check(f) == drh::Fq12::one(); }

int dr_pairing_check(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, int* ok) {
    if (!ok || (n && (!g1_be_xy || !g2_be))) return fail(DR_ERR_INVALID, "null buffer");
    drh::Fq12 f;
    TRY(miller_product(g1_be_xy, g2_be, n, f));
    *ok = drh::final_exponentiation_check(f) == drh::Fq12::one() ? 1 : 0;
    return DR_OK;
}

// diagnostic: the check's Miller loop (prepared G2 points, sparse line products) against the plain affine one, and its
// final exponentiation (Frobenius maps + x-chain with cyclotomic squarings, exponent 3(p^12-1)/r) against the plain
// square-and-multiply one; *consistent = 1 iff the loops agree and fast == reference^3 for this product
int dr_pairing_selfcheck(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, int* consistent) {
    if (!consistent || (n && (!g1_be_xy || !g2_be))) return fail(DR_ERR_INVALID, "null buffer");
    drh::Fq12 f, f_plain;
    TRY(miller_product(g1_be_xy, g2_be, n, f));
    TRY(miller_product(g1_be_xy, g2_be, n, f_plain, false));            // affine Miller loop, slopes computed on the fly
    drh::Fq12 ref = drh::final_exponentiation(f_plain);
    *consistent = (f == f_plain && drh::final_exponentiation_check(f) == ref * ref * ref) ? 1 : 0;
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
