// libdotring_hip.so — C ABI implementation (include/dotring_hip.h): host orchestration of the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dotring_hip.h"
#include "hostmath.hpp"
#include "hostpairing.hpp"
#include "kernels_bsn.cuh"
#include "kernels_g1.cuh"
#include "kernels_ntt.cuh"

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
    Scratch scalars, digits, counts, offsets, cursor, tiles, sorted, buckets, partial, winsum, result, io_a, io_b, io_c;
    dr::TwiddleCache twiddles;
};

struct dr_srs {
    int device = 0;
    size_t count = 0;
    uint32_t* d_bases = nullptr;   // G1Affine[count], Montgomery
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

// ---- window plan for the GPU Pippenger.  Scalars are reduced mod r (< 2^255) on the device and the 256 bits
// are tiled by W = ceil(256/c) windows of width cmax or cmax-1 (see WindowTable).  Work ~ W*n mixed adds +
// W*2^(c-1)*(2 full adds) + per-chunk scalar multiplications; a full add costs ~1.4 mixed adds; pick the c
// minimising that, within [7,16] (W <= 37 fits the table).
bool window_ok(int c) { return c >= 7 && c <= 16; }
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

struct MsmPlan {
    dr::WindowTable wt;
    int W;
    uint32_t H, L, T;
};
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
    p.L = std::min<uint32_t>(p.H, 16);
    p.T = p.H / p.L;
    return p;
}

int g_force_c = 0;   // test hook: DOTRING_MSM_WINDOW

// core: bases/scalars on the device; writes batch results (XYZZ, Montgomery) into host vector
int msm_device(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch,
               std::vector<drh::G1>& results) {
    results.assign(batch, drh::G1::inf());
    if (n == 0 || batch == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "MSM size must be below 2^31");
    const MsmPlan pl = make_plan(n, g_force_c);
    const size_t windows = batch * (size_t)pl.W;
    const size_t nbuckets = windows * pl.H;
    const size_t ndigits = windows * n;
    if (nbuckets >= (1ull << 32) || ndigits >= (1ull << 32))
        return fail(DR_ERR_INVALID, "MSM batch too large for one launch (split the batch)");
    TRY(ctx->digits.reserve(ndigits * 4));
    TRY(ctx->counts.reserve(nbuckets * 4));
    TRY(ctx->offsets.reserve((nbuckets + 1) * 4));
    TRY(ctx->cursor.reserve(nbuckets * 4));
    const unsigned ntiles = div_up(nbuckets, dr::SCAN_TILE);
    TRY(ctx->tiles.reserve((size_t)(ntiles + 1) * 4));
    TRY(ctx->sorted.reserve(ndigits * 4));
    TRY(ctx->buckets.reserve(nbuckets * 192));
    TRY(ctx->partial.reserve(windows * pl.T * 192));
    TRY(ctx->winsum.reserve(windows * 192));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(ctx->counts.p, 0, nbuckets * 4, st));
    HIP_TRY(hipMemsetAsync(ctx->cursor.p, 0, nbuckets * 4, st));

    TRY(launch(ctx, "k_g1_digits", [&] {
        hipLaunchKernelGGL(dr::k_g1_digits, dim3(div_up(n * batch, 256)), dim3(256), 0, st, d_scalars, (uint32_t)n,
                           (uint32_t)batch, pl.wt, ctx->digits.as<int32_t>(), ctx->counts.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_scan", [&] {
        hipLaunchKernelGGL(dr::k_scan_tiles, dim3(ntiles), dim3(dr::SCAN_BLOCK), 0, st, ctx->counts.as<uint32_t>(),
                           ctx->offsets.as<uint32_t>(), ctx->tiles.as<uint32_t>(), nbuckets);
        hipLaunchKernelGGL(dr::k_scan_tile_sums, dim3(1), dim3(dr::SCAN_BLOCK), 0, st, ctx->tiles.as<uint32_t>(), ntiles,
                           ctx->tiles.as<uint32_t>() + ntiles);
        hipLaunchKernelGGL(dr::k_scan_add, dim3(div_up(nbuckets, 256)), dim3(256), 0, st, ctx->offsets.as<uint32_t>(),
                           ctx->tiles.as<uint32_t>(), nbuckets);
    }));
    TRY(launch(ctx, "k_g1_scatter", [&] {
        hipLaunchKernelGGL(dr::k_g1_scatter, dim3(div_up(ndigits, 256)), dim3(256), 0, st, ctx->digits.as<int32_t>(),
                           (uint32_t)n, windows, pl.H, ctx->offsets.as<uint32_t>(), ctx->cursor.as<uint32_t>(),
                           ctx->sorted.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_g1_accumulate", [&] {
        hipLaunchKernelGGL(dr::k_g1_accumulate, dim3(div_up(nbuckets, 256)), dim3(256), 0, st, d_bases,
                           ctx->sorted.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->counts.as<uint32_t>(),
                           ctx->buckets.as<uint32_t>(), nbuckets);
    }));
    TRY(launch(ctx, "k_g1_reduce_chunks", [&] {
        hipLaunchKernelGGL(dr::k_g1_reduce_chunks, dim3(div_up(windows * pl.T, 128)), dim3(128), 0, st,
                           ctx->buckets.as<uint32_t>(), windows, pl.H, pl.L, ctx->partial.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_g1_reduce_windows", [&] {
        hipLaunchKernelGGL(dr::k_g1_reduce_windows, dim3((unsigned)windows), dim3(dr::RW_BLOCK), 0, st,
                           ctx->partial.as<uint32_t>(), pl.T, ctx->winsum.as<uint32_t>());
    }));

    static_assert(sizeof(drh::G1) == 192, "XYZZ layout");
    if (batch == 1) {
        // window combination on the host: a 255-doubling serial chain is ~50x faster on one CPU core
        std::vector<drh::G1> ws(pl.W);
        HIP_TRY(hipMemcpyAsync(ws.data(), ctx->winsum.p, (size_t)pl.W * 192, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
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
        HIP_TRY(hipMemcpyAsync(results.data(), ctx->result.p, batch * 192, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (ctx->prof) TRY(prof_collect(ctx));
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
    *out = ctx;
    return DR_OK;
}

void dr_ctx_destroy(dr_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (Scratch* s : {&ctx->scalars, &ctx->digits, &ctx->counts, &ctx->offsets, &ctx->cursor, &ctx->tiles, &ctx->sorted,
                       &ctx->buckets, &ctx->partial, &ctx->winsum, &ctx->result, &ctx->io_a, &ctx->io_b, &ctx->io_c})
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

int dr_prof_enable(dr_ctx* ctx, int on) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    ctx->prof = on != 0;
    return DR_OK;
}
int dr_prof_reset(dr_ctx* ctx) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    ctx->prof_data.clear();
    return DR_OK;
}
int dr_prof_get(dr_ctx* ctx, const char* name, double* total_ms, int* launches) {
    if (!ctx || !name) return fail(DR_ERR_INVALID, "null argument");
    auto it = ctx->prof_data.find(name);
    if (total_ms) *total_ms = it == ctx->prof_data.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof_data.end() ? 0 : it->second.launches;
    return DR_OK;
}

// ------------------------------------------------------------------------------- seam A
int dr_bsn_scalar_mul_batch_dev(dr_ctx* ctx, const void* d_pts, const void* d_scalars, size_t n, void* d_out) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
        hipLaunchKernelGGL(dr::k_bsn_scalar_mul, dim3(div_up(n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                           (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t*)d_out, (uint32_t)n);
    }));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

static int check_fr_elems(const uint8_t* p, size_t count, const char* what) {
    for (size_t i = 0; i < count; i++) {
        drh::Fr t;
        if (!drh::Fr::load_le(t, p + 32 * i)) return fail(DR_ERR_INVALID, std::string(what) + " coordinate is not a canonical field element");
    }
    return DR_OK;
}

int dr_bsn_scalar_mul_batch(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 32));
    TRY(ctx->io_c.reserve(n * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    TRY(dr_bsn_scalar_mul_batch_dev(ctx, ctx->io_a.p, ctx->io_b.p, n, ctx->io_c.p));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DR_OK;
}

int dr_bsn_msm_groups(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    if (groups == 0) return DR_OK;
    if (m == 0 || m > 64) return fail(DR_ERR_INVALID, "group size must be in 1..64");
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    size_t n = groups * m;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    uint32_t mpad = 1;
    while (mpad < m) mpad <<= 1;
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 32));
    TRY(ctx->io_c.reserve(groups * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t per_block = dr::BSN_BLOCK / mpad;
    TRY(launch(ctx, "k_bsn_msm_groups", [&] {
        hipLaunchKernelGGL(dr::k_bsn_msm_groups, dim3(div_up(groups, per_block)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                           ctx->io_a.as<uint32_t>(), ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>(), (uint32_t)groups,
                           (uint32_t)m, mpad);
    }));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, groups * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}

int dr_bsn_msm(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    TRY(use_ctx(ctx));
    if (!out_xy) return fail(DR_ERR_INVALID, "null buffer");
    if (n == 0) {
        std::memset(out_xy, 0, 64);
        out_xy[32] = 1;
        return DR_OK;
    }
    // fold 64 terms at a time on the device; the (few) partial sums are then combined the same way with scalar 1
    std::vector<uint8_t> pts(pts_xy, pts_xy + n * 64), ks(scalars, scalars + n * 32);
    while (true) {
        size_t cur = pts.size() / 64;
        if (cur <= 64) return dr_bsn_msm_groups(ctx, pts.data(), ks.data(), 1, cur, out_xy);
        size_t full = cur / 64, rem = cur % 64;
        std::vector<uint8_t> next((full + (rem ? 1 : 0)) * 64);
        TRY(dr_bsn_msm_groups(ctx, pts.data(), ks.data(), full, 64, next.data()));
        if (rem) TRY(dr_bsn_msm_groups(ctx, pts.data() + full * 64 * 64, ks.data() + full * 64 * 32, 1, rem, next.data() + full * 64));
        pts.swap(next);
        ks.assign(pts.size() / 2, 0);
        for (size_t i = 0; i < pts.size() / 64; i++) ks[32 * i] = 1;
    }
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

void dr_srs_destroy(dr_srs* srs) {
    if (!srs) return;
    (void)hipSetDevice(srs->device);
    if (srs->d_bases) (void)hipFree(srs->d_bases);
    delete srs;
}

size_t dr_srs_size(const dr_srs* srs) { return srs ? srs->count : 0; }

int dr_g1_msm_batch_dev(dr_ctx* ctx, const dr_srs* srs, const void* d_scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf) {
    TRY(use_ctx(ctx));
    if (!srs || !out_be_xy) return fail(DR_ERR_INVALID, "null argument");
    if (srs->device != ctx->device) return fail(DR_ERR_INVALID, "SRS lives on another device");
    if (n > srs->count) return fail(DR_ERR_INVALID, "polynomial degree exceeds SRS size");
    std::vector<drh::G1> res;
    TRY(msm_device(ctx, srs->d_bases, (const uint32_t*)d_scalars, n, batch, res));
    for (size_t b = 0; b < batch; b++) g1_result_to_bytes(res[b], out_be_xy + 96 * b, is_inf ? is_inf + b : nullptr);
    return DR_OK;
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
    TRY(msm_device(ctx, srs->d_bases + offset * 24, (const uint32_t*)d_scalars, n, 1, res));
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

int dr_pairing_check(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, int* ok) {
    if (!ok || (n && (!g1_be_xy || !g2_be))) return fail(DR_ERR_INVALID, "null buffer");
    std::vector<uint8_t> le;
    TRY(g1_be_to_le_limbs(g1_be_xy, n, le, true));
    drh::Fq12 f = drh::Fq12::one();
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
        drh::Fq px, py;
        drh::Fq::load_le(px, le.data() + 96 * i);
        drh::Fq::load_le(py, le.data() + 96 * i + 48);
        f = f * drh::miller_loop(px, py, Q);
    }
    *ok = drh::final_exponentiation(f) == drh::Fq12::one() ? 1 : 0;
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
