// libdotring_hip.so — C ABI (include/dotring_hip.h), part 1 of 5: contexts, device memory, profiling, seam A (the
// Bandersnatch / twisted Edwards kernels of kernels_bsn.hip.h) and hash-to-curve.  See capi_internal.hpp for the layout.
#include "capi_internal.hpp"
#include "hostsmall.hpp"
#include "kernels_bsn.hip.h"

namespace dri {

thread_local std::string g_err;

int use_ctx(dr_ctx* ctx) {
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return ctx_join_wipe(ctx);
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

int g_force_c = 0;
bool window_ok(int c) { return c >= 7 && c <= 16; }

}  // namespace dri
using namespace dri;

// live contexts (a prover unregisters its helper stream from its context only if that context still exists)
static std::mutex g_live_mutex;
static std::set<dr_ctx*> g_live_ctx;
bool ctx_alive(dr_ctx* c) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return g_live_ctx.count(c) != 0;
}

// =================================================================================== C ABI
const char* dr_version(void) { return "dotring_hip 0.1 (gfx950)"; }
const char* dr_last_error(void) { return g_err.c_str(); }

int dr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
    auto h_owner = std::make_unique<dr::BsnConsts>();        // 200 KB: not on the stack
    dr::BsnConsts& h = *h_owner;
    // the device's twisted Edwards kernels keep Fr in Montgomery form with R = 2^261 (fr29.hip.h), the host with R = 2^256:
    // (32 v) in the host's form has the words of v in the device's
    const Fr thirty_two = Fr::from_u64(32);
    auto put = [&](uint32_t (&w)[8], const Fr& v) { const Fr t = v * thirty_two; std::memcpy(w, t.l, 32); };
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
    static const uint64_t QP1H[4] = {0x7fff2dff80000000ULL, 0x04d0ec02a9ded201ULL, 0x94cebea4199cec04ULL, 0x0000000039f6d3a9ULL};   // (Q + 1) / 2
    put(h.z_q1h, five.pow(QP1H, 4));
    // windows of the discrete logarithm in <c> (kernels_bsn.hip.h: fr_sqrt_core)
    {
        const Fr c0 = five.pow(Q, 4), c_inv = c0.inv();
        Fr step = c_inv;                                     // c^(-2^(8j))
        Fr half0 = c_inv;                                    // c^(-1): dl_half[0][k] = c^(-k/2) for even k
        for (int j = 0; j < 4; j++) {
            Fr hstep = j == 0 ? half0 : Fr::one();
            if (j > 0) {                                     // c^(-2^(8j - 1))
                hstep = c_inv;
                for (int q = 0; q < 8 * j - 1; q++) hstep = hstep.sqr();
            }
            Fr m = Fr::one(), hh = Fr::one();
            for (int k = 0; k < 256; k++) {
                put(h.dl_mul[j][k], m);
                if (j == 0) {
                    put(h.dl_half[0][k], hh);                // entry k holds c^(-(k >> 1)): read for even k only
                    if (k & 1) hh = hh * hstep;
                } else {
                    put(h.dl_half[j][k], hh);
                    hh = hh * hstep;
                }
                m = m * step;
            }
            for (int q = 0; q < 8; q++) step = step.sqr();
        }
        Fr g3 = c0;
        for (int q = 0; q < 24; q++) g3 = g3.sqr();          // order 2^8
        std::vector<uint32_t> low(256);
        Fr v = Fr::one();
        for (int k = 0; k < 256; k++) {
            uint32_t w[8];
            put(w, v);
            low[k] = w[0];
            v = v * g3;
        }
        if (!(v == Fr::one())) return fail(DR_ERR_DEVICE, "bad discrete-logarithm constants");
        bool found = false;
        for (uint32_t a = 0x9e3779b1u; !found && a > 0x9e3779b1u - 200000u; a -= 2) {
            std::memset(h.dl_map, 0xff, sizeof h.dl_map);
            std::vector<uint8_t> used(65536, 0);
            bool ok = true;
            for (int k = 0; k < 256 && ok; k++) {
                const uint32_t idx = (low[k] * a) >> 16;
                if (used[idx]) ok = false;
                used[idx] = 1;
                h.dl_map[idx] = (uint8_t)k;
            }
            if (ok) { h.dl_hash_mul = a; found = true; }
        }
        if (!found) return fail(DR_ERR_DEVICE, "no perfect hash for the discrete-logarithm table");
    }
    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(dr::g_bsn_consts), &h, sizeof h, 0, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DR_OK;
}
}  // namespace

// ---- wiping (capi_internal.hpp)
namespace {
__global__ void k_count_nonzero(const uint32_t* __restrict__ buf, size_t words, unsigned long long* __restrict__ out) {
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) local += buf[i] != 0;
    for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}
std::vector<Scratch*> ctx_scratch_list(dr_ctx* ctx) {
    return {&ctx->scalars, &ctx->digits, &ctx->counts, &ctx->offsets, &ctx->cursor, &ctx->tiles, &ctx->sorted, &ctx->buckets, &ctx->partial,
            &ctx->winsum, &ctx->result, &ctx->io_a, &ctx->io_b, &ctx->io_c, &ctx->perm, &ctx->cells, &ctx->cell_off, &ctx->part_base, &ctx->heavy};
}
}  // namespace

bool dri::wipe_enabled() {
    static const bool on = std::getenv("DOTRING_WIPE") == nullptr || std::atoi(std::getenv("DOTRING_WIPE")) != 0;
    return on;
}
int dri::ctx_join_wipe(dr_ctx* ctx) {
    if (!ctx->wipe_pending) return DR_OK;
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->wipe_done, 0));
    ctx->wipe_pending = false;
    return DR_OK;
}
int dri::ctx_wipe_begin(dr_ctx* ctx, bool in_stream, hipStream_t* out) {
    TRY(use_ctx(ctx));                       // (an earlier wipe still pending: this stream waits for it first, so the two stay in order)
    if (in_stream || ctx->prof) { *out = ctx->stream; return DR_OK; }
    if (!ctx->wipe_stream) {
        // lowest priority: the fills take the CUs the verifier's latency-bound decoding kernels leave (measured: batch_verify 5.2 ms beside
        // fills of equal priority, 4.4 ms beside these)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&ctx->wipe_stream, hipStreamNonBlocking, least));
        HIP_TRY(hipEventCreateWithFlags(&ctx->wipe_from, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->wipe_done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(ctx->wipe_from, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->wipe_stream, ctx->wipe_from, 0));
    *out = ctx->wipe_stream;
    return DR_OK;
}
int dri::ctx_wipe_end(dr_ctx* ctx, hipStream_t wipe_st) {
    if (wipe_st == ctx->stream) return DR_OK;
    HIP_TRY(hipEventRecord(ctx->wipe_done, wipe_st));
    ctx->wipe_pending = true;
    return DR_OK;
}
int dri::ctx_wipe_enqueue_scratch(dr_ctx* ctx, hipStream_t wst) {
    hipError_t e = hipSuccess;
    size_t total = 0;
    TRY(launch(ctx, "wipe", [&] {
        for (Scratch* s : ctx_scratch_list(ctx))
            if (s->p && s->cap && e == hipSuccess) { e = hipMemsetAsync(s->p, 0, s->cap, wst); total += s->cap; }
    }));
    HIP_TRY(e);
    if (std::getenv("DOTRING_TRACE")) {
        std::fprintf(stderr, "[dotring] wipe of context scratch: %.1f MB |", (double)total / 1e6);
        for (Scratch* s : ctx_scratch_list(ctx)) std::fprintf(stderr, " %.0f", (double)s->cap / 1e6);
        std::fprintf(stderr, "\n");
    }
    return DR_OK;
}
int dri::ctx_wipe_scratch(dr_ctx* ctx, bool in_stream) {
    if (!wipe_enabled()) return DR_OK;
    hipStream_t wst = nullptr;
    TRY(ctx_wipe_begin(ctx, in_stream, &wst));
    TRY(ctx_wipe_enqueue_scratch(ctx, wst));
    return ctx_wipe_end(ctx, wst);
}
int dri::count_nonzero_words(dr_ctx* ctx, const void* d_buf, size_t bytes, uint64_t* total) {
    if (!d_buf || bytes < 4) return DR_OK;
    unsigned long long* d_cnt = nullptr;
    HIP_TRY(hipMalloc((void**)&d_cnt, 8));
    hipError_t e = hipMemsetAsync(d_cnt, 0, 8, ctx->stream);
    unsigned long long host = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_count_nonzero, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t*)d_buf, bytes / 4, d_cnt);
        e = hipMemcpyAsync(&host, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_cnt);
    if (e != hipSuccess) return fail(DR_ERR_DEVICE, std::string("residue count: ") + hipGetErrorString(e));
    *total += host;
    return DR_OK;
}
int dr_ctx_scratch_residue(dr_ctx* ctx, uint64_t* words) {
    if (!ctx || !words) return fail(DR_ERR_INVALID, "null argument");
    *words = 0;
    return ctx_scratch_residue(ctx, words);
}
int dri::ctx_scratch_residue(dr_ctx* ctx, uint64_t* words) {
    TRY(use_ctx(ctx));
    for (Scratch* s : ctx_scratch_list(ctx)) TRY(count_nonzero_words(ctx, s->p, s->cap, words));
    return DR_OK;
}

int dr_ctx_create(int device_id, dr_ctx** out) { return ctx_create_role(device_id, 0, out); }

int dri::ctx_create_role(int device_id, int role, dr_ctx** out) {
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
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(DR_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    const char* fc = std::getenv("DOTRING_MSM_WINDOW");
    g_force_c = fc ? std::atoi(fc) : 0;
    if (!window_ok(g_force_c)) g_force_c = 0;
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
    if (ctx->wipe_stream) {
        (void)hipStreamSynchronize(ctx->wipe_stream);
        (void)hipStreamDestroy(ctx->wipe_stream);
        (void)hipEventDestroy(ctx->wipe_from);
        (void)hipEventDestroy(ctx->wipe_done);
    }
    for (Scratch* s : {&ctx->scalars, &ctx->digits, &ctx->counts, &ctx->offsets, &ctx->cursor, &ctx->tiles, &ctx->sorted,
                       &ctx->buckets, &ctx->partial, &ctx->winsum, &ctx->result, &ctx->io_a, &ctx->io_b, &ctx->io_c, &ctx->perm, &ctx->cells,
                       &ctx->cell_off, &ctx->part_base, &ctx->heavy, &ctx->flag, &ctx->vfy_bases, &ctx->vfy_in, &ctx->vfy_std, &ctx->vfy_te_in,
                       &ctx->vfy_te_out, &ctx->vfy_flags})
        s->release();
    for (auto& it : ctx->prof_pending) {
        (void)hipEventDestroy(it.second.first);
        (void)hipEventDestroy(it.second.second);
    }
    for (auto& e : ctx->twiddles.entries) (void)hipFree(e.d_tw);
    for (auto& fb : ctx->fixed_bases) (void)hipFree(fb.d_table);
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
int te_scalar_mul_batch_dev(dr_ctx* ctx, int cv, const void* d_pts, const void* d_scalars, size_t n, void* d_out) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    // 4-bit windows (64 KiB of LDS per wave, 2 waves per CU) while the launch is latency-bound; 2-bit windows
    // (16 KiB, 10 waves per CU) once there are more waves than the 4-bit kernel can keep resident
    constexpr long w2_from = 32768;
    if (drh::te_curve(cv) && drh::te_curve(cv)->glv && n < 16384) {
        // latency-bound launch: GLV on lane pairs, the scalars reduced and decomposed by the lanes themselves (k_bsn_scalar_mul_glv<true>)
        TRY(ctx->flag.reserve(64));              // (not io_b: internal callers pass the context's io buffers as operands)
        HIP_TRY(hipMemsetAsync(ctx->flag.p, 0, 4, ctx->stream));
        TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
            hipLaunchKernelGGL(dr::k_bsn_scalar_mul_glv<true>, dim3(div_up(2 * n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
                               (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t*)d_out, (uint32_t)n, ctx->flag.as<uint32_t>());
        }));
        uint32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, ctx->flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
        if (bad) return fail(DR_ERR_DEVICE, "GLV decomposition out of range");
        return DR_OK;
    }
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
int check_curve(int cv) { return drh::te_curve(cv) ? DR_OK : fail(DR_ERR_INVALID, "unknown curve id"); }
int dr_bsn_scalar_mul_batch_dev(dr_ctx* ctx, const void* d_pts, const void* d_scalars, size_t n, void* d_out) {
    return te_scalar_mul_batch_dev(ctx, dr::CV_BANDERSNATCH, d_pts, d_scalars, n, d_out);
}

int check_fr_elems(const uint8_t* p, size_t count, const char* what) {
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
int glv_split_scalars(const uint8_t* scalars, size_t n, std::vector<uint32_t>& out) {
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

int te_scalar_mul_batch(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (n == 0) return DR_OK;
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    if (n <= drh::small_host_max()) {
        // a few multiplications (a key pair, a proof's handful): ~0.09 ms each on a host core against a ~1 ms kernel chain.  The scalars may
        // be secret keys: fixed-schedule multiplication (hostsmall.hpp: te_mul_secret), reduced mod n first as the kernels do
        const drh::TeCurveHost* hc = drh::te_curve(cv);
        const drh::TeHostParams hp = drh::te_host_params(*hc);
        drh::parallel_for(n, [&](size_t i) {
            drh::TeExt P;
            uint64_t k[4];
            (void)drh::te_load_affine(pts_xy + 64 * i, P);                    // (canonical: checked above)
            hc->n.reduce_bytes(scalars + 32 * i, 32, false, k);
            drh::TeExt r = drh::te_mul_secret(P, k, hp);
            drh::te_store_affine(r, out_xy + 64 * i);
            explicit_bzero(k, sizeof k);
            explicit_bzero(&r, sizeof r);
        }, 1);
        return DR_OK;
    }
    if (drh::te_curve(cv)->glv && n < 16384) {       // latency-bound launch: halve the chain with GLV on lane pairs
        std::vector<uint32_t> split;
        TRY(glv_split_scalars(scalars, n, split));
        TRY(ctx->io_a.reserve(n * 64));
        TRY(ctx->io_b.reserve(n * 48));
        TRY(ctx->io_c.reserve(n * 64));
        HIP_TRY(hipMemcpyAsync(ctx->io_a.p, pts_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->io_b.p, split.data(), n * 48, hipMemcpyHostToDevice, ctx->stream));
        TRY(launch(ctx, "k_bsn_scalar_mul", [&] {
            hipLaunchKernelGGL(dr::k_bsn_scalar_mul_glv<false>, dim3(div_up(2 * n, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream,
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

int te_msm_groups(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (groups == 0) return DR_OK;
    if (m == 0 || m > 64) return fail(DR_ERR_INVALID, "group size must be in 1..64");
    if (!pts_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    size_t n = groups * m;
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(check_fr_elems(pts_xy, 2 * n, "point"));
    if (drh::te_curve(cv)->glv && m <= 32 && n < 16384) {
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
// ---- fixed-base multiplication (kernels_te.hip.h: k_te_fixed_table / k_te_fixed_base_groups)
static int te_fixed_table(dr_ctx* ctx, int cv, const uint8_t base_xy[64], const uint32_t** out) {
    for (auto& fb : ctx->fixed_bases)
        if (fb.cv == cv && std::memcmp(fb.base_xy, base_xy, 64) == 0) { *out = fb.d_table; return DR_OK; }
    TRY(check_fr_elems(base_xy, 2, "base point"));
    uint32_t *d_base = nullptr, *d_table = nullptr;
    HIP_TRY(hipMalloc((void**)&d_table, (size_t)dr::TE_FIXED_TABLE_WORDS * 4));
    hipError_t e = hipMalloc((void**)&d_base, 64);
    if (e == hipSuccess) e = hipMemcpyAsync(d_base, base_xy, 64, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        LAUNCH_CV(cv, dr::k_te_fixed_table, dim3(1), dim3(64), 0, ctx->stream, d_base, d_table);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (d_base) (void)hipFree(d_base);
    if (e != hipSuccess) {
        (void)hipFree(d_table);
        return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, std::string("fixed-base table: ") + hipGetErrorString(e));
    }
    dr_ctx::FixedBase fb;
    fb.cv = cv;
    std::memcpy(fb.base_xy, base_xy, 64);
    fb.d_table = d_table;
    ctx->fixed_bases.push_back(fb);
    *out = d_table;
    return DR_OK;
}

int te_fixed_base_groups(dr_ctx* ctx, int cv, const uint8_t* bases_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy, bool sync) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (groups == 0) return DR_OK;
    if (m == 0 || m > 4) return fail(DR_ERR_INVALID, "1..4 fixed bases per group");
    if (!bases_xy || !scalars || !out_xy) return fail(DR_ERR_INVALID, "null buffer");
    if (groups * m >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    if (groups * m <= drh::small_host_max()) {
        // a few groups: 64 table additions per term on a host core (~0.03 ms), entries picked by mask (the scalars are secrets and nonces)
        const drh::TeCurveHost* hc = drh::te_curve(cv);
        const drh::TeHostParams hp = drh::te_host_params(*hc);
        std::shared_ptr<const drh::TeFixedTable> tabs_h[4];
        TRY(check_fr_elems(bases_xy, 2 * m, "point"));
        for (size_t j = 0; j < m; j++) {
            tabs_h[j] = drh::te_fixed_table_cached(*hc, bases_xy + 64 * j);
            if (!tabs_h[j]) return fail(DR_ERR_INVALID, "fixed base out of range");
        }
        drh::parallel_for(groups, [&](size_t g) {
            drh::TeExt acc = drh::te_identity();
            for (size_t j = 0; j < m; j++) {
                uint64_t k[4];
                hc->n.reduce_bytes(scalars + 32 * (g * m + j), 32, false, k);
                acc = drh::te_add(acc, drh::te_mul_fixed(*tabs_h[j], k, hp, true), hp);
                explicit_bzero(k, sizeof k);
            }
            drh::te_store_affine(acc, out_xy + 64 * g);
            explicit_bzero(&acc, sizeof acc);
        }, 1);
        return DR_OK;
    }
    dr::TeFixedTables tabs{};
    for (size_t j = 0; j < m; j++) TRY(te_fixed_table(ctx, cv, bases_xy + 64 * j, &tabs.t[j]));
    for (size_t j = m; j < 4; j++) tabs.t[j] = tabs.t[0];
    uint32_t mpad = 1;
    while (mpad < m) mpad <<= 1;
    const uint32_t per_block = 64 / (mpad * dr::TE_FIXED_LANES);
    TRY(ctx->io_b.reserve(groups * m * 32));
    TRY(ctx->io_c.reserve(groups * 64));
    HIP_TRY(hipMemcpyAsync(ctx->io_b.p, scalars, groups * m * 32, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch(ctx, "k_bsn_fixed_base", [&] {
        LAUNCH_CV(cv, dr::k_te_fixed_base_groups, dim3(div_up(groups, per_block)), dim3(64), 0, ctx->stream, tabs, ctx->io_b.as<uint32_t>(),
                  ctx->io_c.as<uint32_t>(), (uint32_t)groups, (uint32_t)m, mpad);
    }));
    HIP_TRY(hipMemcpyAsync(out_xy, ctx->io_c.p, groups * 64, hipMemcpyDeviceToHost, ctx->stream));
    if (sync) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) TRY(prof_collect(ctx));
    }
    return DR_OK;
}
int dr_te_fixed_base_msm_groups(dr_ctx* ctx, int curve, const uint8_t* bases_xy, size_t m, const uint8_t* scalars, size_t groups, uint8_t* out_xy) {
    return te_fixed_base_groups(ctx, curve, bases_xy, scalars, groups, m, out_xy, true);
}

int dr_bsn_msm_groups(dr_ctx* ctx, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    return te_msm_groups(ctx, dr::CV_BANDERSNATCH, pts_xy, scalars, groups, m, out_xy);
}
int dr_te_msm_groups(dr_ctx* ctx, int curve, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy) {
    return te_msm_groups(ctx, curve, pts_xy, scalars, groups, m, out_xy);
}

int te_msm(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (!out_xy) return fail(DR_ERR_INVALID, "null buffer");
    if (n == 0) {
        std::memset(out_xy, 0, 64);
        out_xy[32] = 1;
        return DR_OK;
    }
    // from a few hundred terms: the bucket method (K4, capi_msm.hip) — ~W additions per term instead of a full scalar
    // multiplication (~250 doublings + 64 additions) per term
    constexpr size_t pip_from = 256;
    if (pip_from > 0 && n >= pip_from) {
        if (!pts_xy || !scalars) return fail(DR_ERR_INVALID, "null buffer");
        TRY(check_fr_elems(pts_xy, 2 * n, "point"));
        return te_msm_pippenger(ctx, cv, pts_xy, scalars, n, out_xy);
    }
    if (n <= drh::small_host_max() && n <= 64) {
        // a handful of terms (the 7 / 12 points of a one- or two-proof verifier, a sigma protocol's relation): one fixed-schedule multiplication
        // per term on the worker pool (~0.09 ms each, side by side) and their sum — a kernel chain is ~0.8 ms whatever the count
        if (!pts_xy || !scalars) return fail(DR_ERR_INVALID, "null buffer");
        TRY(check_fr_elems(pts_xy, 2 * n, "point"));
        const drh::TeCurveHost* hc = drh::te_curve(cv);
        const drh::TeHostParams hp = drh::te_host_params(*hc);
        std::vector<drh::TeExt> terms(n);
        drh::parallel_for(n, [&](size_t i) {
            drh::TeExt P;
            uint64_t k[4];
            (void)drh::te_load_affine(pts_xy + 64 * i, P);
            hc->n.reduce_bytes(scalars + 32 * i, 32, false, k);
            terms[i] = drh::te_mul_secret(P, k, hp);
            explicit_bzero(k, sizeof k);
        }, 1);
        drh::TeExt acc = terms[0];
        for (size_t i = 1; i < n; i++) acc = drh::te_add(acc, terms[i], hp);
        drh::te_store_affine(acc, out_xy);
        explicit_bzero(terms.data(), n * sizeof(drh::TeExt));
        return DR_OK;
    }
    // below that: fold 64 terms at a time on the device (one launch); the n/64 partial sums are then added on the host in
    // extended coordinates — a second device pass would pay a full scalar-multiplication latency for scalars that are all 1
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
void launch_decode_points(dr_ctx* ctx, hipStream_t st, int cv, bool tai, const uint32_t* d_enc, uint32_t* d_xy, uint32_t* d_ok, size_t n) {
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

int te_decode_points(dr_ctx* ctx, int cv, bool tai, const uint8_t* enc, size_t n, uint8_t* out_xy, uint8_t* ok) {
    TRY(use_ctx(ctx));
    TRY(check_curve(cv));
    if (n == 0) return DR_OK;
    if (!enc || !out_xy || !ok) return fail(DR_ERR_INVALID, "null buffer");
    if (n >= (1ull << 31)) return fail(DR_ERR_INVALID, "batch too large");
    if (!tai && n <= drh::small_host_max()) {
        // a few points (a proof's own, a public key, a small ring): one host core decodes and subgroup-checks a point in ~0.13 ms, the
        // kernel's dependent chain takes ~0.9 ms whatever the count (hostsmall.hpp: te_decode_checked, the same verdicts)
        const drh::TeCurveHost* hc = drh::te_curve(cv);
        drh::parallel_for(n, [&](size_t i) {
            ok[i] = drh::te_decode_checked(*hc, enc + 32 * i, out_xy + 64 * i) ? 1 : 0;
            if (!ok[i]) std::memset(out_xy + 64 * i, 0, 64);
        }, 1);
        return DR_OK;
    }
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
    if (n <= drh::small_host_max()) {            // a few inputs: ~0.06 ms each on a host core (hostsmall.hpp, the kernel's steps one by one)
        drh::parallel_for(n, [&](size_t i) { (void)drh::te_encode_to_curve_host(u_pairs + 64 * i, out_xy + 64 * i); }, 1);
        return DR_OK;
    }
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

int dr_fr_ops_selftest(dr_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* is_square) {
    TRY(use_ctx(ctx));
    if (n == 0) return DR_OK;
    if (!a || !b || !out || !is_square) return fail(DR_ERR_INVALID, "null buffer");
    if (n >= (1ull << 24)) return fail(DR_ERR_INVALID, "batch too large");
    TRY(check_fr_elems(a, n, "field element"));
    TRY(check_fr_elems(b, n, "field element"));
    TRY(ctx->io_a.reserve(n * 64));
    TRY(ctx->io_b.reserve(n * 384));
    TRY(ctx->io_c.reserve(n * 4));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->io_a.as<uint8_t>() + n * 32, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(dr::k_fr_ops_selftest, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, ctx->io_a.as<uint32_t>(),
                       ctx->io_a.as<uint32_t>() + n * 8, (uint32_t)n, ctx->io_b.as<uint32_t>(), ctx->io_c.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> flags(n);
    HIP_TRY(hipMemcpyAsync(out, ctx->io_b.p, n * 384, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(flags.data(), ctx->io_c.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; i++) is_square[i] = flags[i] ? 1 : 0;
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
        case DR_HASH_SHAKE128_X4: {
            if (len % 4 || out_len % 4 || out_len / 4 > 168) return fail(DR_ERR_INVALID, "four equal messages, four digests of at most 168 bytes");
            const size_t m = len / 4, o = out_len / 4;
            drh::Shake128x4 s;
            const uint8_t* in[4] = {data, data + m, data + 2 * m, data + 3 * m};
            // absorbed in two pieces so that a block boundary inside an update is exercised as well
            const uint8_t* in2[4] = {in[0] + m / 3, in[1] + m / 3, in[2] + m / 3, in[3] + m / 3};
            s.update(in, m / 3);
            s.update(in2, m - m / 3);
            uint8_t* outs[4] = {out, out + o, out + 2 * o, out + 3 * o};
            s.digest(outs, o);
            return DR_OK;
        }
    }
    return fail(DR_ERR_INVALID, "unknown hash kind");
}

int dr_host_random_expand(const uint8_t seed[32], uint8_t* out, size_t len) {
    if (!seed || (len && !out)) return fail(DR_ERR_INVALID, "null buffer");
    try {
        const size_t block = 576, blocks = (len + block - 1) / block;
        drh::parallel_for(blocks, [&](size_t j) {
            drh::Shake256 sh;
            sh.update(seed, 32);
            uint8_t ctr[8];
            for (int i = 0; i < 8; i++) ctr[i] = (uint8_t)((uint64_t)j >> (8 * i));
            sh.update(ctr, 8);
            const size_t lo = j * block, n = std::min(block, len - lo);
            sh.digest(out + lo, n);
        });
    } catch (const std::exception& e) {
        return fail(DR_ERR_NOMEM, std::string("random expand: ") + e.what());
    }
    return DR_OK;
}

int dr_ringvrf_aux_take_blindings(uint8_t* aux, size_t batch, uint8_t* out_blind) {
    if (batch && (!aux || !out_blind)) return fail(DR_ERR_INVALID, "null buffer");
    for (size_t i = 0; i < batch; i++) {
        uint8_t* b = aux + (size_t)DR_RINGVRF_AUX_BYTES * i + 256;
        std::memcpy(out_blind + 32 * i, b, 32);
        explicit_bzero(b, 32);
    }
    return DR_OK;
}

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
                   const uint64_t* salt_off, const uint8_t* xs, uint8_t* inputs_xy, uint8_t* outs_xy, const std::function<void()>* while_waiting) {
    if (su.cv->tai || !su.cv->glv || B == 0 || B >= 16384) {
        if (while_waiting) (*while_waiting)();
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
    // Up to 64 proofs: the two kernels below are dependent chains of ~1.2 ms whatever the batch; a host core maps and multiplies one input in
    // ~0.06 ms (hostsmall.hpp, the same steps as the kernels), sixteen of them 64 inputs in ~0.25 ms.  DOTRING_HEAD_HOST_MAX, default 64 — the
    // knob test proves five proofs both ways.  RingVRF.prove of one proof: 4.85 -> 3.8 ms.
    static const size_t head_host_max = std::getenv("DOTRING_HEAD_HOST_MAX") ? (size_t)std::atol(std::getenv("DOTRING_HEAD_HOST_MAX")) : 64;
    if (su.cv->id == dr::CV_BANDERSNATCH && B <= head_host_max) {
        std::vector<int> bad(B, 0);
        drh::parallel_for(B, [&](size_t i) {
            if (!drh::te_encode_to_curve_host(us.data() + 64 * i, inputs_xy + 64 * i) ||
                !drh::te_scalar_mul_host(inputs_xy + 64 * i, xs + 32 * i, outs_xy + 64 * i))
                bad[i] = 1;
        }, 1);
        if (while_waiting) (*while_waiting)();
        for (size_t i = 0; i < B; i++)
            if (bad[i]) return fail(DR_ERR_DEVICE, "hash to curve on the host: field element out of range");
        return DR_OK;
    }
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
        hipLaunchKernelGGL(dr::k_bsn_scalar_mul_glv<false>, dim3(div_up(2 * B, dr::BSN_BLOCK)), dim3(dr::BSN_BLOCK), 0, ctx->stream, d_in,
                           ctx->io_b.as<uint32_t>(), d_out, (uint32_t)B);
    }));
    if (while_waiting) (*while_waiting)();                 // the two latency chains above run ~1.7 ms: host work that does not need them goes here
    HIP_TRY(hipMemcpyAsync(inputs_xy, d_in, B * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(outs_xy, d_out, B * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) TRY(prof_collect(ctx));
    return DR_OK;
}
