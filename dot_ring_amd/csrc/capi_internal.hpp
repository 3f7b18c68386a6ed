// Internal declarations shared by the translation units of libdotring_hip.so (not part of the C ABI):
//   capi_core.hip   contexts, device memory, profiling, seam A (Bandersnatch kernels), hash-to-curve
//   capi_msm.hip    G1 Pippenger pipeline, seam B (SRS, MSM, G1 codecs), pairing entry points
//   capi_ring.hip   seam C (NTT) and the batched ring prover's phases
//   capi_batch.hip  native batch orchestration: Pedersen / IETF / Ring-VRF prove and verify
//   capi_comm.hip   RCCL communicator for the base-sharded MSM
// Each kernel header is included by exactly one of them; the others reach its kernels through the launch wrappers below.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <array>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dotring_hip.h"
#include "dev_types.hpp"
#include "hostmath.hpp"
#include "hostpairing.hpp"
#include "hostproto.hpp"

namespace dri {

extern thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
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

}  // namespace dri
using dri::ProfEntry;
using dri::Scratch;

struct dr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, ProfEntry> prof_data;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    // MSM workspaces
    Scratch scalars, digits, counts, offsets, cursor, tiles, sorted, buckets, partial, winsum, result, io_a, io_b, io_c, perm, cells, cell_off, part_base, heavy;
    Scratch flag;                            // one word for kernels that report a condition (never an operand buffer of a caller)
    Scratch vfy_bases, vfy_in, vfy_std;      // dr_ringvrf_verify_batch: decompressed G1 points stay resident between its steps
    Scratch vfy_te_in, vfy_te_out, vfy_flags;   // ... and its Bandersnatch decoding (public data: none of the six is ever wiped)
    // The zeroing of secret-derived buffers runs on a stream of its own, behind the work that used them (ctx_wipe_begin / _end); whoever
    // touches the context next waits for it on the device (use_ctx -> ctx_join_wipe), the caller that enqueued it never does.
    hipStream_t wipe_stream = nullptr;
    hipEvent_t wipe_from = nullptr, wipe_done = nullptr;
    bool wipe_pending = false;
    dr_ctx* aux = nullptr;                   // second stream for the latency-bound Bandersnatch side of the batch verifier
    dr_ctx* aux2 = nullptr;                  // third stream: the verifier's two G1 MSMs run side by side
    std::vector<dr_ctx*> helpers;            // further streams working for this context (a prover's Pedersen stream): profiling only
    dr::TwiddleCache twiddles;
    // fixed-base window tables of constant points (generator, blinding base): built on first use, 48 KB each
    struct FixedBase {
        int cv;
        uint8_t base_xy[64];
        uint32_t* d_table;
    };
    std::vector<FixedBase> fixed_bases;
};

struct dr_srs {
    int device = 0;
    size_t count = 0;
    uint32_t* d_bases = nullptr;   // G1Affine[count], Montgomery
    // optional fixed-base window table: table[w][i] = 2^(start_w) * base[i]; all windows share one bucket set
    uint32_t* d_table = nullptr;
    dr::WindowTable table_wt{};
    // small SRS: the table has one row per BIT (table[s][i] = 2^s * base[i]), table_wt.row[w] = start[w]; batched MSMs then tile the
    // scalar in non-adjacent form with buckets for odd digit multiples only (tiling_for)
    bool table_bit_rows = false;
    uint32_t table_pt_words = 24;        // words per table record: 24 (packed) or 32 (one point per 128-byte line)
    int table_naf_delta = -2;            // width of the non-adjacent form of batched MSMs: -2 = chosen per call (tiling_for), -1 = never, >= 0: window_bits + this
    // derived bases for summation-by-parts commitments, keyed by log2(domain size): PS_j = sum_{i<=j} L_i(tau) G
    std::map<unsigned, dr_srs*> lagrange_prefix;
    std::mutex derive_mutex;             // provers for the same SRS may be created from different threads
};

namespace dri {

int use_ctx(dr_ctx* ctx);
int prof_collect(dr_ctx* ctx);
// Secrets do not stay in HBM past the call that used them: zero every scratch buffer of the context (scalar uploads, MSM digit rows,
// sorted entries, buckets and partial sums — all of them functions of the scalars), stream-ordered behind the work already enqueued;
// the caller does not wait.  DOTRING_WIPE=0 turns every wipe of the library off (the A/B of its cost).
bool wipe_enabled();
// `in_stream`: the memsets go into the context's own stream (helper contexts whose next user is another thread's plain launch);
// otherwise onto the wipe stream.  With the per-kernel timers on they always go in-stream, so that the "wipe" timer holds them.
int ctx_wipe_scratch(dr_ctx* ctx, bool in_stream = false);
// the stream a wipe's memsets go to (ordered behind everything enqueued on ctx->stream so far); ctx_wipe_end marks them enqueued
int ctx_wipe_begin(dr_ctx* ctx, bool in_stream, hipStream_t* out);
int ctx_wipe_end(dr_ctx* ctx, hipStream_t wipe_st);
int ctx_wipe_enqueue_scratch(dr_ctx* ctx, hipStream_t wipe_st);     // the memsets of ctx_wipe_scratch, between a begin and an end
// make ctx->stream wait (on the device) for a wipe enqueued earlier; nothing to do when there is none
int ctx_join_wipe(dr_ctx* ctx);
// the two halves of dr_pairing_check (capi_msm.hip), for callers that have their pairs at different times: the product of the Miller
// loops of n (G1, G2) pairs (encodings as dr_pairing_check takes them; pairs with an infinite member contribute 1), and the final
// exponentiation's verdict on a product of such values
int pairing_miller(const uint8_t* g1_be_xy, const uint8_t* g2_be, size_t n, drh::Fq12& f);
bool pairing_product_is_one(const drh::Fq12& f);
// non-zero 32-bit words in the context's scratch buffers (the test of the wipe)
int ctx_scratch_residue(dr_ctx* ctx, uint64_t* words);
int count_nonzero_words(dr_ctx* ctx, const void* d_buf, size_t bytes, uint64_t* total);   // adds to *total
// dr_ctx_create for a helper context of another one (second / third stream of the same GPU); `role` is kept for the call sites' sake
int ctx_create_role(int device_id, int role, dr_ctx** out);

// kernel launch wrapper with optional hipEvent timing on the ctx stream
template <class F>
int launch(dr_ctx* ctx, const char* name, F&& f) {
    if (ctx->prof) {
        hipEvent_t a = nullptr, b = nullptr;
        hipError_t e = hipEventCreate(&a);
        if (e == hipSuccess) e = hipEventCreate(&b);
        if (e == hipSuccess) e = hipEventRecord(a, ctx->stream);
        if (e == hipSuccess) {
            f();
            e = hipEventRecord(b, ctx->stream);
        }
        if (e != hipSuccess) {            // no event may outlive a failed launch
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
            return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, std::string("kernel timer: ") + hipGetErrorString(e));
        }
        ctx->prof_pending.push_back({name, {a, b}});
    } else {
        f();
    }
    HIP_TRY(hipGetLastError());
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

// launch kernel template K<CV> for the curve id cv (dr::CV_BANDERSNATCH / dr::CV_JUBJUB)
#define LAUNCH_CV(cv, K, ...)                                                                  \
    do {                                                                                        \
        if ((cv) == dr::CV_JUBJUB) hipLaunchKernelGGL((K<dr::CV_JUBJUB>), __VA_ARGS__);         \
        else hipLaunchKernelGGL((K<dr::CV_BANDERSNATCH>), __VA_ARGS__);                         \
    } while (0)

// ---- knobs (environment, read in dr_ctx_create; defined in capi_core.hip)
extern int g_force_c;              // test hook: DOTRING_MSM_WINDOW
bool window_ok(int c);

}  // namespace dri

// body of a helper std::thread: status and message go to the joining thread; no exception may escape (std::terminate)
template <class F>
void run_guarded(int& rc, std::string& err, F&& f) {
    try {
        rc = f();
        if (rc != DR_OK) err = dr_last_error();
    } catch (const std::bad_alloc&) {
        rc = DR_ERR_NOMEM;
        err = "out of host memory";
    } catch (const std::exception& e) {
        rc = DR_ERR_DEVICE;
        err = e.what();
    }
}

// ---- helpers shared between translation units (global scope, hidden visibility)
bool ctx_alive(dr_ctx* c);

// ---- capi_core.hip
int check_curve(int cv);
int check_fr_elems(const uint8_t* p, size_t count, const char* what);
int glv_split_scalars(const uint8_t* scalars, size_t n, std::vector<uint32_t>& out);
int te_scalar_mul_batch(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t* out_xy);
int te_msm(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]);
int te_msm_groups(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t groups, size_t m, uint8_t* out_xy);
// sum_j k[g*m+j] * Base_j for `groups` groups over m <= 4 CONSTANT bases (fixed-base window tables, cached per context)
int te_fixed_base_groups(dr_ctx* ctx, int cv, const uint8_t* bases_xy /* m*64 */, const uint8_t* scalars /* groups*m*32 */, size_t groups,
                         size_t m, uint8_t* out_xy /* groups*64 */, bool sync = true);
int te_decode_points(dr_ctx* ctx, int cv, bool tai, const uint8_t* enc, size_t n, uint8_t* out_xy, uint8_t* ok);
void launch_decode_points(dr_ctx* ctx, hipStream_t st, int cv, bool tai, const uint32_t* d_enc, uint32_t* d_xy, uint32_t* d_ok, size_t n);
int load_suite(const dr_vrf_suite* s, drh::VrfSuite& out);
int encode_to_curve_msgs(dr_ctx* ctx, const drh::VrfSuite& su, size_t B, const uint8_t* data, const uint64_t* off, const uint8_t* salts,
                         const uint64_t* salt_off, uint8_t* out_xy);
// while_waiting (optional): host work that needs nothing from these kernels, run on the calling thread after the launches and before the wait
int encode_and_mul(dr_ctx* ctx, const drh::VrfSuite& su, size_t B, const uint8_t* data, const uint64_t* off, const uint8_t* salts,
                   const uint64_t* salt_off, const uint8_t* xs, uint8_t* inputs_xy, uint8_t* outs_xy,
                   const std::function<void()>* while_waiting = nullptr);

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

// ---- capi_msm.hip
// Fixed-base table descriptor for msm_device (table == nullptr: plain bases, one bucket set per window).
struct MsmTable {
    const uint32_t* table = nullptr;
    dr::WindowTable wt{};
    uint32_t pt_words = 24;              // words per table record
    bool bit_rows = false;               // the table has a row per bit: a call may recode the scalars as it likes (non-adjacent form)
    int naf_delta = -2;                  // see dr_srs::table_naf_delta
    uint32_t stride = 0, offset = 0;
    uint32_t short_from = 0xffffffffu, n_short = 0;   // batched MSM: vectors from this index on are zero beyond n_short (sort hint)
    bool fold_sign = false;              // scalars above r / 2 enter as their negatives (difference columns: r - 1 becomes -1, one digit)
};
// exact_streams: internal — the second run of a call whose partition sort overfilled a stream (see msm_device)
int msm_device(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch, std::vector<drh::G1>& results,
               const MsmTable* tbl = nullptr, bool exact_streams = false);
MsmTable srs_table(const dr_srs* srs, size_t offset);
// the tiling msm_device takes for `batch` MSMs of n points over this table: mode 0 = the table's window rows, 2 = width-c non-adjacent
// form (bit-row tables, hundreds of MSMs); slots = digit rows per scalar, digits = expected non-zero digits per scalar
struct Tiling {
    int mode, c, slots;
    double digits;
};
Tiling tiling_for(const MsmTable& t, size_t n, size_t batch);
int srs_precompute(dr_ctx* ctx, dr_srs* srs, int window_bits, bool allow_bit_rows);   // dr_srs_precompute with the table shape chosen
void g1_result_to_bytes(const drh::G1& r, uint8_t* out96, int* is_inf);
int msm_batch_results_to_bytes(dr_ctx* ctx, size_t batch, uint8_t* out_be_xy, int* is_inf);
int msm_to_bytes(dr_ctx* ctx, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, size_t batch, uint8_t* out_be_xy, int* is_inf,
                 const MsmTable* tbl = nullptr);
int g1_be_to_le_limbs(const uint8_t* be, size_t m, std::vector<uint8_t>& le, bool check_curve);
// K4: one twisted Edwards MSM by the bucket method (n from a few hundred terms)
int te_msm_pippenger(dr_ctx* ctx, int cv, const uint8_t* pts_xy, const uint8_t* scalars, size_t n, uint8_t out_xy[64]);
// launch wrappers around kernels_g1.hip.h for the batch verifier
void g1_launch_decompress(hipStream_t st, const uint8_t* d_enc, uint32_t* d_bases, uint32_t* d_ok, size_t n);
void g1_launch_bases_to_mont(hipStream_t st, uint32_t* d_bases, size_t n);
void g1_launch_bases_from_mont(hipStream_t st, const uint32_t* d_bases, uint32_t* d_out, size_t n);

// ---- capi_ring.hip
dr_ctx* ring_prover_ctx(dr_ring_prover* p);
int ring_prover_curve(const dr_ring_prover* p);
int ring_prover_aux_ctx(dr_ring_prover* p, dr_ctx** out);      // the prover's second stream (created on first use)


