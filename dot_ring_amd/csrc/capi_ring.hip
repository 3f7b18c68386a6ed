// libdotring_hip.so — C ABI, part 3 of 5: seam C (the Fr NTT of kernels_ntt.hip.h) and the phases of the batched ring
// prover (kernels_ring.hip.h): everything between two Fiat-Shamir hashes stays in HBM.
#include "capi_internal.hpp"
#include "kernels_ntt.hip.h"
#include "kernels_ring.hip.h"

using namespace dri;

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
    Scratch fixed4, lag4, not_last;     // tables on the 4N domain, FS9 records (raw 9-limb Montgomery 2^261): on the three
                                        // non-trivial cosets, coset-major ([poly][c - 1][j])
    Scratch coset_scale;                // [3][N] FS9: zeta^(c m) R^2, the multipliers of the scaled N-point NTT input
    Scratch special;                    // [B][3] FS9: the aggregated constraint polynomial at the three hidden rows of coset 0
    Scratch hid;                        // [B][4][4][8] std: rows N-3, N-2, N-1, 0 of the witness columns' evaluations (saved by the witness phase)
    bool fwd_pending = false;           // the witness phase has launched the coset transforms of this batch (they run beside the host's hashing)
    bool quot2_pending = false;         // the evaluation phase has launched the synthetic division of the linearisation polynomial
    uint8_t root[3 * 96];
    int root_inf[3];
    // per-batch state
    size_t batch = 0;
    Scratch idx, blind, zk, chain_ext, prefix, chain_aff, cnt, relation, rps, cols, wit4, alphas, alphas9, alpha_aux, agg, q, zetas, evals, ks, lin,
        nus, nus9, aggo, chunkv, quot1, quot2, diffs;
};

namespace {

// fmt_in / fmt_out: dr::NTT_FMT_STD8 (8 canonical words, standard form) or dr::NTT_FMT_FS9 (raw 9-limb Montgomery records)
int ring_ntt(dr_ring_prover* p, uint32_t* d_data, unsigned log2n, size_t batch, bool inverse, int fmt_in = dr::NTT_FMT_STD8,
             int fmt_out = dr::NTT_FMT_STD8, const uint32_t* d_src = nullptr, int pad = 0, uint32_t src_div = 1,
             const uint32_t* d_in_scale = nullptr, const uint32_t* d_special = nullptr) {
    dr_ctx* ctx = p->ctx;
    const drh::Fr& w = log2n == p->rc.log2n ? p->omega_n : p->omega_4n;
    drh::Fr wi = inverse ? w.inv() : w;
    drh::Fr scale;
    if (inverse) scale = drh::Fr::from_u64((uint64_t)1 << log2n).inv();
    const size_t out_words = fmt_out == dr::NTT_FMT_FS9 ? dr::L29 : 8;
    // source words per TRANSFORM (the source pointer advances by whole launches): a scaled source is shared by src_div transforms,
    // a coset-major source holds 3/4 of the points
    const size_t in_words_x4 = fmt_in == dr::NTT_FMT_FS9 ? 4 * dr::L29 : fmt_in == dr::NTT_FMT_FS9_COSETS ? 3 * dr::L29 : 32;
    // dr_ntt limits one launch to 65535 transforms (grid.y): split larger batches (in multiples of src_div)
    const size_t per_launch = 65535 / src_div * src_div;
    for (size_t done = 0; done < batch;) {
        size_t take = std::min<size_t>(batch - done, per_launch);
        int rc = dr::ntt_run(ctx->stream, [&](const char* name, auto&& f) { return launch(ctx, name, f); }, ctx->twiddles, ctx->io_b,
                             d_data + done * (out_words << log2n), log2n, take, wi, inverse ? &scale : nullptr,
                             [&]() -> int { return DR_OK; }, fmt_in, fmt_out,
                             d_src ? d_src + (done / src_div) * ((in_words_x4 << (log2n - pad)) / 4) : nullptr, pad, src_div, d_in_scale,
                             d_special ? d_special + done * 3 * dr::L29 : nullptr);
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
    // window rows only: vectors of a few hundred scalars, most of them +-1, gain nothing from the non-adjacent form of a bit-row table
    // (measured in round 4: bit rows + width-11/12 non-adjacent digits for these bases, 16.8-17.0k against 16.9-17.1k proofs/s)
    rc = srs_precompute(ctx, ps, ps_bits, false);
    if (rc != DR_OK) { dr_srs_destroy(ps); return rc; }
    srs->lagrange_prefix[log2n] = ps;
    *out = ps;
    return DR_OK;
}

}  // namespace

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
    const uint32_t n = 1u << log2n;
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
    TRY(lagrange_prefix_srs(ctx, srs, log2n, p->omega_n, &p->ps_srs));       // the witness commitments go by summation by parts
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
    rc.tail9[0] = dr::fs_arg_mont(e3.neg()); rc.tail9[1] = dr::fs_arg_mont(e2); rc.tail9[2] = dr::fs_arg_mont(e1.neg());
    rc.tail9[3] = dr::fs_arg_mont(drh::Fr::one());
    rc.omega9 = dr::fs_arg_mont(p->omega_n);
    rc.nl_hidden[0] = dr::fs_arg_mont(w3 - w4); rc.nl_hidden[1] = dr::fs_arg_mont(w2 - w4); rc.nl_hidden[2] = dr::fs_arg_mont(w1 - w4);
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
    Scratch lagc;
    TRY(lagc.reserve((size_t)2 * n * 32));
    hipLaunchKernelGGL(dr::k_ring_lagrange, dim3(div_up(n, 256)), dim3(256), 0, st, lagc.as<uint32_t>(), n,
                       arg_of(drh::Fr::from_u64(n).inv()), arg_of(w4.inv()));
    // tables on the three non-trivial cosets, coset-major: per polynomial three N-point NTTs of its coefficients scaled by zeta^(c m)
    TRY(p->coset_scale.reserve((size_t)3 * n * dr::L29 * 4));
    hipLaunchKernelGGL(dr::k_ring_coset_scale, dim3(div_up((size_t)3 * n, 256)), dim3(256), 0, st, p->coset_scale.as<uint32_t>(), n,
                       dr::fs_arg_mont(p->omega_4n));
    TRY(p->fixed4.reserve((size_t)3 * 3 * n * dr::L29 * 4));
    TRY(ring_ntt(p, p->fixed4.as<uint32_t>(), log2n, 9, false, dr::NTT_FMT_STD8_SCALED, dr::NTT_FMT_FS9, p->fixed_coef.as<uint32_t>(), 0, 3,
                 p->coset_scale.as<uint32_t>()));
    TRY(p->lag4.reserve((size_t)2 * 3 * n * dr::L29 * 4));
    TRY(ring_ntt(p, p->lag4.as<uint32_t>(), log2n, 6, false, dr::NTT_FMT_STD8_SCALED, dr::NTT_FMT_FS9, lagc.as<uint32_t>(), 0, 3,
                 p->coset_scale.as<uint32_t>()));
    TRY(p->not_last.reserve((size_t)3 * n * dr::L29 * 4));
    hipLaunchKernelGGL(dr::k_ring_not_last3, dim3(div_up((size_t)3 * n, 256)), dim3(256), 0, st, p->not_last.as<uint32_t>(), n, arg_of(p->omega_4n),
                       arg_of(w4));
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
    for (Scratch* s : {&p->ring_pts_mont, &p->fixed_coef, &p->fixed4, &p->lag4, &p->not_last, &p->coset_scale, &p->special, &p->hid, &p->idx, &p->blind, &p->zk, &p->chain_ext,
                       &p->prefix, &p->chain_aff, &p->cnt, &p->relation, &p->rps, &p->cols, &p->wit4, &p->alphas, &p->alphas9, &p->alpha_aux, &p->agg, &p->q, &p->zetas,
                       &p->evals, &p->ks, &p->lin, &p->nus, &p->nus9, &p->aggo, &p->chunkv, &p->quot1, &p->quot2, &p->diffs})
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
    p->quot2_pending = false;
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
    TRY(p->wit4.reserve(batch * 4 * (size_t)n * 4 * dr::L29 * 4));
    uint32_t* col_evals = p->wit4.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(p->idx.p, producer_index, batch * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->blind.p, blinding, batch * 32, hipMemcpyHostToDevice, st));
    if (zk_rows) {
        TRY(p->zk.reserve(batch * 12 * 32));
        HIP_TRY(hipMemcpyAsync(p->zk.p, zk_rows, batch * 12 * 32, hipMemcpyHostToDevice, st));
    }
    TRY(launch(ctx, "k_ring_chain", [&] {
        LAUNCH_CV(p->curve, dr::k_ring_chain_wave, dim3((unsigned)batch), dim3(64), 0, st, p->ring_pts_mont.as<uint32_t>(), p->idx.as<uint32_t>(),
                               p->blind.as<uint32_t>(), rc, (uint32_t)batch, p->chain_ext.as<uint32_t>(), p->chain_aff.as<uint32_t>(),
                               p->cnt.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_ring_columns", [&] {
        hipLaunchKernelGGL(dr::k_ring_relations, dim3(div_up(batch, 64)), dim3(64), 0, st, p->chain_aff.as<uint32_t>(), p->cnt.as<uint32_t>(),
                           (uint32_t)batch, p->relation.as<uint32_t>(), p->rps.as<uint32_t>());
        hipLaunchKernelGGL(dr::k_ring_columns, dim3(div_up(batch * n, 256)), dim3(256), 0, st, p->idx.as<uint32_t>(), p->blind.as<uint32_t>(),
                           p->chain_aff.as<uint32_t>(), zk_rows ? p->zk.as<uint32_t>() : nullptr, rc, (uint32_t)batch, col_evals);
    }));
    HIP_TRY(hipMemcpyAsync(out_relation_xy, p->relation.p, batch * 64, hipMemcpyDeviceToHost, st));
    {
        // commit in evaluation form by summation by parts (sparse scalars), then interpolate for the later phases
        TRY(p->diffs.reserve(batch * 4 * (size_t)n * 32));
        TRY(launch(ctx, "k_ring_diff", [&] {
            hipLaunchKernelGGL(dr::k_ring_diff, dim3(div_up(batch * 4 * n, 256)), dim3(256), 0, st, col_evals, n, batch * 4,
                               p->diffs.as<uint32_t>());
        }));
        TRY(ring_ntt(p, p->cols.as<uint32_t>(), rc.log2n, batch * 4, true, dr::NTT_FMT_STD8, dr::NTT_FMT_STD8, col_evals, 0));
        MsmTable t = srs_table(p->ps_srs, 0);
        t.fold_sign = true;                 // first differences of bit columns: +-1
        TRY(msm_to_bytes(ctx, p->ps_srs->d_bases, p->diffs.as<uint32_t>(), n, batch * 4, out_commitments, is_inf, &t));
    }
    p->fwd_pending = false;
    // The commitments are on the host; the caller now hashes them into the transcript (0.3 - 0.45 ms for 1024 proofs).  What the
    // quotient phase does first needs no challenge: the four columns on the three cosets.  Launched here and NOT waited for, the
    // transforms run beside that hashing; the hidden rows of coset 0 are saved first, the transforms overwrite the evaluation columns.
    TRY(p->hid.reserve(batch * 16 * 32));
    hipLaunchKernelGGL(dr::k_ring_save_rows, dim3(div_up(batch * 16, 256)), dim3(256), 0, st, col_evals, n, (uint32_t)batch, p->hid.as<uint32_t>());
    TRY(ring_ntt(p, p->wit4.as<uint32_t>(), rc.log2n, batch * 12, false, dr::NTT_FMT_STD8_SCALED, dr::NTT_FMT_FS9, p->cols.as<uint32_t>(), 0, 3,
                 p->coset_scale.as<uint32_t>()));
    HIP_TRY(hipGetLastError());
    p->fwd_pending = true;
    return DR_OK;
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
    TRY(p->alphas9.reserve(batch * 7 * dr::L29 * 4));
    TRY(p->wit4.reserve(batch * 4 * (size_t)m * dr::L29 * 4));
    TRY(p->agg.reserve(batch * (size_t)m * dr::L29 * 4));
    TRY(p->q.reserve(batch * (size_t)qn * 32));
    HIP_TRY(hipMemcpyAsync(p->alphas.p, alphas, batch * 7 * 32, hipMemcpyHostToDevice, st));
    // 7 alphas per proof are read by every point of the 4N domain: converted once per batch — FS9 records for the constraint
    // kernel, the 8-word Montgomery form for the two small per-proof kernels (k_ring_alpha_aux, k_ring_lin_scalars)
    hipLaunchKernelGGL(dr::k_fr_std_to_fs9, dim3(div_up(batch * 7, 256)), dim3(256), 0, st, p->alphas.as<uint32_t>(), p->alphas9.as<uint32_t>(), batch * 7);
    hipLaunchKernelGGL(dr::k_fr_to_mont, dim3(div_up(batch * 7, 256)), dim3(256), 0, st, p->alphas.as<uint32_t>(), batch * 7);
    TRY(p->alpha_aux.reserve(batch * 2 * dr::L29 * 4));
    hipLaunchKernelGGL(dr::k_ring_alpha_aux, dim3(div_up(batch, 64)), dim3(64), 0, st, p->alphas.as<uint32_t>(), p->rps.as<uint32_t>(), rc,
                       (uint32_t)batch, p->alpha_aux.as<uint32_t>());
    // coset 0: the three hidden rows from the N-domain evaluations the witness phase saved
    TRY(p->special.reserve(batch * 3 * dr::L29 * 4));
    if (!p->fwd_pending) return fail(DR_ERR_INVALID, "quotient phase without a witness phase for this batch");
    p->fwd_pending = false;
    // (the forward transforms — N coefficients per column -> evaluations on the cosets zeta^c H, c = 1..3: three N-point NTTs of the
    //  coefficients scaled by zeta^(c m), raw 9-limb records, coset-major — were launched by the witness phase and are in this stream)
    TRY(launch(ctx, "k_ring_constraints", [&] {
        LAUNCH_CV(p->curve, dr::k_ring_hidden_rows, dim3(div_up(batch * 3, 64)), dim3(64), 0, st, p->hid.as<uint32_t>(), p->ring_pts_mont.as<uint32_t>(),
                  p->alphas9.as<uint32_t>(), rc, (uint32_t)batch, p->special.as<uint32_t>());
    }));
    TRY(launch(ctx, "k_ring_constraints", [&] {
        LAUNCH_CV(p->curve, dr::k_ring_constraints3, dim3(div_up(batch * 3 * (size_t)n, 256)), dim3(256), 0, st, p->wit4.as<uint32_t>(),
                  p->fixed4.as<uint32_t>(), p->lag4.as<uint32_t>(), p->not_last.as<uint32_t>(), p->alphas9.as<uint32_t>(),
                  p->alpha_aux.as<uint32_t>(), rc, (uint32_t)batch, p->agg.as<uint32_t>());
    }));
    // inverse 4N-point transform of (zeros and the hidden rows on coset 0, the three evaluated cosets): standard-form coefficients
    TRY(ring_ntt(p, p->wit4.as<uint32_t>(), rc.log2n + 2, batch, true, dr::NTT_FMT_FS9_COSETS, dr::NTT_FMT_STD8, p->agg.as<uint32_t>(), 0, 1, nullptr,
                 p->special.as<uint32_t>()));
    TRY(launch(ctx, "k_ring_quotient", [&] {
        hipLaunchKernelGGL(dr::k_ring_quotient, dim3(div_up(batch * qn, 256)), dim3(256), 0, st, p->wit4.as<uint32_t>(), rc, (uint32_t)batch,
                           p->q.as<uint32_t>());
    }));
    MsmTable t = srs_table(p->srs, 0);
    return msm_to_bytes(ctx, p->srs->d_bases, p->q.as<uint32_t>(), qn, batch, out_cq, is_inf, &t);
}

// synthetic division of [batch] polynomials of len coefficients by (X - zeta) or (X - zeta*omega): chunk-local pass, link, write
static int ring_syndiv(dr_ring_prover* p, size_t batch, const uint32_t* poly, uint32_t len, int mul_omega, uint32_t* quot) {
    dr_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    const dr::RingConsts& rc = p->rc;
    const uint32_t nch = (len + dr::SD_CHUNK - 1) / dr::SD_CHUNK;
    return launch(ctx, "k_syndiv", [&] {
        hipLaunchKernelGGL(dr::k_syndiv_local, dim3(div_up(batch * nch, 128)), dim3(128), 0, st, poly, len, p->zetas.as<uint32_t>(), mul_omega, rc,
                           (uint32_t)batch, p->chunkv.as<uint32_t>());
        hipLaunchKernelGGL(dr::k_syndiv_link, dim3(div_up(batch, 64)), dim3(64), 0, st, p->chunkv.as<uint32_t>(), len, p->zetas.as<uint32_t>(),
                           mul_omega, rc, (uint32_t)batch);
        hipLaunchKernelGGL(dr::k_syndiv_write, dim3(div_up(batch * nch, 128)), dim3(128), 0, st, poly, len, p->zetas.as<uint32_t>(), mul_omega, rc,
                           (uint32_t)batch, p->chunkv.as<uint32_t>(), quot);
    });
}

// the second opening quotient, lin / (X - zeta*omega), zero-padded into the second half of the shared MSM's scalar vectors; needs no nu
static int ring_quot2(dr_ring_prover* p, size_t batch) {
    const uint32_t n = p->rc.n, qn = 3 * n + 1;
    TRY(ring_syndiv(p, batch, p->lin.as<uint32_t>(), n, 1, p->quot2.as<uint32_t>()));
    return launch(p->ctx, "k_ring_pad", [&] {
        hipLaunchKernelGGL(dr::k_ring_pad, dim3(div_up(batch * (size_t)(qn - 1), 256)), dim3(256), 0, p->ctx->stream, p->quot2.as<uint32_t>(), n - 1,
                           p->quot1.as<uint32_t>() + batch * (size_t)(qn - 1) * 8, qn - 1, batch);
    });
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
    TRY(p->ks.reserve(batch * 3 * dr::L29 * 4));
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
    // The caller hashes the evaluations into the transcript to draw the nus; the second opening quotient needs none of them: launched
    // here and not waited for, it runs beside that hashing.  Both quotients of all proofs share ONE batched MSM: [2*batch][3N] scalar
    // vectors, the short second quotient zero-padded (zero scalars produce no digits) — one sort / accumulate / reduce / affine pipeline.
    const uint32_t qn = 3 * n + 1;
    TRY(p->chunkv.reserve(batch * (size_t)((qn + dr::SD_CHUNK - 1) / dr::SD_CHUNK) * 32));
    TRY(p->quot1.reserve(2 * batch * (size_t)(qn - 1) * 32));
    TRY(p->quot2.reserve(batch * (size_t)(n - 1) * 32));
    TRY(ring_quot2(p, batch));
    HIP_TRY(hipGetLastError());
    p->quot2_pending = true;
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
    TRY(p->nus9.reserve(batch * 8 * dr::L29 * 4));
    HIP_TRY(hipMemcpyAsync(p->nus.p, nus, batch * 8 * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dr::k_fr_std_to_fs9, dim3(div_up(batch * 8, 256)), dim3(256), 0, st, p->nus.as<uint32_t>(), p->nus9.as<uint32_t>(), batch * 8);     // multipliers: Montgomery form
    TRY(launch(ctx, "k_ring_aggpoly", [&] {
        hipLaunchKernelGGL(dr::k_ring_aggpoly, dim3(div_up(batch * qn, 256)), dim3(256), 0, st, p->fixed_coef.as<uint32_t>(), p->cols.as<uint32_t>(),
                           p->q.as<uint32_t>(), p->nus9.as<uint32_t>(), n, (uint32_t)batch, p->aggo.as<uint32_t>());
    }));
    if (!p->quot2_pending) return fail(DR_ERR_INVALID, "openings phase without an evaluation phase for this batch");
    p->quot2_pending = false;               // (lin's quotient was launched by the evaluation phase and is in this stream)
    TRY(ring_syndiv(p, batch, p->aggo.as<uint32_t>(), qn, 0, p->quot1.as<uint32_t>()));
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
    // the last phase of a batch: its results are on the host, nothing of the witness has to outlive it
    return dr_ring_prover_wipe(p);
}

namespace {
std::vector<Scratch*> prover_batch_state(dr_ring_prover* p) {
    return {&p->idx, &p->blind, &p->zk, &p->chain_ext, &p->prefix, &p->chain_aff, &p->cnt, &p->relation, &p->rps, &p->cols, &p->wit4, &p->alphas,
            &p->alphas9, &p->alpha_aux, &p->agg, &p->q, &p->zetas, &p->evals, &p->ks, &p->lin, &p->nus, &p->nus9, &p->aggo, &p->chunkv, &p->quot1,
            &p->quot2, &p->diffs, &p->special, &p->hid};
}
}  // namespace

// Blinding factors, hidden rows, the bit column and every polynomial derived from them (witness columns and their evaluations, the
// by-parts differences, quotient, linearisation and opening quotients — 2.6 GB for 1024 proofs at N = 2048) are zeroed when the last
// phase of a batch has delivered its results, and so is the MSM scratch of the prover's context (digit rows, sorted entries, buckets of
// the witness MSMs): 6.6 GB for 1024 proofs, 1.1 ms of memsets.  They go onto the context's wipe stream behind the batch's last kernel
// and nobody waits for them on the host; the next call that uses the context waits for them on the device (use_ctx) — except the
// batch verifier, whose decoding phase works in buffers of its own and runs beside them (capi_batch.hip).  Per-ring tables (public)
// stay.  Also callable on request (a Ring about to be parked).
int dr_ring_prover_wipe(dr_ring_prover* p) {
    if (!p) return fail(DR_ERR_INVALID, "null argument");
    if (!wipe_enabled()) return DR_OK;
    hipStream_t wst = nullptr;
    TRY(ctx_wipe_begin(p->ctx, false, &wst));
    hipError_t e = hipSuccess;
    size_t total = 0;
    TRY(launch(p->ctx, "wipe", [&] {
        for (Scratch* s : prover_batch_state(p))
            if (s->p && s->cap && e == hipSuccess) { e = hipMemsetAsync(s->p, 0, s->cap, wst); total += s->cap; }
    }));
    HIP_TRY(e);
    if (std::getenv("DOTRING_TRACE")) {
        std::fprintf(stderr, "[dotring] wipe of prover state: %.1f MB |", (double)total / 1e6);
        for (Scratch* s : prover_batch_state(p)) std::fprintf(stderr, " %.0f", (double)s->cap / 1e6);
        std::fprintf(stderr, "\n");
    }
    p->fwd_pending = p->quot2_pending = false;
    TRY(ctx_wipe_enqueue_scratch(p->ctx, wst));
    return ctx_wipe_end(p->ctx, wst);
}

// test hook: non-zero 32-bit words left in the prover's per-batch state, its context's scratch and its helper context's scratch
int dr_ring_prover_residue(dr_ring_prover* p, uint64_t* words) {
    if (!p || !words) return fail(DR_ERR_INVALID, "null argument");
    *words = 0;
    TRY(use_ctx(p->ctx));
    for (Scratch* s : prover_batch_state(p)) TRY(count_nonzero_words(p->ctx, s->p, s->cap, words));
    TRY(ctx_scratch_residue(p->ctx, words));
    if (p->aux_ctx) {
        TRY(ctx_scratch_residue(p->aux_ctx, words));
        TRY(use_ctx(p->ctx));
    }
    return DR_OK;
}

dr_ctx* ring_prover_ctx(dr_ring_prover* p) { return p->ctx; }
int ring_prover_curve(const dr_ring_prover* p) { return p->curve; }
int ring_prover_aux_ctx(dr_ring_prover* p, dr_ctx** out) {
    if (!p->aux_ctx) {
        TRY(ctx_create_role(p->ctx->device, 1, &p->aux_ctx));
        p->ctx->helpers.push_back(p->aux_ctx);
    }
    *out = p->aux_ctx;
    return DR_OK;
}
