// libdotring_hip.so — C ABI, part 4 of 5: native batch orchestration.  Pedersen / IETF / Ring-VRF prove_batch and
// batch_verify run the whole protocol in the library: GPU phases through the entry points of the other parts, the hashing
// between them on worker threads (hosthash.hpp, hostproto.hpp).
#include "capi_internal.hpp"
#include <atomic>
#include "hostsigma.hpp"

using namespace dri;

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
    ~PedersenBatch() {          // secret scalars, blinding factors and nonces (and the scalar vectors built from them) do not outlive the call
        for (std::vector<uint8_t>* v : {&xs, &blind, &ks, &kbs, &sc, &sc3})
            if (!v->empty()) explicit_bzero(v->data(), v->size());
    }

    int head(dr_ctx* ctx, const uint8_t* alphas, const uint64_t* alpha_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
             const uint64_t* salt_off, const uint8_t* secret_scalars, PhaseTrace& tr_, const std::function<void()>* while_waiting = nullptr) {
        const drh::Mod256& mn = su.cv->n;
        // 1. secrets mod n
        xs.resize(B * 32);
        for (size_t i = 0; i < B; i++) {
            uint64_t x[4];
            mn.reduce_bytes(secret_scalars + 32 * i, 32, false, x);
            drh::store_le32(x, xs.data() + 32 * i);
        }
        // 2. I_i = encode_to_curve(salt || alpha), O_i = x_i * I_i
        inputs.resize(B * 64); outs.resize(B * 64);
        TRY(encode_and_mul(ctx, su, B, alphas, alpha_off, salts, salt_off, xs.data(), inputs.data(), outs.data(), while_waiting));
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
        // device copies of x, b, k, k_b (scalar uploads of the fixed-base and variable-base launches) do not outlive the call — on ANY exit path
        struct ScratchWipe {
            dr_ctx* c;
            bool armed = true;
            ~ScratchWipe() { if (armed) (void)ctx_wipe_scratch(c, true); }
        } scratch_wipe{actx};
        const drh::Mod256& mn = su.cv->n;
        const int cv = su.cv->id;
        ybar.resize(B * 64); ks.resize(B * 32); kbs.resize(B * 32); pts3.resize(2 * B * 128); sc3.resize(2 * B * 64); third.resize(2 * B * 64);
        // 4. blinded public keys  Y_bar_i = x_i*G + b_i*B: both bases are constants of the suite -> fixed-base window tables
        constexpr bool fixed = true;            // (te_msm_groups, the variable-base launches, stays for callers with other bases)
        uint8_t gb[128];
        std::memcpy(gb, su.generator, 64);
        std::memcpy(gb + 64, su.blinding_base, 64);
        if (fixed) TRY(te_fixed_base_groups(actx, cv, gb, sc.data(), B, 2, ybar.data()));
        else TRY(te_msm_groups(actx, cv, gb_pts.data(), sc.data(), B, 2, ybar.data()));
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
        if (fixed) {
            // R_i = k_i*G + kb_i*B from the tables; O_k,i = k_i * I_i is the one variable-base multiplication left
            TRY(te_fixed_base_groups(actx, cv, gb, sc3.data(), B, 2, third.data()));
            TRY(te_scalar_mul_batch(actx, cv, inputs.data(), ks.data(), B, third.data() + 64 * B));
        } else {
            TRY(te_msm_groups(actx, cv, pts3.data(), sc3.data(), 2 * B, 2, third.data()));
        }
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
        scratch_wipe.armed = false;                          // the success path reports the wipe's own status
        return ctx_wipe_scratch(actx, true);
    }
};

int ringvrf_prove_batch_impl(dr_ring_prover* p, const dr_vrf_suite* suite, size_t batch, const uint8_t* alphas, const uint64_t* alpha_off,
                                    const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                                    const uint8_t* secret_scalars, const uint32_t* producer_index, const uint8_t* fs_prefix, size_t fs_prefix_len,
                                    const uint8_t* zk_random48, uint8_t* out_proofs, uint8_t* out_aux) {
    if (!p || !alpha_off || !ad_off || !secret_scalars || !producer_index || !fs_prefix || !out_proofs) return fail(DR_ERR_INVALID, "null argument");
    if (batch == 0) return DR_OK;
    if (batch > 4096) return fail(DR_ERR_INVALID, "batch must be at most 4096 per call");
    drh::VrfSuite su;
    TRY(load_suite(suite, su));
    if (su.cv->id != ring_prover_curve(p)) return fail(DR_ERR_INVALID, "VRF suite and ring prover are on different curves");
    for (size_t i = 0; i < batch; i++)
        if (alpha_off[i + 1] < alpha_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
            return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
    dr_ctx* ctx = ring_prover_ctx(p);
    TRY(use_ctx(ctx));
    const size_t B = batch;
    PhaseTrace tr_("prove_batch");

    // hidden rows of the ring proof: 12 x 48 random bytes per proof reduced mod p — needs nothing from the GPU, so it runs on this
    // thread (and the pool) while the Elligator and x*I kernels of the Pedersen head are in flight
    std::vector<uint8_t> zk;
    const std::function<void()> reduce_zk = [&] {
        if (!zk_random48) return;
        zk.resize(B * 12 * 32);
        drh::parallel_for(B * 12, [&](size_t j) {
            uint64_t v[4];
            drh::mod_p().reduce_bytes(zk_random48 + 48 * j, 48, false, v);
            drh::store_le32(v, zk.data() + 32 * j);
        });
    };
    PedersenBatch ped(su, B);
    TRY(ped.head(ctx, alphas, alpha_off, ads, ad_off, salts, salt_off, secret_scalars, tr_, &reduce_zk));
    std::vector<uint8_t>& blind = ped.blind;
    // 4.-6. the rest of the Pedersen part needs nothing from the ring proof and the ring proof needs only the blinding
    // factors: it runs on a second stream (own context: scratch + stream) from a helper thread while this thread drives
    // the ring phases.  Its kernels are latency-bound (16..64 waves) and hide under the chip-filling MSMs.
    dr_ctx* actx = nullptr;
    TRY(ring_prover_aux_ctx(p, &actx));
    actx->prof = ctx->prof;
    int ped_rc = DR_OK;
    std::string ped_err;
    std::thread ped_thread([&] { run_guarded(ped_rc, ped_err, [&] { return ped.tail(actx, out_proofs, 784, out_aux, DR_RINGVRF_AUX_BYTES); }); });
    struct WipeGuard {       // an exit before the last ring phase (which wipes on its own) still leaves no witness state in HBM
        dr_ring_prover* p;
        bool armed = true;
        ~WipeGuard() { if (armed) (void)dr_ring_prover_wipe(p); }
    } wipe_guard{p};
    struct Joiner {          // every exit path below must wait for the helper before the buffers it uses go away
        std::thread& t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{ped_thread};

    // 7. ring proof: witness columns
    std::vector<uint8_t> relation(B * 64), wit(B * 4 * 96), cq(B * 96), evals(B * 256), opens(B * 192);
    std::vector<int> wit_inf(B * 4), cq_inf(B), open_inf(B * 2);
    tr_.mark("spawn+zk");
    TRY(dr_ring_prove_witness(p, B, producer_index, blind.data(), zk_random48 ? zk.data() : nullptr, relation.data(), wit.data(), wit_inf.data()));
    tr_.mark("witness");
    // transcripts in groups of four proofs (FsTranscript4: four Keccak states per vector instruction); the last group repeats
    // proof B - 1 in its spare slots, whose (identical) results land on the same bytes
    drh::FsTranscript4 base;
    base.sh.update_same(fs_prefix, fs_prefix_len);
    const size_t G = (B + 3) / 4;
    std::vector<drh::FsTranscript4> fs(G, base);
    auto member = [&](size_t g, int k) { return std::min(4 * g + (size_t)k, B - 1); };
    std::vector<uint8_t> alphas7(B * 7 * 32), zetas(B * 32), nus(B * 8 * 32);
    drh::parallel_for(G, [&](size_t g) {
        const uint8_t* in[4];
        uint8_t* out[4];
        uint8_t ser[4][4 * 96];
        for (int k = 0; k < 4; k++) in[k] = relation.data() + 64 * member(g, k);
        fs[g].absorb_labeled("instance", in, 64);
        for (int k = 0; k < 4; k++) {
            const size_t i = member(g, k);
            for (int c = 0; c < 4; c++) drh::g1_serialized(wit.data() + 96 * (4 * i + c), wit_inf[4 * i + c], ser[k] + 96 * c);
            in[k] = ser[k];
            out[k] = alphas7.data() + 224 * i;
        }
        fs[g].absorb_labeled("committed_cols", in, 4 * 96);
        fs[g].challenges("constraints_aggregation", 7, out);
    }, 4);        // (an item is four proofs' sponges, 10 - 20 us: worth a thread from four items on)
    tr_.mark("fs1");
    TRY(dr_ring_prove_quotient(p, B, alphas7.data(), cq.data(), cq_inf.data()));
    tr_.mark("quotient");
    drh::parallel_for(G, [&](size_t g) {
        const uint8_t* in[4];
        uint8_t* out[4];
        uint8_t ser[4][96];
        for (int k = 0; k < 4; k++) {
            const size_t i = member(g, k);
            drh::g1_serialized(cq.data() + 96 * i, cq_inf[i], ser[k]);
            in[k] = ser[k];
            out[k] = zetas.data() + 32 * i;
        }
        fs[g].absorb_labeled("quotient", in, 96);
        fs[g].challenges("evaluation_point", 1, out);
    }, 4);        // (an item is four proofs' sponges, 10 - 20 us: worth a thread from four items on)
    tr_.mark("fs2");
    TRY(dr_ring_prove_evals(p, B, zetas.data(), evals.data()));
    tr_.mark("evals");
    drh::parallel_for(G, [&](size_t g) {
        const uint8_t *in[4], *in2[4];
        uint8_t* out[4];
        for (int k = 0; k < 4; k++) {
            const size_t i = member(g, k);
            in[k] = evals.data() + 256 * i;
            in2[k] = evals.data() + 256 * i + 224;
            out[k] = nus.data() + 256 * i;
        }
        fs[g].absorb_labeled("register_evaluations", in, 224);
        fs[g].absorb_labeled("shifted_linearization_evaluation", in2, 32);
        fs[g].challenges("kzg_aggregation", 8, out);
    }, 4);        // (an item is four proofs' sponges, 10 - 20 us: worth a thread from four items on)
    tr_.mark("fs3");
    TRY(dr_ring_prove_openings(p, B, nus.data(), opens.data(), open_inf.data()));
    wipe_guard.armed = false;        // (the openings phase ended with the wipe)
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
int pedersen_verify_core(dr_ctx* actx, const drh::VrfSuite& su, size_t B, const uint8_t* proofs, size_t stride,
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

// a host fold's result as dr_pairing_check takes it: affine big-endian x || y (zeros and *is_inf for the point at infinity); `negate`: -P
static void g1_pair_operand(const drh::G1& pt, bool negate, uint8_t out[96], int* is_inf) {
    drh::Fq ax, ay;
    if (!drh::g1_to_affine(pt, ax, ay)) { std::memset(out, 0, 96); *is_inf = 1; return; }
    ax.store_be(out);
    (negate ? ay.neg() : ay).store_be(out + 48);
    *is_inf = 0;
}

int ringvrf_verify_batch_impl(dr_ctx* ctx, const dr_vrf_suite* suite, const dr_ring_verifier_key* vk, size_t batch, const uint8_t* proofs,
                                     const uint8_t* inputs, const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
                                     const uint64_t* salt_off, const uint8_t seed32[32], int* ok) {
    // (not use_ctx: a prover that ran on this context may have left the zeroing of its buffers on the wipe stream; the decoding phase
    //  below works in verifier-owned buffers and runs beside it, the MSM phase — the first to touch the context's scratch — joins it)
    if (!ctx) return fail(DR_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
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
    // verifier randomness: two non-zero coefficients per proof (kzg.py:84-108) — they depend on the seed alone, and they are the only scalars of
    // the rhs fold (r1, r2 on the two opening proofs): drawn here, so that that fold can start as soon as the points are decoded
    std::vector<uint8_t> rhs_sc(2 * B * 32);
    std::atomic<bool> canonical{true};
    drh::parallel_for(B, [&](size_t i) {
        const uint8_t* pr = proofs + 784 * i;
        for (int k = 0; k < 2; k++) {
            drh::Shake256 sh;
            sh.update(seed32, 32);
            uint8_t ctr[9] = {0};
            for (int j = 0; j < 8; j++) ctr[j] = (uint8_t)((uint64_t)(2 * i + k) >> (8 * j));
            sh.update(ctr, 8);
            uint8_t raw[48];
            sh.digest(raw, 48);
            uint64_t r[4];
            mp.reduce_bytes(raw, 48, true, r);
            if (mp.is_zero(r)) mp.set_u64(1, r);
            drh::store_le32(r, rhs_sc.data() + 64 * i + 32 * k);
        }
        std::memcpy(te_enc.data() + 128 * i, pr, 128);
        uint64_t v[4];
        bool ok_i = true;
        for (int k = 0; k < 2; k++) { drh::load_le32(pr + 128 + 32 * k, v); if (drh::Mod256::geq(v, mn.m)) ok_i = false; }      // dec_scalar
        const uint8_t* pl = pr + 192;
        for (int k = 0; k < 7; k++) { drh::load_le32(pl + 192 + 32 * k, v); if (drh::Mod256::geq(v, mp.m)) ok_i = false; }
        drh::load_le32(pl + 464, v); if (drh::Mod256::geq(v, mp.m)) ok_i = false;
        if (!ok_i) canonical.store(false, std::memory_order_relaxed);
        uint8_t* g = g1_enc.data() + 336 * i;
        std::memcpy(g, pl, 192);                   // C_b, C_accip, C_accx, C_accy
        std::memcpy(g + 192, pl + 416, 48);        // C_q
        std::memcpy(g + 240, pl + 496, 96);        // Phi_zeta, Phi_zeta_omega
    });
    if (!canonical.load()) return DR_OK;
    tr_.mark("gather");

    // ---- 2. GPU: decode + validate the 4B Bandersnatch points, decompress the 7B G1 points; meanwhile a helper thread
    // hashes the inputs to the curve on a second stream (all three kernels are latency-bound: a few dozen waves)
    if (!ctx->aux) TRY(ctx_create_role(ctx->device, 1, &ctx->aux));
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
        run_guarded(side_rc, side_err, [&]() -> int {
            TRY(encode_to_curve_msgs(actx, su, B, inputs, in_off, salts, salt_off, in_pts.data()));
            {
                std::unique_lock<std::mutex> lk(gate_m);
                gate_cv.wait(lk, [&] { return gate >= 0; });
                if (gate == 0) return DR_OK;
            }
            return pedersen_verify_core(actx, su, B, proofs, 784, te_xy, in_pts, ads, ad_off, ped_ok);
        });
    });
    struct Joiner {
        std::thread& t;
        std::function<void()> give_up;
        ~Joiner() { give_up(); if (t.joinable()) t.join(); }
    } joiner{side, [&] { open_gate(0); }};
    // Up to eight proofs: every launch chain below would be pure latency (0.7 - 2 ms each), so the points are decoded and the two
    // G1 folds done on the host (hostsmall.hpp: ~80 us per point on the worker pool, ~0.4 ms per fold); the Pedersen side keeps its
    // stream.  Measured at the end of round 4: 1.2 ms for one proof + 0.13 - 0.2 ms per further one (the host's load decides) against a flat 2.9 ms
    // through the kernels.
    static const size_t host_max = std::getenv("DOTRING_VERIFY_HOST_MAX") ? (size_t)std::atol(std::getenv("DOTRING_VERIFY_HOST_MAX")) : 8;
    const bool small = B <= host_max;
    Scratch &g1_bases = ctx->vfy_bases, &g1_in = ctx->vfy_in, &g1_std = ctx->vfy_std;
    std::vector<uint8_t> g1_le(7 * B * 96);
    std::vector<drh::G1AffineHost> host_bases;
    if (small) {
        std::vector<int> good(n_te + 7 * B, 0);
        auto decode_one = [&](size_t j) {
            if (j < n_te) {
                good[j] = drh::te_decode_checked(*su.cv, te_enc.data() + 32 * j, te_xy.data() + 64 * j) ? 1 : 0;
                return;
            }
            const size_t k = j - n_te;
            uint8_t be[96];
            int inf = 0;
            if (dr_g1_decompress(g1_enc.data() + 48 * k, be, &inf) != DR_OK) return;
            uint8_t* le = g1_le.data() + 96 * k;
            if (inf) std::memset(le, 0, 96);
            else for (int q = 0; q < 48; q++) { le[q] = be[47 - q]; le[48 + q] = be[95 - q]; }
            good[j] = 1;
        };
        // 4B + 7B decodings of 35 - 80 us each, one per item of the worker pool (its threads are already there: starting seven of our
        // own cost more than a decoding)
        drh::parallel_for(n_te + 7 * B, decode_one, 1);
        for (int g : good) if (!g) return DR_OK;                     // malformed / invalid point: ok = 0
        host_bases.resize(n_g1);
        for (size_t k = 0; k < 7 * B; k++) {
            const uint8_t* le = g1_le.data() + 96 * k;
            bool inf = true;
            for (int q = 0; q < 96; q++) if (le[q]) { inf = false; break; }
            host_bases[k].inf = inf;
            if (!inf && (!drh::Fq::load_le(host_bases[k].x, le) || !drh::Fq::load_le(host_bases[k].y, le + 48))) return DR_OK;
        }
        for (int k = 0; k < 4; k++) {
            const uint8_t* be = k < 3 ? vk->fixed_commitments + 96 * k : vk->g1_generator;
            drh::G1AffineHost& hb = host_bases[7 * B + k];
            hb.inf = (be[0] & 0x40) != 0;
            if (!hb.inf) {
                if (be[0] & 0xe0) return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
                if (!drh::Fq::load_be(hb.x, be) || !drh::Fq::load_be(hb.y, be + 48) || !drh::g1_on_curve(hb.x, hb.y))
                    return fail(DR_ERR_INVALID, "invalid BLS12-381 G1 encoding");
            }
        }
    } else {
        Scratch &te_in = ctx->vfy_te_in, &te_out = ctx->vfy_te_out, &vflags = ctx->vfy_flags;
        TRY(te_in.reserve(n_te * 32));
        TRY(te_out.reserve(n_te * 64));
        TRY(vflags.reserve(n_te * 4 + n_g1 * 4));
        hipStream_t st = ctx->stream;
        // the third stream (G1 decompression, below) writes its verdicts behind this stream's: it starts when this stream is done with
        // whatever an earlier call left in it, but not behind the decoding kernel launched next
        hipEvent_t scratch_ready;
        HIP_TRY(hipEventCreateWithFlags(&scratch_ready, hipEventDisableTiming));
        struct EventGuard { hipEvent_t e; ~EventGuard() { (void)hipEventDestroy(e); } } scratch_ready_guard{scratch_ready};
        HIP_TRY(hipEventRecord(scratch_ready, st));
        HIP_TRY(hipMemcpyAsync(te_in.p, te_enc.data(), n_te * 32, hipMemcpyHostToDevice, st));
        uint32_t* d_ok = vflags.as<uint32_t>();
        TRY(launch(ctx, "k_bsn_decode_points", [&] {
            launch_decode_points(ctx, st, su.cv->id, false, te_in.as<uint32_t>(), te_out.as<uint32_t>(), d_ok, n_te);
        }));
        std::vector<uint32_t> flags(n_te + n_g1);
        // G1: bases buffer = 7B decompressed points followed by C_px, C_py, C_s and G1[0]
        TRY(g1_bases.reserve(n_g1 * 96));
        TRY(g1_in.reserve(7 * B * 48));
        TRY(g1_std.reserve(n_g1 * 96));
        // the G1 decompression (a 380-squaring chain on ~100 waves) runs next to the Bandersnatch decoding (~1 ms on 128 waves) on
        // the third stream; an event brings it back into this stream before anything reads the bases
        if (!ctx->aux2) TRY(ctx_create_role(ctx->device, 2, &ctx->aux2));
        dr_ctx* dctx = ctx->aux2;
        hipStream_t st2 = dctx->stream;
        ctx->aux2->prof = ctx->prof;
        if (st2 != st) HIP_TRY(hipStreamWaitEvent(st2, scratch_ready, 0));
        HIP_TRY(hipMemcpyAsync(g1_in.p, g1_enc.data(), 7 * B * 48, hipMemcpyHostToDevice, st2));
        TRY(launch(dctx, "k_g1_decompress", [&] {
            g1_launch_decompress(st2, g1_in.as<uint8_t>(), g1_bases.as<uint32_t>(), d_ok + n_te, 7 * B);
        }));
        // only now the copy back of the decoded Bandersnatch points: into pageable memory it blocks this thread until the decoding
        // kernel is done, and issued before the G1 launch (as it was until round 3) it kept the two decoders from running side by
        // side — 1.06 + 0.96 ms in a row instead of 1.06
        HIP_TRY(hipMemcpyAsync(te_xy.data(), te_out.p, n_te * 64, hipMemcpyDeviceToHost, st));
        {
            hipEvent_t done;
            HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            hipError_t e1 = hipEventRecord(done, st2), e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, done, 0) : e1;
            (void)hipEventDestroy(done);
            HIP_TRY(e2);
        }
        {
            uint8_t tail_be[4 * 96];
            std::memcpy(tail_be, vk->fixed_commitments, 3 * 96);
            std::memcpy(tail_be + 288, vk->g1_generator, 96);
            for (int k = 0; k < 3; k++) if (tail_be[96 * k] & 0x40) std::memset(tail_be + 96 * k, 0, 96);       // serialised infinity
            std::vector<uint8_t> le;
            TRY(g1_be_to_le_limbs(tail_be, 4, le, true));
            HIP_TRY(hipMemcpyAsync(g1_bases.as<uint32_t>() + 7 * B * 24, le.data(), 4 * 96, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));          // `le` is a stack-lifetime staging buffer
            tr_.mark("decode kernels");
            g1_launch_bases_to_mont(st, g1_bases.as<uint32_t>() + 7 * B * 24, 4);
        }
        g1_launch_bases_from_mont(st, g1_bases.as<uint32_t>(), g1_std.as<uint32_t>(), 7 * B);
        HIP_TRY(hipMemcpyAsync(g1_le.data(), g1_std.p, 7 * B * 96, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(flags.data(), d_ok, (n_te + 7 * B) * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i < n_te + 7 * B; i++) if (!flags[i]) return DR_OK;                  // malformed / invalid point: ok = 0
    }
    tr_.mark("decode");

    // ---- 3a. the rhs fold — r1, r2 on the two opening proofs of every proof, zero scalars (no digits, no cost) on everything else — needs
    // nothing from the transcripts: it starts now, on the third stream from a helper thread, and runs under the two host passes below;
    // the thread then takes the Miller loop of its pair (-rhs, G2[1]) as well, so that after the lhs fold only one loop, one product in
    // Fq12 and the final exponentiation are left (the two loops' product is what the joint loop of dr_pairing_check computes).
    std::vector<uint8_t> rhs_full(n_g1 * 32, 0);
    for (size_t i = 0; i < B; i++) std::memcpy(rhs_full.data() + 224 * i + 160, rhs_sc.data() + 64 * i, 64);
    uint8_t pair_g1[2 * 96];
    int pair_inf[2] = {0, 0};
    drh::Fq12 f_rhs = drh::Fq12::one();
    drh::G1 rhs_pt = drh::G1::inf();             // (the host path's rhs fold, one or two proofs)
    int rhs_rc = DR_OK;
    std::string rhs_err;
    std::thread rhs_thread;
    struct RhsJoiner {               // (declared after everything the thread touches: joined before any of it goes away)
        std::thread& t;
        ~RhsJoiner() { if (t.joinable()) t.join(); }
    } rhs_joiner{rhs_thread};
    if (small) {
        // one or two proofs: the same on the host (hostsmall.hpp: ~0.3 ms for the 2B live terms)
        rhs_thread = std::thread([&] {
            run_guarded(rhs_rc, rhs_err, [&]() -> int {
                rhs_pt = drh::g1_msm_small(host_bases.data(), rhs_full.data(), n_g1, 1);
                g1_pair_operand(rhs_pt, true, pair_g1 + 96, pair_inf + 1);
                return pairing_miller(pair_g1 + 96, vk->g2 + 192, 1, f_rhs);
            });
        });
    } else {
        if (!ctx->aux2) TRY(ctx_create_role(ctx->device, 2, &ctx->aux2));
        dr_ctx* bctx = ctx->aux2;
        bctx->prof = ctx->prof;
        rhs_thread = std::thread([&, bctx] {
            run_guarded(rhs_rc, rhs_err, [&]() -> int {
                TRY(use_ctx(bctx));
                TRY(bctx->scalars.reserve(n_g1 * 32));
                HIP_TRY(hipMemcpyAsync(bctx->scalars.p, rhs_full.data(), n_g1 * 32, hipMemcpyHostToDevice, bctx->stream));
                TRY(msm_to_bytes(bctx, g1_bases.as<uint32_t>(), bctx->scalars.as<uint32_t>(), n_g1, 1, pair_g1 + 96, pair_inf + 1));
                // (a vanishing rhs can only come from r1 = r2 = 0 or infinity openings: its pair then contributes 1 and the equation demands lhs = O)
                if (!pair_inf[1]) {                                              // e(lhs, G2[0]) * e(-rhs, G2[1]) == 1
                    drh::Fq y;
                    if (!drh::Fq::load_be(y, pair_g1 + 144)) return fail(DR_ERR_DEVICE, "MSM result out of range");
                    y.neg().store_be(pair_g1 + 144);
                }
                return pairing_miller(pair_g1 + 96, vk->g2 + 192, 1, f_rhs);
            });
        });
    }

    // ---- 3. (the Pedersen helper may go on only once the two host passes below have their slices of the worker pool: its challenge hashing
    //          queued first would be drained first — the pool serves the oldest job — and the critical path would wait behind it)

    // ---- 4. ring proofs: transcript replay + verifier scalar pass per proof, random linear combination of all claims
    drh::RingVerifierDomain dm;
    dm.init(vk->log2n, vk->omega_n, vk->seed_xy);
    drh::FsTranscript4 base;
    base.sh.update_same(vk->fs_prefix, vk->fs_prefix_len);
    std::vector<uint8_t> lhs_sc(n_g1 * 32);
    std::vector<uint64_t> fixed_part(B * 16);           // per proof: r1*nu0, r1*nu1, r1*nu2, r1*agg + r2*l_zw
    std::vector<int> bad(B, 0);
    auto be_rec = [&](size_t idx, uint8_t out[96]) {     // device LE limbs -> serialize() form
        const uint8_t* s = g1_le.data() + 96 * idx;
        bool inf = true;
        for (int j = 0; j < 96; j++) if (s[j]) { inf = false; break; }
        if (inf) { std::memset(out, 0, 96); out[0] = 0x40; return; }
        for (int j = 0; j < 48; j++) { out[j] = s[47 - j]; out[48 + j] = s[95 - j]; }
    };
    // transcript replay, four proofs per sponge group (the spare slots of the last group repeat proof B - 1)
    std::vector<uint8_t> al_all(B * 7 * 32), zeta_all(B * 32), nus_all(B * 8 * 32);
    drh::parallel_for((B + 3) / 4, [&](size_t g) {
        drh::FsTranscript4 t = base;
        size_t idx[4];
        const uint8_t *in[4], *in2[4];
        uint8_t* out[4];
        uint8_t ser[4][4 * 96];
        for (int k = 0; k < 4; k++) {
            idx[k] = std::min(4 * g + (size_t)k, B - 1);
            in[k] = te_xy.data() + 64 * (4 * idx[k] + 1);                     // blinded public key
        }
        t.absorb_labeled("instance", in, 64);
        for (int k = 0; k < 4; k++) {
            for (int c = 0; c < 4; c++) be_rec(7 * idx[k] + c, ser[k] + 96 * c);
            in[k] = ser[k];
            out[k] = al_all.data() + 224 * idx[k];
        }
        t.absorb_labeled("committed_cols", in, 4 * 96);
        t.challenges("constraints_aggregation", 7, out);
        for (int k = 0; k < 4; k++) {
            be_rec(7 * idx[k] + 4, ser[k]);
            out[k] = zeta_all.data() + 32 * idx[k];
        }
        t.absorb_labeled("quotient", in, 96);
        t.challenges("evaluation_point", 1, out);
        for (int k = 0; k < 4; k++) {
            const uint8_t* pl = proofs + 784 * idx[k] + 192;
            in[k] = pl + 192;
            in2[k] = pl + 464;
            out[k] = nus_all.data() + 256 * idx[k];
        }
        t.absorb_labeled("register_evaluations", in, 224);
        t.absorb_labeled("shifted_linearization_evaluation", in2, 32);
        t.challenges("kzg_aggregation", 8, out);
    }, 4);        // (an item is four proofs' sponges, 10 - 20 us: worth a thread from four items on)
    tr_.mark("replay");
    // Every proof needs two field inversions (seed + relation in affine form; the Lagrange / vanishing denominators at zeta) — at 7 us
    // each they were two thirds of this pass.  Sixteen proofs share ONE (Montgomery's trick, drh::batch_inv): both halves of the
    // arithmetic are split around their inversion.
    constexpr size_t VS = 16;
    drh::parallel_for((B + VS - 1) / VS, [&](size_t slice) {
      const size_t lo = slice * VS, hi = std::min(B, lo + VS);
      drh::TeAddPending ta[VS];
      drh::RingTermsPending rt[VS];
      uint64_t den[2 * VS][4];
      for (size_t i = lo; i < hi; i++) {
          const size_t k = i - lo;
          drh::te_add_affine_prep(*su.cv, vk->seed_xy, te_xy.data() + 64 * (4 * i + 1), ta[k]);
          std::memcpy(den[2 * k], ta[k].den, 32);
          if (drh::ring_verifier_terms_prep(dm, zeta_all.data() + 32 * i, rt[k])) std::memcpy(den[2 * k + 1], rt[k].prod, 32);
          else { bad[i] = 1; std::memset(den[2 * k + 1], 0, 32); }
      }
      drh::batch_inv(mp, den, 2 * (hi - lo));
      for (size_t i = lo; i < hi; i++) {
        if (bad[i]) continue;
        const size_t k = i - lo;
        const uint8_t* pr = proofs + 784 * i;
        const uint8_t* pl = pr + 192;
        uint8_t result_seed[64];
        const uint8_t *al = al_all.data() + 224 * i, *zeta = zeta_all.data() + 32 * i, *nus = nus_all.data() + 256 * i;
        drh::te_add_affine_finish(ta[k], den[2 * k], result_seed);          // seed + blinded public key
        drh::RingClaimScalars cl;
        drh::ring_verifier_terms_finish(*su.cv, dm, al, nus, zeta, pl + 192, pl + 464, result_seed, rt[k], den[2 * k + 1], cl);
        uint64_t r[2][4];                                                    // this proof's verifier randomness (drawn in pass 1)
        for (int k = 0; k < 2; k++) drh::load_le32(rhs_sc.data() + 64 * i + 32 * k, r[k]);
        uint64_t v[4], w[4];
        uint8_t* L = lhs_sc.data() + 224 * i;
        mp.mul(r[0], cl.nus[3], v); drh::store_le32(v, L);                                                      // C_b
        mp.mul(r[0], cl.nus[4], v); mp.mul(r[1], cl.k_ip, w); mp.add(v, w, v); drh::store_le32(v, L + 32);       // C_accip
        mp.mul(r[0], cl.nus[5], v); mp.mul(r[1], cl.k_x, w); mp.add(v, w, v); drh::store_le32(v, L + 64);        // C_accx
        mp.mul(r[0], cl.nus[6], v); mp.mul(r[1], cl.k_y, w); mp.add(v, w, v); drh::store_le32(v, L + 96);        // C_accy
        mp.mul(r[0], cl.nus[7], v); drh::store_le32(v, L + 128);                                                 // C_q
        mp.mul(r[0], cl.zeta, v); drh::store_le32(v, L + 160);                                                   // Phi_zeta
        mp.mul(r[1], cl.zeta_omega, v); drh::store_le32(v, L + 192);                                             // Phi_zeta_omega
        uint64_t* fp = &fixed_part[16 * i];
        for (int k = 0; k < 3; k++) mp.mul(r[0], cl.nus[k], fp + 4 * k);
        mp.mul(r[0], cl.agg_zeta, v); mp.mul(r[1], cl.l_zw, w); mp.add(v, w, fp + 12);
      }
    }, 1);
    open_gate(1);                // Pedersen part: te_xy has been complete since the decode; its hashing and MSM run beside the G1 folds and the pairing
    for (size_t i = 0; i < B; i++) if (bad[i]) return DR_OK;
    tr_.mark("transcripts");
    {
        uint64_t acc[4][4] = {{0}};
        for (size_t i = 0; i < B; i++)
            for (int k = 0; k < 4; k++) mp.add(acc[k], &fixed_part[16 * i + 4 * k], acc[k]);
        mp.neg(acc[3], acc[3]);                                                   // - sum_v on G1[0]
        for (int k = 0; k < 4; k++) drh::store_le32(acc[k], lhs_sc.data() + 224 * B + 32 * k);
    }
    // two MSMs over the decompressed bases (already resident): lhs over all 7B+4 points here, rhs (started above) over the 2B opening
    // proofs.  Two single MSMs rather than a batch of two: the final 255-doubling window combination of a single MSM runs on the host
    // (0.2 ms), a batch leaves it to one GPU lane per MSM (4 ms).
    if (small) {
        // the lhs fold on the host, three or four points per worker thread (the rhs has been running since the decode)
        const drh::G1 lhs_pt = drh::g1_msm_small(host_bases.data(), lhs_sc.data(), n_g1, (unsigned)std::min<size_t>(8, std::max<size_t>(2, n_g1 / 3)));
        g1_pair_operand(lhs_pt, false, pair_g1, pair_inf);
    } else {
        TRY(use_ctx(ctx));           // from here on the context's scratch is used: behind a pending wipe of it
        TRY(ctx->scalars.reserve(n_g1 * 32));
        HIP_TRY(hipMemcpyAsync(ctx->scalars.p, lhs_sc.data(), n_g1 * 32, hipMemcpyHostToDevice, st));
        TRY(msm_to_bytes(ctx, g1_bases.as<uint32_t>(), ctx->scalars.as<uint32_t>(), n_g1, 1, pair_g1, pair_inf));
    }
    tr_.mark("g1 msms");
    // e(lhs, G2[0]) * e(-rhs, G2[1]) == 1: the second loop comes with the rhs fold — joined only now: with one proof that thread (fold +
    // loop, ~0.45 ms from the decode on) ends after this one's lhs fold, and this loop needs nothing from it
    drh::Fq12 f_lhs;
    TRY(pairing_miller(pair_g1, vk->g2, 1, f_lhs));
    rhs_thread.join();
    if (rhs_rc != DR_OK) return fail(rhs_rc, rhs_err.empty() ? "rhs fold failed" : rhs_err);
    const int pok = pairing_product_is_one(f_lhs * f_rhs) ? 1 : 0;
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
        if (drh::small_host_serves(su, batch)) {
            // a handful of proofs: the whole protocol on host cores (hostsigma.hpp), one proof per worker thread
            const auto tables = drh::te_suite_tables(su);
            if (!tables) return fail(DR_ERR_INVALID, "suite base point out of range");
            std::vector<int> rcs(batch, 0);
            drh::parallel_for(batch, [&](size_t i) {
                rcs[i] = drh::pedersen_prove_one(su, *tables, drh::span_of(alphas, alpha_off, i), drh::span_of(ads, ad_off, i),
                                                 drh::span_of(salts, salt_off, i), secret_scalars + 32 * i, out_proofs + 192 * i,
                                                 out_aux ? out_aux + DR_PEDERSEN_AUX_BYTES * i : nullptr);
            }, 1);
            tr_.mark("host");
            for (size_t i = 0; i < batch; i++)
                if (rcs[i]) return fail(rcs[i] == 2 ? DR_ERR_INVALID : DR_ERR_DEVICE, rcs[i] == 2 ? "nonce scalar is zero" : "hash to curve on the host: field element out of range");
            return DR_OK;
        }
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
        if (drh::small_host_serves(su, B)) {
            // a handful of proofs: decoding, hash-to-curve and both relations of every proof on host cores (hostsigma.hpp)
            const auto tables = drh::te_suite_tables(su);
            if (!tables) return fail(DR_ERR_INVALID, "suite base point out of range");
            std::vector<int> verdict(B, 0);
            drh::parallel_for(B, [&](size_t i) {
                verdict[i] = drh::pedersen_verify_one(su, *tables, proofs + 192 * i, drh::span_of(inputs, in_off, i), drh::span_of(ads, ad_off, i),
                                                      drh::span_of(salts, salt_off, i), B <= 2);
            }, 1);
            int all = 1;
            for (size_t i = 0; i < B; i++) all &= verdict[i] == drh::SIGMA_OK ? 1 : 0;
            *ok = all;
            return DR_OK;
        }
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
        if (drh::small_host_serves(su, B)) {
            // a handful of proofs: the whole protocol on host cores (hostsigma.hpp), one proof per worker thread
            const auto tables = drh::te_suite_tables(su);
            if (!tables) return fail(DR_ERR_INVALID, "suite base point out of range");
            std::vector<int> rcs(B, 0);
            drh::parallel_for(B, [&](size_t i) {
                rcs[i] = drh::ietf_prove_one(su, *tables, thin != 0, drh::span_of(alphas, alpha_off, i), drh::span_of(ads, ad_off, i),
                                             drh::span_of(salts, salt_off, i), secret_scalars + 32 * i, out_proofs + plen * i,
                                             out_aux ? out_aux + 128 * i : nullptr);
            }, 1);
            for (size_t i = 0; i < B; i++)
                if (rcs[i]) return fail(rcs[i] == 2 ? DR_ERR_INVALID : DR_ERR_DEVICE, rcs[i] == 2 ? "nonce scalar is zero" : "hash to curve on the host: field element out of range");
            return DR_OK;
        }
        std::vector<uint8_t> xs(B * 32), inputs(B * 64), pts(2 * B * 64), sc(2 * B * 32), firsts(2 * B * 64), ks(B * 32);
        struct SecretGuard {         // secret scalars and nonces, on the host and in the context's scratch, do not outlive the call
            std::vector<uint8_t>*a, *b, *c;
            dr_ctx* ctx;
            ~SecretGuard() {
                for (std::vector<uint8_t>* v : {a, b, c})
                    if (!v->empty()) explicit_bzero(v->data(), v->size());
                (void)ctx_wipe_scratch(ctx);
            }
        } secret_guard{&xs, &sc, &ks, ctx};
        for (size_t i = 0; i < B; i++) {
            uint64_t x[4];
            mn.reduce_bytes(secret_scalars + 32 * i, 32, false, x);
            drh::store_le32(x, xs.data() + 32 * i);
        }
        TRY(encode_to_curve_msgs(ctx, su, B, alphas, alpha_off, salts, salt_off, inputs.data()));
        // pk_i = x_i G and O_i = x_i I_i in one launch
        for (size_t i = 0; i < B; i++) {
            std::memcpy(pts.data() + 64 * i, su.generator, 64);
            std::memcpy(pts.data() + 64 * (B + i), inputs.data() + 64 * i, 64);
            std::memcpy(sc.data() + 32 * i, xs.data() + 32 * i, 32);
            std::memcpy(sc.data() + 32 * (B + i), xs.data() + 32 * i, 32);
        }
        constexpr bool fixed = true;            // (te_msm_groups, the variable-base launches, stays for callers with other bases)
        if (fixed) {                    // pk = x G from the generator's window table; O = x I is variable-base
            TRY(te_fixed_base_groups(ctx, cv, su.generator, xs.data(), B, 1, firsts.data()));
            TRY(te_scalar_mul_batch(ctx, cv, inputs.data(), xs.data(), B, firsts.data() + 64 * B));
        } else {
            TRY(te_scalar_mul_batch(ctx, cv, pts.data(), sc.data(), 2 * B, firsts.data()));
        }
        const uint8_t* pks = firsts.data();
        const uint8_t* outs = firsts.data() + 64 * B;
        // transcripts, delinearisation scalar z, nonces
        std::vector<drh::Bytes> tr(B);
        std::vector<uint8_t> gpts(B * 128), gsc(B * 64);
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


// TinyVRF.verify / ThinVRF.verify (vrf/ietf/tiny.py:72-88, thin.py:96-118) for `batch` ENCODED proofs (80 / 96 bytes each), each under its
// own compressed public key: verdict[i] = 1 verifies, 0 does not, 2 the public key is not a valid point, 3 the proof is malformed (a
// point that does not decode to a prime-order point, a non-canonical scalar) — the cases the reference raises ValueError for.  Every
// proof is checked on its own (no random linear combination), one proof per worker thread, on host cores: this is the single-proof
// entry point — one proof costs ~0.6 ms here against three kernel launch chains (2.5 ms); ThinVRF.batch_verify of many proofs is the
// one MSM on the GPU (dr_te_msm).  Elligator suites of Bandersnatch only (DR_ERR_INVALID otherwise).
int dr_ietf_verify_batch(dr_ctx* ctx, const dr_vrf_suite* suite, int thin, size_t batch, const uint8_t* proofs, const uint8_t* public_keys,
                         const uint8_t* inputs, const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts,
                         const uint64_t* salt_off, uint8_t* verdict) {
    try {
        if (!ctx) return fail(DR_ERR_INVALID, "null context");
        if (batch == 0) return DR_OK;
        if (!proofs || !public_keys || !in_off || !ad_off || !verdict) return fail(DR_ERR_INVALID, "null argument");
        drh::VrfSuite su;
        TRY(load_suite(suite, su));
        if (su.cv->id != 0 || su.cv->tai) return fail(DR_ERR_INVALID, "dr_ietf_verify_batch serves the Elligator suites of Bandersnatch");
        for (size_t i = 0; i < batch; i++)
            if (in_off[i + 1] < in_off[i] || ad_off[i + 1] < ad_off[i] || (salt_off && salt_off[i + 1] < salt_off[i]))
                return fail(DR_ERR_INVALID, "offsets must be non-decreasing");
        const auto tables = drh::te_suite_tables(su);
        if (!tables) return fail(DR_ERR_INVALID, "suite base point out of range");
        const size_t plen = thin ? 96 : 80;
        drh::parallel_for(batch, [&](size_t i) {
            verdict[i] = (uint8_t)drh::ietf_verify_one(su, *tables, thin != 0, proofs + plen * i, public_keys + 32 * i, drh::span_of(inputs, in_off, i),
                                                       drh::span_of(ads, ad_off, i), drh::span_of(salts, salt_off, i), batch <= 2);
        }, 1);
        return DR_OK;
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native verifier: ") + e.what());
    }
}

// ---- ONE batch over several GPUs of this process (SURVEY 8(b)'s additive row, 8(e) first mode; the reference's process-sharded bench,
// tests/benchmark/bench_ring_proof.py:168-182, hands index ranges out and takes every result back the same way).  Proofs are
// independent units: device g takes proofs [g B / G, (g + 1) B / G) — sizes differ by at most one, empty shards are fine — on a host
// thread of its own, and writes its 784-byte proofs straight into the caller's buffer, so the "gather" is the memory the caller
// already holds.  No collective, nothing exchanged between the devices.
namespace {
struct Shard {
    size_t lo, hi;
};
Shard shard_of(size_t n, size_t g, size_t G) {
    const size_t base = n / G, rem = n % G;
    const size_t lo = g * base + std::min(g, rem);
    return {lo, lo + base + (g < rem ? 1 : 0)};
}
// run f(g) for g < G on G threads (the calling thread takes shard 0); first failure by shard order wins
template <class F>
int run_shards(size_t G, F&& f) {
    std::vector<int> rc(G, DR_OK);
    std::vector<std::string> err(G);
    std::vector<std::thread> th;
    th.reserve(G);
    for (size_t g = 1; g < G; g++) th.emplace_back([&, g] { run_guarded(rc[g], err[g], [&] { return f(g); }); });
    run_guarded(rc[0], err[0], [&] { return f(0); });
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (rc[g] != DR_OK) return fail(rc[g], "device shard " + std::to_string(g) + ": " + err[g]);
    return DR_OK;
}
}  // namespace

int dr_ringvrf_prove_batch_multi(dr_ring_prover* const* provers, size_t n_provers, const dr_vrf_suite* suite, size_t batch, const uint8_t* alphas,
                                 const uint64_t* alpha_off, const uint8_t* ads, const uint64_t* ad_off, const uint8_t* salts, const uint64_t* salt_off,
                                 const uint8_t* secret_scalars, const uint32_t* producer_index, const uint8_t* fs_prefix, size_t fs_prefix_len,
                                 const uint8_t* zk_random48, uint8_t* out_proofs, uint8_t* out_aux) {
    try {
        if (!provers || n_provers == 0 || n_provers > 64) return fail(DR_ERR_INVALID, "1..64 provers");
        if (!alpha_off || !ad_off || !secret_scalars || !producer_index || !fs_prefix || !out_proofs) return fail(DR_ERR_INVALID, "null argument");
        for (size_t g = 0; g < n_provers; g++) {
            if (!provers[g]) return fail(DR_ERR_INVALID, "null prover");
            for (size_t h = 0; h < g; h++)
                if (provers[h] == provers[g] || ring_prover_ctx(provers[h]) == ring_prover_ctx(provers[g]))
                    return fail(DR_ERR_INVALID, "every shard needs a prover and a context of its own");
        }
        if (batch == 0) return DR_OK;
        return run_shards(n_provers, [&](size_t g) -> int {
            const Shard sh = shard_of(batch, g, n_provers);
            for (size_t lo = sh.lo; lo < sh.hi; lo += 4096) {                 // (one native call proves at most 4096 proofs)
                const size_t n = std::min<size_t>(4096, sh.hi - lo);
                TRY(ringvrf_prove_batch_impl(provers[g], suite, n, alphas, alpha_off + lo, ads, ad_off + lo, salts, salt_off ? salt_off + lo : nullptr,
                                             secret_scalars + 32 * lo, producer_index + lo, fs_prefix, fs_prefix_len,
                                             zk_random48 ? zk_random48 + 576 * lo : nullptr, out_proofs + 784 * lo,
                                             out_aux ? out_aux + DR_RINGVRF_AUX_BYTES * lo : nullptr));
            }
            return DR_OK;
        });
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native prover: ") + e.what());
    }
}

int dr_ringvrf_verify_batch_multi(dr_ctx* const* ctxs, size_t n_ctxs, const dr_vrf_suite* suite, const dr_ring_verifier_key* vk, size_t batch,
                                  const uint8_t* proofs, const uint8_t* inputs, const uint64_t* in_off, const uint8_t* ads, const uint64_t* ad_off,
                                  const uint8_t* salts, const uint64_t* salt_off, const uint8_t seed32[32], int* ok) {
    try {
        if (!ctxs || n_ctxs == 0 || n_ctxs > 64) return fail(DR_ERR_INVALID, "1..64 contexts");
        if (!vk || !proofs || !in_off || !ad_off || !seed32 || !ok) return fail(DR_ERR_INVALID, "null argument");
        for (size_t g = 0; g < n_ctxs; g++) {
            if (!ctxs[g]) return fail(DR_ERR_INVALID, "null context");
            for (size_t h = 0; h < g; h++)
                if (ctxs[h] == ctxs[g]) return fail(DR_ERR_INVALID, "every shard needs a context of its own");
        }
        *ok = 0;
        if (batch == 0) { *ok = 1; return DR_OK; }
        std::vector<int> verdict(n_ctxs, 1);
        TRY(run_shards(n_ctxs, [&](size_t g) -> int {
            const Shard sh = shard_of(batch, g, n_ctxs);
            for (size_t lo = sh.lo; lo < sh.hi && verdict[g]; lo += 4096) {
                const size_t n = std::min<size_t>(4096, sh.hi - lo);
                // every shard (and chunk) folds its claims with randomness of its own: SHAKE256(seed || LE64(first proof))
                uint8_t mix[40], sub[32];
                std::memcpy(mix, seed32, 32);
                for (int k = 0; k < 8; k++) mix[32 + k] = (uint8_t)((uint64_t)lo >> (8 * k));
                drh::Shake256 sh256;
                sh256.update(mix, 40);
                sh256.digest(sub, 32);
                int one = 0;
                TRY(ringvrf_verify_batch_impl(ctxs[g], suite, vk, n, proofs + 784 * lo, inputs, in_off + lo, ads, ad_off + lo, salts,
                                              salt_off ? salt_off + lo : nullptr, sub, &one));
                verdict[g] = one;
            }
            return DR_OK;
        }));
        int all = 1;
        for (size_t g = 0; g < n_ctxs; g++) all &= verdict[g];
        *ok = all;
        return DR_OK;
    } catch (const std::bad_alloc&) {
        return fail(DR_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DR_ERR_DEVICE, std::string("native verifier: ") + e.what());
    }
}
