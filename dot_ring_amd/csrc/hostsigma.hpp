// The sigma protocols of the small VRFs for a HANDFUL of proofs, entirely on host cores.
//
// One Tiny / Thin / Pedersen proof through the kernels is three dependent launch chains (hash-to-curve, the x * I and fixed-base
// multiplications, the nonce multiplications): 2.4 - 2.5 ms whatever the batch, slower than the reference's interpreted CPU path
// (2.2 ms prove, 2.0 ms verify, docs/BENCHMARK.md:20-47).  One x86 core does the same arithmetic in 0.3 - 0.6 ms, sixteen of them
// 64 proofs in about that time; so calls of up to DOTRING_SMALL_HOST_MAX proofs (default 64; 0 = always the kernels) on the Elligator
// suites of Bandersnatch run here, every larger batch on the GPU.  Both routes produce the same bytes and verdicts — the tests run the
// reference's vectors through each.
//
//   ietf_prove_one / ietf_verify_one           TinyVRF / ThinVRF  (dot_ring/vrf/ietf/tiny.py:53-88, thin.py:38-152)
//   pedersen_prove_one / pedersen_verify_one   PedersenVRF        (dot_ring/vrf/pedersen/vrf.py:86-169)
//
// Secret scalars (x, the blinding factor, the nonces) only meet te_mul_secret / te_mul_fixed(secret = true): fixed schedules, table
// entries picked by mask.  The verifier's combinations are public and use Straus with indexed tables.
#pragma once
#include "hostsmall.hpp"

namespace drh {

// the Elligator suites of Bandersnatch (SHA-512 and SHAKE128); JubJub hashes to the curve by try-and-increment and stays on the kernels
inline bool small_host_serves(const VrfSuite& su, size_t B) { return su.cv->id == 0 && !su.cv->tai && B > 0 && B <= small_host_max(); }

struct Span {
    const uint8_t* p;
    size_t n;
};
inline Span span_of(const uint8_t* blob, const uint64_t* off, size_t i) {
    return off ? Span{blob + off[i], (size_t)(off[i + 1] - off[i])} : Span{nullptr, 0};
}

// encode_to_curve(salt || msg) -> affine x || y  (curve.py:110-185 + te_affine_point.py:212-295)
inline bool sigma_encode_to_curve(const VrfSuite& su, Span salt, Span msg, uint8_t out_xy[64]) {
    Bytes m;
    if (salt.n) put(m, salt.p, salt.n);
    if (msg.n) put(m, msg.p, msg.n);
    uint8_t u[64];
    hash_to_field2(su, m.data(), m.size(), u);
    return te_encode_to_curve_host(u, out_xy);
}

// transcript of the IETF schemes over the pairs (G, pk), (I, O), then the delinearisation scalar z (primitives.py:26-55,92-122)
inline void ietf_transcript(const VrfSuite& su, bool thin, const uint8_t enc_g[32], const uint8_t enc_pk[32], const uint8_t enc_i[32],
                            const uint8_t enc_o[32], Span ad, Bytes& t, uint64_t z[4]) {
    t = su.suite_id;
    put8(t, thin ? 0x01 : 0x00);
    put_le64(t, 2);
    put(t, enc_g, 32);
    put(t, enc_pk, 32);
    put(t, enc_i, 32);
    put(t, enc_o, 32);
    put_le64(t, ad.n);
    if (ad.n) put(t, ad.p, ad.n);
    Bytes d = t;
    put8(d, 0x30);                                         // DELINEARIZE
    uint8_t raw[16];
    vrf_squeeze(su.xof, d.data(), d.size(), raw, 16);
    su.cv->n.reduce_bytes(raw, 16, false, z);
}

// TinyVRF.prove / ThinVRF.prove of one proof.  out: 80 (O || c || s) or 96 (O || R || s) bytes; aux (nullable): O, R affine.
inline int ietf_prove_one(const VrfSuite& su, const TeSuiteTables& tb, bool thin, Span alpha, Span ad, Span salt, const uint8_t secret[32],
                          uint8_t* out, uint8_t* aux) {
    const Mod256& mn = su.cv->n;
    uint64_t x[4], k[4], z[4], c[4], s[4];
    mn.reduce_bytes(secret, 32, false, x);
    uint8_t in_xy[64], o_xy[64], pk_xy[64], r_xy[64];
    if (!sigma_encode_to_curve(su, salt, alpha, in_xy)) return 1;
    TeExt I;
    if (!te_load_affine(in_xy, I)) return 1;
    TeExt O = te_mul_secret(I, x, tb.c);
    te_store_affine(O, o_xy);
    te_store_affine(te_mul_fixed(tb.tg, x, tb.c, true), pk_xy);
    uint8_t enc_g[32], enc_pk[32], enc_i[32], enc_o[32], enc_r[32];
    enc_te_point(su.generator, enc_g);
    enc_te_point(pk_xy, enc_pk);
    enc_te_point(in_xy, enc_i);
    enc_te_point(o_xy, enc_o);
    Bytes t;
    ietf_transcript(su, thin, enc_g, enc_pk, enc_i, enc_o, ad, t, z);
    int rc = 0;
    if (!vrf_nonce(su, t, x, k)) rc = 2;
    if (rc == 0) {
        const uint64_t zs[1][4] = {{z[0], z[1], z[2], z[3]}};
        const TeExt M = te_add(tb.g, te_msm_public(&I, zs, 1, tb.c), tb.c);      // G + z I (public)
        TeExt R = te_mul_secret(M, k, tb.c);
        te_store_affine(R, r_xy);
        enc_te_point(r_xy, enc_r);
        vrf_challenge(su, t, enc_r, 1, c);
        mn.mul(c, x, s);
        mn.add(s, k, s);
        std::memcpy(out, enc_o, 32);
        if (thin) {
            std::memcpy(out + 32, enc_r, 32);
            store_le32(s, out + 64);
        } else {
            uint8_t cb[32];
            store_le32(c, cb);
            std::memcpy(out + 32, cb, 16);
            store_le32(s, out + 48);
        }
        if (aux) {
            std::memcpy(aux, o_xy, 64);
            std::memcpy(aux + 64, r_xy, 64);
        }
        explicit_bzero(&R, sizeof R);
    }
    explicit_bzero(x, sizeof x);
    explicit_bzero(k, sizeof k);
    explicit_bzero(s, sizeof s);
    explicit_bzero(&O, sizeof O);
    return rc;
}

// verdicts of the one-proof verifiers
enum { SIGMA_OK = 1, SIGMA_REJECT = 0, SIGMA_BAD_PUBLIC_KEY = 2, SIGMA_BAD_PROOF = 3 };

// TinyVRF.verify / ThinVRF.verify of one ENCODED proof (80 / 96 bytes) under a compressed public key.  Every point is decoded and
// validated here (canonical y, on the curve, not the identity, prime-order subgroup), scalars must be canonical.
//   R' = s (G + z I) - c (pk + z O) = s G + (s z) I - c pk - (c z) O ;  Tiny: c == challenge(R'),  Thin: R' == R with c = challenge(R)
// `fan_out`: the independent first steps (point decodings with their subgroup checks, the hash to the curve: ~0.1 ms each) go to
// the worker pool — for a call with one or two proofs; a larger batch already has a proof per thread.
inline int ietf_verify_one(const VrfSuite& su, const TeSuiteTables& tb, bool thin, const uint8_t* proof, const uint8_t pk_enc[32], Span input,
                           Span ad, Span salt, bool fan_out = false) {
    const Mod256& mn = su.cv->n;
    uint8_t pk_xy[64], o_xy[64], r_xy[64], in_xy[64];
    bool good[4] = {false, false, true, false};
    const auto first = [&](size_t j) {
        if (j == 0) good[0] = te_decode_checked(*su.cv, pk_enc, pk_xy);
        else if (j == 1) good[1] = te_decode_checked(*su.cv, proof, o_xy);
        else if (j == 2) good[2] = !thin || te_decode_checked(*su.cv, proof + 32, r_xy);
        else good[3] = sigma_encode_to_curve(su, salt, input, in_xy);
    };
    if (fan_out) parallel_for(4, first, 1);
    else for (size_t j = 0; j < 4; j++) first(j);
    if (!good[0]) return SIGMA_BAD_PUBLIC_KEY;
    if (!good[1] || !good[2]) return SIGMA_BAD_PROOF;
    uint64_t c[4] = {0, 0, 0, 0}, s[4], z[4];
    if (thin) {
        load_le32(proof + 64, s);
    } else {
        uint8_t cb[32] = {0};
        std::memcpy(cb, proof + 32, 16);
        load_le32(cb, c);                                   // 128 bits: below n
        load_le32(proof + 48, s);
    }
    if (Mod256::geq(s, mn.m)) return SIGMA_BAD_PROOF;         // dec_scalar (codec.py:9-34)
    if (!good[3]) return SIGMA_REJECT;
    uint8_t enc_g[32], enc_i[32];
    enc_te_point(su.generator, enc_g);
    enc_te_point(in_xy, enc_i);
    Bytes t;
    ietf_transcript(su, thin, enc_g, pk_enc, enc_i, proof, ad, t, z);      // (a validated encoding is the canonical one: enc(dec(e)) == e)
    if (thin) vrf_challenge(su, t, proof + 32, 1, c);
    TeExt pts[3];
    uint64_t ks[3][4];
    if (!te_load_affine(in_xy, pts[0]) || !te_load_affine(pk_xy, pts[1]) || !te_load_affine(o_xy, pts[2])) return SIGMA_REJECT;
    mn.mul(s, z, ks[0]);
    mn.neg(c, ks[1]);
    mn.mul(c, z, ks[2]);
    mn.neg(ks[2], ks[2]);
    const TeExt rp = te_add(te_mul_fixed(tb.tg, s, tb.c, false), te_msm_public(pts, ks, 3, tb.c), tb.c);
    if (thin) {
        TeExt R;
        return te_load_affine(r_xy, R) && te_equal(rp, R) ? SIGMA_OK : SIGMA_REJECT;
    }
    uint8_t rp_xy[64], enc_rp[32];
    te_store_affine(rp, rp_xy);
    enc_te_point(rp_xy, enc_rp);
    uint64_t c2[4];
    vrf_challenge(su, t, enc_rp, 1, c2);
    return Mod256::eq(c, c2) ? SIGMA_OK : SIGMA_REJECT;
}

// transcript of the Pedersen scheme after the (I, O) pair (pedersen/vrf.py:86-104)
inline void pedersen_transcript(const VrfSuite& su, const uint8_t enc_i[32], const uint8_t enc_o[32], Span ad, Bytes& t) {
    t = su.suite_id;
    put8(t, 0x02);                                         // PEDERSEN_VRF
    put_le64(t, 1);
    put(t, enc_i, 32);
    put(t, enc_o, 32);
    put_le64(t, ad.n);
    if (ad.n) put(t, ad.p, ad.n);
}

// PedersenVRF.prove of one proof: 192 bytes O || Y_bar || R || O_k || s || s_b; aux (nullable, DR_PEDERSEN_AUX_BYTES): the four
// affine points and the blinding factor
inline int pedersen_prove_one(const VrfSuite& su, const TeSuiteTables& tb, Span alpha, Span ad, Span salt, const uint8_t secret[32], uint8_t* out,
                              uint8_t* aux) {
    const Mod256& mn = su.cv->n;
    uint64_t x[4], b[4], k[4], kb[4], c[4], s[4], sb[4];
    mn.reduce_bytes(secret, 32, false, x);
    uint8_t in_xy[64], o_xy[64], yb_xy[64], r_xy[64], ok_xy[64];
    if (!sigma_encode_to_curve(su, salt, alpha, in_xy)) return 1;
    TeExt I;
    if (!te_load_affine(in_xy, I)) return 1;
    te_store_affine(te_mul_secret(I, x, tb.c), o_xy);
    uint8_t enc_i[32];
    enc_te_point(in_xy, enc_i);
    enc_te_point(o_xy, out);
    Bytes t;
    pedersen_transcript(su, enc_i, out, ad, t);
    Bytes tbl = t;
    put8(tbl, 0x12);                                       // PEDERSEN_BLINDING
    int rc = 0;
    if (!vrf_nonce(su, tbl, x, b)) rc = 2;
    if (rc == 0) {
        te_store_affine(te_add(te_mul_fixed(tb.tg, x, tb.c, true), te_mul_fixed(tb.tb, b, tb.c, true), tb.c), yb_xy);
        enc_te_point(yb_xy, out + 32);
        put(t, out + 32, 32);
        if (!vrf_nonce(su, t, x, k) || !vrf_nonce(su, t, b, kb)) rc = 2;
    }
    if (rc == 0) {
        te_store_affine(te_add(te_mul_fixed(tb.tg, k, tb.c, true), te_mul_fixed(tb.tb, kb, tb.c, true), tb.c), r_xy);
        te_store_affine(te_mul_secret(I, k, tb.c), ok_xy);
        enc_te_point(r_xy, out + 64);
        enc_te_point(ok_xy, out + 96);
        vrf_challenge(su, t, out + 64, 2, c);
        mn.mul(c, x, s);
        mn.add(s, k, s);
        mn.mul(c, b, sb);
        mn.add(sb, kb, sb);
        store_le32(s, out + 128);
        store_le32(sb, out + 160);
        if (aux) {
            std::memcpy(aux, o_xy, 64);
            std::memcpy(aux + 64, yb_xy, 64);
            std::memcpy(aux + 128, r_xy, 64);
            std::memcpy(aux + 192, ok_xy, 64);
            store_le32(b, aux + 256);
        }
    }
    for (uint64_t* v : {x, b, k, kb, s, sb}) explicit_bzero(v, 32);
    return rc;
}

// PedersenVRF.verify of one ENCODED proof (192 bytes): every point decoded and validated, both relations checked directly
//   s I - c O == O_k   and   s G + s_b B - c Y_bar == R     (pedersen/vrf.py:128-169)
inline int pedersen_verify_one(const VrfSuite& su, const TeSuiteTables& tb, const uint8_t* proof, Span input, Span ad, Span salt,
                               bool fan_out = false) {
    const Mod256& mn = su.cv->n;
    uint8_t xy[4][64], in_xy[64];
    bool good[5] = {false, false, false, false, false};
    const auto first = [&](size_t j) {
        if (j < 4) good[j] = te_decode_checked(*su.cv, proof + 32 * j, xy[j]);
        else good[4] = sigma_encode_to_curve(su, salt, input, in_xy);
    };
    if (fan_out) parallel_for(5, first, 1);
    else for (size_t j = 0; j < 5; j++) first(j);
    if (!good[0] || !good[1] || !good[2] || !good[3]) return SIGMA_BAD_PROOF;
    uint64_t s[4], sb[4], c[4];
    load_le32(proof + 128, s);
    load_le32(proof + 160, sb);
    if (Mod256::geq(s, mn.m) || Mod256::geq(sb, mn.m)) return SIGMA_BAD_PROOF;
    if (!good[4]) return SIGMA_REJECT;
    uint8_t enc_i[32];
    enc_te_point(in_xy, enc_i);
    Bytes t;
    pedersen_transcript(su, enc_i, proof, ad, t);
    put(t, proof + 32, 32);
    vrf_challenge(su, t, proof + 64, 2, c);
    TeExt O, Yb, R, Ok, I;
    if (!te_load_affine(xy[0], O) || !te_load_affine(xy[1], Yb) || !te_load_affine(xy[2], R) || !te_load_affine(xy[3], Ok) || !te_load_affine(in_xy, I))
        return SIGMA_REJECT;
    uint64_t nc[4];
    mn.neg(c, nc);
    const TeExt p1[2] = {I, O};
    const uint64_t k1[2][4] = {{s[0], s[1], s[2], s[3]}, {nc[0], nc[1], nc[2], nc[3]}};
    if (!te_equal(te_msm_public(p1, k1, 2, tb.c), Ok)) return SIGMA_REJECT;
    const uint64_t k2[1][4] = {{nc[0], nc[1], nc[2], nc[3]}};
    TeExt rhs = te_add(te_mul_fixed(tb.tg, s, tb.c, false), te_mul_fixed(tb.tb, sb, tb.c, false), tb.c);
    rhs = te_add(rhs, te_msm_public(&Yb, k2, 1, tb.c), tb.c);
    return te_equal(rhs, R) ? SIGMA_OK : SIGMA_REJECT;
}

}  // namespace drh
