/*
 * dotring_hip.h — C ABI of libdotring_hip.so: the MI355X (gfx950) replacement for the native arithmetic on
 * dot-ring's Ring-VRF prove/verify hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative dr_status; dr_last_error() gives the text
 *     (thread-local).  The Python shim maps DR_ERR_INVALID -> ValueError, DR_ERR_NOMEM -> MemoryError,
 *     the same exception types the reference raises at these seams.
 *   - the caller owns every buffer; the library keeps no caller pointer after return.  Handles made by
 *     *_create / *_load are freed by *_destroy.  A dr_ctx is bound to one GPU and one HIP stream; calls on
 *     one ctx are synchronous (results are on the host when the call returns) and must not be issued
 *     concurrently from several threads; different ctx objects are independent.
 *   - Bandersnatch field elements / scalars: 32 bytes little-endian, standard form (as the reference's
 *     bls_scalar_from_bytes, dot_ring/curve/native_field/bls12_381_scalar.c:266).  A TE affine point is
 *     x(32) || y(32).
 *   - BLS12-381 G1 affine points: 96 bytes, big-endian x(48) || y(48) — the SRS file record
 *     (dot_ring/ring_proof/pcs/srs.py:61-70) and blst's serialize() format; all-zero coordinates or the
 *     0x40 flag in byte 0 denote infinity.  KZG scalars: 32 bytes little-endian, any value < 2^256.
 *   - *_dev variants take pointers into GPU memory obtained from dr_dev_alloc (same layouts) so that a
 *     pipeline — or a benchmark — can keep its operands resident in HBM.
 */
#ifndef DOTRING_HIP_H
#define DOTRING_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define DR_API __attribute__((visibility("default")))
#else
#define DR_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dr_ctx dr_ctx;
typedef struct dr_srs dr_srs;

enum dr_status {
    DR_OK = 0,
    DR_ERR_INVALID = -1,   /* bad length / encoding / argument  (reference: ValueError) */
    DR_ERR_NOMEM = -2,     /* host or device allocation failed  (reference: MemoryError) */
    DR_ERR_DEVICE = -3,    /* HIP runtime failure, no usable gfx950 device */
    DR_ERR_NOTSQUARE = -4  /* dr_fr_sqrt: input is a quadratic non-residue (reference: ValueError) */
};

/* ---- library / context ------------------------------------------------------------------------- */
DR_API const char *dr_version(void);
DR_API const char *dr_last_error(void);
DR_API int dr_device_count(void);
DR_API int dr_ctx_create(int device_id, dr_ctx **out);
DR_API void dr_ctx_destroy(dr_ctx *ctx);
DR_API int dr_ctx_sync(dr_ctx *ctx);

/* HBM buffers for the *_dev entry points */
DR_API int dr_dev_alloc(dr_ctx *ctx, size_t bytes, void **dptr);
DR_API int dr_dev_free(dr_ctx *ctx, void *dptr);
DR_API int dr_dev_upload(dr_ctx *ctx, void *dptr, const void *host, size_t bytes);
DR_API int dr_dev_download(dr_ctx *ctx, void *host, const void *dptr, size_t bytes);

/* Per-kernel timing with HIP events on the ctx stream (used by bench.py for the roofline line).
 * dr_prof_get: accumulated milliseconds and launch count of the kernel called `name` since the last reset. */
DR_API int dr_prof_enable(dr_ctx *ctx, int on);
DR_API int dr_prof_reset(dr_ctx *ctx);
DR_API int dr_prof_get(dr_ctx *ctx, const char *name, double *total_ms, int *launches);

/* ---- seam A: Bandersnatch kernels ---------------------------------------------------------------
 * Replaces dot_ring/curve/native_field/bandersnatch_te.pyx:
 *   scalar_mult_windowed_native_w2_cy :480 (+ GLV callers dot_ring/curve/glv.py:191, specs/bandersnatch.py:177)
 *   scalar_mult_4_native_w2_cy :557, scalar_mult_6_native_w2_cy :669, msm_pippenger_signed_native_cy :257
 *   sqrt_mod_bls_scalar_cy :421, projective_to_affine_cy :244
 * Outputs are canonical affine coordinates (the reference normalises its projective tuples immediately,
 * glv.py:243-248), so any internal windowing gives identical bytes.
 */

/* out[i] = scalars[i] * pts[i]  for i < n.  Scalars are taken mod the group order. */
DR_API int dr_bsn_scalar_mul_batch(dr_ctx *ctx, const uint8_t *pts_xy /* n*64 */, const uint8_t *scalars /* n*32 */,
                            size_t n, uint8_t *out_xy /* n*64 */);
DR_API int dr_bsn_scalar_mul_batch_dev(dr_ctx *ctx, const void *d_pts_xy, const void *d_scalars, size_t n, void *d_out_xy);

/* out = sum_i scalars[i] * pts[i]   (n may be 0: identity (0,1)) */
DR_API int dr_bsn_msm(dr_ctx *ctx, const uint8_t *pts_xy, const uint8_t *scalars, size_t n, uint8_t out_xy[64]);

/* groups[g] = sum of `m` consecutive terms: out[g] = sum_{j<m} scalars[g*m+j] * pts[g*m+j], g < groups.
 * One launch for the many small fixed-arity MSMs of the sigma protocols (m = 2, 3, 4). */
DR_API int dr_bsn_msm_groups(dr_ctx *ctx, const uint8_t *pts_xy, const uint8_t *scalars, size_t groups, size_t m,
                      uint8_t *out_xy /* groups*64 */);

/* Elligator 2 hash-to-curve field work for n inputs: out[i] = clear_cofactor(map(u[2i]) + map(u[2i+1])), i.e.
 * TEAffinePoint._e2c_ell2_ro (dot_ring/curve/twisted_edwards/te_affine_point.py:212-295, te_curve.py:48-95) after
 * hash_to_field, which stays on the host.  u_pairs: n*2 canonical field elements (32-byte LE). */
DR_API int dr_bsn_encode_to_curve_batch(dr_ctx *ctx, const uint8_t *u_pairs, size_t n, uint8_t *out_xy);

/* dec_point for n compressed points (dot_ring/vrf/codec.py:39-45, curve/point.py:150-214, curve/curve.py:56-67):
 * decompression (y < p, x^2 = (1-y^2)/(a-d y^2), the sign bit picks the larger root) and validation (not the
 * identity, prime-order subgroup) on the GPU.  ok[i] = 1 for valid points; out_xy[i] is meaningful only then. */
DR_API int dr_bsn_decode_points(dr_ctx *ctx, const uint8_t *enc /* n*32 */, size_t n, uint8_t *out_xy /* n*64 */, uint8_t *ok /* n */);

/* The same four operations on any twisted Edwards curve over this base field the library knows:
 * DR_CURVE_BANDERSNATCH (a = -5, cofactor 4; identical to the dr_bsn_* calls) or DR_CURVE_JUBJUB (a = -1, cofactor 8,
 * dot_ring/curve/specs/jubjub.py:17-29 — no endomorphism, so the plain 64-window kernels).  The sigma-protocol and
 * ring entry points below take the curve from dr_vrf_suite.curve. */
enum { DR_CURVE_BANDERSNATCH = 0, DR_CURVE_JUBJUB = 1 };
DR_API int dr_te_scalar_mul_batch(dr_ctx *ctx, int curve, const uint8_t *pts_xy, const uint8_t *scalars, size_t n, uint8_t *out_xy);
DR_API int dr_te_msm(dr_ctx *ctx, int curve, const uint8_t *pts_xy, const uint8_t *scalars, size_t n, uint8_t out_xy[64]);
DR_API int dr_te_msm_groups(dr_ctx *ctx, int curve, const uint8_t *pts_xy, const uint8_t *scalars, size_t groups, size_t m, uint8_t *out_xy);
DR_API int dr_te_decode_points(dr_ctx *ctx, int curve, const uint8_t *enc, size_t n, uint8_t *out_xy, uint8_t *ok);

/* Fixed-base multiplication for CONSTANT points — the generator G and the Pedersen blinding base B of the sigma protocols
 * (dot_ring/vrf/pedersen/vrf.py:94,104,111: x*G + b*B, k*G + k_b*B; curve.py:384 pk = sk*G):
 * out[g] = sum_{j<m} scalars[g*m+j] * bases[j], m <= 4.  The first call with a base builds its window table in HBM
 * (every multiple (e+1)*16^w*P, 48 KB, cached in the context); a multiplication is then 64 table additions and no doubling,
 * spread over four lanes.  Results are the canonical affine points, as dr_bsn_msm_groups gives them. */
DR_API int dr_te_fixed_base_msm_groups(dr_ctx *ctx, int curve, const uint8_t *bases_xy /* m*64 */, size_t m,
                                       const uint8_t *scalars /* groups*m*32 */, size_t groups, uint8_t *out_xy /* groups*64 */);

/* Diagnostic: the device's field arithmetic on the base field of the twisted Edwards curves (the 9 x 29-bit representation of
 * csrc/fr29.hip.h that every kernel of this seam computes in), one lane per pair of canonical little-endian elements.
 * out: n x 12 x 32 bytes — a b, a^2, a + b, a - b, a^-1 (0 for 0), (a + b)(a - b), -5 a, 2 a b (fused product), sqrt(a) or 0;
 * then, each times 2^261 (the form the kernels compute in): a b + a, a + 27 b, a - 28 b through the lazy-sum helpers of the NTT /
 * polynomial kernels (canon29_small, reduce_small).  is_square[i] = 1 iff a[i] is a square.  The reference has no counterpart: its field is Python / C big integers
 * (dot_ring/curve/native_field/scalar.pyx); the tests check this entry point against the same integers. */
DR_API int dr_fr_ops_selftest(dr_ctx *ctx, const uint8_t *a /* n*32 */, const uint8_t *b /* n*32 */, size_t n, uint8_t *out /* n*384 */,
                              uint8_t *is_square /* n */);

/* square root in the Bandersnatch base field; DR_ERR_NOTSQUARE if none exists. Host-side, no ctx. */
DR_API int dr_fr_sqrt(const uint8_t in[32], uint8_t out[32]);

/* ---- seam B: KZG / BLS12-381 G1 -----------------------------------------------------------------
 * Replaces the blst calls behind dot_ring/ring_proof/pcs/kzg.py: commit :152-175 (mult_pippenger over
 * srs.blst_g1_memory[:n]), msm_g1 :147, compress_g1 :129, serialize_g1_uncompressed :133, decompress_g1 :137,
 * and the SRS memory built in dot_ring/ring_proof/pcs/srs.py:98-114.
 */
DR_API int dr_srs_load(dr_ctx *ctx, const uint8_t *g1_be_xy /* m*96 */, size_t m, dr_srs **out);
/* Synthetic bases for sizes no SRS file covers: base[i] = (first+i) * seed (first >= 1), generated on
 * the GPU.  With these bases an MSM has the closed form [sum_i k_i*(first+i)] * seed — a size-independent check. */
DR_API int dr_srs_synthetic(dr_ctx *ctx, const uint8_t seed_be_xy[96], uint32_t first, size_t count, dr_srs **out);
/* Known-tau SRS for domains the shipped file does not cover (test/bench use only — tau is public):
 * bases[i] = tau^i * base, i < count, generated on the GPU.  dr_g2_mul (host) gives the matching tau * G2. */
DR_API int dr_srs_powers(dr_ctx *ctx, const uint8_t base_be_xy[96], const uint8_t tau_le[32], size_t count, dr_srs **out);
DR_API int dr_g2_mul(const uint8_t g2_be[192], const uint8_t scalar_le[32], uint8_t out_be[192]);
/* Fixed-base window table for this SRS, kept in HBM: table[w][i] = 2^(start_w) * base[i] for the W = ceil(256/c)
 * windows of width ~c = window_bits (7..22; 0 drops the table).  Costs W * count * 96 bytes (13 MB for the shipped
 * 6145-point SRS at c = 12, 1.6 GB for 2^20 bases at c = 16) and makes every later MSM over this SRS use ONE bucket
 * set per MSM: a single bucket reduction instead of W and no window combination.  Results are unchanged. */
DR_API int dr_srs_precompute(dr_ctx *ctx, dr_srs *srs, int window_bits);
/* Shape of the table dr_srs_precompute built (all zero: none) and the tiling `batch` MSMs of n points over it would take.  An SRS
 * small enough (256 * count * 128 bytes within DOTRING_SRS_BIT_ROWS_MB, default 512: 201 MB for 6145 points) gets a row for EVERY bit,
 * table[s][i] = 2^s * base[i]; batches of hundreds of MSMs then recode every scalar in width-w non-adjacent form (w chosen per call from
 * n and batch; DOTRING_SRS_TILING=rows keeps the window rows): 256 / (w + 1) odd digits per scalar on average, each one a
 * point of the row of its bit position added to one of 2^(w-2) odd-multiple buckets — against 256 / window_bits additions over the
 * window rows.  Results are unchanged.
 *   info[0] window_bits, info[1] rows of the table (W or 256), info[2] digit rows (windows / slots) per scalar for this (n, batch),
 *   info[3] width w of the per-call tiling (0 = the window rows), info[4] tiling: 0 = window rows, 2 = width-w non-adjacent
 *   form, info[5] expected non-zero digits per scalar x 1000 */
DR_API int dr_srs_table_info(const dr_srs *srs, size_t n, size_t batch, int info[6]);
/* copy `count` bases starting at `offset` back to the host as BE x||y records */
DR_API int dr_srs_download(dr_ctx *ctx, const dr_srs *srs, size_t offset, size_t count, uint8_t *out_be_xy);
DR_API void dr_srs_destroy(dr_srs *srs);
DR_API size_t dr_srs_size(const dr_srs *srs);

/* out = sum_{i<n} scalars[i] * SRS[offset+i] ; *is_inf = 1 and out = zeros when the sum is the identity */
DR_API int dr_g1_msm(dr_ctx *ctx, const dr_srs *srs, size_t offset, const uint8_t *scalars /* n*32 */, size_t n,
              uint8_t out_be_xy[96], int *is_inf);
DR_API int dr_g1_msm_dev(dr_ctx *ctx, const dr_srs *srs, size_t offset, const void *d_scalars, size_t n,
                  uint8_t out_be_xy[96], int *is_inf);
/* `batch` independent MSMs over the same bases SRS[0..n): scalars is batch*n*32, out is batch*96, is_inf batch ints */
DR_API int dr_g1_msm_batch(dr_ctx *ctx, const dr_srs *srs, const uint8_t *scalars, size_t n, size_t batch,
                    uint8_t *out_be_xy, int *is_inf);
DR_API int dr_g1_msm_batch_dev(dr_ctx *ctx, const dr_srs *srs, const void *d_scalars, size_t n, size_t batch,
                        uint8_t *out_be_xy, int *is_inf);
/* MSM over caller-supplied points (verifier folds, kzg.py:295-301,332-338) */
DR_API int dr_g1_msm_points(dr_ctx *ctx, const uint8_t *pts_be_xy /* n*96 */, const uint8_t *scalars, size_t n,
                     uint8_t out_be_xy[96], int *is_inf);

/* host-side sum of a few affine points (combining per-GPU partial MSM results after an all-gather) */
DR_API int dr_g1_sum(const uint8_t *pts_be_xy /* n*96 */, size_t n, uint8_t out_be_xy[96], int *is_inf);

/* ---- multi-GPU: the base-sharded MSM (second mode; the shape of the one MSM to shard is the reference's
 * KZG.commit, dot_ring/ring_proof/pcs/kzg.py:152-175; its process-sharded bench is tests/benchmark/bench_ring_proof.py:168-182).
 * One process per GPU.  Rank g reduces its shard of (base, scalar) pairs to one point; the points are exchanged with RCCL
 * ncclAllGather over xGMI (97 bytes per rank) and every rank folds them with the group law.  librccl is dlopen'ed on first
 * use; no PyTorch anywhere.  dr_comm_unique_id runs on ONE rank, its 128 bytes reach the others through the launcher's
 * channel (dot_ring_amd/parallel.py: a TCP socket on MASTER_ADDR), then every rank calls dr_comm_create (a collective). */
#define DR_COMM_ID_BYTES 128
typedef struct dr_comm dr_comm;
DR_API int dr_comm_unique_id(uint8_t out_id[DR_COMM_ID_BYTES]);
DR_API int dr_comm_create(dr_ctx *ctx, const uint8_t id[DR_COMM_ID_BYTES], int rank, int world, dr_comm **out);
DR_API void dr_comm_destroy(dr_comm *comm);
DR_API int dr_comm_rank(const dr_comm *comm);
DR_API int dr_comm_world(const dr_comm *comm);
/* ranks RCCL itself counts in the communicator (ncclCommCount) */
DR_API int dr_comm_count(const dr_comm *comm, int *out_ranks);
/* host-to-host all-gather of `bytes` bytes per rank (staged through HBM, ncclAllGather on the context's stream) */
DR_API int dr_comm_all_gather(dr_comm *comm, const void *send, size_t bytes, void *recv /* world*bytes */);
/* this rank's n_local pairs (srs[offset..], device-resident scalars) of one MSM sharded over the communicator;
 * the result is the whole MSM, identical on every rank.  A rank whose local part fails still enters the all-gather
 * (status byte in its 97-byte record), so the others return DR_ERR_DEVICE naming it instead of blocking in the collective */
DR_API int dr_g1_msm_sharded_dev(dr_ctx *ctx, dr_comm *comm, const dr_srs *srs, size_t offset, const void *d_scalars, size_t n_local,
                                 uint8_t out_be_xy[96], int *is_inf);

/* Host-side pairing product check: *ok = 1 iff prod_i e(P_i, Q_i) == 1.  G2 points are 192-byte records in the SRS
 * file layout x.c1 || x.c0 || y.c1 || y.c0 (big-endian, dot_ring/ring_proof/pcs/srs.py:78-88).  Replaces
 * blst.PT + PT.finalverify (dot_ring/ring_proof/pcs/pairing.py:24-31); stays on the CPU (2 Miller loops per batch). */
DR_API int dr_pairing_check(const uint8_t *g1_be_xy /* n*96 */, const uint8_t *g2_be /* n*192 */, size_t n, int *ok);
/* diagnostic for tests: *consistent = 1 iff the fast final exponentiation used by dr_pairing_check (Frobenius maps +
 * x-chain, exponent 3(p^12-1)/r) equals the cube of the plain square-and-multiply one on this Miller-loop product */
DR_API int dr_pairing_selfcheck(const uint8_t *g1_be_xy, const uint8_t *g2_be, size_t n, int *consistent);

/* zcash encodings, host-side */
DR_API int dr_g1_compress(const uint8_t xy[96], int is_inf, uint8_t out[48]);
DR_API int dr_g1_decompress(const uint8_t in[48], uint8_t out_xy[96], int *is_inf);   /* on-curve check, no subgroup check (as blst P1_Affine(bytes)) */
DR_API int dr_g1_serialize_check(const uint8_t xy[96]);
/* dr_g1_decompress for n encodings in one kernel launch: out = n BE x||y records (all zero for infinity), ok[i] = 0 for
 * malformed encodings (compression flag missing, x >= p, x not on the curve, non-canonical infinity). */
DR_API int dr_g1_decompress_batch(dr_ctx *ctx, const uint8_t *enc /* n*48 */, size_t n, uint8_t *out_be_xy /* n*96 */, uint8_t *ok /* n */);                              /* DR_OK iff on curve or infinity */

/* ---- seam C: NTT over Fr ------------------------------------------------------------------------
 * Replaces BlsScalarNTTPlan.transform / transform_scaled (dot_ring/ring_proof/polynomial/ntt.pyx:104-163,
 * bls_scalar_ntt_round in bls12_381_scalar.c:333): `batch` in-place radix-2 transforms of size 2^log2n with
 * the primitive root `omega`; when scale != NULL every output is multiplied by it (inverse transform: pass
 * omega^-1 and n^-1).  data: batch * 2^log2n * 32 bytes, natural order in and out.
 */
DR_API int dr_ntt(dr_ctx *ctx, uint8_t *data, unsigned log2n, size_t batch, const uint8_t omega[32], const uint8_t *scale);
DR_API int dr_ntt_dev(dr_ctx *ctx, void *d_data, unsigned log2n, size_t batch, const uint8_t omega[32], const uint8_t *scale);

/* ---- batched ring prover ---------------------------------------------------------------------------
 * Device-resident prover for MANY proofs over ONE ring (additive API): replaces the interpreted loops of
 * dot_ring/ring_proof/proof_builder.py:38-315, columns/columns.py:111-167 and constraints/constraints.py:43-151.
 * The Fiat-Shamir transcript stays on the host (hashlib), so proving is four phases, each ending where the
 * reference squeezes challenges; everything between the hashes stays in HBM.
 *   create    ring rows nm_points (N * 64 bytes, x||y LE; rows as built by Ring(): keys, padding, 2^i*B, 4 x (0,0)),
 *             omega_n / omega_4n primitive roots of the N and 4N domains, seed = accumulator base
 *   root      the three fixed-column commitments px, py, s (RingRoot.from_ring, vrf/ring/root.py:21-44)
 *   witness   in: producer row index and blinding factor per proof, optional 12 hidden-row values per proof
 *             (columns b, accip, accx, accy x 3 rows; NULL = test-vector mode, zeros)
 *             out: relation point Y_bar (x||y LE) and the four witness commitments (b, accip, accx, accy)
 *   quotient  in: seven alphas per proof; out: quotient commitment C_q
 *   evals     in: zeta per proof; out: px,py,s,b,accip,accx,accy at zeta and l(zeta*omega)   (8 x 32 bytes LE)
 *   openings  in: eight nus per proof; out: the two opening proofs (at zeta, at zeta*omega)
 * All four phase calls of one batch must use the same `batch`.  Scalars are 32-byte LE canonical field elements.
 */
typedef struct dr_ring_prover dr_ring_prover;
DR_API int dr_ring_prover_create(dr_ctx *ctx, const dr_srs *srs, unsigned log2n, uint32_t max_ring, const uint8_t omega_n[32],
                                 const uint8_t omega_4n[32], const uint8_t *nm_points_xy, const uint8_t seed_xy[64],
                                 dr_ring_prover **out);
/* the same for a ring whose keys live on `curve` (DR_CURVE_*): the constraint system uses that curve's coefficient a and
 * max_ring + bit length of its group order + 4 must fit the domain */
DR_API int dr_ring_prover_create_te(dr_ctx *ctx, int curve, const dr_srs *srs, unsigned log2n, uint32_t max_ring,
                                    const uint8_t omega_n[32], const uint8_t omega_4n[32], const uint8_t *nm_points_xy,
                                    const uint8_t seed_xy[64], dr_ring_prover **out);
DR_API void dr_ring_prover_destroy(dr_ring_prover *p);
DR_API int dr_ring_prover_root(const dr_ring_prover *p, uint8_t out_commitments[3 * 96], int is_inf[3]);
DR_API int dr_ring_prover_fixed_coeffs(dr_ring_prover *p, uint8_t *out /* 3*N*32: px, py, s coefficients */);
DR_API int dr_ring_prove_witness(dr_ring_prover *p, size_t batch, const uint32_t *producer_index, const uint8_t *blinding,
                                 const uint8_t *zk_rows, uint8_t *out_relation_xy, uint8_t *out_commitments, int *is_inf);
DR_API int dr_ring_prove_quotient(dr_ring_prover *p, size_t batch, const uint8_t *alphas, uint8_t *out_cq, int *is_inf);
DR_API int dr_ring_prove_evals(dr_ring_prover *p, size_t batch, const uint8_t *zetas, uint8_t *out_evals);
DR_API int dr_ring_prove_openings(dr_ring_prover *p, size_t batch, const uint8_t *nus, uint8_t *out_openings, int *is_inf);
/* Secrets do not stay in HBM.  dr_ring_prove_openings — the last phase of a batch — ends by zeroing the prover's per-batch state
 * (blinding factors, hidden rows, the witness columns with their bit column, every polynomial derived from them) and the MSM scratch
 * of its context; the batch entry points (dr_ringvrf_prove_batch, dr_pedersen_prove_batch, dr_ietf_prove_batch) also zero the device
 * copies of secret scalars and nonces.  The memsets are ordered behind the batch's last kernel on a stream of their own; the call does not
 * wait for them, and whatever uses the context next is ordered behind them on the device.  dr_ring_prover_wipe does the same on
 * request; dr_ring_prover_residue counts the non-zero 32-bit words left in those buffers (0 after a wipe; a test hook).
 * DOTRING_WIPE=0 disables the wipes (to measure their cost). */
DR_API int dr_ring_prover_wipe(dr_ring_prover *p);
DR_API int dr_ctx_scratch_residue(dr_ctx *ctx, uint64_t *words);    /* the same count for one context's scratch buffers */
DR_API int dr_ring_prover_residue(dr_ring_prover *p, uint64_t *words);


/* ---- native batch orchestration ------------------------------------------------------------------
 * The reference runs the hashing between the arithmetic steps of a proof in interpreted Python (hashlib):
 * hash_to_field (dot_ring/curve/curve.py:110-185), the VRF transcript / nonces / challenge
 * (dot_ring/vrf/primitives.py:26-122, pedersen/vrf.py:86-126) and the ring proof's Fiat-Shamir transcript
 * (ring_proof/transcript/transcript.py:21-136, phases.py:18-69).  For a batch that is ~40 small hashes per proof; here
 * they run on worker threads (DOTRING_HOST_THREADS, default min(16, cores)) inside ONE call per batch, with the GPU
 * phases above in between.  Results are byte-identical to prove() called per proof.
 */
enum { DR_HASH_SHA512 = 0, DR_HASH_SHAKE128 = 1, DR_HASH_SHAKE256 = 2,
       /* diagnostic: `data` = four messages of len / 4 bytes each, `out` = their four SHAKE128 digests of out_len / 4 bytes each
        * (at most 168), computed by the four-in-lockstep sponge the batch transcripts use */
       DR_HASH_SHAKE128_X4 = 3 };
DR_API int dr_host_hash(int kind, const uint8_t *data, size_t len, uint8_t *out, size_t out_len);
/* out = SHAKE256(seed || LE64(0))[0..576) || SHAKE256(seed || LE64(1))[0..576) || ... (len bytes), hashed on the worker threads.
 * dr_ringvrf_prove_batch's zk_random48 for a batch (12 x 48 bytes per proof: the hidden rows of columns/columns.py:139-146, drawn
 * with `secrets` in the reference) comes from one 32-byte OS seed this way. */
DR_API int dr_host_random_expand(const uint8_t seed[32], uint8_t *out, size_t len);
/* moves the secret part of dr_ringvrf_prove_batch's auxiliary records out: the 32-byte blinding factor of record i (offset 256
 * of its DR_RINGVRF_AUX_BYTES) is copied to out_blind + 32 i and zeroed in the record, so that what callers keep per batch holds
 * no secret and what they keep per proof is that proof's own factor only */
DR_API int dr_ringvrf_aux_take_blindings(uint8_t *aux, size_t batch, uint8_t *out_blind /* batch*32 */);

typedef struct dr_vrf_suite {
    const uint8_t *suite_id;        /* e.g. "Bandersnatch-SHA512-ELL2-v1" (bandersnatch.py:74-87) */
    size_t suite_id_len;
    int xof;                        /* 1: SHAKE128 suite, 0: SHA-512 (counter-mode squeeze, expand_message_xmd) */
    uint8_t generator_xy[64];       /* group generator, x||y little-endian */
    uint8_t blinding_base_xy[64];   /* Pedersen blinding base (bandersnatch.py:89-102) */
    int curve;                      /* DR_CURVE_BANDERSNATCH (Elligator 2 hash-to-curve) or DR_CURVE_JUBJUB (try-and-increment) */
} dr_vrf_suite;

/* hash_to_field(msg, 2) for `count` messages msgs[off[i]..off[i+1]): out = count * 2 field elements (32-byte LE),
 * the input format of dr_bsn_encode_to_curve_batch. */
DR_API int dr_hash_to_field_batch(const dr_vrf_suite *suite, const uint8_t *msgs, const uint64_t *off /* count+1 */, size_t count,
                                  uint8_t *out_u_pairs);

/* encode_to_curve(salt_i || msg_i) for `count` messages (salts / salt_off nullable), whichever way the suite's curve hashes:
 * Elligator 2 (hash_to_field here + dr_bsn_encode_to_curve_batch) or try-and-increment (dot_ring/curve/point.py:252-296:
 * candidates hashed on worker threads, decompressed and cofactor-cleared on the GPU, several counters per launch). */
DR_API int dr_encode_to_curve_batch(dr_ctx *ctx, const dr_vrf_suite *suite, const uint8_t *msgs, const uint64_t *off /* count+1 */,
                                    const uint8_t *salts, const uint64_t *salt_off, size_t count, uint8_t *out_xy /* count*64 */);

/* RingVRF.prove for `batch` (<= 4096) proofs over the prover's ring: out_proofs = batch * 784 bytes
 * (Pedersen 192 || ring payload 592, dot_ring/vrf/ring/vrf.py:51-58).  alphas/ads/salts are concatenated with
 * (batch+1) offsets (salts may be NULL); secret_scalars batch*32 LE; producer_index = position of each signer's key in
 * the ring; fs_prefix = the bytes the ring-proof transcript has absorbed before "instance" (initial label and verifier
 * key, vrf/ring/root.py:54-72); zk_random48 = batch*12 48-byte random strings for the hidden rows (reduced mod p
 * here), or NULL for the deterministic test-vector mode.  out_aux (nullable, batch * DR_RINGVRF_AUX_BYTES): per proof
 * the affine Pedersen points O, Y_bar, R, O_k (4*64), the blinding factor (32) and the seven G1 commitments
 * serialised uncompressed (7*96: C_b, C_accip, C_accx, C_accy, C_q, Phi_zeta, Phi_zeta_omega). */
#define DR_RINGVRF_AUX_BYTES 960
DR_API int dr_ringvrf_prove_batch(dr_ring_prover *p, const dr_vrf_suite *suite, size_t batch, const uint8_t *alphas,
                                  const uint64_t *alpha_off, const uint8_t *ads, const uint64_t *ad_off, const uint8_t *salts,
                                  const uint64_t *salt_off, const uint8_t *secret_scalars, const uint32_t *producer_index,
                                  const uint8_t *fs_prefix, size_t fs_prefix_len, const uint8_t *zk_random48, uint8_t *out_proofs,
                                  uint8_t *out_aux);


/* PedersenVRF.prove / batch_verify for a batch (dot_ring/vrf/pedersen/vrf.py:86-126, 171-242): the Pedersen halves of the
 * two Ring-VRF calls on their own.  Proofs are 192 bytes (gamma || Y_bar || R || O_k || s || s_b).  out_aux (nullable,
 * batch * DR_PEDERSEN_AUX_BYTES): O, Y_bar, R, O_k affine (4*64) and the blinding factor (32).  The verifier decodes and
 * subgroup-checks the proof points on the GPU; *ok = 1 iff every proof verifies (malformed input: *ok = 0, DR_OK). */
#define DR_PEDERSEN_AUX_BYTES 288
DR_API int dr_pedersen_prove_batch(dr_ctx *ctx, const dr_vrf_suite *suite, size_t batch, const uint8_t *alphas, const uint64_t *alpha_off,
                                   const uint8_t *ads, const uint64_t *ad_off, const uint8_t *salts, const uint64_t *salt_off,
                                   const uint8_t *secret_scalars, uint8_t *out_proofs, uint8_t *out_aux);
DR_API int dr_pedersen_verify_batch(dr_ctx *ctx, const dr_vrf_suite *suite, size_t batch, const uint8_t *proofs, const uint8_t *inputs,
                                    const uint64_t *in_off, const uint8_t *ads, const uint64_t *ad_off, const uint8_t *salts,
                                    const uint64_t *salt_off, int *ok);

/* TinyVRF.prove (thin = 0: 80-byte proofs O || c || s, dot_ring/vrf/ietf/tiny.py:35-70) or ThinVRF.prove (thin = 1: 96-byte
 * proofs O || R || s) for a batch; arguments as for dr_pedersen_prove_batch.  out_aux (nullable, batch * 128): O and R affine. */
DR_API int dr_ietf_prove_batch(dr_ctx *ctx, const dr_vrf_suite *suite, int thin, size_t batch, const uint8_t *alphas,
                               const uint64_t *alpha_off, const uint8_t *ads, const uint64_t *ad_off, const uint8_t *salts,
                               const uint64_t *salt_off, const uint8_t *secret_scalars, uint8_t *out_proofs, uint8_t *out_aux);

/* TinyVRF.verify / ThinVRF.verify (dot_ring/vrf/ietf/tiny.py:72-88, thin.py:96-118) for `batch` ENCODED proofs (80 bytes O || c || s, or
 * 96 bytes O || R || s with thin = 1), proof i under the compressed public key public_keys[32 i ..]: verdict[i] = 1 verifies, 0 does
 * not, 2 the public key does not decode to a prime-order point, 3 the proof is malformed (a point that does not decode, a scalar >= n)
 * — the cases the reference raises ValueError for.  Each proof is checked on its own, one per worker thread, on HOST cores: this is the
 * single-proof entry point (one proof: ~0.6 ms against three kernel launch chains); the relation of MANY Thin proofs at once is
 * ThinVRF.batch_verify's one MSM on the GPU (dr_te_msm).  Elligator suites of Bandersnatch only.
 *
 * Small calls in general: up to DOTRING_SMALL_HOST_MAX proofs (default 64; 0 = never) dr_ietf_prove_batch, dr_pedersen_prove_batch and
 * dr_pedersen_verify_batch run the same protocol on host cores too (csrc/hostsigma.hpp) — secret scalars on fixed-schedule
 * multiplications — and give the same bytes and verdicts as the kernels. */
DR_API int dr_ietf_verify_batch(dr_ctx *ctx, const dr_vrf_suite *suite, int thin, size_t batch, const uint8_t *proofs,
                                const uint8_t *public_keys /* batch*32 */, const uint8_t *inputs, const uint64_t *in_off, const uint8_t *ads,
                                const uint64_t *ad_off, const uint8_t *salts, const uint64_t *salt_off, uint8_t *verdict /* batch */);

/* What a verifier knows about one ring (RingRoot + RingProofParams + SRS verifier part). */
typedef struct dr_ring_verifier_key {
    unsigned log2n;                      /* domain size N = 2^log2n */
    uint8_t omega_n[32];                 /* primitive N-th root of unity, LE */
    uint8_t seed_xy[64];                 /* accumulator base point (bandersnatch.py:89-102), x||y LE */
    uint8_t fixed_commitments[3 * 96];   /* C_px, C_py, C_s serialised uncompressed (BE x||y; infinity = 0x40 || 0) */
    uint8_t g1_generator[96];            /* SRS G1[0], BE x||y */
    uint8_t g2[2 * 192];                 /* SRS [1]G2, [tau]G2 in file byte order (pcs/srs.py:78-88) */
    const uint8_t *fs_prefix;            /* as for dr_ringvrf_prove_batch */
    size_t fs_prefix_len;
} dr_ring_verifier_key;

/* RingVRF.batch_verify (dot_ring/vrf/ring/vrf.py:239-283) over `batch` (<= 4096) ENCODED proofs (784 bytes each):
 * decoding and validating every point (Bandersnatch: canonical, on curve, prime-order subgroup; G1: zcash
 * decompression) runs on the GPU, the transcript replay and the verifier's scalar pass on worker threads, then one
 * (5B+2)-point Bandersnatch MSM (Pedersen part) and two G1 MSMs + one pairing equation (ring part).  seed32 = fresh
 * verifier randomness for the random linear combination.  *ok = 1 iff every proof verifies; malformed proofs give
 * *ok = 0 with DR_OK. */
DR_API int dr_ringvrf_verify_batch(dr_ctx *ctx, const dr_vrf_suite *suite, const dr_ring_verifier_key *vk, size_t batch,
                                   const uint8_t *proofs, const uint8_t *inputs, const uint64_t *in_off, const uint8_t *ads,
                                   const uint64_t *ad_off, const uint8_t *salts, const uint64_t *salt_off,
                                   const uint8_t seed32[32], int *ok);

/* ---- one batch over several GPUs of ONE process (SURVEY 8(b) additive row: a device set instead of one device; 8(e) first mode) -----
 * The same two calls over a set of devices: provers[g] / ctxs[g] live on the devices the caller chose (dr_ctx_create(device_id), one
 * dr_srs and one dr_ring_prover of the SAME ring per device — the SRS and the per-ring tables are replicated in each HBM, < 250 MB).
 * Device g takes proofs [g B / G, (g + 1) B / G) (sizes differ by at most one; B < G leaves shards empty) on a host thread of its own
 * and writes its results straight into the caller's buffers: all B proofs come back in order, byte-identical to the one-device call
 * (in test-vector mode; with hidden rows, zk_random48 is consumed by proof index, so the bytes do not depend on G either).  No collective
 * and no exchange between the devices.  The verifier ANDs the shards' verdicts; every shard folds with randomness of its own derived
 * from seed32.  Entries of the set may be several contexts on ONE device (how the one-GPU tests run it).  The host worker pool
 * (DOTRING_HOST_THREADS) is shared by the shards: give it about 16 threads per device.
 * One process PER GPU (torch.distributed / any launcher) shards the same way through dot_ring_amd.parallel.prove_batch_sharded, with the
 * 784-byte proofs gathered over the communicator; the reference's own multi-worker shape is tests/benchmark/bench_ring_proof.py:168-182. */
DR_API int dr_ringvrf_prove_batch_multi(dr_ring_prover *const *provers, size_t n_provers, const dr_vrf_suite *suite, size_t batch,
                                        const uint8_t *alphas, const uint64_t *alpha_off, const uint8_t *ads, const uint64_t *ad_off,
                                        const uint8_t *salts, const uint64_t *salt_off, const uint8_t *secret_scalars,
                                        const uint32_t *producer_index, const uint8_t *fs_prefix, size_t fs_prefix_len,
                                        const uint8_t *zk_random48, uint8_t *out_proofs, uint8_t *out_aux);
DR_API int dr_ringvrf_verify_batch_multi(dr_ctx *const *ctxs, size_t n_ctxs, const dr_vrf_suite *suite, const dr_ring_verifier_key *vk,
                                         size_t batch, const uint8_t *proofs, const uint8_t *inputs, const uint64_t *in_off,
                                         const uint8_t *ads, const uint64_t *ad_off, const uint8_t *salts, const uint64_t *salt_off,
                                         const uint8_t seed32[32], int *ok);

#ifdef __cplusplus
}
#endif
#endif /* DOTRING_HIP_H */
