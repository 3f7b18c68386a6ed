// Batched ring-proof prover kernels (K7 constraint evaluation, K8 polynomial passes, witness generation).
// Replaces, for a batch of proofs over ONE ring, the interpreted loops of the reference's prover:
//   witness columns            dot_ring/ring_proof/columns/columns.py:111-146
//   constraints c1..c7 + alpha dot_ring/ring_proof/constraints/constraints.py:64-151, proof_builder.py:165-180
//   tail factor / vanishing    dot_ring/ring_proof/proof_builder.py:182-195, polynomial/ops.py:207-224
//   Horner evaluations         dot_ring/ring_proof/polynomial/ops.py:170-176 (proof_builder.py:235-250)
//   linearisation / nu-aggregation  proof_builder.py:197-233, 288-315
//   synthetic division         dot_ring/ring_proof/pcs/utils.py:27-35
// Conventions: coefficient arrays cross kernel boundaries in STANDARD form (32-byte little-endian field elements — what the
// MSM kernels consume).  The hot kernels compute on the unsaturated field of fr29.hip.h (round 3: K7 constraints, K8 quotient /
// evaluation / linearisation / aggregation; bodies and their bounds: ring_body.hip.h): the 4N-domain evaluations between the
// forward NTT, the constraint kernel and the inverse NTT, the per-ring tables on the 4N domain and the per-proof scalars that
// multiply whole vectors are raw 9-limb records in Montgomery form (2^261) — "FS9", 36 bytes.  Witness generation, the setup
// kernels and synthetic division keep the saturated 8-word field (Montgomery 2^256).  Batch-major layouts: [proof][column][index].
#pragma once
#include "kernels_te.hip.h"
#include "kernels_ntt.hip.h"

namespace dr {

// a*v on the saturated representation (the constraint and linearisation kernels): a = -5 (Bandersnatch) or -1 (JubJub)
template <int CV>
DR_DEV Fr te_mul_a_fr(const Fr& v) {
    if (CV == CV_JUBJUB) return neg(v);
    Fr t = dbl(v);
    t = dbl(t);
    t = add(t, v);
    return neg(t);
}
DR_DEV Fr ld_std(const uint32_t* p) { return to_mont(gload_fr(p)); }
DR_DEV void st_std(uint32_t* p, const Fr& v) { gstore_fr(p, from_mont(v)); }
DR_DEV Fr fr_from_u32(uint32_t v) {       // small integer -> Montgomery
    Fr r = Fr::zero();
    r.l[0] = v;
    return to_mont(r);
}
DR_DEV Fr fr_pow_u32(Fr base, uint32_t e) {
    Fr r = Fr::one();
    for (; e; e >>= 1) {
        if (e & 1) r = mul(r, base);
        base = sqr(base);
    }
    return r;
}

DR_DEV Fs fs_pow_u32_early(Fs base, uint32_t e) {          // base^e on the unsaturated field (Montgomery in and out)
    Fs r = Fs::one();
    for (; e; e >>= 1) {
        if (e & 1) r = mul(r, base);
        base = sqr(base);
    }
    return r;
}

// ---- per-ring setup ------------------------------------------------------------------------------------------
// split the ring points into the px / py evaluation columns (standard form) and the selector column
__global__ void k_ring_fixed_evals(const uint32_t* __restrict__ pts_std /* N*16 */, uint32_t n, uint32_t max_ring,
                                   uint32_t* __restrict__ cols /* [3][N][8]: px, py, s */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gstore_fr(cols + ((size_t)0 * n + i) * 8, gload_fr(pts_std + (size_t)i * 16));
    gstore_fr(cols + ((size_t)1 * n + i) * 8, gload_fr(pts_std + (size_t)i * 16 + 8));
    Fr s = Fr::zero();
    s.l[0] = i < max_ring ? 1u : 0u;
    gstore_fr(cols + ((size_t)2 * n + i) * 8, s);
}
// zero-padded copy of `count` coefficient vectors of length n into vectors of length m >= n (std form)
__global__ void k_ring_pad(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ dst, uint32_t m, size_t count) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count * m) return;
    size_t v = gid / m;
    uint32_t j = (uint32_t)(gid % m);
    Fr x = j < n ? gload_fr(src + (v * n + j) * 8) : Fr::zero();
    gstore_fr(dst + gid * 8, x);
}
// Lagrange basis coefficients over the size-n domain for rows 0 and `last`: L_i(X) = (1/n) sum_j (x_i^-1)^j X^j
__global__ void k_ring_lagrange(uint32_t* __restrict__ out /* [2][n][8] std */, uint32_t n, FrArg inv_n_mont, FrArg inv_xlast_mont) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Fr inv_n = from_arg(inv_n_mont);
    st_std(out + (size_t)j * 8, inv_n);                                         // row 0: x_0 = 1
    st_std(out + ((size_t)n + j) * 8, mul(inv_n, fr_pow_u32(from_arg(inv_xlast_mont), j)));
}
// in place: standard -> Montgomery
__global__ void k_fr_to_mont(uint32_t* data, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) gstore_fr(data + i * 8, to_mont(gload_fr(data + i * 8)));
}
// standard-form 8-word elements -> FS9 records (Montgomery 2^261, normal limbs): per-ring tables once per ring, alphas / nus once per batch
__global__ void k_fr_std_to_fs9(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) fs_store9(dst + i * L29, fs_from_std(gload_fr(src + i * 8)));
}

// ---- the three non-trivial cosets of H in the 4N domain (round 3) ---------------------------------------------------
// The 4N-domain point i = 4 j + c is w^j zeta^c (zeta = w_4N).  Everything the constraint kernel touches is kept coset-major,
// [c - 1][j] for c = 1..3: a polynomial's evaluations on coset c are ONE N-point NTT of its coefficients scaled by zeta^(c m), the
// row shift i -> i + 4 is j -> j + 1 within a coset, and coset 0 — H itself, where the aggregated constraint polynomial vanishes
// outside the three hidden rows — is never transformed or evaluated at all.
// multipliers of the scaled NTT input: out[c - 1][m] = zeta^(c m) R^2 (FS9), so that unpack(a_m) * out = Montgomery(a_m zeta^(c m))
__global__ void k_ring_coset_scale(uint32_t* __restrict__ out /* [3][n] FS9 */, uint32_t n, FsArg zeta_mont) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 3 * n) return;
    const uint32_t c = gid / n + 1, m = gid % n;
    fs_store9(out + (size_t)gid * L29, mul(fs_pow_u32_early(from_arg(zeta_mont), c * m), Fs::constant<Fr29Params::R2>()));
}
// not_last on the three cosets: out[c - 1][j] = zeta^(4 j + c) - w^(N-4)   (FS9)
__global__ void k_ring_not_last3(uint32_t* __restrict__ out /* [3][n] FS9 */, uint32_t n, FrArg zeta_mont256, FrArg last_root_mont256) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 3 * n) return;
    const uint32_t c = gid / n + 1, j = gid % n;
    fs_store9(out + (size_t)gid * L29, from_mont256(sub(fr_pow_u32(from_arg(zeta_mont256), 4 * j + c), from_arg(last_root_mont256))));
}

// ---- summation-by-parts commitments of the witness columns ------------------------------------------------------
// The witness columns are piecewise constant in EVALUATION form (one-hot / bit rows, an accumulator that changes at
// <= 254 rows), so   sum_j e_j * L_j(tau) G  =  sum_j (e_j - e_{j+1}) * PS_j   with  PS_j = sum_{i<=j} L_i(tau) G
// has ~270 non-zero scalars per column instead of N dense coefficients — the same group element, ~8x fewer
// bucket additions.  PS_j = sum_m S[j][m] * [tau^m] G with S[j][m] = (1/N) sum_{i<=j} w^(-i*m): one batched MSM over
// the monomial SRS at setup (k_ring_ps_scalars builds the N x N scalar matrix, lane m walks column m).
__global__ void k_ring_ps_scalars(uint32_t* __restrict__ out /* [N][N][8] std, row j = scalars of PS_j */, uint32_t n,
                                  FrArg winv_mont, FrArg inv_n_mont) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    Fr step = fr_pow_u32(from_arg(winv_mont), m);        // w^-m
    Fr term = from_arg(inv_n_mont);                      // (1/N) * w^(-i*m), i = 0
    Fr acc = Fr::zero();
#pragma unroll 1
    for (uint32_t j = 0; j < n; j++) {
        acc = add(acc, term);
        st_std(out + ((size_t)j * n + m) * 8, acc);
        term = mul(term, step);
    }
}
// first differences of the evaluation columns: d_j = e_j - e_{j+1} (e_N = 0); standard form in and out
__global__ void k_ring_diff(const uint32_t* __restrict__ cols /* [count][n][8] */, uint32_t n, size_t count, uint32_t* __restrict__ out) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count * n) return;
    uint32_t j = (uint32_t)(gid % n);
    Fr e = gload_fr(cols + gid * 8);
    Fr nx = j + 1 < n ? gload_fr(cols + (gid + 1) * 8) : Fr::zero();
    gstore_fr(out + gid * 8, sub(e, nx));
}

// ---- witness generation --------------------------------------------------------------------------------------
constexpr int RING_CHAIN = 256;    // seed, +PK_k, one per set blinding bit (<= 253), relation

struct RingConsts {
    uint32_t log2n, n, max_ring, rows;   // rows = n - 4
    FrArg seed_x, seed_y;                // accumulator base, Montgomery
    FrArg tail[4];                       // (X - w^-1)(X - w^-2)(X - w^-3) coefficients, Montgomery, low first
    FrArg omega;                         // w_N, Montgomery
    FrArg last_x;                        // w_N^(N-4), Montgomery
    FsArg tail9[4];                      // the tail coefficients and w_N again as FS9 arguments (Montgomery 2^261)
    FsArg omega9;
    FsArg nl_hidden[3];                  // w^(N-3+r) - w^(N-4), r = 0..2: the factor (x - w^(N-4)) at the three hidden rows
};


// The same chain with one WAVE per proof: lane l owns blinding bits 4l..4l+3.  Local sums of the selected bit points,
// an inclusive scan across the wave (6 shuffle steps), then every lane walks its own <= 4 additions from
// seed + PK_k + (sum of all lower lanes) — 16 dependent additions instead of ~128, and one division-step inversion per lane
// (lock-step, so it costs the time of one) to normalise the lane's own values.  Same outputs as k_ring_chain.
DR_DEV TePoint te_shfl_up(const TePoint& p, unsigned delta) {
    TePoint o;
#pragma unroll
    for (int t = 0; t < L29; t++) {
        o.x.l[t] = __shfl_up(p.x.l[t], delta, 64);
        o.y.l[t] = __shfl_up(p.y.l[t], delta, 64);
        o.z.l[t] = __shfl_up(p.z.l[t], delta, 64);
        o.t.l[t] = __shfl_up(p.t.l[t], delta, 64);
    }
    return o;
}
template <int CV>
__global__ __launch_bounds__(64) void k_ring_chain_wave(const uint32_t* __restrict__ ring_pts_mont, const uint32_t* __restrict__ producer_idx,
                                                        const uint32_t* __restrict__ blinding, RingConsts rc, uint32_t batch,
                                                        uint32_t* __restrict__ chain_ext /* B*256*32 scratch */,
                                                        uint32_t* __restrict__ chain_aff /* B*256*16 */, uint32_t* __restrict__ cnt_out) {
    const uint32_t pid = blockIdx.x, lane = threadIdx.x;
    if (pid >= batch) return;
    uint32_t* ext = chain_ext + (size_t)pid * RING_CHAIN * 32;
    uint32_t* aff = chain_aff + (size_t)pid * RING_CHAIN * 16;
    // (the curve arithmetic runs on Fs values, fr29.hip.h; the ring table, the seed and the affine results are in the 2^256
    //  Montgomery form the column kernels read: one product per coordinate at the boundary.  The scratch keeps packed words.)
    auto put = [&](uint32_t idx, const TePoint& p) {
        gstore_fr(ext + idx * 32, pack(p.x)); gstore_fr(ext + idx * 32 + 8, pack(p.y));
        gstore_fr(ext + idx * 32 + 16, pack(p.z)); gstore_fr(ext + idx * 32 + 24, pack(p.t));
    };
    auto ring_point = [&](uint32_t row) {
        TePoint p;
        p.x = from_mont256(gload_fr(ring_pts_mont + (size_t)row * 16));
        p.y = from_mont256(gload_fr(ring_pts_mont + (size_t)row * 16 + 8));
        p.z = Fs::one();
        p.t = mul(p.x, p.y);
        return p;
    };
    uint32_t t[8];
    {
        Fr tt = gload_fr(blinding + (size_t)pid * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = tt.l[j];
    }
    t[7] &= 0x1fffffffu;                                   // bits 0..252 only (k_ring_chain's loop bound)
    const uint32_t j0 = lane * 4;
    const uint32_t mine = (t[j0 >> 5] >> (j0 & 31)) & 0xfu;    // this lane's four bits
    // number of set bits below j0 -> chain index of this lane's first value
    uint32_t below = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t lo = w * 32;
        if (j0 >= lo + 32) below += __popc(t[w]);
        else if (j0 > lo) below += __popc(t[w] & ((1u << (j0 - lo)) - 1));
    }
    uint32_t total_bits = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) total_bits += __popc(t[w]);
    // 1. local sum of the selected bit points
    TePoint loc = te_identity();
#pragma unroll 1
    for (uint32_t b = 0; b < 4; b++)
        if ((mine >> b) & 1) loc = te_add<CV>(loc, ring_point(rc.max_ring + j0 + b));
    // 2. inclusive scan across the wave
    TePoint inc = loc;
#pragma unroll 1
    for (unsigned d = 1; d < 64; d <<= 1) {
        TePoint o = te_shfl_up(inc, d);
        if (lane >= d) inc = te_add<CV>(inc, o);
    }
    TePoint exc = te_shfl_up(inc, 1);                      // sum over all lower lanes
    if (lane == 0) exc = te_identity();
    // 3. seed, seed + PK_k, then this lane's own values
    TePoint seed;
    seed.x = from_mont256(from_arg(rc.seed_x)); seed.y = from_mont256(from_arg(rc.seed_y)); seed.z = Fs::one(); seed.t = mul(seed.x, seed.y);
    TePoint base = te_add<CV>(seed, ring_point(producer_idx[pid]));
    uint32_t idx[6];
    int nv = 0;
    if (lane == 0) { put(0, seed); put(1, base); idx[nv++] = 0; idx[nv++] = 1; }
    TePoint cur = te_add<CV>(base, exc);
    uint32_t pos = 2 + below;
#pragma unroll 1
    for (uint32_t b = 0; b < 4; b++)
        if ((mine >> b) & 1) {
            cur = te_add<CV>(cur, ring_point(rc.max_ring + j0 + b));
            put(pos, cur);
            idx[nv++] = pos++;
        }
    const uint32_t cnt = 2 + total_bits;
    if (lane == 63) {                                      // cur = the final accumulator value on the last lane
        put(cnt, te_add<CV>(cur, te_cneg(seed, true)));        // relation = result - seed
        idx[nv++] = cnt;
        cnt_out[pid] = cnt;
    }
    // 4. normalise this lane's values with one inversion
    Fs pre[6];
    Fs run = Fs::one();
#pragma unroll 1
    for (int i = 0; i < nv; i++) {
        pre[i] = run;
        run = mul(run, unpack(gload_fr(ext + idx[i] * 32 + 16)));
    }
    Fs inv_run = inv(run);                                 // lock-step across the wave (inv(1) on idle lanes)
#pragma unroll 1
    for (int i = nv - 1; i >= 0; i--) {
        const uint32_t* e = ext + idx[i] * 32;
        const Fs zi = mul(inv_run, pre[i]);
        inv_run = mul(inv_run, unpack(gload_fr(e + 16)));
        gstore_fr(aff + idx[i] * 16, to_mont256(mul(unpack(gload_fr(e)), zi)));
        gstore_fr(aff + idx[i] * 16 + 8, to_mont256(mul(unpack(gload_fr(e + 8)), zi)));
    }
}

// One lane per (proof, row): the four witness columns in transcript order  b, accip, accx, accy  (standard form).
// Rows [0, n-4] follow columns.py:111-146; the last three rows are the hidden rows (zk != NULL) or zero.
__global__ void k_ring_columns(const uint32_t* __restrict__ producer_idx, const uint32_t* __restrict__ blinding,
                               const uint32_t* __restrict__ chain_aff, const uint32_t* __restrict__ zk /* B*4*3*8 std or NULL */,
                               RingConsts rc, uint32_t batch, uint32_t* __restrict__ cols /* [B][4][n][8] */) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * rc.n) return;
    uint32_t pid = (uint32_t)(gid / rc.n), i = (uint32_t)(gid % rc.n);
    uint32_t* out = cols + (size_t)pid * 4 * rc.n * 8;
    Fr vb = Fr::zero(), vip = Fr::zero(), vx = Fr::zero(), vy = Fr::zero();
    if (i <= rc.rows) {
        uint32_t k = producer_idx[pid];
        uint32_t t[8];
        {
            Fr tt = gload_fr(blinding + (size_t)pid * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = tt.l[j];
        }
        // b[i]
        uint32_t bit = 0;
        if (i < rc.max_ring) bit = i == k;
        else if (i - rc.max_ring < 253) bit = (t[(i - rc.max_ring) >> 5] >> ((i - rc.max_ring) & 31)) & 1;
        if (i == rc.rows) bit = 0;                      // the appended padding bit
        vb.l[0] = bit;
        // number of additions applied before row i
        uint32_t adds = k < i ? 1u : 0u;
        if (i > rc.max_ring) {
            uint32_t nb = i - rc.max_ring;              // blinding bits j < nb are below row i
            if (nb > 253) nb = 253;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                uint32_t lo = w * 32;
                if (nb >= lo + 32) adds += __popc(t[w]);
                else if (nb > lo) adds += __popc(t[w] & ((1u << (nb - lo)) - 1));
            }
        }
        vip.l[0] = k < i ? 1u : 0u;
        const uint32_t* a = chain_aff + ((size_t)pid * RING_CHAIN + adds) * 16;
        vx = from_mont(gload_fr(a));
        vy = from_mont(gload_fr(a + 8));
    } else if (zk != nullptr) {
        uint32_t r = i - (rc.rows + 1);                 // hidden row 0..2
        const uint32_t* z = zk + (size_t)pid * 4 * 3 * 8;
        vb = gload_fr(z + (0 * 3 + r) * 8);
        vip = gload_fr(z + (1 * 3 + r) * 8);
        vx = gload_fr(z + (2 * 3 + r) * 8);
        vy = gload_fr(z + (3 * 3 + r) * 8);
    }
    gstore_fr(out + ((size_t)0 * rc.n + i) * 8, vb);
    gstore_fr(out + ((size_t)1 * rc.n + i) * 8, vip);
    gstore_fr(out + ((size_t)2 * rc.n + i) * 8, vx);
    gstore_fr(out + ((size_t)3 * rc.n + i) * 8, vy);
}

// result points for the host transcript: relation (x,y) and result+seed, standard form
__global__ void k_ring_relations(const uint32_t* __restrict__ chain_aff, const uint32_t* __restrict__ cnt, uint32_t batch,
                                 uint32_t* __restrict__ relation_std /* B*16 */, uint32_t* __restrict__ rps_mont /* B*16 */) {
    uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= batch) return;
    uint32_t c = cnt[pid];
    const uint32_t* rel = chain_aff + ((size_t)pid * RING_CHAIN + c) * 16;
    const uint32_t* last = chain_aff + ((size_t)pid * RING_CHAIN + c - 1) * 16;
    gstore_fr(relation_std + (size_t)pid * 16, from_mont(gload_fr(rel)));
    gstore_fr(relation_std + (size_t)pid * 16 + 8, from_mont(gload_fr(rel + 8)));
    gstore_fr(rps_mont + (size_t)pid * 16, gload_fr(last));
    gstore_fr(rps_mont + (size_t)pid * 16 + 8, gload_fr(last + 8));
}


// per proof: A = a5 seed_x + a6 seed_y, B = a5 r_x + a6 r_y + a7 for k_ring_constraints (computed on the 8-word field from the
// Montgomery-256 alphas, written as FS9 records)
__global__ void k_ring_alpha_aux(const uint32_t* __restrict__ alphas /* [B][7][8] Montgomery */, const uint32_t* __restrict__ rps_mont /* [B][16] */,
                                 RingConsts rc, uint32_t batch, uint32_t* __restrict__ out /* [B][2] FS9 */) {
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= batch) return;
    const uint32_t* al = alphas + (size_t)pid * 7 * 8;
    const Fr a5 = gload_fr(al + 4 * 8), a6 = gload_fr(al + 5 * 8), a7 = gload_fr(al + 6 * 8);
    const Fr rx = gload_fr(rps_mont + (size_t)pid * 16), ry = gload_fr(rps_mont + (size_t)pid * 16 + 8);
    fs_store9(out + (size_t)pid * 2 * L29, from_mont256(add(mul(a5, from_arg(rc.seed_x)), mul(a6, from_arg(rc.seed_y)))));
    fs_store9(out + ((size_t)pid * 2 + 1) * L29, from_mont256(add(add(mul(a5, rx), mul(a6, ry)), a7)));
}

// K7 on the three cosets (coset-major layouts, see above): one lane per (proof, coset, row); the shifted row is j + 1 of the same coset
template <int CV>
__global__ __launch_bounds__(256) void k_ring_constraints3(const uint32_t* __restrict__ wit3 /* [B][4][3][n] FS9: b, accip, accx, accy */,
                                                           const uint32_t* __restrict__ fixed3 /* [3][3][n] FS9: px, py, s */,
                                                           const uint32_t* __restrict__ lag3 /* [2][3][n] FS9: L0, Llast */,
                                                           const uint32_t* __restrict__ nl3 /* [3][n] FS9 */, const uint32_t* __restrict__ alphas,
                                                           const uint32_t* __restrict__ alpha_aux, RingConsts rc, uint32_t batch,
                                                           uint32_t* __restrict__ agg3 /* [B][3][n] FS9, raw sums */) {
    const uint32_t n = rc.n, m3 = 3 * n;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * m3) return;
    const uint32_t pid = (uint32_t)(gid / m3), r = (uint32_t)(gid % m3), c = r / n, j = r % n, jn = j + 1 == n ? 0 : j + 1;
    const uint32_t* w = wit3 + (size_t)pid * 4 * m3 * L29;
    auto at = [&](uint32_t col, uint32_t row) { return fs_load9(w + (((size_t)col * 3 + c) * n + row) * L29); };
    auto tab = [&](const uint32_t* t, uint32_t k) { return fs_load9(t + (((size_t)k * 3 + c) * n + j) * L29); };
    const uint32_t* al = alphas + (size_t)pid * 7 * L29;
    const uint32_t* ax = alpha_aux + (size_t)pid * 2 * L29;
    const Fs acc = body_constraints<CV>(at(0, j), at(1, j), at(1, jn), at(2, j), at(2, jn), at(3, j), at(3, jn), tab(fixed3, 0), tab(fixed3, 1),
                                        tab(fixed3, 2), tab(lag3, 0), tab(lag3, 1), fs_load9(nl3 + ((size_t)c * n + j) * L29), fs_load9(al),
                                        fs_load9(al + L29), fs_load9(al + 2 * L29), fs_load9(al + 3 * L29), fs_load9(al + 4 * L29), fs_load9(al + 5 * L29),
                                        fs_load9(al + 6 * L29), fs_load9(ax), fs_load9(ax + L29));
    fs_store9(agg3 + gid * L29, acc);
}
// coset 0 = H: the aggregated constraint polynomial is zero there except at the three hidden rows N-3 .. N-1 (the quotient's tail
// factor takes care of those).  k_ring_save_rows keeps the N-domain evaluations of rows N-3, N-2, N-1 and 0 of the four witness
// columns (standard form) before the coset transforms overwrite the evaluation columns; k_ring_hidden_rows, one lane per (proof,
// hidden row), runs the same body on them — the ring point of that row, selector 0, L0 = Llast = 0.
__global__ void k_ring_save_rows(const uint32_t* __restrict__ col_evals /* [B][4][n][8] std */, uint32_t n, uint32_t batch,
                                 uint32_t* __restrict__ hid /* [B][4][4][8]: rows n-3, n-2, n-1, 0 */) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= batch * 16) return;
    const uint32_t pid = gid / 16, col = (gid / 4) % 4, r = gid % 4, row = r == 3 ? 0 : n - 3 + r;
    gstore_fr(hid + (size_t)gid * 8, gload_fr(col_evals + (((size_t)pid * 4 + col) * n + row) * 8));
}
template <int CV>
__global__ void k_ring_hidden_rows(const uint32_t* __restrict__ hid /* [B][4][4][8] std */, const uint32_t* __restrict__ ring_pts_mont /* [n][16] Montgomery 256 */,
                                   const uint32_t* __restrict__ alphas /* [B][7] FS9 */, RingConsts rc, uint32_t batch,
                                   uint32_t* __restrict__ special /* [B][3] FS9 */) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= batch * 3) return;
    const uint32_t pid = gid / 3, r = gid % 3, n = rc.n, j = n - 3 + r;
    const uint32_t* w = hid + (size_t)pid * 16 * 8;
    // reduce_small: the body's witness inputs are the forward NTT's outputs elsewhere (|value| < 0.51 p)
    auto at = [&](uint32_t col, uint32_t idx) { return reduce_small(fs_from_std(gload_fr(w + ((size_t)col * 4 + idx) * 8))); };
    const Fs x2 = from_mont256(gload_fr(ring_pts_mont + (size_t)j * 16)), y2 = from_mont256(gload_fr(ring_pts_mont + (size_t)j * 16 + 8));
    const Fs zero = Fs::zero();
    const uint32_t* al = alphas + (size_t)pid * 7 * L29;
    const Fs acc = body_constraints<CV>(at(0, r), at(1, r), at(1, r + 1), at(2, r), at(2, r + 1), at(3, r), at(3, r + 1), carry(x2), carry(y2), zero, zero, zero,
                                        from_arg(rc.nl_hidden[r]), fs_load9(al), fs_load9(al + L29), fs_load9(al + 2 * L29), fs_load9(al + 3 * L29),
                                        fs_load9(al + 4 * L29), fs_load9(al + 5 * L29), fs_load9(al + 6 * L29), zero, zero);
    fs_store9(special + (size_t)gid * L29, acc);
}

// ---- K8: coefficient-space passes ---------------------------------------------------------------------------
// These kernels combine standard-form coefficient vectors with a few per-proof scalars.  A Montgomery product of a
// Montgomery-form scalar with a STANDARD-form value is the standard form of the product (x R * c / R = x c), and sums of
// standard-form values are standard-form: so the scalars (tail, k_i, nu_i, the evaluation point) are kept in Montgomery
// form — FS9 records or kernel arguments — and the coefficients are used exactly as they lie in memory: their 8 canonical words
// unpacked into limbs (20 instructions), no conversion per coefficient; a result is packed once (canon29_small / canon29).
DR_DEV Fs ld_coef(const uint32_t* p) { return unpack29(gload_fr(p).l); }            // a standard-form coefficient as limbs
DR_DEV void st_coef_small(uint32_t* p, const Fs& v) {                               // value in (-p, 3p) -> canonical words
    Fr o;
    canon29_small(v, o.l);
    gstore_fr(p, o);
}
DR_DEV Fs fs_pow_u32(const Fs& base, uint32_t e) { return fs_pow_u32_early(base, e); }
// quotient: c_agg = tail (cubic) * agg_poly ;  q_j = sum_{i>=1} c_agg[j + i*N],  j < 3N+1
__global__ void k_ring_quotient(const uint32_t* __restrict__ agg_poly /* [B][4N][8] std */, RingConsts rc, uint32_t batch,
                                uint32_t* __restrict__ q /* [B][3N+1][8] std */) {
    const uint32_t n = rc.n, m = 4 * n, qn = 3 * n + 1;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * qn) return;
    uint32_t pid = (uint32_t)(gid / qn), j = (uint32_t)(gid % qn);
    const uint32_t* a = agg_poly + (size_t)pid * m * 8;
    // regrouped: q_j = sum_d tail[d] * (sum_i a[j + i N - d]) — the four foldings first, then 4 products (two fused pairs)
    // instead of up to 16
    Fs fold[4];
#pragma unroll
    for (uint32_t d = 0; d < 4; d++) {
        fold[d] = Fs::zero();
#pragma unroll 1
        for (uint32_t i = 1; i <= 4; i++) {
            const uint32_t kidx = j + i * n;             // index into c_agg (length m + 3)
            if (kidx >= m + 3) break;
            if (kidx >= d && kidx - d < m) fold[d] = add(fold[d], ld_coef(a + (size_t)(kidx - d) * 8));
        }
    }
    st_coef_small(q + gid * 8, body_quotient(from_arg(rc.tail9[0]), fold[0], from_arg(rc.tail9[1]), fold[1], from_arg(rc.tail9[2]), fold[2],
                                             from_arg(rc.tail9[3]), fold[3]));
}

// Horner evaluation of `npoly` polynomials per proof at that proof's point: one workgroup per (proof, poly).
// poly p < nfixed comes from the per-ring array `fixed` (std, [nfixed][len]); the others from `perproof`
// ([B][nper][len] std).  out[pid*out_stride + out_off + p] (std).  A lane runs Horner over its slice with a lazy accumulator
// (one product + 29 instructions per coefficient), scales by x^lo, and the 256 partial values are added up in a tree through
// LDS (a carry per level, reduce_small after the fourth and the last level: ring_body / ring_bounds_check).
constexpr int EV_BLOCK = 256;
__global__ __launch_bounds__(EV_BLOCK) void k_ring_eval(const uint32_t* __restrict__ fixed, uint32_t nfixed,
                                                        const uint32_t* __restrict__ perproof, uint32_t nper, uint32_t len,
                                                        const uint32_t* __restrict__ points /* [B][8] std */, int mul_omega, RingConsts rc,
                                                        uint32_t* __restrict__ out, uint32_t out_stride, uint32_t out_off) {
    __shared__ int32_t red[EV_BLOCK * L29];
    const uint32_t pid = blockIdx.y, p = blockIdx.x;
    const uint32_t* src = p < nfixed ? fixed + (size_t)p * len * 8 : perproof + ((size_t)pid * nper + (p - nfixed)) * len * 8;
    Fs x = fs_from_std(gload_fr(points + (size_t)pid * 8));
    if (mul_omega) x = mul(x, from_arg(rc.omega9));
    const uint32_t per = (len + EV_BLOCK - 1) / EV_BLOCK;
    const uint32_t lo = threadIdx.x * per;
    Fs acc = Fs::zero();
    if (lo < len) {
        uint32_t hi = lo + per < len ? lo + per : len;
#pragma unroll 1
        for (int j = (int)hi - 1; j >= (int)lo; j--) acc = body_horner(acc, x, ld_coef(src + (size_t)j * 8));     // acc in standard form, x Montgomery
        acc = mul(acc, fs_pow_u32(x, lo));
    }
#pragma unroll
    for (int l = 0; l < L29; l++) red[l * EV_BLOCK + threadIdx.x] = acc.l[l];
    __syncthreads();
    int level = 0;
    for (int s = EV_BLOCK / 2; s > 0; s >>= 1, level++) {
        if ((int)threadIdx.x < s) {
            Fs a, b;
#pragma unroll
            for (int l = 0; l < L29; l++) { a.l[l] = red[l * EV_BLOCK + threadIdx.x]; b.l[l] = red[l * EV_BLOCK + threadIdx.x + s]; }
            a = carry(add(a, b));
            if (level == 3 || level == 7) a = reduce_small(a);
#pragma unroll
            for (int l = 0; l < L29; l++) red[l * EV_BLOCK + threadIdx.x] = a.l[l];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Fs r;
#pragma unroll
        for (int l = 0; l < L29; l++) r.l[l] = red[l * EV_BLOCK];
        st_coef_small(out + ((size_t)pid * out_stride + out_off + p) * 8, r);
    }
}

// linearisation scalars per proof (proof_builder.py:253-286): k0 = a0*term, k1 = a1*fx*term, k2 = a2*fy*term (computed on the
// 8-word field, written as FS9 records for k_ring_linpoly)
template <int CV>
__global__ void k_ring_lin_scalars(const uint32_t* __restrict__ evals /* [B][8][8] std: px,py,s,b,accip,accx,accy,(l) */,
                                   const uint32_t* __restrict__ alphas, const uint32_t* __restrict__ zetas, RingConsts rc,
                                   uint32_t batch, uint32_t* __restrict__ ks /* [B][3] FS9 */) {
    uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= batch) return;
    const uint32_t* e = evals + (size_t)pid * 8 * 8;
    Fr pxz = ld_std(e), pyz = ld_std(e + 8), bz = ld_std(e + 3 * 8), axz = ld_std(e + 5 * 8), ayz = ld_std(e + 6 * 8);
    Fr term = sub(ld_std(zetas + (size_t)pid * 8), from_arg(rc.last_x));
    Fr omb = sub(Fr::one(), bz);
    Fr fx = mul(add(mul(bz, add(mul(ayz, pyz), te_mul_a_fr<CV>(mul(axz, pxz)))), omb), term);
    Fr fy = mul(add(mul(bz, sub(mul(axz, pyz), mul(pxz, ayz))), omb), term);
    const uint32_t* al = alphas + (size_t)pid * 7 * 8;
    fs_store9(ks + ((size_t)pid * 3 + 0) * L29, from_mont256(mul(gload_fr(al), term)));          // alphas: Montgomery 256 (k_fr_to_mont)
    fs_store9(ks + ((size_t)pid * 3 + 1) * L29, from_mont256(mul(gload_fr(al + 8), fx)));
    fs_store9(ks + ((size_t)pid * 3 + 2) * L29, from_mont256(mul(gload_fr(al + 16), fy)));
}
// lin[j] = k0*accip[j] + k1*accx[j] + k2*accy[j]
__global__ void k_ring_linpoly(const uint32_t* __restrict__ cols /* [B][4][n][8] std coefficient columns */,
                               const uint32_t* __restrict__ ks /* [B][3] FS9 */, uint32_t n, uint32_t batch, uint32_t* __restrict__ lin /* [B][n][8] */) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * n) return;
    uint32_t pid = (uint32_t)(gid / n), j = (uint32_t)(gid % n);
    const uint32_t* c = cols + (size_t)pid * 4 * n * 8;
    const uint32_t* k = ks + (size_t)pid * 3 * L29;
    st_coef_small(lin + gid * 8, body_lin3(fs_load9(k), ld_coef(c + ((size_t)1 * n + j) * 8), fs_load9(k + L29), ld_coef(c + ((size_t)2 * n + j) * 8),
                                           fs_load9(k + 2 * L29), ld_coef(c + ((size_t)3 * n + j) * 8)));
}
// aggregated opening polynomial: sum of nu_i * poly_i over (px, py, s, b, accip, accx, accy, q)
__global__ void k_ring_aggpoly(const uint32_t* __restrict__ fixed /* [3][n][8] std */, const uint32_t* __restrict__ cols,
                               const uint32_t* __restrict__ q /* [B][3n+1][8] */, const uint32_t* __restrict__ nus /* [B][8] FS9 (converted once per batch) */,
                               uint32_t n, uint32_t batch, uint32_t* __restrict__ out /* [B][3n+1][8] */) {
    const uint32_t qn = 3 * n + 1;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * qn) return;
    uint32_t pid = (uint32_t)(gid / qn), j = (uint32_t)(gid % qn);
    const uint32_t* nu = nus + (size_t)pid * 8 * L29;
    const Fs qj = ld_coef(q + gid * 8);
    if (j >= n) {                                    // beyond degree N - 1 only the quotient contributes
        st_coef_small(out + gid * 8, mul(fs_load9(nu + 7 * L29), qj));
        return;
    }
    const uint32_t* c = cols + (size_t)pid * 4 * n * 8;
    Fs nv[8], cv[8];
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) nv[t] = fs_load9(nu + t * L29);
#pragma unroll
    for (uint32_t t = 0; t < 3; t++) cv[t] = ld_coef(fixed + ((size_t)t * n + j) * 8);
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) cv[3 + t] = ld_coef(c + ((size_t)t * n + j) * 8);
    cv[7] = qj;
    Fr o;
    canon29(body_agg8(nv, cv), o.l);
    gstore_fr(out + gid * 8, o);
}

// Synthetic division by (X - x): quotient Q_{i-1} = S_i with S_i = a_i + x*S_{i+1} (suffix Horner values).
// Chunked scan: pass 1 gives every chunk's local Horner value, pass 2 links the chunks serially per proof,
// pass 3 replays each chunk with its incoming value and writes the quotient.
constexpr uint32_t SD_CHUNK = 32;
__global__ void k_syndiv_local(const uint32_t* __restrict__ poly, uint32_t len, const uint32_t* __restrict__ points, int mul_omega,
                               RingConsts rc, uint32_t batch, uint32_t* __restrict__ chunk_val /* [B][nchunks][8] std */) {
    const uint32_t nch = (len + SD_CHUNK - 1) / SD_CHUNK;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * nch) return;
    uint32_t pid = (uint32_t)(gid / nch), ch = (uint32_t)(gid % nch);
    Fr x = ld_std(points + (size_t)pid * 8);
    if (mul_omega) x = mul(x, from_arg(rc.omega));
    uint32_t lo = ch * SD_CHUNK, hi = lo + SD_CHUNK < len ? lo + SD_CHUNK : len;
    const uint32_t* a = poly + (size_t)pid * len * 8;
    Fr acc = Fr::zero();
#pragma unroll 1
    for (int j = (int)hi - 1; j >= (int)lo; j--) acc = add(mul(acc, x), gload_fr(a + (size_t)j * 8));
    gstore_fr(chunk_val + gid * 8, acc);
}
__global__ void k_syndiv_link(uint32_t* __restrict__ chunk_val, uint32_t len, const uint32_t* __restrict__ points, int mul_omega,
                              RingConsts rc, uint32_t batch) {
    const uint32_t nch = (len + SD_CHUNK - 1) / SD_CHUNK;
    uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= batch) return;
    Fr x = ld_std(points + (size_t)pid * 8);
    if (mul_omega) x = mul(x, from_arg(rc.omega));
    Fr xc = fr_pow_u32(x, SD_CHUNK);
    uint32_t* cv = chunk_val + (size_t)pid * nch * 8;
    // after this pass cv[ch] = S at the END of chunk ch (the value entering it from above)
    Fr incoming = Fr::zero();
#pragma unroll 1
    for (int ch = (int)nch - 1; ch >= 0; ch--) {
        Fr local = gload_fr(cv + (size_t)ch * 8);
        gstore_fr(cv + (size_t)ch * 8, incoming);
        uint32_t lo = ch * SD_CHUNK, hi = lo + SD_CHUNK < len ? lo + SD_CHUNK : len;
        Fr shift = (hi - lo) == SD_CHUNK ? xc : fr_pow_u32(x, hi - lo);
        incoming = add(local, mul(shift, incoming));
    }
}
__global__ void k_syndiv_write(const uint32_t* __restrict__ poly, uint32_t len, const uint32_t* __restrict__ points, int mul_omega,
                               RingConsts rc, uint32_t batch, const uint32_t* __restrict__ chunk_val,
                               uint32_t* __restrict__ quot /* [B][len-1][8] std */) {
    const uint32_t nch = (len + SD_CHUNK - 1) / SD_CHUNK;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)batch * nch) return;
    uint32_t pid = (uint32_t)(gid / nch), ch = (uint32_t)(gid % nch);
    Fr x = ld_std(points + (size_t)pid * 8);
    if (mul_omega) x = mul(x, from_arg(rc.omega));
    uint32_t lo = ch * SD_CHUNK, hi = lo + SD_CHUNK < len ? lo + SD_CHUNK : len;
    const uint32_t* a = poly + (size_t)pid * len * 8;
    uint32_t* qo = quot + (size_t)pid * (len - 1) * 8;
    Fr s = gload_fr(chunk_val + gid * 8);
#pragma unroll 1
    for (int j = (int)hi - 1; j >= (int)lo; j--) {
        s = add(mul(s, x), gload_fr(a + (size_t)j * 8));      // S_j
        if (j >= 1) gstore_fr(qo + (size_t)(j - 1) * 8, s);
    }
}

}  // namespace dr
