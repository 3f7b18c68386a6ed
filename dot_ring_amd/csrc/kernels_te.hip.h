// Twisted Edwards kernels (seam A), part 1: the curve-templated scalar-multiplication kernels and the device helpers the
// ring prover's kernels share (kernels_ring.hip.h).  Part 2 — Elligator, square roots, GLV, point decoding, everything that
// reads the per-context constant block — is kernels_bsn.hip.h.  One lane = one variable-base scalar multiplication.
//
// k_bsn_scalar_mul: signed fixed 4-bit windows (digits in [-8,7], 64 windows over the 253-bit scalar),
// the per-lane table {1..8}P in LDS laid out [entry][word][lane] so that a data-dependent entry index
// still hits bank = lane (conflict-free ds_read_b32), 3 of every 4 doublings skip the T coordinate,
// final affine conversion by one division-step inversion per lane (fr29.hip.h).  Replaces the reference's GLV + joint 2-bit
// window kernel (dot_ring/curve/native_field/bandersnatch_te.pyx:480-554 via specs/bandersnatch.py:177-191):
// the output is the canonical affine point, so the different window schedule is invisible in the bytes.
#pragma once
#include "curve.hip.h"

namespace dr {

constexpr int BSN_BLOCK = 64;          // one wave per workgroup: 64 KiB of LDS table per wave
constexpr int BSN_TABLE = 8;           // entries 1P..8P
constexpr int BSN_PT_WORDS = 32;       // X,Y,Z,T x 8 limbs

// (the table keeps canonical 8-word coordinates: 64 KiB per wave as before; packing costs ~120 instructions per coordinate
//  when the 8 entries are built, unpacking ~20 per coordinate per lookup — against ~2200 for the addition that follows)
DR_DEV void lds_store_point(uint32_t* tab, int entry, int lane, const TePoint& p) {
    uint32_t* base = tab + (size_t)entry * BSN_PT_WORDS * BSN_BLOCK + lane;
    const Fr x = pack(p.x), y = pack(p.y), z = pack(p.z), t = pack(p.t);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        base[(0 + i) * BSN_BLOCK] = x.l[i];
        base[(8 + i) * BSN_BLOCK] = y.l[i];
        base[(16 + i) * BSN_BLOCK] = z.l[i];
        base[(24 + i) * BSN_BLOCK] = t.l[i];
    }
}
DR_DEV TePoint lds_load_point(const uint32_t* tab, int entry, int lane) {
    const uint32_t* base = tab + (size_t)entry * BSN_PT_WORDS * BSN_BLOCK + lane;
    Fr x, y, z, t;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x.l[i] = base[(0 + i) * BSN_BLOCK];
        y.l[i] = base[(8 + i) * BSN_BLOCK];
        z.l[i] = base[(16 + i) * BSN_BLOCK];
        t.l[i] = base[(24 + i) * BSN_BLOCK];
    }
    TePoint p;
    p.x = unpack(x); p.y = unpack(y); p.z = unpack(z); p.t = unpack(t);
    return p;
}

DR_DEV Fr load_fr_std(const uint32_t* p) {
    Fr r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
DR_DEV void store_fr_std(uint32_t* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

DR_DEV TePoint te_shfl_down(const TePoint& p, unsigned delta) {
    TePoint o;
#pragma unroll
    for (int t = 0; t < L29; t++) {
        o.x.l[t] = __shfl_down(p.x.l[t], delta, 64);
        o.y.l[t] = __shfl_down(p.y.l[t], delta, 64);
        o.z.l[t] = __shfl_down(p.z.l[t], delta, 64);
        o.t.l[t] = __shfl_down(p.t.l[t], delta, 64);
    }
    return o;
}

// k mod n for a 256-bit k by conditional subtractions: floor(2^256 / n) = 8 for Bandersnatch (n > 2^252), 17 for JubJub
// (n > 2^251)
template <int CV = CV_BANDERSNATCH>
DR_DEV void reduce_mod_order(uint32_t (&k)[8]) {
    constexpr uint32_t ORD_B[8] = {0x2876e7e1u, 0x74fd06b5u, 0x74190471u, 0xff8f8700u,
                                   0x02687600u, 0x0cce7602u, 0xca675f52u, 0x1cfb69d4u};
    constexpr uint32_t ORD_J[8] = {0xd6f72cb7u, 0xd0970e5eu, 0xccc81082u, 0xa6682093u,
                                   0x01343b00u, 0x06673b01u, 0x6533afa9u, 0x0e7db4eau};
    constexpr int ROUNDS = CV == CV_JUBJUB ? 18 : 9;
#pragma unroll 1
    for (int it = 0; it < ROUNDS; it++) {
        uint32_t d[8], borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = subb(k[i], CV == CV_JUBJUB ? ORD_J[i] : ORD_B[i], borrow);
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = borrow ? k[i] : d[i];
    }
}

// scalar multiplication core shared by the batch and the grouped-MSM kernels: returns k*P (extended coords)
template <int CV = CV_BANDERSNATCH>
DR_DEV TePoint bsn_scalar_mul_core(uint32_t* tab, int lane, const Fs& px, const Fs& py, uint32_t (&k)[8]) {
    TePoint P;
    P.x = px; P.y = py; P.z = Fs::one(); P.t = mul(px, py);
    // table 1P..8P
    lds_store_point(tab, 0, lane, P);
    TePoint Q = te_dbl<true, CV>(P);
    lds_store_point(tab, 1, lane, Q);
#pragma unroll 1
    for (int e = 2; e < BSN_TABLE; e++) {
        Q = te_add<CV>(Q, P);
        lds_store_point(tab, e, lane, Q);
    }
    // signed recoding, LSB first: nibble + carry in [0,16]; >= 8 -> minus 16 with carry
    uint32_t dig[8];   // 64 digits, stored as (d + 8) in 4 bits
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = ((k[w] >> (4 * j)) & 15u) + carry;
            carry = v >= 8u ? 1u : 0u;
            uint32_t d = (v + 8u) & 15u;     // (v - 16*carry) + 8
            packed |= d << (4 * j);
        }
        dig[w] = packed;
    }
    // (k < n < 2^253: the top nibble is <= 1, so the final carry is 0)
    TePoint acc = te_identity();
#pragma unroll 1
    for (int w = 63; w >= 0; w--) {
#pragma unroll 1
        for (int j = 0; j < 3; j++) acc = te_dbl<false, CV>(acc);   // rolled: keeps the loop body inside the I-cache
        acc = te_dbl<true, CV>(acc);
        int d = (int)((dig[w >> 3] >> (4 * (w & 7))) & 15u) - 8;
        int mag = d < 0 ? -d : d;
        TePoint T = lds_load_point(tab, mag == 0 ? 0 : mag - 1, lane);
        T = te_cneg(T, d < 0);
        if (mag == 0) T = te_identity();
        acc = te_add<CV>(acc, T);
    }
    return acc;
}

DR_DEV void te_store_affine(uint32_t* out, const TePoint& acc) {
    const Fs zi = inv(acc.z);
    store_fr_std(out, fs_to_std(mul(acc.x, zi)));
    store_fr_std(out + 8, fs_to_std(mul(acc.y, zi)));
}

// out[i] = k[i] * P[i].  pts: n x 16 u32 (x||y, standard form LE), ks: n x 8 u32, out: n x 16 u32.
template <int CV>
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_scalar_mul(const uint32_t* __restrict__ pts,
                                                              const uint32_t* __restrict__ ks,
                                                              uint32_t* __restrict__ out, uint32_t n) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    uint32_t i = blockIdx.x * BSN_BLOCK + lane;
    const bool live = i < n;
    if (!live) i = n - 1;             // keep the wave converged; the duplicate result is not stored
    Fs px = fs_from_std(load_fr_std(pts + (size_t)i * 16));
    Fs py = fs_from_std(load_fr_std(pts + (size_t)i * 16 + 8));
    uint32_t k[8];
    {
        Fr kk = load_fr_std(ks + (size_t)i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = kk.l[j];
    }
    reduce_mod_order<CV>(k);
    TePoint acc = bsn_scalar_mul_core<CV>(tab, lane, px, py, k);
    if (live) te_store_affine(out + (size_t)i * 16, acc);
}

// High-occupancy variant for large batches: signed 2-bit windows, table {P, 2P} = 16 KiB of LDS per wave instead of
// 64 KiB, i.e. 10 resident waves per CU instead of 2.  18 % more field products per scalar multiplication (128
// additions instead of 64 + table), but the 4-bit kernel leaves half the SIMDs empty and the rest with one wave:
// from ~32 k scalar multiplications per launch this one is ~3x faster; below that the shorter chain of the 4-bit
// kernel wins (both are latency-bound there).
constexpr int BSN2_TABLE = 2;
template <int CV>
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_scalar_mul_w2(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ ks,
                                                                 uint32_t* __restrict__ out, uint32_t n) {
    __shared__ uint32_t tab[BSN2_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    uint32_t i = blockIdx.x * BSN_BLOCK + lane;
    const bool live = i < n;
    if (!live) i = n - 1;
    TePoint P;
    P.x = fs_from_std(load_fr_std(pts + (size_t)i * 16));
    P.y = fs_from_std(load_fr_std(pts + (size_t)i * 16 + 8));
    P.z = Fs::one();
    P.t = mul(P.x, P.y);
    uint32_t k[8];
    {
        Fr kk = load_fr_std(ks + (size_t)i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = kk.l[j];
    }
    reduce_mod_order<CV>(k);
    lds_store_point(tab, 0, lane, P);
    lds_store_point(tab, 1, lane, te_dbl<true, CV>(P));
    // signed recoding, LSB first: pair + carry in [0,4]; >= 2 -> minus 4 with carry; digits stored as (d + 2) in 2 bits.
    // k < n < 2^253: the top pairs are 0, so the final carry is absorbed (bits 252..253 -> at most 1 + carry = 2 -> d = -2,
    // carry into pair 127 which is 0 -> 1).
    uint32_t dig[8];
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t v = ((k[w] >> (2 * j)) & 3u) + carry;
            carry = v >= 2u ? 1u : 0u;
            packed |= ((v + 2u) & 3u) << (2 * j);            // (v - 4*carry) + 2
        }
        dig[w] = packed;
    }
    TePoint acc = te_identity();
#pragma unroll 1
    for (int w = 127; w >= 0; w--) {
        acc = te_dbl<false, CV>(acc);
        acc = te_dbl<true, CV>(acc);
        int d = (int)((dig[w >> 4] >> (2 * (w & 15))) & 3u) - 2;
        int mag = d < 0 ? -d : d;
        TePoint T = lds_load_point(tab, mag == 2 ? 1 : 0, lane);
        T = te_cneg(T, d < 0);
        if (mag == 0) T = te_identity();
        acc = te_add<CV>(acc, T);
    }
    if (live) te_store_affine(out + (size_t)i * 16, acc);
}

// out[g] = sum_{j<m} k[g*m+j] * P[g*m+j]: the lanes of a group (m a power-of-two-padded width <= 64) each do
// one scalar multiplication, then the group is folded with wave shuffles.  Covers msm-2/3/4 of the sigma
// protocols (bandersnatch_te.pyx:557,669) and, with one group, small Pippenger inputs (:257).
template <int CV>
__global__ __launch_bounds__(BSN_BLOCK) void k_bsn_msm_groups(const uint32_t* __restrict__ pts,
                                                              const uint32_t* __restrict__ ks,
                                                              uint32_t* __restrict__ out, uint32_t groups,
                                                              uint32_t m, uint32_t mpad) {
    __shared__ uint32_t tab[BSN_TABLE * BSN_PT_WORDS * BSN_BLOCK];
    const int lane = threadIdx.x;
    const uint32_t per_block = BSN_BLOCK / mpad;
    const uint32_t g = blockIdx.x * per_block + lane / mpad;
    const uint32_t j = lane % mpad;
    const bool live = g < groups && j < m;
    TePoint acc = te_identity();
    // every lane runs the core (wave-uniform control flow); dead lanes recompute element 0 and are masked out
    size_t idx = live ? (size_t)g * m + j : 0;
    Fs px = fs_from_std(load_fr_std(pts + idx * 16));
    Fs py = fs_from_std(load_fr_std(pts + idx * 16 + 8));
    uint32_t k[8];
    {
        Fr kk = load_fr_std(ks + idx * 8);
#pragma unroll
        for (int t = 0; t < 8; t++) k[t] = kk.l[t];
    }
    reduce_mod_order<CV>(k);
    TePoint r = bsn_scalar_mul_core<CV>(tab, lane, px, py, k);
    if (live) acc = r;
    // fold within the group: lane j += lane j+s
#pragma unroll 1
    for (uint32_t s = mpad >> 1; s > 0; s >>= 1) {
        acc = te_add<CV>(acc, te_shfl_down(acc, s));
    }
    if (g < groups && j == 0) te_store_affine(out + (size_t)g * 16, acc);
}

// ---- Elligator 2 hash-to-curve, device side --------------------------------------------------------------------
// out = clear_cofactor(map(u0) + map(u1))  — the field work of TEAffinePoint._e2c_ell2_ro
// (dot_ring/curve/twisted_edwards/te_affine_point.py:212-295, te_curve.py:48-95); hash_to_field stays on the host.
// One lane per input.  Square roots: Tonelli-Shanks with p-1 = Q*2^32 and the non-residue 5, as the reference's
// sqrt_mod_bls_scalar_cy (bandersnatch_te.pyx:421); which root comes out is irrelevant (the map fixes the sign).
// a^e for a 256-bit exponent given as plain limbs, the same for every lane (control flow is scalar): MSB-first with sliding windows
// of up to three bits over the odd powers a, a^3, a^5, a^7 — one product per ~4 exponent bits instead of one per 2
DR_DEV Fs fr_pow_limbs(const Fs& a, const uint32_t (&e)[8]) {
    const Fs a2 = sqr(a), a3 = mul(a, a2), a5 = mul(a3, a2), a7 = mul(a5, a2);
    auto bit = [&](int i) -> uint32_t { return (e[i >> 5] >> (i & 31)) & 1u; };
    int i = 255;
    while (i >= 0 && !bit(i)) i--;
    if (i < 0) return Fs::one();
    Fs r = Fs::one();
    bool started = false;
#pragma unroll 1
    while (i >= 0) {
        if (!bit(i)) { r = sqr(r); i--; continue; }
        int l = i >= 2 ? 3 : i + 1;
        while (!bit(i - l + 1)) l--;
        uint32_t v = 0;
        for (int k = 0; k < l; k++) v = (v << 1) | bit(i - k);
        if (started) {
#pragma unroll 1
            for (int k = 0; k < l; k++) r = sqr(r);
        }
        Fs m;
#pragma unroll
        for (int t = 0; t < L29; t++) m.l[t] = v == 1 ? a.l[t] : v == 3 ? a3.l[t] : v == 5 ? a5.l[t] : a7.l[t];
        r = started ? mul(r, m) : m;
        started = true;
        i -= l;
    }
    return r;
}



// acc + (x2, y2, d*t2) with Z2 = 1  (add-2008-hwcd: D = Z1, C = T1 * d t2)
template <int CV>
DR_DEV TePoint te_madd(const TePoint& p, const Fs& x2, const Fs& y2, const Fs& dt2) {
    const Fs A = mul(p.x, x2), B = mul(p.y, y2), C = mul(p.t, dt2);
    const Fs E = mul2(p.x, y2, p.y, x2);
    const Fs F = sub(p.z, C), G = add(p.z, C), H = te_b_minus_aA<CV>(A, B);      // bounds as in te_add (D = Z1)
    TePoint r;
    r.x = mul(E, F);
    r.y = mul(G, H);
    r.t = mul(E, H);
    r.z = mul(F, G);
    return r;
}


// ---- fixed-base scalar multiplication -------------------------------------------------------------------------------
// The sigma protocols multiply two CONSTANT points all the time — the generator G and the Pedersen blinding base B
// (vrf/pedersen/vrf.py:94,104,111: x*G + b*B, k*G + k_b*B; pk = sk*G) — and their launches are latency chains, not
// throughput: a variable-base multiplication is ~250 dependent doublings.  With a table of every window multiple,
//     table[w][e] = (e + 1) * 16^w * P   (w < 64, e < 8; affine, as (x, y, d x y) in Montgomery form, canonical words: 48 KB per base),
// k*P is the sum of 64 signed table entries — no doublings — and the sum splits over 4 lanes of 16 windows each plus two
// shuffle additions: a dependent chain of 18 additions instead of ~320 operations.
constexpr int TE_FIXED_WINDOWS = 64, TE_FIXED_ENTRIES = 8, TE_FIXED_LANES = 4;
constexpr int TE_FIXED_TABLE_WORDS = TE_FIXED_WINDOWS * TE_FIXED_ENTRIES * 24;

// one block of 64 lanes: lane w derives 16^w * P by 4w doublings, then its 8 multiples, each normalised to affine
template <int CV>
__global__ __launch_bounds__(64) void k_te_fixed_table(const uint32_t* __restrict__ base_xy /* 16 words std */, uint32_t* __restrict__ table) {
    const int w = threadIdx.x;
    TePoint P;
    P.x = fs_from_std(load_fr_std(base_xy)); P.y = fs_from_std(load_fr_std(base_xy + 8)); P.z = Fs::one(); P.t = mul(P.x, P.y);
#pragma unroll 1
    for (int i = 0; i < 4 * w; i++) P = te_dbl<true, CV>(P);
    TePoint cur = P;
#pragma unroll 1
    for (int e = 0; e < TE_FIXED_ENTRIES; e++) {
        if (e > 0) cur = te_add<CV>(cur, P);
        const Fs zi = inv(cur.z);
        const Fs x = mul(cur.x, zi), y = mul(cur.y, zi);
        uint32_t* o = table + ((size_t)w * TE_FIXED_ENTRIES + e) * 24;
        store_fr_std(o, pack(x));
        store_fr_std(o + 8, pack(y));
        store_fr_std(o + 16, pack(mul(te_d_mont<CV>(), mul(x, y))));
    }
}

struct TeFixedTables {              // up to 4 bases per group (kernel argument)
    const uint32_t* t[4];
};

// out[g] = sum_{j<m} k[g*m + j] * Base_j for `groups` groups; term j of every group multiplies base j (m <= 4).
// Lane layout inside a wave: group-major, then term, then the 4 window quarters; a group is mpad * 4 lanes (mpad = m
// rounded up to a power of two), folded with shuffles.
template <int CV>
__global__ __launch_bounds__(64) void k_te_fixed_base_groups(TeFixedTables tabs, const uint32_t* __restrict__ ks, uint32_t* __restrict__ out,
                                                             uint32_t groups, uint32_t m, uint32_t mpad) {
    const uint32_t lane = threadIdx.x;
    const uint32_t width = mpad * TE_FIXED_LANES;                    // lanes per group: 4, 8 or 16
    const uint32_t per_block = 64 / width;
    const uint32_t g = blockIdx.x * per_block + lane / width;
    const uint32_t j = (lane % width) / TE_FIXED_LANES, q = lane % TE_FIXED_LANES;
    const bool live = g < groups && j < m;
    uint32_t k[8];
    {
        const Fr kk = load_fr_std(ks + (live ? (size_t)g * m + j : 0) * 8);
#pragma unroll
        for (int t = 0; t < 8; t++) k[t] = kk.l[t];
    }
    reduce_mod_order<CV>(k);
    // signed 4-bit recoding of the whole scalar (the carry chain starts at the bottom), as in bsn_scalar_mul_core
    uint32_t dig[8];
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v = ((k[w] >> (4 * i)) & 15u) + carry;
            carry = v >= 8u ? 1u : 0u;
            packed |= ((v + 8u) & 15u) << (4 * i);
        }
        dig[w] = packed;
    }
    const uint32_t* table = tabs.t[live ? j : 0];
    TePoint acc = te_identity();
    // this lane's 16 windows are the two digit words 2q, 2q + 1 (selected without indexing the register array)
    const uint32_t d_lo = q == 0 ? dig[0] : q == 1 ? dig[2] : q == 2 ? dig[4] : dig[6];
    const uint32_t d_hi = q == 0 ? dig[1] : q == 1 ? dig[3] : q == 2 ? dig[5] : dig[7];
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
        const uint32_t word = i < 8 ? d_lo : d_hi;
        const int d = (int)((word >> (4 * (i & 7))) & 15u) - 8;
        if (d == 0) continue;
        const int mag = d < 0 ? -d : d;
        const uint32_t* e = table + ((size_t)(16 * q + i) * TE_FIXED_ENTRIES + (mag - 1)) * 24;
        Fs x = unpack(load_fr_std(e)), dt = unpack(load_fr_std(e + 16));
        const Fs y = unpack(load_fr_std(e + 8));
        x = cneg(x, d < 0);
        dt = cneg(dt, d < 0);
        acc = te_madd<CV>(acc, x, y, dt);
    }
    if (!live) acc = te_identity();
#pragma unroll 1
    for (uint32_t s = width >> 1; s > 0; s >>= 1) acc = te_add<CV>(acc, te_shfl_down(acc, s));
    if (g < groups && (lane % width) == 0) te_store_affine(out + (size_t)g * 16, acc);
}

}  // namespace dr
