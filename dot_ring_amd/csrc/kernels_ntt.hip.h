// Batched radix-2 NTT over Fr (seam C, kernel K6).  Replaces BlsScalarNTTPlan._transform_core
// (dot_ring/ring_proof/polynomial/ntt.pyx:116-163) + bls_scalar_ntt_round (bls12_381_scalar.c:333-356):
// out[i] = sum_j in[j] * omega^(i*j), natural order in and out, optional scaling of every output.
//
// Structure: decimation in time over the bit-reversed input, exactly the butterfly network of the reference, staged through
// LDS: pass A loads a tile of 2^10 consecutive positions of the bit-reversed array (a gather of whole elements), runs stages
// 1..10 in LDS (limb-major layout: lane i touches word [limb][i], conflict-free) and streams the tile out; each later pass takes
// the next <= 4 index bits as "rows" of a (rows x 64 columns) tile so that global accesses stay 2 KiB-contiguous.
//
// Arithmetic (round 3): the UNSATURATED field of fr29.hip.h — nine signed 29-bit limbs, Montgomery R = 2^261, lazy reduction.
// A butterfly is one 206-instruction product, nine additions, nine subtractions and ~1.5 carry passes (25 instructions each)
// where the saturated 8 x 32-bit field needed a 305-instruction product and two ~24-instruction modular add / sub: ~262
// against ~353 instructions.  Nothing is canonicalised between stages:
//   * LDS tiles and the intermediate arrays between passes hold the nine limbs RAW (36 bytes per element);
//   * values grow by at most 1.04 p per stage along a run of "upper" butterfly inputs — after 24 stages still below 26 p, and a
//     product takes operands up to 33 p (|a| |b| <= 35 p^2 with a normal twiddle);
//   * limbs: the operand of the twiddle product is carried on load in every stage, the other operand in every second stage, so a
//     limb is the sum of at most three 29-bit values and a carried one (|limb| < 2^31);
//   * the last stage's outputs leave through a product (scaled / standard-form output: mul by the factor, then the canonical
//     8 words) or through reduce_small (raw 9-limb output for the constraint kernel: |value| < 0.51 p).
// Element formats at the kernel boundary: STD8 = 8 canonical words, standard form (the C ABI, coefficient arrays); FS9 = raw
// 9-limb records in Montgomery form (prover-internal: the 4N-domain evaluations).  Twiddles omega^j, j < n/2, are a per-(n, omega)
// table of 48-byte records (9 limbs + padding, three 16-byte loads) in HBM, cached in the context.
// Radix-4 butterflies would not save products here: a prime field has no free multiplication by the fourth root of unity, so the
// four points of two fused stages cost four products either way; what fusing saves is LDS traffic, which is not the bound.
#pragma once
#include "dev_types.hpp"
#include "curve.hip.h"
#include "hostmath.hpp"
#include "ring_body.hip.h"

namespace dr {

constexpr int NTT_LOG_TILE = 10, NTT_TILE = 1 << NTT_LOG_TILE, NTT_BLOCK = 256;
constexpr int NTT_STRIDED_TILE = 1024, NTT_MIN_COLS = 64, NTT_MAX_ROW_BITS = 4;
constexpr int NTT_TW_WORDS = 12;                     // a twiddle record: 9 limbs + 3 words of padding (16-byte loads)
enum { NTT_FMT_STD8 = 0, NTT_FMT_FS9 = 1, NTT_FMT_STD8_SCALED = 2, NTT_FMT_FS9_COSETS = 3 };
// Where pass A finds element `idx` (natural order) of transform `xform`:
//   STD8 / FS9       base[xform][idx]                                           (n >> pad elements per transform)
//   STD8_SCALED      base[xform / div][idx] * scale[xform % div][idx]           — `div` transforms read ONE coefficient vector, each with
//                    its own table of multipliers (FS9, premultiplied by R^2: the product that converts to Montgomery form applies
//                    it): the evaluations of a polynomial on the cosets zeta^c H are NTT_N(a_m zeta^(c m))
//   FS9_COSETS       the 4N-point vector whose coset c = idx mod 4, row j = idx / 4 lives coset-major in base[xform][c - 1][j] for
//                    c = 1..3; coset 0 is zero except its last three rows, special[xform][0..2] (the aggregated constraint polynomial
//                    vanishes on H outside the hidden rows)
struct NttSource {
    const uint32_t* base;
    const uint32_t* scale;
    const uint32_t* special;
    int fmt;
    uint32_t div;
};

template <int STRIDE = NTT_TILE>
DR_DEV Fs lds_get9(const int32_t* t, int i) {
    Fs r;
#pragma unroll
    for (int l = 0; l < L29; l++) r.l[l] = t[l * STRIDE + i];
    return r;
}
template <int STRIDE = NTT_TILE>
DR_DEV void lds_put9(int32_t* t, int i, const Fs& v) {
#pragma unroll
    for (int l = 0; l < L29; l++) t[l * STRIDE + i] = v.l[l];
}
DR_DEV Fr gload_fr(const uint32_t* p) {
    Fr r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
DR_DEV void gstore_fr(uint32_t* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
DR_DEV Fs tw_load(const uint32_t* p) {              // a 48-byte twiddle record
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    Fs r;
    r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
    r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
    r.l[8] = (int32_t)c.x;
    return r;
}

struct FrArg {   // kernel-argument copy of a field element (8 words: whatever form the caller says)
    uint32_t l[8];
};
DR_DEV Fr from_arg(const FrArg& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}

// element `idx` of transform `xform` (n_src elements per source vector) -> Fs, Montgomery form
DR_DEV Fs ntt_load(const NttSource& in, size_t xform, size_t idx, size_t n_src) {
    if (in.fmt == NTT_FMT_FS9) return fs_load9(in.base + (xform * n_src + idx) * L29);
    if (in.fmt == NTT_FMT_STD8) return fs_from_std(gload_fr(in.base + (xform * n_src + idx) * 8));
    if (in.fmt == NTT_FMT_STD8_SCALED) {
        const size_t poly = xform / in.div, c = xform % in.div;
        return mul(unpack29(gload_fr(in.base + (poly * n_src + idx) * 8).l), fs_load9(in.scale + (c * n_src + idx) * L29));
    }
    // NTT_FMT_FS9_COSETS
    const size_t c = idx & 3, j = idx >> 2, nq = n_src >> 2;
    if (c != 0) return fs_load9(in.base + ((xform * 3 + (c - 1)) * nq + j) * L29);
    if (j + 3 >= nq) return fs_load9(in.special + (xform * 3 + (j + 3 - nq)) * L29);
    return Fs::zero();
}
// What a pass leaves behind.  Not the last pass: the raw limbs.  The last pass — STD8: v * factor with the factor in STANDARD form
// (Montgomery x standard -> standard), canonical words; FS9: v * factor (Montgomery) or, without a factor, reduce_small(v).
// (Staging the 36-byte records through the tile as whole 16-byte chunks per lane was measured and dropped: the strided pass went
// from 1.26 to 1.40 ms per 1024 proofs — the limb-major scatter in LDS costs more than nine 4-byte accesses per record.)
DR_DEV void ntt_store_std8(uint32_t* base, size_t idx, const Fs& v, const Fs& factor) {
    Fr o;
    canon29_small(mul(carry(v), factor), o.l);
    gstore_fr(base + idx * 8, o);
}
DR_DEV Fs ntt_final_fs9(const Fs& v, int has_factor, const Fs& factor) { return has_factor ? mul(carry(v), factor) : reduce_small(v); }

// tw[j] = omega^j (Montgomery 2^261), j < count, as 12-word records
__global__ void k_ntt_twiddles(uint32_t* tw, uint32_t count, FsArg omega_mont) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fs w = from_arg(omega_mont), r = Fs::one();
    for (uint32_t e = j; e; e >>= 1) {
        if (e & 1) r = mul(r, w);
        w = sqr(w);
    }
    uint32_t* o = tw + (size_t)j * NTT_TW_WORDS;
#pragma unroll
    for (int i = 0; i < L29; i++) o[i] = (uint32_t)r.l[i];
    o[9] = o[10] = o[11] = 0;
}

// one butterfly on (u, v) with twiddle w (trivial: w = 1, no product); carry_u: also carry the upper operand (every second
// stage) — ring_body.hip.h: body_butterfly, whose limb and value bounds the host interval check walks through 24 stages
DR_DEV void ntt_butterfly(Fs& u, Fs& v, const Fs& w, bool trivial, bool carry_u) { body_butterfly(u, v, w, trivial, carry_u); }

// columns of a strided-pass tile: 1024 / rows, at least 64 (host and device agree through this one function)
__host__ __device__ inline int ntt_strided_col_bits(int row_bits) { return row_bits >= 4 ? 6 : 10 - row_bits; }

// Pass A: stages 1..S (S = min(k, LOG_TILE)) on tile `blockIdx.x` of transform `blockIdx.y`.  Two instances: 2^10 elements per tile
// (36 KB of LDS, 256 lanes) and 2^11 (72 KB, 512 lanes) — the second one for k = 11, the prover's domain at ring 1024, whose 16
// transforms per proof then take ONE pass over HBM instead of a ten-stage pass plus a one-stage strided pass (round 4: NTT kernels
// 3.08 -> 2.95 ms per 1024 proofs; with 256 lanes per transform 3.1, with 1024 lanes 3.4 — the eleventh stage costs the local pass
// twice a normal stage: 1024 distinct twiddle records per workgroup).
// src is read at bit-reversed positions, dst written contiguously (src == dst is allowed only when S == k, where one workgroup
// owns the whole transform and the formats have the same element size or the whole tile is loaded before anything is stored).
template <int LOG_TILE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ntt_local(NttSource in, uint32_t* dst,
                                                         const uint32_t* __restrict__ tw, int k, int S, int final_pass, int fmt_out,
                                                         int has_factor, FsArg out_factor, int pad) {
    constexpr int TILE = 1 << LOG_TILE;
    __shared__ int32_t tile[L29 * TILE];
    const size_t n = (size_t)1 << k;
    const int tsize = 1 << S;
    const size_t xform = blockIdx.y, tbase = (size_t)blockIdx.x * tsize;
    const size_t n_src = n >> pad;
    uint32_t* out = dst + xform * n * (final_pass && fmt_out == NTT_FMT_STD8 ? 8 : L29);
    if (pad == 0) {
        for (int i = threadIdx.x; i < tsize; i += BLOCK) {
            size_t pos = tbase + i;
            size_t rev = (size_t)(__brevll((unsigned long long)pos) >> (64 - k));
            lds_put9<TILE>(tile, i, ntt_load(in, xform, rev, n_src));
        }
    } else {
        // the input is a polynomial of n / 2^pad coefficients, zero-padded to n: in bit-reversed order only every 2^pad-th
        // position is non-zero (the low `pad` bits of a position are the top bits of its source index), and the first `pad`
        // stages — butterflies with a zero lower half and twiddles that never multiply anything but zero — just copy that
        // value over its group.  So: load n / 2^pad values, replicate, start at stage pad + 1 (no padded copy in HBM either).
        for (int i = threadIdx.x; i < (tsize >> pad); i += BLOCK) {
            size_t grp = (tbase >> pad) + i;
            size_t rev = (size_t)(__brevll((unsigned long long)grp) >> (64 - (k - pad)));
            const Fs v = ntt_load(in, xform, rev, n_src);
            for (int c = 0; c < (1 << pad); c++) lds_put9<TILE>(tile, (i << pad) + c, v);
        }
    }
    __syncthreads();
    for (int s = 1 + pad; s <= S; s++) {
        const int half = 1 << (s - 1);
        const bool cu = ((s - pad) & 1) != 0;       // the first stage run here and every second one after it carry both operands
        // while half <= BLOCK every butterfly of a lane has the same j (BLOCK is a multiple of half): one twiddle load per stage and lane
        const bool one_tw = half <= BLOCK;
        Fs w = Fs::zero();
        if (s > 1 && one_tw) w = tw_load(tw + ((size_t)(threadIdx.x & (half - 1)) << (k - s)) * NTT_TW_WORDS);
        for (int t = threadIdx.x; t < tsize / 2; t += BLOCK) {
            int j = t & (half - 1);
            int base = (t >> (s - 1)) << s;
            Fs u = lds_get9<TILE>(tile, base + j), v = lds_get9<TILE>(tile, base + j + half);
            if (s > 1 && !one_tw) w = tw_load(tw + ((size_t)j << (k - s)) * NTT_TW_WORDS);   // stage 1: every twiddle is w^0 = 1 (wave-uniform: no product at all)
            ntt_butterfly(u, v, w, s == 1, cu);
            lds_put9<TILE>(tile, base + j, u);
            lds_put9<TILE>(tile, base + j + half, v);
        }
        __syncthreads();
    }
    const Fs f = from_arg(out_factor);
    if (final_pass && fmt_out == NTT_FMT_STD8) {
        for (int i = threadIdx.x; i < tsize; i += BLOCK) ntt_store_std8(out, tbase + i, lds_get9<TILE>(tile, i), f);
        return;
    }
    for (int i = threadIdx.x; i < tsize; i += BLOCK) {
        const Fs v = lds_get9<TILE>(tile, i);
        fs_store9(out + (tbase + i) * L29, final_pass ? ntt_final_fs9(v, has_factor, f) : v);
    }
}

// Later passes: stages lo+1..hi (hi - lo <= 4).  Tile = 2^(hi-lo) rows x `cols` columns (cols = 1024 / rows, at least 64); element
// position p = (high * 2^(hi-lo) + r) * 2^lo + low, the workgroup owns a fixed `high` and `cols` consecutive `low` values.  Reads
// raw 9-limb records from `src`, writes the same positions of `dst` (src == dst unless this is the final pass of a STD8 transform).
__global__ __launch_bounds__(NTT_BLOCK) void k_ntt_strided(const uint32_t* src, uint32_t* dst, const uint32_t* __restrict__ tw,
                                                           int k, int lo, int hi, int final_pass, int fmt_out, int has_factor, FsArg out_factor) {
    __shared__ int32_t tile[L29 * NTT_STRIDED_TILE];
    const size_t n = (size_t)1 << k;
    const int rb = hi - lo, rows = 1 << rb;
    const int cb = ntt_strided_col_bits(rb), cols = 1 << cb;
    const size_t low_blocks = ((size_t)1 << lo) >> cb;
    const size_t high = blockIdx.x / low_blocks, low0 = (blockIdx.x % low_blocks) << cb;
    const uint32_t* in = src + (size_t)blockIdx.y * n * L29;
    uint32_t* out = dst + (size_t)blockIdx.y * n * (final_pass && fmt_out == NTT_FMT_STD8 ? 8 : L29);
    auto pos_of = [&](int r, int c) -> size_t { return (((high << rb) + r) << lo) + low0 + c; };
    for (int e = threadIdx.x; e < rows * cols; e += NTT_BLOCK) lds_put9<NTT_STRIDED_TILE>(tile, e, fs_load9(in + pos_of(e >> cb, e & (cols - 1)) * L29));
    __syncthreads();
    for (int s = lo + 1; s <= hi; s++) {
        const int hb = s - 1 - lo;                 // row bit that this stage pairs
        const bool cu = ((s - lo) & 1) != 0;
        for (int t = threadIdx.x; t < rows * cols / 2; t += NTT_BLOCK) {
            int c = t & (cols - 1), rr = t >> cb;               // rr enumerates rows with bit hb cleared
            int r0 = ((rr >> hb) << (hb + 1)) | (rr & ((1 << hb) - 1));
            int r1 = r0 | (1 << hb);
            size_t p0 = pos_of(r0, c);
            size_t j = p0 & (((size_t)1 << (s - 1)) - 1);
            Fs u = lds_get9<NTT_STRIDED_TILE>(tile, r0 * cols + c), v = lds_get9<NTT_STRIDED_TILE>(tile, r1 * cols + c);
            const Fs w = tw_load(tw + (j << (k - s)) * NTT_TW_WORDS);
            ntt_butterfly(u, v, w, false, cu);
            lds_put9<NTT_STRIDED_TILE>(tile, r0 * cols + c, u);
            lds_put9<NTT_STRIDED_TILE>(tile, r1 * cols + c, v);
        }
        __syncthreads();
    }
    const Fs f = from_arg(out_factor);
    if (final_pass && fmt_out == NTT_FMT_STD8) {
        for (int e = threadIdx.x; e < rows * cols; e += NTT_BLOCK) ntt_store_std8(out, pos_of(e >> cb, e & (cols - 1)), lds_get9<NTT_STRIDED_TILE>(tile, e), f);
        return;
    }
    for (int e = threadIdx.x; e < rows * cols; e += NTT_BLOCK) {
        const Fs v = lds_get9<NTT_STRIDED_TILE>(tile, e);
        fs_store9(out + pos_of(e >> cb, e & (cols - 1)) * L29, final_pass ? ntt_final_fs9(v, has_factor, f) : v);
    }
}

// ---- host driver ---------------------------------------------------------------------------------------------
inline FrArg to_arg(const drh::Fr& v) {   // raw limbs (whatever form v is in)
    FrArg a;
    for (int i = 0; i < 4; i++) {
        a.l[2 * i] = (uint32_t)v.l[i];
        a.l[2 * i + 1] = (uint32_t)(v.l[i] >> 32);
    }
    return a;
}
// 256-bit little-endian limbs (a canonical value below p) -> the nine 29-bit limbs of an Fs
inline FsArg fs_limbs_of(const uint64_t (&w)[4]) {
    FsArg a;
    for (int i = 0; i < L29; i++) {
        const int bit = 29 * i, j = bit >> 6, sh = bit & 63;
        uint64_t v = w[j] >> sh;
        if (sh > 35 && j + 1 < 4) v |= w[j + 1] << (64 - sh);
        a.l[i] = (int32_t)(v & M29);
    }
    return a;
}
// a host field element -> kernel argument in the device's Montgomery form (2^261): the host keeps x 2^256, and (32 x) 2^256 has
// the words of x 2^261
inline FsArg fs_arg_mont(const drh::Fr& v) {
    const drh::Fr t = v * drh::Fr::from_u64(32);
    uint64_t w[4] = {t.l[0], t.l[1], t.l[2], t.l[3]};
    return fs_limbs_of(w);
}
// ... and as the plain integer (standard form) in limbs
inline FsArg fs_arg_std(const drh::Fr& v) {
    const drh::Fr t = v.from_mont();
    uint64_t w[4] = {t.l[0], t.l[1], t.l[2], t.l[3]};
    return fs_limbs_of(w);
}

// One batched transform.  fmt_in / fmt_out: element format of the input and of the output (NTT_FMT_STD8: 8 canonical words in
// standard form — the C ABI and the coefficient arrays; NTT_FMT_FS9: raw 9-limb Montgomery records — what the constraint kernel
// reads and writes).  d_src (optional): the input lives in another buffer, n / 2^pad elements per transform (zero-padded to n in
// effect); without it the transform runs on d_data in place.  Intermediate arrays between passes are raw 9-limb records: they
// live in d_data when the output is FS9 and the input comes from d_src, in `tmp` otherwise.
template <class Launch, class ScratchT, class Sync>
int ntt_run(hipStream_t st, Launch&& launch, TwiddleCache& cache, ScratchT& tmp, uint32_t* d_data, unsigned k, size_t batch,
            const drh::Fr& omega_mont, const drh::Fr* scale_mont, Sync&& sync, int fmt_in = NTT_FMT_STD8, int fmt_out = NTT_FMT_STD8,
            const uint32_t* d_src = nullptr, int pad = 0, uint32_t src_div = 1, const uint32_t* d_in_scale = nullptr,
            const uint32_t* d_special = nullptr) {
    const size_t n = (size_t)1 << k;
    if (batch > 65535) return DR_ERR_INVALID;
    // value growth (ring_bounds_check.cpp): raw sums as input, or raw output without a final product, are covered up to 16 stages
    if ((fmt_in == NTT_FMT_FS9 || fmt_in == NTT_FMT_FS9_COSETS || (fmt_out == NTT_FMT_FS9 && !scale_mont)) && k > 16) return DR_ERR_INVALID;
    if ((fmt_in == NTT_FMT_STD8_SCALED && (!d_in_scale || !d_src || src_div == 0 || pad)) || (fmt_in == NTT_FMT_FS9_COSETS && (!d_special || !d_src || k < 4 || pad)))
        return DR_ERR_INVALID;
    uint32_t* d_tw = nullptr;
    for (auto& e : cache.entries)
        if (e.log2n == k && e.omega == omega_mont) d_tw = e.d_tw;
    if (!d_tw) {
        if (cache.entries.size() >= 16) {
            (void)hipFree(cache.entries.front().d_tw);
            cache.entries.erase(cache.entries.begin());
        }
        size_t cnt = n / 2;
        if (hipMalloc((void**)&d_tw, cnt * NTT_TW_WORDS * 4) != hipSuccess) return DR_ERR_NOMEM;
        FsArg wa = fs_arg_mont(omega_mont);
        int rc = launch("k_ntt_twiddles", [&] {
            hipLaunchKernelGGL(k_ntt_twiddles, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_tw, (uint32_t)cnt, wa);
        });
        if (rc != DR_OK) return rc;
        cache.entries.push_back({k, omega_mont, d_tw});
    }
    // STD8 output: the factor in STANDARD form (Montgomery value x standard factor = standard(value x factor)), 1 without a scale;
    // FS9 output: the factor in Montgomery form, or none
    const int has_factor = (fmt_out == NTT_FMT_STD8 || scale_mont) ? 1 : 0;
    const FsArg fa = fmt_out == NTT_FMT_STD8 ? fs_arg_std(scale_mont ? *scale_mont : drh::Fr::one())
                                             : fs_arg_mont(scale_mont ? *scale_mont : drh::Fr::one());
    const bool wide = k == NTT_LOG_TILE + 1;                           // one workgroup of 512 lanes owns a whole 2048-point transform
    const int S = wide ? (int)k : (int)std::min<unsigned>(k, NTT_LOG_TILE);
    NttSource src{d_src ? d_src : d_data, d_in_scale, d_special, fmt_in, src_div ? src_div : 1u};
    if ((int)k == S) {
        int rc = launch("k_ntt_local", [&] {
            if (wide)
                hipLaunchKernelGGL((k_ntt_local<NTT_LOG_TILE + 1, 2 * NTT_BLOCK>), dim3(1, (unsigned)batch), dim3(2 * NTT_BLOCK), 0, st, src, d_data, d_tw, (int)k, S, 1,
                                   fmt_out, has_factor, fa, pad);
            else
                hipLaunchKernelGGL((k_ntt_local<NTT_LOG_TILE, NTT_BLOCK>), dim3(1, (unsigned)batch), dim3(NTT_BLOCK), 0, st, src, d_data, d_tw, (int)k, S, 1, fmt_out,
                                   has_factor, fa, pad);
        });
        if (rc != DR_OK) return rc;
        return sync();
    }
    // several passes: where do the raw intermediates live?
    uint32_t* d_mid = d_data;
    if (fmt_out == NTT_FMT_STD8 || !d_src) {
        int rc = tmp.reserve(n * batch * (size_t)L29 * 4);
        if (rc != DR_OK) return rc;
        d_mid = reinterpret_cast<uint32_t*>(tmp.p);
    }
    int rc = launch("k_ntt_local", [&] {
        hipLaunchKernelGGL((k_ntt_local<NTT_LOG_TILE, NTT_BLOCK>), dim3((unsigned)(n >> S), (unsigned)batch), dim3(NTT_BLOCK), 0, st, src, d_mid, d_tw, (int)k, S, 0, fmt_out,
                           has_factor, fa, pad);
    });
    if (rc != DR_OK) return rc;
    for (int lo = S; lo < (int)k; lo += NTT_MAX_ROW_BITS) {
        int hi = std::min<int>((int)k, lo + NTT_MAX_ROW_BITS);
        int fin = hi == (int)k ? 1 : 0;
        unsigned blocks = (unsigned)(n >> (hi - lo) >> ntt_strided_col_bits(hi - lo));    // tiles per transform
        uint32_t* d_out = fin ? d_data : d_mid;
        rc = launch("k_ntt_strided", [&] {
            hipLaunchKernelGGL(k_ntt_strided, dim3(blocks, (unsigned)batch), dim3(NTT_BLOCK), 0, st, d_mid, d_out, d_tw, (int)k, lo, hi, fin, fmt_out, has_factor, fa);
        });
        if (rc != DR_OK) return rc;
    }
    return sync();
}

}  // namespace dr
