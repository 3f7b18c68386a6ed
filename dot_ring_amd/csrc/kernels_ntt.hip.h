// Batched radix-2 NTT over Fr (seam C, kernel K6).  Replaces BlsScalarNTTPlan._transform_core
// (dot_ring/ring_proof/polynomial/ntt.pyx:116-163) + bls_scalar_ntt_round (bls12_381_scalar.c:333-356):
// out[i] = sum_j in[j] * omega^(i*j), natural order in and out, optional scaling of every output.
//
// Structure: decimation in time over the bit-reversed input, exactly the butterfly network of the reference,
// but staged through LDS: pass A loads a tile of 2^10 consecutive positions of the bit-reversed array (a gather
// of 32-B elements), runs stages 1..10 in LDS (limb-major layout: lane i touches word [limb][i], conflict-free),
// and streams the tile out; each later pass takes the next <= 4 index bits as "rows" of a (rows x 64 columns)
// tile so that global accesses stay 2 KiB-contiguous.  Twiddles omega^j, j < n/2, are a per-(n, omega) table in
// HBM (Montgomery form), cached in the context.  Elements are converted to Montgomery form on the first load
// and back (fused with the optional scale) on the last store, so a transform moves 2 x 32 B per element per pass.
#pragma once
#include "dev_types.hpp"
#include "curve.hip.h"
#include "hostmath.hpp"

namespace dr {

constexpr int NTT_LOG_TILE = 10, NTT_TILE = 1 << NTT_LOG_TILE, NTT_BLOCK = 256;
constexpr int NTT_COLS = 64, NTT_MAX_ROW_BITS = 4;

DR_DEV Fr lds_get(const uint32_t* t, int i) {
    Fr r;
#pragma unroll
    for (int l = 0; l < 8; l++) r.l[l] = t[l * NTT_TILE + i];
    return r;
}
DR_DEV void lds_put(uint32_t* t, int i, const Fr& v) {
#pragma unroll
    for (int l = 0; l < 8; l++) t[l * NTT_TILE + i] = v.l[l];
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

struct FrArg {   // kernel-argument copy of a field element
    uint32_t l[8];
};
DR_DEV Fr from_arg(const FrArg& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}

// tw[j] = omega^j (Montgomery), j < count
__global__ void k_ntt_twiddles(uint32_t* tw, uint32_t count, FrArg omega_mont) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr w = from_arg(omega_mont), r = Fr::one();
    for (uint32_t e = j; e; e >>= 1) {
        if (e & 1) r = mul(r, w);
        w = sqr(w);
    }
    gstore_fr(tw + (size_t)j * 8, r);
}

// Pass A: stages 1..S (S = min(k, 10)) on tile `blockIdx.x` of transform `blockIdx.y`.
// src is read at bit-reversed positions, dst written contiguously (src == dst is allowed only when S == k,
// where one workgroup owns the whole transform).
__global__ __launch_bounds__(NTT_BLOCK) void k_ntt_local(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                         const uint32_t* __restrict__ tw, int k, int S, int final_pass,
                                                         FrArg out_factor, int in_mont, int pad) {
    __shared__ uint32_t tile[8 * NTT_TILE];
    const size_t n = (size_t)1 << k;
    const int tsize = 1 << S;
    const size_t xform = blockIdx.y, tbase = (size_t)blockIdx.x * tsize;
    const uint32_t* in = src + xform * (n >> pad) * 8;
    uint32_t* out = dst + xform * n * 8;
    if (pad == 0) {
        for (int i = threadIdx.x; i < tsize; i += NTT_BLOCK) {
            size_t pos = tbase + i;
            size_t rev = (size_t)(__brevll((unsigned long long)pos) >> (64 - k));
            Fr v = gload_fr(in + rev * 8);
            lds_put(tile, i, in_mont ? v : to_mont(v));     // in_mont: the producer already wrote Montgomery form
        }
    } else {
        // the input is a polynomial of n / 2^pad coefficients, zero-padded to n: in bit-reversed order only every 2^pad-th
        // position is non-zero (the low `pad` bits of a position are the top bits of its source index), and the first `pad`
        // stages — butterflies with a zero lower half and twiddles that never multiply anything but zero — just copy that
        // value over its group.  So: load n / 2^pad values, replicate, start at stage pad + 1 (no padded copy in HBM either).
        for (int i = threadIdx.x; i < (tsize >> pad); i += NTT_BLOCK) {
            size_t grp = (tbase >> pad) + i;
            size_t rev = (size_t)(__brevll((unsigned long long)grp) >> (64 - (k - pad)));
            Fr v = gload_fr(in + rev * 8);
            if (!in_mont) v = to_mont(v);
            for (int c = 0; c < (1 << pad); c++) lds_put(tile, (i << pad) + c, v);
        }
    }
    __syncthreads();
    for (int s = 1 + pad; s <= S; s++) {
        const int half = 1 << (s - 1);
        for (int t = threadIdx.x; t < tsize / 2; t += NTT_BLOCK) {
            int j = t & (half - 1);
            int base = (t >> (s - 1)) << s;
            Fr u = lds_get(tile, base + j), v = lds_get(tile, base + j + half);
            if (s > 1) {                        // stage 1: every twiddle is w^0 = 1 (wave-uniform: no product at all)
                Fr w = gload_fr(tw + ((size_t)j << (k - s)) * 8);
                v = mul(w, v);
            }
            lds_put(tile, base + j, add(u, v));
            lds_put(tile, base + j + half, sub(u, v));
        }
        __syncthreads();
    }
    Fr f = from_arg(out_factor);
    for (int i = threadIdx.x; i < tsize; i += NTT_BLOCK) {
        Fr v = lds_get(tile, i);
        if (final_pass) v = mul(v, f);          // Montgomery * standard-form factor -> standard form of v*factor
        gstore_fr(out + (tbase + i) * 8, v);
    }
}

// Later passes: stages lo+1..hi (hi - lo <= 4) in place.  Tile = 2^(hi-lo) rows x 64 columns; element position
// p = (high * 2^(hi-lo) + r) * 2^lo + low, the workgroup owns a fixed `high`, 64 consecutive `low` values.
__global__ __launch_bounds__(NTT_BLOCK) void k_ntt_strided(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw,
                                                           int k, int lo, int hi, int final_pass, FrArg out_factor) {
    __shared__ uint32_t tile[8 * NTT_TILE];
    const size_t n = (size_t)1 << k;
    const int rb = hi - lo, rows = 1 << rb;
    const size_t low_blocks = ((size_t)1 << lo) / NTT_COLS;
    const size_t high = blockIdx.x / low_blocks, low0 = (blockIdx.x % low_blocks) * NTT_COLS;
    uint32_t* d = data + (size_t)blockIdx.y * n * 8;
    auto pos_of = [&](int r, int c) -> size_t { return (((high << rb) + r) << lo) + low0 + c; };
    for (int e = threadIdx.x; e < rows * NTT_COLS; e += NTT_BLOCK) {
        int r = e / NTT_COLS, c = e % NTT_COLS;
        lds_put(tile, e, gload_fr(d + pos_of(r, c) * 8));
    }
    __syncthreads();
    for (int s = lo + 1; s <= hi; s++) {
        const int hb = s - 1 - lo;                 // row bit that this stage pairs
        for (int t = threadIdx.x; t < rows * NTT_COLS / 2; t += NTT_BLOCK) {
            int c = t % NTT_COLS, rr = t / NTT_COLS;            // rr enumerates rows with bit hb cleared
            int r0 = ((rr >> hb) << (hb + 1)) | (rr & ((1 << hb) - 1));
            int r1 = r0 | (1 << hb);
            size_t p0 = pos_of(r0, c);
            size_t j = p0 & (((size_t)1 << (s - 1)) - 1);
            Fr u = lds_get(tile, r0 * NTT_COLS + c), v = lds_get(tile, r1 * NTT_COLS + c);
            Fr w = gload_fr(tw + (j << (k - s)) * 8);
            v = mul(w, v);
            lds_put(tile, r0 * NTT_COLS + c, add(u, v));
            lds_put(tile, r1 * NTT_COLS + c, sub(u, v));
        }
        __syncthreads();
    }
    Fr f = from_arg(out_factor);
    for (int e = threadIdx.x; e < rows * NTT_COLS; e += NTT_BLOCK) {
        int r = e / NTT_COLS, c = e % NTT_COLS;
        Fr v = lds_get(tile, e);
        if (final_pass) v = mul(v, f);
        gstore_fr(d + pos_of(r, c) * 8, v);
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

// in_mont / out_mont: the data is (left) in Montgomery form instead of standard form — producers and consumers inside the
// prover that work in Montgomery form anyway save the two conversions (one product per element each).
template <class Launch, class ScratchT, class Sync>
int ntt_run(hipStream_t st, Launch&& launch, TwiddleCache& cache, ScratchT& tmp, uint32_t* d_data, unsigned k, size_t batch,
            const drh::Fr& omega_mont, const drh::Fr* scale_mont, Sync&& sync, bool in_mont = false, bool out_mont = false,
            const uint32_t* d_src = nullptr, int pad = 0) {
    // d_src (optional): the input lives in another buffer, n / 2^pad coefficients per transform (zero-padded to n in effect);
    // the first pass then writes straight into d_data and no temporary / copy back is needed
    const size_t n = (size_t)1 << k;
    if (batch > 65535) return DR_ERR_INVALID;
    uint32_t* d_tw = nullptr;
    for (auto& e : cache.entries)
        if (e.log2n == k && e.omega == omega_mont) d_tw = e.d_tw;
    if (!d_tw) {
        if (cache.entries.size() >= 16) {
            (void)hipFree(cache.entries.front().d_tw);
            cache.entries.erase(cache.entries.begin());
        }
        size_t cnt = n / 2;
        if (hipMalloc((void**)&d_tw, cnt * 32) != hipSuccess) return DR_ERR_NOMEM;
        FrArg wa = to_arg(omega_mont);
        int rc = launch("k_ntt_twiddles", [&] {
            hipLaunchKernelGGL(k_ntt_twiddles, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_tw, (uint32_t)cnt, wa);
        });
        if (rc != DR_OK) return rc;
        cache.entries.push_back({k, omega_mont, d_tw});
    }
    // output factor in STANDARD form: Montgomery value * standard factor = standard(value * factor)
    // (Montgomery output: Montgomery value * Montgomery factor = Montgomery(value * factor); without a scale the final
    // product is skipped altogether)
    drh::Fr factor_std = out_mont ? (scale_mont ? *scale_mont : drh::Fr::one()) : (scale_mont ? scale_mont->from_mont() : drh::Fr::one().from_mont());
    FrArg fa = to_arg(factor_std);
    const int do_final = (out_mont && !scale_mont) ? 0 : 1;
    const int im = in_mont ? 1 : 0;
    const int S = (int)std::min<unsigned>(k, NTT_LOG_TILE);
    if ((int)k == S) {
        int rc = launch("k_ntt_local", [&] {
            hipLaunchKernelGGL(k_ntt_local, dim3(1, (unsigned)batch), dim3(NTT_BLOCK), 0, st, d_src ? d_src : d_data, d_data, d_tw, (int)k, S, do_final, fa, im, pad);
        });
        if (rc != DR_OK) return rc;
        return sync();
    }
    int rc = DR_OK;
    uint32_t* d_tmp = d_data;
    if (!d_src) {
        rc = tmp.reserve(n * batch * 32);
        if (rc != DR_OK) return rc;
        d_tmp = reinterpret_cast<uint32_t*>(tmp.p);
    }
    rc = launch("k_ntt_local", [&] {
        hipLaunchKernelGGL(k_ntt_local, dim3((unsigned)(n >> S), (unsigned)batch), dim3(NTT_BLOCK), 0, st, d_src ? d_src : d_data, d_tmp, d_tw, (int)k, S, 0, fa, im, pad);
    });
    if (rc != DR_OK) return rc;
    for (int lo = S; lo < (int)k; lo += NTT_MAX_ROW_BITS) {
        int hi = std::min<int>((int)k, lo + NTT_MAX_ROW_BITS);
        int fin = hi == (int)k ? do_final : 0;
        unsigned blocks = (unsigned)(n >> (hi - lo) >> 6);    // tiles per transform
        rc = launch("k_ntt_strided", [&] {
            hipLaunchKernelGGL(k_ntt_strided, dim3(blocks, (unsigned)batch), dim3(NTT_BLOCK), 0, st, d_tmp, d_tw, (int)k, lo, hi, fin, fa);
        });
        if (rc != DR_OK) return rc;
    }
    if (d_tmp != d_data && hipMemcpyAsync(d_data, d_tmp, n * batch * 32, hipMemcpyDeviceToDevice, st) != hipSuccess) return DR_ERR_DEVICE;
    return sync();
}

}  // namespace dr
