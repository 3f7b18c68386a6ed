// Twisted Edwards Pippenger (kernel K4): the bucket method for ONE variable-base MSM of n >= a few hundred terms on
// Bandersnatch / JubJub — the reference's msm_pippenger_signed_native_cy (dot_ring/curve/native_field/bandersnatch_te.pyx:257-418,
// window rule curve/specs/bandersnatch.py:23-36), used by PedersenVRF.batch_verify (5B + 2 terms, vrf/pedersen/vrf.py:171-242)
// and ThinVRF.batch_verify (vrf/ietf/thin.py:108).
//
// Signed window digits and the per-set counting sort are the G1 pipeline's (k_g1_sort_sets with one bucket set per
// (window, index group), k_size_hist / k_size_place): they only look at scalars.  The kernels here are the group-law part:
//   k_te_msm_prepare     affine standard-form points -> (x, y, d*x*y) in Montgomery form, 96 B per point
//   k_te_msm_accumulate  one lane per bucket walks its list with mixed additions (8 Fr products each); buckets with more than
//                        64 entries are walked by a whole wave (k_te_msm_accumulate_heavy)
//   k_te_msm_reduce      per (set, chunk of L buckets): running sums + the chunk's offset multiple
//   k_te_msm_fold        one lane per set adds its chunk results
// The W x G set sums go back to the host, which combines index groups and windows (a 250-doubling serial chain: one CPU
// core is ~50x faster at that than one GPU lane).  Data: points 64 B in, table 96 B, buckets / partials 128 B (X, Y, Z, T).
#pragma once
#include "kernels_te.hip.h"

namespace dr {

// (load_fr_std / store_fr_std copy 8 words as they are; pack / unpack keep the Montgomery form: canonical words in memory)
DR_DEV TePoint load_te_ext(const uint32_t* arr, size_t idx) {
    const uint32_t* p = arr + idx * 32;
    TePoint r;
    r.x = unpack(load_fr_std(p)); r.y = unpack(load_fr_std(p + 8)); r.z = unpack(load_fr_std(p + 16)); r.t = unpack(load_fr_std(p + 24));
    return r;
}
DR_DEV void store_te_ext(uint32_t* arr, size_t idx, const TePoint& v) {
    uint32_t* p = arr + idx * 32;
    store_fr_std(p, pack(v.x)); store_fr_std(p + 8, pack(v.y)); store_fr_std(p + 16, pack(v.z)); store_fr_std(p + 24, pack(v.t));
}

template <int CV>
__global__ void k_te_msm_prepare(const uint32_t* __restrict__ pts /* n*16 std */, uint32_t n, uint32_t* __restrict__ table /* n*24 */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fs x = fs_from_std(load_fr_std(pts + (size_t)i * 16)), y = fs_from_std(load_fr_std(pts + (size_t)i * 16 + 8));
    uint32_t* o = table + (size_t)i * 24;
    store_fr_std(o, pack(x));
    store_fr_std(o + 8, pack(y));
    store_fr_std(o + 16, pack(mul(te_d_mont<CV>(), mul(x, y))));
}

// A bucket longer than this is not walked by one lane but by a whole wave (k_te_msm_accumulate_heavy): skewed scalars — many
// equal ones, or values much shorter than the windows cover — put thousands of points into one bucket, and a single lane
// adding them one after the other would be the whole kernel's run time.
constexpr uint32_t TE_HEAVY_BUCKET = 64;

template <int CV>
DR_DEV TePoint te_msm_walk(const uint32_t* __restrict__ table, const uint32_t* __restrict__ sorted, uint32_t beg, uint32_t len, uint32_t first,
                           uint32_t stride) {
    TePoint acc = te_identity();
#pragma unroll 1
    for (uint32_t p = first; p < len; p += stride) {
        const uint32_t e = sorted[beg + p];
        const uint32_t* q = table + (size_t)(e & 0x7fffffffu) * 24;
        const bool minus = (e >> 31) != 0;                       // -(x, y) = (-x, y), t = x y changes sign with x
        const Fs x = cneg(unpack(load_fr_std(q)), minus), y = unpack(load_fr_std(q + 8)), dt = cneg(unpack(load_fr_std(q + 16)), minus);
        acc = te_madd<CV>(acc, x, y, dt);
    }
    return acc;
}

template <int CV>
__global__ __launch_bounds__(256) void k_te_msm_accumulate(const uint32_t* __restrict__ table, const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ perm, uint32_t* __restrict__ buckets, size_t nbuckets) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbuckets) return;
    const size_t b = perm[t];
    const uint32_t len = counts[b];
    if (len > TE_HEAVY_BUCKET) return;                               // the wave kernel takes it
    store_te_ext(buckets, b, te_msm_walk<CV>(table, sorted, offsets[b], len, 0, 1));
}

// one wave per bucket; returns at once unless the bucket is heavy: lanes stride over the list, a shuffle tree folds the 64 sums
template <int CV>
__global__ __launch_bounds__(64) void k_te_msm_accumulate_heavy(const uint32_t* __restrict__ table, const uint32_t* __restrict__ sorted,
                                                                const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                                uint32_t* __restrict__ buckets, size_t nbuckets) {
    const size_t b = blockIdx.x;
    if (b >= nbuckets) return;
    const uint32_t len = counts[b];
    if (len <= TE_HEAVY_BUCKET) return;
    TePoint acc = te_msm_walk<CV>(table, sorted, offsets[b], len, threadIdx.x, 64);
#pragma unroll 1
    for (unsigned d = 32; d >= 1; d >>= 1) acc = te_add<CV>(acc, te_shfl_down(acc, d));
    if (threadIdx.x == 0) store_te_ext(buckets, b, acc);
}

// value of a bucket set = sum_j (j+1) B_j.  Chunk [s, s+L): running sums give sum_j (j-s+1) B_j and A = sum_j B_j; the
// chunk contributes that plus s * A (double-and-add over the bits of s < H).
template <int CV>
__global__ __launch_bounds__(128) void k_te_msm_reduce(const uint32_t* __restrict__ buckets, size_t sets, uint32_t H, uint32_t L,
                                                       uint32_t* __restrict__ partial) {
    const uint32_t T = H / L;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= sets * T) return;
    const size_t set = gid / T;
    const uint32_t s = (uint32_t)(gid % T) * L;
    TePoint run = te_identity(), sum = te_identity();
#pragma unroll 1
    for (int j = (int)L - 1; j >= 0; j--) {
        run = te_add<CV>(run, load_te_ext(buckets, set * H + s + (uint32_t)j));
        sum = te_add<CV>(sum, run);
    }
    if (s != 0) {
        TePoint t = te_identity();
#pragma unroll 1
        for (int bit = 31 - __clz(s); bit >= 0; bit--) {
            t = te_dbl<true, CV>(t);
            if ((s >> bit) & 1) t = te_add<CV>(t, run);
        }
        sum = te_add<CV>(sum, t);
    }
    store_te_ext(partial, gid, sum);
}

template <int CV>
__global__ __launch_bounds__(64) void k_te_msm_fold(const uint32_t* __restrict__ partial, size_t sets, uint32_t T, uint32_t* __restrict__ setsum) {
    size_t set = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (set >= sets) return;
    TePoint acc = load_te_ext(partial, set * T);
#pragma unroll 1
    for (uint32_t t = 1; t < T; t++) acc = te_add<CV>(acc, load_te_ext(partial, set * T + t));
    store_te_ext(setsum, set, acc);
}

}  // namespace dr
