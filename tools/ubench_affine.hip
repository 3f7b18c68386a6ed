// Micro-benchmark behind DESIGN §8 "batched-affine bucket accumulation": how fast can gfx950 add INDEPENDENT pairs of affine
// G1 points when every lane shares one inversion over its own K pairs (Montgomery's trick), next to the XYZZ mixed-addition
// chain k_g1_accumulate runs today?  Same memory behaviour as the real thing would have: operands are gathered by index from
// a 13 MB table of affine points (the size of the prover's window table), prefix products and results stream through HBM.
//   build: hipcc --offload-arch=gfx950 -O3 -I dot_ring_amd/csrc -I tools tools/ubench_affine.hip -o tools/ubench_affine
//   run:   tools/ubench_affine [K ...]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "legacy_g1_fq32.hip.h"   // round-1 arithmetic (12 x 32-bit limbs): this experiment was run on it

using namespace legacy;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// baseline: one lane = one chain of K mixed additions over gathered table points (what k_g1_accumulate does per bucket)
__global__ __launch_bounds__(256) void k_chain(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx_a, uint32_t K, uint32_t lanes,
                                               uint32_t* __restrict__ out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    G1Xyzz acc = g1_inf();
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) acc = g1_madd(acc, load_affine(table, idx_a[(size_t)j * lanes + lane]));
    store_xyzz(out, lane, acc);
}

// batched affine: out[j][lane] = table[a] + table[b] for the lane's K pairs, one inversion per lane.
// Exceptional pairs (equal x: doubling or opposite points) are flagged and skipped here (out = (0,0)): the ubench measures the
// common path; a production kernel resolves them with the doubling formula / infinity in the same two passes.
__global__ __launch_bounds__(256) void k_affine_batch(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx_a,
                                                      const uint32_t* __restrict__ idx_b, uint32_t K, uint32_t lanes,
                                                      uint32_t* __restrict__ prefix /* [K][lanes][12] */, uint32_t* __restrict__ out /* [K][lanes][24] */) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    Fq acc = Fq::one();
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) {
        const size_t s = (size_t)j * lanes + lane;
        Fq ax = load_fq(table + (size_t)idx_a[s] * 24), bx = load_fq(table + (size_t)idx_b[s] * 24);
        Fq d = sub(bx, ax);
        if (d.is_zero()) d = Fq::one();
        store_fq(prefix + s * 12, acc);
        acc = mul(acc, d);
    }
    Fq run = inv(acc);
#pragma unroll 1
    for (uint32_t j = K; j-- > 0;) {
        const size_t s = (size_t)j * lanes + lane;
        G1Affine a = load_affine(table, idx_a[s]), b = load_affine(table, idx_b[s]);
        Fq d = sub(b.x, a.x);
        const bool special = d.is_zero();
        if (special) d = Fq::one();
        Fq dinv = mul(run, load_fq(prefix + s * 12));
        run = mul(run, d);
        Fq lam = mul(sub(b.y, a.y), dinv);
        Fq x3 = sub(sub(sqr(lam), a.x), b.x);
        Fq y3 = sub(mul(lam, sub(a.x, x3)), a.y);
        if (special) { x3 = Fq::zero(); y3 = Fq::zero(); }
        store_fq(out + s * 24, x3);
        store_fq(out + s * 24 + 12, y3);
    }
}

// check: recompute sampled pairs with the XYZZ formulas and compare the affine results
__global__ void k_check(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx_a, const uint32_t* __restrict__ idx_b, uint32_t K,
                        uint32_t lanes, const uint32_t* __restrict__ out, uint32_t stride, uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t s = (size_t)t * stride;
    if (s >= (size_t)K * lanes) return;
    G1Affine a = load_affine(table, idx_a[s]), b = load_affine(table, idx_b[s]);
    if (a.x == b.x) return;
    G1Affine want = g1_to_affine_dev(g1_madd(g1_from_affine(a), b));
    if (!(want.x == load_fq(out + s * 24)) || !(want.y == load_fq(out + s * 24 + 12))) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
    const uint32_t T = 6145 * 22;                       // entries of the prover's window table
    const uint32_t lanes = 256 * 1024;                  // 4096 waves: 4 per SIMD
    // G1 generator, standard form, little-endian limbs
    static const uint32_t GEN[24] = {
        0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu, 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u,
        0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu, 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    uint32_t *d_seed, *d_table, *d_a, *d_b, *d_prefix, *d_out, *d_chain, *d_bad;
    CK(hipMalloc(&d_seed, sizeof GEN));
    CK(hipMemcpy(d_seed, GEN, sizeof GEN, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_g1_bases_to_mont, dim3(1), dim3(64), 0, 0, d_seed, 1u);
    CK(hipMalloc(&d_table, (size_t)T * 96));
    hipLaunchKernelGGL(k_g1_synth_bases, dim3((T + 255) / 256), dim3(256), 0, 0, d_table, T, 1u, d_seed);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> ks;
    for (int i = 1; i < argc; i++) ks.push_back((uint32_t)atoi(argv[i]));
    if (ks.empty()) ks = {32, 64, 128, 256, 512};
    uint32_t kmax = 0;
    for (uint32_t k : ks) kmax = k > kmax ? k : kmax;
    const size_t slots = (size_t)kmax * lanes;
    std::vector<uint32_t> ha(slots), hb(slots);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); };
    for (size_t i = 0; i < slots; i++) { ha[i] = rnd() % T; do hb[i] = rnd() % T; while (hb[i] == ha[i]); }
    CK(hipMalloc(&d_a, slots * 4)); CK(hipMalloc(&d_b, slots * 4));
    CK(hipMemcpy(d_a, ha.data(), slots * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, hb.data(), slots * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_prefix, slots * 48)); CK(hipMalloc(&d_out, slots * 96)); CK(hipMalloc(&d_chain, (size_t)lanes * 192)); CK(hipMalloc(&d_bad, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("table %u points (%.1f MB), %u lanes; additions per second, kernel time by HIP events\n", T, T * 96 / 1e6, lanes);
    for (uint32_t K : ks) {
        float ms_chain = 0, ms_aff = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain, dim3(lanes / 256), dim3(256), 0, 0, d_table, d_a, K, lanes, d_chain);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_chain, e0, e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_affine_batch, dim3(lanes / 256), dim3(256), 0, 0, d_table, d_a, d_b, K, lanes, d_prefix, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_aff, e0, e1));
        }
        CK(hipMemset(d_bad, 0, 4));
        const uint32_t stride = 4099, checks = (uint32_t)(((size_t)K * lanes + stride - 1) / stride);
        hipLaunchKernelGGL(k_check, dim3((checks + 63) / 64), dim3(64), 0, 0, d_table, d_a, d_b, K, lanes, d_out, stride, d_bad);
        uint32_t bad = 0;
        CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        const double adds = (double)K * lanes;
        printf("K = %4u pairs per lane: XYZZ chain %7.2f ms = %5.2f G add/s | batched affine %7.2f ms = %5.2f G add/s (x%.2f), %u of %u sampled results wrong\n",
               K, ms_chain, adds / ms_chain / 1e6, ms_aff, adds / ms_aff / 1e6, ms_chain / ms_aff, bad, checks);
    }
    return 0;
}
