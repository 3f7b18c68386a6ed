// Micro-benchmark behind DESIGN "unsaturated limbs": the bucket walk of k_g1_accumulate (a chain of XYZZ mixed additions
// over points gathered from a 13 MB table — the prover's window table) with the Fq product as it is today
// (12 x 32-bit limbs: v_mad_u64_u32 + v_addc_co_u32 per partial product, field.hip.h) against 14 x 28-bit signed limbs
// (one v_mad_i64_i32 per partial product, lazy reduction, fq28.hip.h).  Results of the two chains are compared word for word.
//   build: hipcc --offload-arch=gfx950 -O3 -I dot_ring_amd/csrc -I tools tools/ubench_limbs.hip -o tools/ubench_limbs
//   run:   tools/ubench_limbs [K ...]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "legacy_g1_fq32.hip.h"   // the 12 x 32-bit arithmetic of round 1, namespace legacy
#include "g1.hip.h"               // the 14 x 28-bit arithmetic the library uses

using namespace dr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_chain32(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t K, uint32_t lanes,
                                                 uint32_t* __restrict__ out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    legacy::G1Xyzz acc = legacy::g1_inf();
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) {
        const uint32_t e = idx[(size_t)j * lanes + lane];
        acc = legacy::g1_madd(acc, legacy::g1_neg_affine(legacy::load_affine(table, e & 0x7fffffffu), (e >> 31) != 0));
    }
    legacy::store_xyzz(out, lane, acc);
}

// table of 32-bit-limb Montgomery values (x 2^384) -> 28-bit-limb Montgomery values (x 2^392), canonical words
__global__ void k_table_to28(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_coords) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coords) return;
    store_fq28(out + (size_t)i * 12, mul(load_fq28(in + (size_t)i * 12), Fq28::constant<Fq28Params::K400>()));
}

#define DEFINE_CHAIN28(NAME, ATTR) \
__global__ __launch_bounds__(256) ATTR void NAME(const uint32_t* __restrict__ table28, const uint32_t* __restrict__ idx, uint32_t K, uint32_t lanes, \
                                                 uint32_t* __restrict__ out) { \
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x; \
    if (lane >= lanes) return; \
    G1Xyzz acc = g1_inf(); \
_Pragma("unroll 1") \
    for (uint32_t j = 0; j < K; j++) { \
        const uint32_t e = idx[(size_t)j * lanes + lane]; \
        acc = g1_madd(acc, g1_neg_affine(load_affine(table28, e & 0x7fffffffu), (e >> 31) != 0)); \
    } \
    const Fq28 k384 = Fq28::constant<Fq28Params::K384>(); \
    uint32_t* o = out + (size_t)lane * 48; \
    if (acc.inf) acc = g1_inf(); \
    store_fq28(o, mul(acc.x, k384)); \
    store_fq28(o + 12, mul(carry(acc.y), k384)); \
    store_fq28(o + 24, mul(acc.zz, k384)); \
    store_fq28(o + 36, mul(acc.zzz, k384)); \
}
DEFINE_CHAIN28(k_chain28, )
DEFINE_CHAIN28(k_chain28_w3, __attribute__((amdgpu_waves_per_eu(3, 3))))

int main(int argc, char** argv) {
    const uint32_t T = 6145 * 22;                       // entries of the prover's window table
    const uint32_t lanes = 256 * 1024;                  // 4096 waves: 4 per SIMD
    static const uint32_t GEN[24] = {
        0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu, 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u,
        0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu, 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    uint32_t *d_seed, *d_table, *d_table28, *d_idx, *d_o32, *d_o28;
    CK(hipMalloc(&d_seed, sizeof GEN));
    CK(hipMemcpy(d_seed, GEN, sizeof GEN, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(legacy::k_g1_bases_to_mont, dim3(1), dim3(64), 0, 0, d_seed, 1u);
    CK(hipMalloc(&d_table, (size_t)T * 96));
    CK(hipMalloc(&d_table28, (size_t)T * 96));
    hipLaunchKernelGGL(legacy::k_g1_synth_bases, dim3((T + 255) / 256), dim3(256), 0, 0, d_table, T, 1u, d_seed);
    hipLaunchKernelGGL(k_table_to28, dim3((2 * T + 255) / 256), dim3(256), 0, 0, d_table, d_table28, 2 * T);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> ks;
    for (int i = 1; i < argc; i++) ks.push_back((uint32_t)atoi(argv[i]));
    if (ks.empty()) ks = {32, 128, 512};
    uint32_t kmax = 0;
    for (uint32_t k : ks) kmax = k > kmax ? k : kmax;
    const size_t slots = (size_t)kmax * lanes;
    std::vector<uint32_t> h(slots);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); };
    for (size_t i = 0; i < slots; i++) h[i] = (rnd() % T) | ((rnd() & 1u) << 31);
    // exceptional cases in a few lanes: the same point twice in a row (doubling), then its negative (-> infinity), then on
    for (uint32_t lane = 0; lane < 64; lane += 7) {
        h[(size_t)1 * lanes + lane] = h[(size_t)0 * lanes + lane];
        if (lane & 1) { h[(size_t)2 * lanes + lane] = h[lane] ^ 0x80000000u; h[(size_t)3 * lanes + lane] = h[lane] ^ 0x80000000u; }
    }
    CK(hipMalloc(&d_idx, slots * 4));
    CK(hipMemcpy(d_idx, h.data(), slots * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_o32, (size_t)lanes * 192)); CK(hipMalloc(&d_o28, (size_t)lanes * 192));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("table %u points (%.1f MB), %u lanes; mixed additions per second, kernel time by HIP events\n", T, T * 96 / 1e6, lanes);
    std::vector<uint32_t> o32((size_t)lanes * 48), o28((size_t)lanes * 48);
    for (uint32_t K : ks) {
        float ms32 = 0, ms28 = 0, ms28w3 = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain32, dim3(lanes / 256), dim3(256), 0, 0, d_table, d_idx, K, lanes, d_o32);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms32, e0, e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain28, dim3(lanes / 256), dim3(256), 0, 0, d_table28, d_idx, K, lanes, d_o28);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms28, e0, e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain28_w3, dim3(lanes / 256), dim3(256), 0, 0, d_table28, d_idx, K, lanes, d_o28);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms28w3, e0, e1));
        }
        CK(hipMemcpy(o32.data(), d_o32, o32.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o28.data(), d_o28, o28.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t l = 0; l < lanes; l++) {
            bool same = true;
            for (int w = 0; w < 48; w++) same &= o32[l * 48 + w] == o28[l * 48 + w];
            bad += !same;
        }
        const double adds = (double)K * lanes;
        printf("K = %4u additions per lane: 12 x 32-bit limbs %7.2f ms = %5.2f G add/s | 14 x 28-bit limbs %7.2f ms = %5.2f G add/s (x%.2f), at 3 waves/SIMD %7.2f ms = %5.2f G add/s (x%.2f), %zu of %u lanes differ\n",
               K, ms32, adds / ms32 / 1e6, ms28, adds / ms28 / 1e6, ms32 / ms28, ms28w3, adds / ms28w3 / 1e6, ms32 / ms28w3, bad, lanes);
    }
    return 0;
}
