// Micro-benchmark: how fast does ONE wave per SIMD run a dependent chain of field operations?  The latency-bound kernels of a batch
// (Elligator, point decoding, G1 decompression, the scalar-multiplication chains: 16 - 112 waves on a 1024-SIMD chip) are such chains;
// the bucket walk reaches ~4.2 cycles per instruction with two waves per SIMD.  Chains of squarings and products of Fr (9 x 29 bits,
// fr29.hip.h) and Fq (14 x 28 bits, fq28.hip.h), with 1, 2 and 4 waves per SIMD resident (blocks of 64 lanes, one block per wave).
//   build: hipcc --offload-arch=gfx950 -O3 -I dot_ring_amd/csrc tools/ubench_chain.hip -o tools/ubench_chain
//   run:   tools/ubench_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fq28.hip.h"
#include "fr29.hip.h"

using namespace dr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t iters, uint64_t* __restrict__ cycles) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    if (OP < 2) {
        Fr w;
        for (int i = 0; i < 8; i++) w.l[i] = in[(gid * 8 + i) % 4096];
        w.l[7] &= 0x3fffffffu;
        Fs x = unpack(w), y = x;
#pragma unroll 1
        for (uint32_t k = 0; k < iters; k++) x = OP == 0 ? sqr(x) : mul(x, y);
        for (int i = 0; i < L29; i++) acc ^= (uint32_t)x.l[i];
    } else {
        uint32_t w[12];
        for (int i = 0; i < 12; i++) w[i] = in[(gid * 12 + i) % 4096];
        w[11] &= 0x0fffffffu;
        Fq28 x = unpack28(w), y = x;
#pragma unroll 1
        for (uint32_t k = 0; k < iters; k++) x = OP == 2 ? sqr(x) : mul(x, y);
        for (int i = 0; i < L28; i++) acc ^= (uint32_t)x.l[i];
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[gid] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    uint32_t *d_in, *d_out;
    uint64_t* d_cyc;
    const uint32_t iters = 4096;
    CK(hipMalloc(&d_in, 4096 * 4));
    CK(hipMalloc(&d_out, 8192 * 64 * 4));
    CK(hipMalloc(&d_cyc, 8192 * 8));
    uint32_t h[4096];
    for (int i = 0; i < 4096; i++) h[i] = 0x9e3779b9u * (i + 1);
    CK(hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice));
    const char* names[4] = {"Fr sqr (178 instr)", "Fr mul (206 instr)", "Fq sqr", "Fq mul"};
    for (int op = 0; op < 4; op++) {
        for (unsigned blocks : {32u, 1024u, 2048u, 4096u, 8192u}) {        // 1024 SIMDs: 1024 blocks = one wave per SIMD
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (op == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters, d_cyc);
                if (op == 1) hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters, d_cyc);
                if (op == 2) hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters, d_cyc);
                if (op == 3) hipLaunchKernelGGL(k_chain<3>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters, d_cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t cyc[64];
            CK(hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost));
            printf("%-20s blocks %5u (%.1f waves/SIMD): %7.3f ms, %.2f us per op, wave 0: %.0f shader-clock ticks per op\n", names[op], blocks, blocks / 1024.0, ms,
                   ms * 1e3 / iters, (double)cyc[0] / iters);
        }
    }
    return 0;
}
