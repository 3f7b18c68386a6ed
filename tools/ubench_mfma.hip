// VERDICT r2 item 6, measured: can the matrix pipe take the constant half of the Montgomery product of the bucket walk?
//
// Half of a 14 x 28-bit Montgomery product is m x p with p CONSTANT — a product of a (lanes x digits) matrix with a fixed Toeplitz
// matrix, and v_mfma_i32_16x16x64_i8 co-issues with the VALU.  What the matrix pipe needs, though, is i8 digits in ITS lane layout
// and what it returns is 32-bit column sums in its output layout.  This benchmark measures a LOWER BOUND of such a product
// against the library's product (fq28.hip.h mul: 392 v_mad_i64_i32 + 69 = 461 instructions, reduction interleaved in the same
// 64-bit column accumulators):
//   chain A: x <- x * y with the library's product.
//   chain B: per product ONLY the steps no MFMA scheme can avoid — a * b on the VALU (196 multiply-adds, 27 columns carried to
//            28-bit limbs), the low half split into 56 seven-bit digits (bytes, four per dword), sixteen
//            v_mfma_i32_16x16x64_i8 on those bytes against a constant operand (the 64 x 64 block of the Toeplitz matrix that
//            yields the high half), the 64 column sums of the lane recombined into 16 limbs and added to the high half of a * b.
//            LEFT OUT (each would add instructions): computing m = t_lo * (-1/p) mod 2^392 at all (105 more multiply-adds, or
//            a second MFMA round with its own split and recombination), moving digits into and column sums out of the MFMA
//            lane layout (a 4 x 4 block transpose among lanes 16 apart, ~16 + ~48 cross-lane moves), the carry of the low
//            half.  Chain B therefore computes NO correct product — its value is only its cost: a floor for every real variant.
//   chain C: sixteen MFMAs per step alone, and interleaved with 196 independent v_mad_i64_i32, to show what the matrix pipe
//            itself costs beside the VALU.
//   build: python3 tools/gen_ubench_mfma.py && hipcc --offload-arch=gfx950 -O3 -I dot_ring_amd/csrc -I tools tools/ubench_mfma.hip -o tools/ubench_mfma
//   run:   tools/ubench_mfma [products per lane]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fq28.hip.h"
#include "ubench_mfma_gen.hip.h"

using namespace dr;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_chain_lib(uint32_t K, uint32_t* __restrict__ out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    Fq28 x = Fq28::constant<Fq28Params::R2>(), y = Fq28::constant<Fq28Params::K384>();
    x.l[0] += (int32_t)(lane & 0xffff);
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) x = mul(x, y);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L28; i++) acc ^= (uint32_t)x.l[i];
    out[lane] = acc;
}

// seven-bit digits of a 28-bit limb as four bytes
__device__ __forceinline__ uint32_t digits_of(int32_t limb) {
    const uint32_t u = (uint32_t)limb;
    return (u & 0x7fu) | ((u << 1) & 0x7f00u) | ((u << 2) & 0x7f0000u) | ((u << 3) & 0x7f000000u);
}

template <bool WITH_MFMA>
__global__ __launch_bounds__(256) void k_chain_floor(uint32_t K, const uint32_t* __restrict__ bmat, uint32_t* __restrict__ out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    Fq28 x = Fq28::constant<Fq28Params::R2>(), y = Fq28::constant<Fq28Params::K384>();
    x.l[0] += (int32_t)(lane & 0xffff);
    // the constant operand: four 16 x 64 blocks, 16 bytes per lane each (what it holds does not matter for the cost)
    v4i bm[4];
#pragma unroll
    for (int t = 0; t < 4; t++) bm[t] = *reinterpret_cast<const v4i*>(bmat + ((threadIdx.x & 63) * 4 + t) * 4);
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) {
        int32_t t[28];
        plainmul14x28_asm(t, x.l, y.l);                              // a * b, carried: 251 instructions
        uint32_t dg[14];
#pragma unroll
        for (int i = 0; i < 14; i++) dg[i] = digits_of(t[i]);        // 56 digits of the low half (m would be split the same way)
        // the lane's 16-byte share of the digit matrix for each of the four 16-element groups of the wave (here: its own
        // bytes — the real thing first moves them across lanes)
        v4i cs[4][4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4i a = {(int)dg[(4 * g) % 14], (int)dg[(4 * g + 1) % 14], (int)dg[(4 * g + 2) % 14], (int)dg[(4 * g + 3) % 14]};
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const v4i zero = {0, 0, 0, 0};
                if (WITH_MFMA) cs[g][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bm[tt], zero, 0, 0, 0);
                else cs[g][tt] = a + bm[tt];                         // same data flow without the matrix pipe
            }
        }
        // 64 column sums -> 16 limbs: four 7-bit-spaced sums per limb, 64-bit because a sum has up to 21 bits
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const v4i c = cs[i & 3][(i >> 2) & 3];
            const int64_t v = (int64_t)c.x + ((int64_t)c.y << 7) + ((int64_t)c.z << 14) + ((int64_t)c.w << 21);
            r.l[i] = t[14 + i] + (int32_t)(v & M28) + (int32_t)(v >> 28);      // plus the high half of a * b
        }
#pragma unroll
        for (int i = 14; i < 16; i++) {                                        // columns 56..63: the two limbs above (carry-out side)
            const v4i c = cs[i & 3][(i >> 2) & 3];
            const int64_t v = (int64_t)c.x + ((int64_t)c.y << 7) + ((int64_t)c.z << 14) + ((int64_t)c.w << 21);
            r.l[i - 14] += (int32_t)(v >> 28) + (int32_t)(v & 0xff);
        }
        x = carry(r);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L28; i++) acc ^= (uint32_t)x.l[i];
    out[lane] = acc;
}

// the matrix pipe on its own and beside independent multiply-adds
template <int MADS>
__global__ __launch_bounds__(256) void k_mfma_rate(uint32_t K, const uint32_t* __restrict__ bmat, uint32_t* __restrict__ out) {
    v4i a = *reinterpret_cast<const v4i*>(bmat + (threadIdx.x & 63) * 4), b = *reinterpret_cast<const v4i*>(bmat + 256 + (threadIdx.x & 63) * 4);
    v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int64_t m[4] = {1, 2, 3, 4};
    const int32_t u = (int32_t)threadIdx.x * 3 + 1, w = (int32_t)blockIdx.x * 7 + 5;
#pragma unroll 1
    for (uint32_t j = 0; j < K; j++) {
#pragma unroll
        for (int s = 0; s < 16; s++) {
            acc[s & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[s & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MADS / 16; q++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(m[(s + q) & 3]) : "v"(u), "v"(w) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc[0].x + acc[1].y + acc[2].z + acc[3].w) ^ (uint32_t)(m[0] + m[1] + m[2] + m[3]);
}

int main(int argc, char** argv) {
    const uint32_t K = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048;
    const uint32_t lanes = 256 * 1024 * 2;              // 8192 waves: 8 per SIMD in flight over the run, two resident like the bucket walk
    uint32_t *d_out, *d_b;
    CK(hipMalloc(&d_out, (size_t)lanes * 4));
    CK(hipMalloc(&d_b, 4096 * 4));
    {
        uint32_t h[4096];
        for (int i = 0; i < 4096; i++) h[i] = 0x01020304u * (uint32_t)(i % 31 + 1) & 0x7f7f7f7fu;
        CK(hipMemcpy(d_b, h, sizeof h, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_ms = [&](auto&& launch) -> float {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        return best;
    };
    const float a = time_ms([&] { hipLaunchKernelGGL(k_chain_lib, dim3(lanes / 256), dim3(256), 0, 0, K, d_out); });
    const float b = time_ms([&] { hipLaunchKernelGGL(k_chain_floor<true>, dim3(lanes / 256), dim3(256), 0, 0, K, d_b, d_out); });
    const float b0 = time_ms([&] { hipLaunchKernelGGL(k_chain_floor<false>, dim3(lanes / 256), dim3(256), 0, 0, K, d_b, d_out); });
    const float c0 = time_ms([&] { hipLaunchKernelGGL(k_mfma_rate<0>, dim3(lanes / 256), dim3(256), 0, 0, K, d_b, d_out); });
    const float c1 = time_ms([&] { hipLaunchKernelGGL(k_mfma_rate<192>, dim3(lanes / 256), dim3(256), 0, 0, K, d_b, d_out); });
    CK(hipGetLastError());
    const double prods = (double)K * lanes;
    printf("%u lanes x %u dependent products per lane (kernel time by HIP events, best of 3)\n", lanes, K);
    printf("A  library product (461 instructions, reduction interleaved)        %8.2f ms  %6.2f G products/s\n", a, prods / a / 1e6);
    printf("B  floor of an MFMA variant (a*b %d + digit split + 16 MFMA + recombination; no m, no lane moves, no low-half carry)\n", PLAINMUL_INSTRUCTIONS);
    printf("                                                                     %8.2f ms  %6.2f G 'products'/s   x%.2f of A\n", b, prods / b / 1e6, b / a);
    printf("B' the same data flow with vector adds in place of the MFMAs        %8.2f ms  %6.2f G/s               x%.2f of A\n", b0, prods / b0 / 1e6, b0 / a);
    printf("C  16 v_mfma_i32_16x16x64_i8 per step alone                          %8.2f ms  = %.1f cycles per MFMA per SIMD at 2.4 GHz\n", c0,
           c0 * 1e-3 * 2.4e9 / ((double)K * 16 * (lanes / 64) / 1024));
    printf("C' the same 16 MFMAs with 192 independent v_mad_i64_i32 between them  %8.2f ms  (192 multiply-adds alone would take ~%.2f ms at 4.2 cycles each)\n", c1,
           (double)K * 192 * (lanes / 64) / 1024 * 4.2 / 2.4e9 * 1e3);
    printf("verdict: %s\n", b >= a ? "the floor of the MFMA variant is already slower than the library's product — the matrix pipe cannot pay for its operand conversions"
                                    : "the floor is faster than the library's product: the left-out steps decide");
    return 0;
}
