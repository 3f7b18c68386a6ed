// Host build of dot_ring_amd/csrc/divstep28.hip.h for tests/test_divstep_cpu.py: reads one hexadecimal x (< p) per line,
// prints the signed output limbs and the number of batches.  argv[1] = "fq" (14 x 28 bits, BLS12-381 base field) or "fr"
// (9 x 29 bits, its scalar field).
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include "divstep28.hip.h"

static const uint32_t FQ_P[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                  0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
static const uint32_t FQ_N0 = 0xffcfffdu;
static const uint32_t FR_P[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
static const uint32_t FR_N0 = 0x1fffffffu;

template <int N, int BITS, int MAXB>
static void run(const uint32_t (&P)[N], uint32_t n0) {
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        // hex -> big number -> N limbs of BITS bits
        unsigned char nib[128] = {0};
        const int len = (int)line.size();
        for (int i = 0; i < len && i < 128; i++) {
            const char c = line[len - 1 - i];
            nib[i] = (unsigned char)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
        }
        int32_t x[N], out[N];
        for (int i = 0; i < N; i++) {
            uint64_t v = 0;
            for (int b = 0; b < BITS; b++) {
                const int bit = BITS * i + b;
                if (bit / 4 < 128 && ((nib[bit / 4] >> (bit % 4)) & 1)) v |= (uint64_t)1 << b;
            }
            x[i] = (int32_t)v;
        }
        const int batches = dr::inv_divsteps<N, BITS, MAXB>(P, n0, x, out);
        for (int i = 0; i < N; i++) std::printf("%d ", out[i]);
        std::printf("%d\n", batches);
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "fr") == 0) run<9, 29, 26>(FR_P, FR_N0);
    else run<14, 28, 40>(FQ_P, FQ_N0);
    return 0;
}
