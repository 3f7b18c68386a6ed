// Host build of dot_ring_amd/csrc/divstep28.hip.h for tests/test_divstep_cpu.py: reads one hexadecimal x (< p) per line,
// prints the 14 signed output limbs and the number of 28-step batches.
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include "divstep28.hip.h"

static const uint32_t P[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                               0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
static const uint32_t N0 = 0xffcfffdu;

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        // hex -> 14 limbs of 28 bits = 7 hex digits each
        while (line.size() < 98) line = "0" + line;
        int32_t x[14], out[14];
        for (int i = 0; i < 14; i++) x[i] = (int32_t)std::stoul(line.substr(98 - 7 * (i + 1), 7), nullptr, 16);
        int batches = dr::inv_divsteps28(P, N0, x, out);
        for (int i = 0; i < 14; i++) std::printf("%d ", out[i]);
        std::printf("%d\n", batches);
    }
    return 0;
}
