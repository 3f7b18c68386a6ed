// Host-side hash functions for the Fiat-Shamir / VRF transcripts that sit between GPU phases:
// SHA-512 (FIPS 180-4) and the Keccak sponge behind SHAKE128 / SHAKE256 (FIPS 202).
// The reference hashes with hashlib (dot_ring/vrf/primitives.py:26-55, ring_proof/transcript/transcript.py:21-136,
// curve/curve.py:145-185); a batch of proofs needs ~40 small hashes per proof, which the native prover runs on
// worker threads instead of one interpreter call each.  Checked against hashlib in tests/test_native_host.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace drh {

// ---------------------------------------------------------------- SHA-512
struct Sha512 {
    uint64_t h[8];
    uint8_t buf[128];
    size_t fill = 0;
    uint64_t total = 0;

    Sha512() { reset(); }
    void reset() {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        std::memcpy(h, iv, sizeof h);
        fill = 0;
        total = 0;
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void block(const uint8_t* p) {
        static const uint64_t K[80] = {
            0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
            0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
            0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
            0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
            0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
            0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
            0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
            0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
            0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
            0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
            0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
            0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
            0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
            0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
            0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
            0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
        uint64_t w[80];
        for (int i = 0; i < 16; i++) {
            uint64_t v = 0;
            for (int j = 0; j < 8; j++) v = (v << 8) | p[8 * i + j];
            w[i] = v;
        }
        for (int i = 16; i < 80; i++) {
            uint64_t s0 = rotr(w[i - 15], 1) ^ rotr(w[i - 15], 8) ^ (w[i - 15] >> 7);
            uint64_t s1 = rotr(w[i - 2], 19) ^ rotr(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; i++) {
            uint64_t S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41);
            uint64_t ch = (e & f) ^ (~e & g);
            uint64_t t1 = hh + S1 + ch + K[i] + w[i];
            uint64_t S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
            uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint64_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        total += len;
        if (fill) {
            size_t take = 128 - fill < len ? 128 - fill : len;
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; len -= take;
            if (fill == 128) { block(buf); fill = 0; }
        }
        while (len >= 128) { block(p); p += 128; len -= 128; }
        if (len) { std::memcpy(buf, p, len); fill = len; }
    }
    void final(uint8_t out[64]) {
        uint64_t bits = total * 8;
        uint8_t pad[256] = {0x80};
        size_t padlen = (fill < 112 ? 112 : 240) - fill;
        uint8_t lenb[16] = {0};
        for (int i = 0; i < 8; i++) lenb[15 - i] = (uint8_t)(bits >> (8 * i));
        uint64_t keep = total;
        update(pad, padlen);
        update(lenb, 16);
        total = keep;
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
    }
    static void hash(const void* data, size_t len, uint8_t out[64]) {
        Sha512 s;
        s.update(data, len);
        s.final(out);
    }
};

// ---------------------------------------------------------------- Keccak-f[1600] sponge (SHAKE128: rate 168, SHAKE256: rate 136)
inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PIL[i];
            uint64_t keep = st[j];
            st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = keep;
        }
        for (int j = 0; j < 25; j += 5) {
            uint64_t r[5];
            for (int i = 0; i < 5; i++) r[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] = r[i] ^ (~r[(i + 1) % 5] & r[(i + 2) % 5]);
        }
        st[0] ^= RC[round];
    }
}

template <int RATE>
struct Shake {
    uint64_t st[25];
    size_t pos = 0;          // bytes absorbed into the current block

    Shake() { std::memset(st, 0, sizeof st); }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        while (len) {
            size_t take = RATE - pos < len ? RATE - pos : len;
            for (size_t i = 0; i < take; i++) st[(pos + i) >> 3] ^= (uint64_t)p[i] << (8 * ((pos + i) & 7));
            pos += take; p += take; len -= take;
            if (pos == RATE) { keccak_f1600(st); pos = 0; }
        }
    }
    // digest of everything absorbed so far; like hashlib's .digest(n) the absorbing state is left untouched,
    // so the caller may keep absorbing afterwards
    void digest(uint8_t* out, size_t n) const {
        uint64_t s[25];
        std::memcpy(s, st, sizeof s);
        s[pos >> 3] ^= (uint64_t)0x1f << (8 * (pos & 7));
        s[(RATE - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((RATE - 1) & 7));
        keccak_f1600(s);
        size_t off = 0;
        while (n) {
            size_t take = RATE - off < n ? RATE - off : n;
            for (size_t i = 0; i < take; i++) out[i] = (uint8_t)(s[(off + i) >> 3] >> (8 * ((off + i) & 7)));
            out += take; n -= take; off += take;
            if (off == RATE && n) { keccak_f1600(s); off = 0; }
        }
    }
};
using Shake128 = Shake<168>;
using Shake256 = Shake<136>;

}  // namespace drh
