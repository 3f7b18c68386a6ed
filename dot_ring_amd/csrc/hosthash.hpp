// Host-side hash functions for the Fiat-Shamir / VRF transcripts that sit between GPU phases:
// SHA-512 (FIPS 180-4) and the Keccak sponge behind SHAKE128 / SHAKE256 (FIPS 202).
// The reference hashes with hashlib (dot_ring/vrf/primitives.py:26-55, ring_proof/transcript/transcript.py:21-136,
// curve/curve.py:145-185); a batch of proofs needs ~40 small hashes per proof, which the native prover runs on
// worker threads instead of one interpreter call each.  Checked against hashlib in tests/test_native_host.py.
#pragma once
#include <cstdint>
#include <cstdlib>
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
// Fully unrolled rounds on 25 local lanes (theta, rho + pi, chi, iota fused; rotation counts and lane moves are compile-time):
// the transcript replay of a batch verifier is ~20 permutations per proof, i.e. this function is most of its host time.
// The body is a template over the lane type: uint64_t for one state, a 4 x 64-bit vector for FOUR states in lockstep (the
// transcripts of four proofs absorb the same lengths in the same order, so their sponges permute at the same moments; with
// AVX2 one vector instruction serves four permutations, ~3.5x the throughput of the scalar code).
typedef uint64_t u64x4 __attribute__((vector_size(32)));
template <class V>
__attribute__((always_inline)) inline V keccak_rol(V v, int n) { return (v << n) | (v >> (64 - n)); }
template <class V>
__attribute__((always_inline)) inline void keccak_f1600_t(V* st) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    V a00 = st[0], a01 = st[1], a02 = st[2], a03 = st[3], a04 = st[4], a05 = st[5], a06 = st[6], a07 = st[7], a08 = st[8], a09 = st[9],
             a10 = st[10], a11 = st[11], a12 = st[12], a13 = st[13], a14 = st[14], a15 = st[15], a16 = st[16], a17 = st[17], a18 = st[18],
             a19 = st[19], a20 = st[20], a21 = st[21], a22 = st[22], a23 = st[23], a24 = st[24];
    for (int round = 0; round < 24; round++) {
        // theta
        const V c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                       c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        const V d0 = c4 ^ keccak_rol(c1, 1), d1 = c0 ^ keccak_rol(c2, 1), d2 = c1 ^ keccak_rol(c3, 1), d3 = c2 ^ keccak_rol(c4, 1),
                       d4 = c3 ^ keccak_rol(c0, 1);
        // rho + pi: b[y][2x+3y] = rol(a[x][y] ^ d[x], r[x][y])   (lane index = x + 5y)
        const V b00 = a00 ^ d0;
        const V b10 = keccak_rol(a01 ^ d1, 1), b20 = keccak_rol(a02 ^ d2, 62), b05 = keccak_rol(a03 ^ d3, 28), b15 = keccak_rol(a04 ^ d4, 27);
        const V b16 = keccak_rol(a05 ^ d0, 36), b01 = keccak_rol(a06 ^ d1, 44), b11 = keccak_rol(a07 ^ d2, 6), b21 = keccak_rol(a08 ^ d3, 55),
                       b06 = keccak_rol(a09 ^ d4, 20);
        const V b07 = keccak_rol(a10 ^ d0, 3), b17 = keccak_rol(a11 ^ d1, 10), b02 = keccak_rol(a12 ^ d2, 43), b12 = keccak_rol(a13 ^ d3, 25),
                       b22 = keccak_rol(a14 ^ d4, 39);
        const V b23 = keccak_rol(a15 ^ d0, 41), b08 = keccak_rol(a16 ^ d1, 45), b18 = keccak_rol(a17 ^ d2, 15), b03 = keccak_rol(a18 ^ d3, 21),
                       b13 = keccak_rol(a19 ^ d4, 8);
        const V b14 = keccak_rol(a20 ^ d0, 18), b24 = keccak_rol(a21 ^ d1, 2), b09 = keccak_rol(a22 ^ d2, 61), b19 = keccak_rol(a23 ^ d3, 56),
                       b04 = keccak_rol(a24 ^ d4, 14);
        // chi + iota
        a00 = b00 ^ (~b01 & b02) ^ RC[round]; a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
    }
    st[0] = a00; st[1] = a01; st[2] = a02; st[3] = a03; st[4] = a04; st[5] = a05; st[6] = a06; st[7] = a07; st[8] = a08; st[9] = a09;
    st[10] = a10; st[11] = a11; st[12] = a12; st[13] = a13; st[14] = a14; st[15] = a15; st[16] = a16; st[17] = a17; st[18] = a18; st[19] = a19;
    st[20] = a20; st[21] = a21; st[22] = a22; st[23] = a23; st[24] = a24;
}

inline void keccak_f1600(uint64_t st[25]) { keccak_f1600_t<uint64_t>(st); }
__attribute__((target("avx2"))) inline void keccak_f1600_x4_avx2(u64x4* st) { keccak_f1600_t<u64x4>(st); }
inline void keccak_f1600_x4_generic(u64x4* st) { keccak_f1600_t<u64x4>(st); }       // the compiler splits the vectors (SSE2)
inline void keccak_f1600_x4(u64x4* st) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && std::getenv("DOTRING_KECCAK_GENERIC") == nullptr;   // (the knob: tests)
    if (avx2) keccak_f1600_x4_avx2(st); else keccak_f1600_x4_generic(st);
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
            size_t i = 0;
            for (; i < take && ((pos + i) & 7); i++) st[(pos + i) >> 3] ^= (uint64_t)p[i] << (8 * ((pos + i) & 7));
            for (; i + 8 <= take; i += 8) {          // whole lanes (x86-64: little-endian, as Keccak's byte order)
                uint64_t w;
                std::memcpy(&w, p + i, 8);
                st[(pos + i) >> 3] ^= w;
            }
            for (; i < take; i++) st[(pos + i) >> 3] ^= (uint64_t)p[i] << (8 * ((pos + i) & 7));
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
            std::memcpy(out, reinterpret_cast<const uint8_t*>(s) + off, take);      // little-endian lanes
            out += take; n -= take; off += take;
            if (off == RATE && n) { keccak_f1600(s); off = 0; }
        }
    }
};
using Shake128 = Shake<168>;
using Shake256 = Shake<136>;

// four sponges in lockstep: every update gives each of them the same number of bytes
template <int RATE>
struct Shake4 {
    u64x4 st[25];
    uint8_t buf[4][RATE];    // the current block of each stream, XORed into the state when it is full
    size_t pos = 0;

    Shake4() { std::memset(st, 0, sizeof st); }
    static void xor_block(u64x4* s, const uint8_t (*blk)[RATE]) {
        for (size_t l = 0; l < RATE / 8; l++) {
            uint64_t w[4];
            for (int k = 0; k < 4; k++) std::memcpy(&w[k], blk[k] + 8 * l, 8);
            s[l] ^= u64x4{w[0], w[1], w[2], w[3]};
        }
    }
    void update(const uint8_t* const p[4], size_t len) {
        size_t done = 0;
        while (done < len) {
            const size_t take = RATE - pos < len - done ? RATE - pos : len - done;
            for (int k = 0; k < 4; k++) std::memcpy(buf[k] + pos, p[k] + done, take);
            pos += take; done += take;
            if (pos == RATE) { xor_block(st, buf); keccak_f1600_x4(st); pos = 0; }
        }
    }
    void update_same(const void* data, size_t len) {
        const uint8_t* q = (const uint8_t*)data;
        const uint8_t* p[4] = {q, q, q, q};
        update(p, len);
    }
    // digests of everything absorbed so far (n <= RATE bytes each); the absorbing state is left untouched
    void digest(uint8_t* const out[4], size_t n) const {
        u64x4 s[25];
        std::memcpy(s, st, sizeof s);
        uint8_t b[4][RATE];                                    // the partial blocks, padded
        for (int k = 0; k < 4; k++) {
            std::memcpy(b[k], buf[k], pos);
            std::memset(b[k] + pos, 0, RATE - pos);
            b[k][pos] ^= 0x1f;
            b[k][RATE - 1] ^= 0x80;
        }
        xor_block(s, b);
        keccak_f1600_x4(s);
        for (int k = 0; k < 4; k++)
            for (size_t i = 0; i < n; i++) out[k][i] = (uint8_t)(s[i >> 3][k] >> (8 * (i & 7)));
    }
};
using Shake128x4 = Shake4<168>;

}  // namespace drh
