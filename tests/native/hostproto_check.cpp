// Host check of the verifier's split arithmetic (dot_ring_amd/csrc/hostproto.hpp): batch_inv against single inversions (zeros stay
// zero), te_add_affine against te_add_affine_prep + batch_inv + te_add_affine_finish, ring_verifier_terms against its two halves
// around a batched inversion — on pseudo-random inputs, incl. zeta on the special points (1, w^(n-4): zero denominators) and in the
// domain (refused).  Prints "ok <cases>" or the first difference.  Plain g++, no GPU.
#include <cstdio>
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "hostproto.hpp"

using namespace drh;

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t next64() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static void rand_fr(const Mod256& f, uint64_t out[4]) {
    uint8_t b[40];
    for (int i = 0; i < 40; i++) b[i] = (uint8_t)next64();
    f.reduce_bytes(b, 40, false, out);
}

// the Bandersnatch generator (twisted Edwards form), x || y little-endian
static void generator(uint8_t out[64]) {
    const uint64_t gx[4] = {0xe1e71866a252ae18ULL, 0x2b79c022ad998465ULL, 0x743711777bbe42f3ULL, 0x29c132cc2c0b34c5ULL};
    const uint64_t gy[4] = {0x5e3167b6cc974166ULL, 0x358cad81eee46460ULL, 0x157d8b50badcd586ULL, 0x2a6c669eda123e0fULL};
    store_le32(gx, out);
    store_le32(gy, out + 32);
}

int main() {
    const Mod256& f = mod_p();
    const TeCurveHost* cv = te_curve(0);
    size_t cases = 0;
    uint8_t gen_xy[64];
    generator(gen_xy);
    // 1. batch_inv
    for (int round = 0; round < 50; round++) {
        const size_t n = 1 + next64() % 40;
        std::vector<uint64_t> v(4 * n), want(4 * n);
        for (size_t k = 0; k < n; k++) {
            rand_fr(f, &v[4 * k]);
            if (next64() % 7 == 0) std::memset(&v[4 * k], 0, 32);
            if (f.is_zero(&v[4 * k])) std::memset(&want[4 * k], 0, 32); else f.inv(&v[4 * k], &want[4 * k]);
        }
        batch_inv(f, reinterpret_cast<uint64_t(*)[4]>(v.data()), n);
        if (std::memcmp(v.data(), want.data(), 32 * n)) { printf("batch_inv differs (round %d)\n", round); return 1; }
        cases += n;
    }
    // 2. te_add_affine: multiples of the generator (points of the curve)
    {
        uint8_t g[64];
        std::memcpy(g, gen_xy, 64);
        uint8_t p[64], q[64];
        std::memcpy(p, g, 64);
        std::memcpy(q, g, 64);
        const size_t n = 24;
        TeAddPending pd[n];
        uint64_t den[n][4];
        uint8_t want[n][64], a[n][64], b[n][64];
        for (size_t k = 0; k < n; k++) {
            te_add_affine(*cv, p, g, p);            // p = (k + 2) G
            te_add_affine(*cv, q, p, q);            // q = triangular multiples
            std::memcpy(a[k], p, 64); std::memcpy(b[k], q, 64);
            te_add_affine(*cv, a[k], b[k], want[k]);
            te_add_affine_prep(*cv, a[k], b[k], pd[k]);
            std::memcpy(den[k], pd[k].den, 32);
        }
        batch_inv(f, den, n);
        for (size_t k = 0; k < n; k++) {
            uint8_t got[64];
            te_add_affine_finish(pd[k], den[k], got);
            if (std::memcmp(got, want[k], 64)) { printf("te_add_affine differs (%zu)\n", k); return 1; }
        }
        cases += n;
    }
    // 3. ring_verifier_terms
    {
        uint64_t w[4];
        // a 2^11-th root of unity of Fr: 7^((p-1)/2^11)
        static const uint64_t E[4] = {0x7fdfffffffe00000ULL, 0x00aa77b4805fffcbULL, 0xa906673b0101343bULL, 0x000e7db4ea6533afULL};   // (p - 1) >> 11
        uint64_t seven[4] = {7, 0, 0, 0};
        f.pow(seven, E, w);
        uint8_t omega[32], seed[64];
        store_le32(w, omega);
        std::memcpy(seed, gen_xy, 64);
        RingVerifierDomain dm;
        dm.init(11, omega, seed);
        const size_t n = 40;
        std::vector<uint8_t> al(n * 224), nus(n * 256), zeta(n * 32), ev(n * 224), lzw(n * 32);
        RingTermsPending pd[n];
        uint64_t den[n][4];
        bool ok[n];
        uint64_t t[4];
        for (size_t k = 0; k < n; k++) {
            for (int i = 0; i < 7; i++) { rand_fr(f, t); store_le32(t, &al[224 * k + 32 * i]); rand_fr(f, t); store_le32(t, &ev[224 * k + 32 * i]); }
            for (int i = 0; i < 8; i++) { rand_fr(f, t); store_le32(t, &nus[256 * k + 32 * i]); }
            rand_fr(f, t); store_le32(t, &lzw[32 * k]);
            rand_fr(f, t);
            if (k == 3) f.set_u64(1, t);                                  // zeta = 1: in the domain -> refused
            if (k == 5) std::memcpy(t, dm.w_nm4, 32);                     // zeta = w^(n-4): in the domain -> refused
            if (k == 9) std::memcpy(t, dm.omega, 32);
            store_le32(t, &zeta[32 * k]);
            ok[k] = ring_verifier_terms_prep(dm, &zeta[32 * k], pd[k]);
            if (ok[k]) std::memcpy(den[k], pd[k].prod, 32); else std::memset(den[k], 0, 32);
        }
        batch_inv(f, den, n);
        for (size_t k = 0; k < n; k++) {
            RingClaimScalars want, got;
            std::memset(&want, 0, sizeof want); std::memset(&got, 0, sizeof got);
            const bool w_ok = ring_verifier_terms(*cv, dm, &al[224 * k], &nus[256 * k], &zeta[32 * k], &ev[224 * k], &lzw[32 * k], seed, want);
            if (w_ok != ok[k]) { printf("ring_verifier_terms: acceptance differs (%zu)\n", k); return 1; }
            if (w_ok == (k == 3 || k == 5 || k == 9)) { printf("zeta in the domain must be refused, and only then (%zu)\n", k); return 1; }
            if (!w_ok) continue;
            ring_verifier_terms_finish(*cv, dm, &al[224 * k], &nus[256 * k], &zeta[32 * k], &ev[224 * k], &lzw[32 * k], seed, pd[k], den[k], got);
            if (std::memcmp(&want, &got, sizeof want)) { printf("ring_verifier_terms differs (%zu)\n", k); return 1; }
        }
        cases += n;
    }
    // 4. parallel_for (the worker pool every host phase of the batch calls goes through): every index exactly once for any size and grain,
    //    also with two threads posting jobs at the same time; an exception thrown by an item reaches the caller
    {
        const size_t sizes[] = {0, 1, 15, 16, 17, 63, 64, 65, 256, 1000, 1024, 4099};
        const size_t grains[] = {1, 16, 100};
        for (size_t n : sizes)
            for (size_t g : grains) {
                std::vector<std::atomic<int>> hits(n);
                for (auto& h : hits) h.store(0);
                parallel_for(n, [&](size_t i) { hits[i].fetch_add(1); }, g);
                for (size_t i = 0; i < n; i++)
                    if (hits[i].load() != 1) { printf("parallel_for: index %zu of %zu (grain %zu) visited %d times\n", i, n, g, hits[i].load()); return 1; }
                cases++;
            }
        std::vector<std::atomic<int>> a(5000), b(3000);
        for (auto& h : a) h.store(0);
        for (auto& h : b) h.store(0);
        std::thread other([&] { for (int r = 0; r < 20; r++) parallel_for(b.size(), [&](size_t i) { b[i].fetch_add(1); }); });
        for (int r = 0; r < 20; r++) parallel_for(a.size(), [&](size_t i) { a[i].fetch_add(1); });
        other.join();
        for (auto& h : a) if (h.load() != 20) { printf("parallel_for: concurrent jobs lost or repeated an item\n"); return 1; }
        for (auto& h : b) if (h.load() != 20) { printf("parallel_for: concurrent jobs lost or repeated an item\n"); return 1; }
        bool thrown = false;
        try {
            parallel_for(2048, [&](size_t i) { if (i == 1234) throw std::runtime_error("item 1234"); });
        } catch (const std::runtime_error& e) { thrown = std::strcmp(e.what(), "item 1234") == 0; }
        if (!thrown) { printf("parallel_for: the exception of an item did not reach the caller\n"); return 1; }
        std::atomic<size_t> after{0};
        parallel_for(512, [&](size_t) { after.fetch_add(1); });           // the pool still works after a failed job
        if (after.load() != 512) { printf("parallel_for: pool unusable after an exception\n"); return 1; }
        cases += 3;
    }
    printf("ok %zu\n", cases);
    return 0;
}
