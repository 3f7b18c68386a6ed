// Host-side protocol glue of the batched Ring-VRF prover: everything the reference does in interpreted Python
// BETWEEN the GPU phases of a proof — hash_to_field, the VRF transcript (nonces, challenge), the ring proof's
// Fiat-Shamir transcript, scalar arithmetic mod the group order, the byte encodings — for a whole batch at once on
// worker threads.  Follows, as text:
//   hash_to_field / expand_message_xmd       dot_ring/curve/curve.py:110-185 (Z_pad = 48 zero bytes in this suite)
//   VRF transcript, nonce, challenge         dot_ring/vrf/primitives.py:26-122
//   Pedersen prove                           dot_ring/vrf/pedersen/vrf.py:86-126
//   Fiat-Shamir transcript + phases          dot_ring/ring_proof/transcript/transcript.py:21-136, phases.py:18-69
//   point / scalar codecs                    dot_ring/vrf/codec.py:9-45, dot_ring/curve/point.py:150-214
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hosthash.hpp"
#include "hostmath.hpp"

namespace drh {

// ---------------------------------------------------------------- worker threads
inline unsigned host_threads() {
    static unsigned cached = [] {
        const char* e = std::getenv("DOTRING_HOST_THREADS");
        unsigned n = e ? (unsigned)std::atoi(e) : 0;
        if (n == 0) {
            // 16, not more: 24 - 48 threads gave +0.7 % proofs/s on a GPU box whose 256-thread host was lightly loaded (load average 18 - 26)
            // and -1 % on one under load (52) — the other tenants' load is not ours to choose (DOTRING_HOST_THREADS overrides)
            n = std::thread::hardware_concurrency();
            if (n == 0) n = 1;
            if (n > 16) n = 16;
        }
        return n;
    }();
    return cached;
}
// A persistent pool of host_threads() - 1 workers: starting 16 threads costs ~0.35 ms, and one batch goes through a dozen
// short parallel loops (0.1-0.5 ms of hashing each).  Several threads may run parallel loops at once (the prover's main
// thread and its Pedersen helper): jobs queue up, every worker drains the oldest one, and the CALLER works on its own job too,
// so a loop finishes even when no worker is free (or after a fork, when none exists).
class WorkerPool {
    struct Job {
        const std::function<void(size_t)>* f;
        size_t n, chunk, chunks;
        std::atomic<size_t> next{0}, done{0};
        std::mutex m;
        std::condition_variable cv;
        std::exception_ptr error;          // first exception thrown by f on any thread; rethrown by run() on the caller
    };
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;

    static void drain(Job& j) {
        for (;;) {
            size_t c = j.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= j.chunks) return;
            size_t lo = c * j.chunk, hi = std::min(j.n, lo + j.chunk);
            try {
                for (size_t i = lo; i < hi; i++) (*j.f)(i);
            } catch (...) {                 // an exception escaping a worker thread would terminate the host process
                std::lock_guard<std::mutex> lk(j.m);
                if (!j.error) j.error = std::current_exception();
            }
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.chunks) {
                std::lock_guard<std::mutex> lk(j.m);
                j.cv.notify_all();
            }
        }
    }
    void worker() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
                if (stop_) return;
                j = jobs_.front();
            }
            drain(*j);
            std::lock_guard<std::mutex> lk(m_);
            if (!jobs_.empty() && jobs_.front() == j) jobs_.pop_front();          // all its chunks are taken
        }
    }

public:
    explicit WorkerPool(unsigned workers) {
        for (unsigned k = 0; k < workers; k++) workers_.emplace_back([this] { worker(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void run(size_t n, unsigned slices, const std::function<void(size_t)>& f) {
        auto j = std::make_shared<Job>();
        j->f = &f;
        j->n = n;
        j->chunk = (n + slices - 1) / slices;
        j->chunks = (n + j->chunk - 1) / j->chunk;
        {
            std::lock_guard<std::mutex> lk(m_);
            jobs_.push_back(j);
        }
        cv_.notify_all();
        drain(*j);
        {
            std::lock_guard<std::mutex> lk(m_);
            auto it = std::find(jobs_.begin(), jobs_.end(), j);
            if (it != jobs_.end()) jobs_.erase(it);
        }
        std::unique_lock<std::mutex> lk(j->m);
        j->cv.wait(lk, [&] { return j->done.load(std::memory_order_acquire) == j->chunks; });
        if (j->error) std::rethrow_exception(j->error);
    }
};
inline WorkerPool* worker_pool() {
    if (host_threads() <= 1) return nullptr;
    static WorkerPool pool(host_threads() - 1);
    return &pool;
}
// f(i) for i in [0, n).  Short loops stay on the calling thread.  An exception thrown by f on a worker (std::bad_alloc from a
// staging buffer) is carried to the calling thread and rethrown there, where the C entry points' try / catch turn it into a
// status code.
// `grain`: the fewest items worth a thread of their own — 16 for the usual few-microsecond items; 1 where an item is itself a slice of work
// (the batch verifier's claim scalars go in slices of 16 proofs around a shared inversion: 64 items that ran on FOUR threads until round 4)
template <class F>
void parallel_for(size_t n, F f, size_t grain = 16) {
    unsigned t = host_threads();
    if (t > n / grain) t = (unsigned)(n / grain);
    if (t <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    if (WorkerPool* pool = worker_pool()) {
        const std::function<void(size_t)> fn = [&f](size_t i) { f(i); };
        // four chunks per thread, taken on demand: a worker that wakes late or loses its core to another tenant of the host holds up a
        // quarter of a share, not a whole one (the phases between the GPU calls are 0.1 - 0.5 ms long; one late thread used to double them)
        pool->run(n, 4 * t, fn);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(t);
    std::mutex em;
    std::exception_ptr error;
    for (unsigned k = 0; k < t; k++)
        pool.emplace_back([=, &em, &error] {
            size_t lo = n * k / t, hi = n * (k + 1) / t;
            try {
                for (size_t i = lo; i < hi; i++) f(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(em);
                if (!error) error = std::current_exception();
            }
        });
    for (auto& th : pool) th.join();
    if (error) std::rethrow_exception(error);
}

// ---------------------------------------------------------------- arithmetic modulo a runtime odd 256-bit modulus
struct Mod256 {
    uint64_t m[4], r2[4], n0;

    static bool geq(const uint64_t* a, const uint64_t* b) {
        for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
        return true;
    }
    void add(const uint64_t* a, const uint64_t* b, uint64_t* r) const {
        u128 c = 0; uint64_t t[4];
        for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
        if (c || geq(t, m)) { uint64_t bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - m[i] - bw; t[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); } }
        std::memcpy(r, t, 32);
    }
    void sub(const uint64_t* a, const uint64_t* b, uint64_t* r) const {
        uint64_t t[4], bw = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - bw; t[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); }
        if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + m[i]; t[i] = (uint64_t)c; c >>= 64; } }
        std::memcpy(r, t, 32);
    }
    void mont(const uint64_t* a, const uint64_t* b, uint64_t* r) const {      // a*b/R mod m
        uint64_t t[6] = {0};
        for (int i = 0; i < 4; i++) {
            uint64_t c = 0;
            for (int j = 0; j < 4; j++) { u128 p = (u128)a[j] * b[i] + t[j] + c; t[j] = (uint64_t)p; c = (uint64_t)(p >> 64); }
            u128 top = (u128)t[4] + c; t[4] = (uint64_t)top; t[5] = (uint64_t)(top >> 64);
            uint64_t q = t[0] * n0;
            u128 p = (u128)q * m[0] + t[0]; c = (uint64_t)(p >> 64);
            for (int j = 1; j < 4; j++) { p = (u128)q * m[j] + t[j] + c; t[j - 1] = (uint64_t)p; c = (uint64_t)(p >> 64); }
            top = (u128)t[4] + c; t[3] = (uint64_t)top; t[4] = t[5] + (uint64_t)(top >> 64);
        }
        if (t[4] || geq(t, m)) { uint64_t bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - m[i] - bw; t[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); } }
        std::memcpy(r, t, 32);
    }
    void mul(const uint64_t* a, const uint64_t* b, uint64_t* r) const {       // standard form in and out
        uint64_t t[4];
        mont(a, b, t);
        mont(t, r2, r);
    }
    void init(const uint64_t mod[4]) {
        std::memcpy(m, mod, 32);
        uint64_t inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - m[0] * inv;       // m^-1 mod 2^64
        n0 = (uint64_t)0 - inv;
        uint64_t t[4] = {1, 0, 0, 0};
        for (int i = 0; i < 512; i++) add(t, t, t);              // 2^512 mod m = R^2
        std::memcpy(r2, t, 32);
    }
    // big-endian or little-endian byte string of any length -> value mod m
    void reduce_bytes(const uint8_t* p, size_t len, bool big_endian, uint64_t out[4]) const {
        uint64_t acc[4] = {0, 0, 0, 0};
        const uint64_t shift[4] = {0, 1, 0, 0};                  // 2^64
        size_t limbs = (len + 7) / 8;
        for (size_t k = limbs; k-- > 0;) {                       // most significant limb first
            uint64_t w = 0;
            for (int j = 7; j >= 0; j--) {
                size_t idx = 8 * k + j;                          // little-endian byte index
                if (idx >= len) continue;
                w = (w << 8) | (big_endian ? p[len - 1 - idx] : p[idx]);
            }
            uint64_t limb[4] = {w, 0, 0, 0};
            mul(acc, shift, acc);
            add(acc, limb, acc);
        }
        std::memcpy(out, acc, 32);
    }
    bool is_zero(const uint64_t* a) const { return (a[0] | a[1] | a[2] | a[3]) == 0; }
    void pow(const uint64_t* a, const uint64_t e[4], uint64_t* r) const {     // standard form
        uint64_t acc[4] = {1, 0, 0, 0}, base[4];
        std::memcpy(base, a, 32);
        bool started = false;
        for (int i = 3; i >= 0; i--)
            for (int b = 63; b >= 0; b--) {
                if (started) mul(acc, acc, acc);
                if ((e[i] >> b) & 1) { if (started) mul(acc, base, acc); else std::memcpy(acc, base, 32); started = true; }
            }
        std::memcpy(r, acc, 32);
    }
    void inv_fermat(const uint64_t* a, uint64_t* r) const {                   // a^(m-2); 0 -> 0
        uint64_t e[4];
        std::memcpy(e, m, 32);
        uint64_t bw = 2;
        for (int i = 0; i < 4 && bw; i++) { uint64_t old = e[i]; e[i] -= bw; bw = old < bw ? 1 : 0; }
        pow(a, e, r);
    }
    // a^-1 mod m for a < m, m an odd prime (0 -> 0): binary extended Euclid on the standard-form value, ~2 us against ~40 us for
    // the 766 Montgomery products of the Fermat power — the verifier's scalar pass inverts twice per proof
    void inv(const uint64_t* a, uint64_t* r) const {
        if (is_zero(a)) { r[0] = r[1] = r[2] = r[3] = 0; return; }
        uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
        std::memcpy(u, a, 32);
        std::memcpy(v, m, 32);
        auto is_one = [](const uint64_t* t) { return t[0] == 1 && (t[1] | t[2] | t[3]) == 0; };
        auto shr1 = [](uint64_t* t, uint64_t top) {
            t[0] = (t[0] >> 1) | (t[1] << 63); t[1] = (t[1] >> 1) | (t[2] << 63); t[2] = (t[2] >> 1) | (t[3] << 63); t[3] = (t[3] >> 1) | (top << 63);
        };
        auto halve = [&](uint64_t* x) {                                       // x / 2 mod m
            uint64_t top = 0;
            if (x[0] & 1) {
                u128 c = 0;
                for (int i = 0; i < 4; i++) { c += (u128)x[i] + m[i]; x[i] = (uint64_t)c; c >>= 64; }
                top = (uint64_t)c;
            }
            shr1(x, top);
        };
        auto sub_raw = [](uint64_t* x, const uint64_t* y) {                   // x -= y, x >= y
            uint64_t bw = 0;
            for (int i = 0; i < 4; i++) { u128 d = (u128)x[i] - y[i] - bw; x[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); }
        };
        while (!is_one(u) && !is_one(v)) {
            while (!(u[0] & 1)) { shr1(u, 0); halve(x1); }
            while (!(v[0] & 1)) { shr1(v, 0); halve(x2); }
            if (geq(u, v)) { sub_raw(u, v); sub(x1, x2, x1); }
            else { sub_raw(v, u); sub(x2, x1, x2); }
        }
        std::memcpy(r, is_one(u) ? x1 : x2, 32);
    }
    void neg(const uint64_t* a, uint64_t* r) const { const uint64_t z[4] = {0, 0, 0, 0}; sub(z, a, r); }
    void set_u64(uint64_t v, uint64_t* r) const { r[0] = v; r[1] = r[2] = r[3] = 0; }
    static bool eq(const uint64_t* a, const uint64_t* b) { return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) == 0; }
};
inline void store_le32(const uint64_t v[4], uint8_t* out) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(v[i] >> (8 * j));
}
inline void load_le32(const uint8_t* in, uint64_t v[4]) {
    for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 7; j >= 0; j--) w = (w << 8) | in[8 * i + j]; v[i] = w; }
}
inline const Mod256& mod_n() {       // Bandersnatch prime-order subgroup (bandersnatch.py:58-67)
    static Mod256 s = [] { Mod256 t; const uint64_t n[4] = {0x74fd06b52876e7e1ULL, 0xff8f870074190471ULL, 0x0cce760202687600ULL, 0x1cfb69d4ca675f52ULL}; t.init(n); return t; }();
    return s;
}
inline const Mod256& mod_p() {       // Bandersnatch base field = BLS12-381 scalar field
    static Mod256 s = [] { Mod256 t; t.init(FieldParams<4>::P); return t; }();
    return s;
}

// The twisted Edwards curves over that field the library serves (ids as DR_CURVE_* in dotring_hip.h and CV_* in
// curve.hip.h): Bandersnatch (specs/bandersnatch.py:57-72) and JubJub (specs/jubjub.py:17-29).
struct TeCurveHost {
    int id;
    Mod256 n;                 // prime-order subgroup
    uint64_t d[4];            // curve coefficient d, standard form
    uint64_t neg_a[4];        // -a as a small integer: 5 or 1
    unsigned scalar_bits;     // bit length of n
    bool glv;                 // has the endomorphism the lane-pair kernels use
    bool tai;                 // hash-to-curve by try-and-increment (otherwise Elligator 2)
};
inline const TeCurveHost* te_curve(int id) {
    static const TeCurveHost curves[2] = {
        [] { TeCurveHost c{}; c.id = 0; c.n = mod_n();
             const uint64_t d[4] = {0xb369f2f5188d58e7ULL, 0xcb66677177e54f92ULL, 0xc66e3bf86be3b6d8ULL, 0x6389c12633c267cbULL};
             std::memcpy(c.d, d, 32); c.neg_a[0] = 5; c.scalar_bits = 253; c.glv = true; c.tai = false; return c; }(),
        [] { TeCurveHost c{}; c.id = 1;
             const uint64_t n[4] = {0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL};
             c.n.init(n);
             const uint64_t d[4] = {0x01065fd6d6343eb1ULL, 0x292d7f6d37579d26ULL, 0xf5fd9207e6bd7fd4ULL, 0x2a9318e74bfa2b48ULL};
             std::memcpy(c.d, d, 32); c.neg_a[0] = 1; c.scalar_bits = 252; c.glv = false; c.tai = true; return c; }(),
    };
    return id == 0 || id == 1 ? &curves[id] : nullptr;
}

// ---------------------------------------------------------------- GLV decomposition (dot_ring/curve/glv.py:57-160)
// k = k1 + k2*lambda (mod n) with |k1|, |k2| < 2^128, from the lattice basis v1 = (a1, b1), v2 = (a2, -a1) the reference
// finds by extended Euclid on (n, lambda) (det = -n):  c1 = k*a1/n, c2 = k*b1/n (here: floor via 2^256-scaled
// reciprocals — any c within 1 of the rounded value still gives a valid, slightly longer pair; the result of the scalar
// multiplication is the same group element), k1 = k - c1*a1 - c2*a2, k2 = c2*a1 - c1*b1.
struct GlvSplit {
    uint64_t k1[2], k2[2];
    int neg1, neg2;
};
inline void mul_limbs(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out) {
    for (int i = 0; i < na + nb; i++) out[i] = 0;
    for (int i = 0; i < na; i++) {
        uint64_t c = 0;
        for (int j = 0; j < nb; j++) { u128 p = (u128)a[i] * b[j] + out[i + j] + c; out[i + j] = (uint64_t)p; c = (uint64_t)(p >> 64); }
        out[i + nb] = c;
    }
}
inline bool glv_decompose(const uint64_t k[4], GlvSplit& out) {
    static const uint64_t A1[2] = {0x4b02f94a9789181fULL, 0x555fe2004be6928eULL};
    static const uint64_t B1[2] = {0xf8e2591a23d61f44ULL, 0x0814b3eee55e8f5dULL};
    static const uint64_t A2[2] = {0xf1c4b23447ac3e88ULL, 0x102967ddcabd1ebbULL};
    static const uint64_t G1[3] = {0xdebac77a3f4747c1ULL, 0xf21df5b0541cf632ULL, 0x0000000000000002ULL};   // floor(2^256 a1 / n)
    static const uint64_t G2[3] = {0x993b75e7547768aaULL, 0x4760f127d8767bdeULL, 0x0000000000000000ULL};   // floor(2^256 b1 / n)
    uint64_t t7[7], c1[2], c2[2];
    mul_limbs(k, 4, G1, 3, t7); c1[0] = t7[4]; c1[1] = t7[5]; if (t7[6]) return false;
    mul_limbs(k, 4, G2, 3, t7); c2[0] = t7[4]; c2[1] = t7[5]; if (t7[6]) return false;
    uint64_t p1[4], p2[4], s[4], d[4];
    auto sub4 = [](const uint64_t* a, const uint64_t* b, uint64_t* r) {          // r = a - b, returns borrow
        uint64_t bw = 0;
        for (int i = 0; i < 4; i++) { u128 v = (u128)a[i] - b[i] - bw; r[i] = (uint64_t)v; bw = (uint64_t)(v >> 127); }
        return bw;
    };
    auto finish = [](uint64_t* v, uint64_t borrow, uint64_t* mag, int& neg) {     // signed 256-bit -> 128-bit magnitude
        neg = borrow ? 1 : 0;
        if (borrow) { uint64_t c = 1; for (int i = 0; i < 4; i++) { u128 t = (u128)(~v[i]) + c; v[i] = (uint64_t)t; c = (uint64_t)(t >> 64); } }
        mag[0] = v[0]; mag[1] = v[1];
        return (v[2] | v[3]) == 0;
    };
    // k1 = k - (c1*a1 + c2*a2)
    mul_limbs(c1, 2, A1, 2, p1);
    mul_limbs(c2, 2, A2, 2, p2);
    u128 cy = 0;
    for (int i = 0; i < 4; i++) { cy += (u128)p1[i] + p2[i]; s[i] = (uint64_t)cy; cy >>= 64; }
    if (cy) return false;
    uint64_t bw = sub4(k, s, d);
    if (!finish(d, bw, out.k1, out.neg1)) return false;
    // k2 = c2*a1 - c1*b1
    mul_limbs(c2, 2, A1, 2, p1);
    mul_limbs(c1, 2, B1, 2, p2);
    bw = sub4(p1, p2, d);
    return finish(d, bw, out.k2, out.neg2);
}

// ---------------------------------------------------------------- byte helpers
using Bytes = std::vector<uint8_t>;
inline void put(Bytes& b, const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
inline void put8(Bytes& b, uint8_t v) { b.push_back(v); }
inline void put_le64(Bytes& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }

// compressed Twisted-Edwards point: y little-endian, bit 255 set iff x > p - x  (point.py:150-214)
inline void enc_te_point(const uint8_t xy[64], uint8_t out[32]) {
    uint64_t x[4], nx[4], zero[4] = {0, 0, 0, 0};
    load_le32(xy, x);
    mod_p().sub(zero, x, nx);
    std::memcpy(out, xy + 32, 32);
    bool gt = false;
    for (int i = 3; i >= 0; i--) { if (x[i] != nx[i]) { gt = x[i] > nx[i]; break; } }
    if (gt) out[31] |= 0x80;
}

// affine twisted-Edwards addition: the verifier's seed + relation (ring/vrf.py:239-283)
// in two halves around the one inversion, so that a caller with many additions can invert all denominators together (batch_inv)
struct TeAddPending { uint64_t e[4], t[4], dx[4], dy[4], den[4]; };
inline void te_add_affine_prep(const TeCurveHost& cv, const uint8_t p1[64], const uint8_t p2[64], TeAddPending& pd) {
    const uint64_t* D = cv.d;
    const uint64_t* five = cv.neg_a;        // -a (5 on Bandersnatch)
    const Mod256& f = mod_p();
    uint64_t x1[4], y1[4], x2[4], y2[4], a[4], b[4], c[4], t[4], one[4];
    load_le32(p1, x1); load_le32(p1 + 32, y1); load_le32(p2, x2); load_le32(p2 + 32, y2);
    f.set_u64(1, one);
    f.mul(x1, x2, a);                       // x1 x2
    f.mul(y1, y2, b);                       // y1 y2
    f.mul(a, b, c); f.mul(c, D, c);         // d x1 x2 y1 y2
    f.mul(x1, y2, pd.e); f.mul(y1, x2, t); f.add(pd.e, t, pd.e);          // x numerator
    f.mul(a, five, t); f.add(b, t, pd.t);                                 // y numerator: y1 y2 - a x1 x2
    f.add(one, c, pd.dx); f.sub(one, c, pd.dy);
    f.mul(pd.dx, pd.dy, pd.den);                                          // non-zero for points of the curve (complete law)
}
inline void te_add_affine_finish(const TeAddPending& pd, const uint64_t inv_den[4], uint8_t out[64]) {
    const Mod256& f = mod_p();
    uint64_t ix[4], iy[4], x3[4], y3[4];
    f.mul(inv_den, pd.dy, ix); f.mul(inv_den, pd.dx, iy);
    f.mul(pd.e, ix, x3); f.mul(pd.t, iy, y3);
    store_le32(x3, out); store_le32(y3, out + 32);
}
inline void te_add_affine(const TeCurveHost& cv, const uint8_t p1[64], const uint8_t p2[64], uint8_t out[64]) {
    TeAddPending pd;
    te_add_affine_prep(cv, p1, p2, pd);
    uint64_t inv[4];
    mod_p().inv(pd.den, inv);                                             // 1 / (dx dy)
    te_add_affine_finish(pd, inv, out);
}
// vals[k] <- vals[k]^-1 for n values with ONE inversion (Montgomery's trick); zeros stay zero
inline void batch_inv(const Mod256& f, uint64_t (*vals)[4], size_t n) {
    std::vector<uint64_t> pre(4 * (n + 1));
    uint64_t one[4], acc[4];
    f.set_u64(1, one);
    std::memcpy(acc, one, 32);
    for (size_t k = 0; k < n; k++) {
        std::memcpy(&pre[4 * k], acc, 32);
        if (!f.is_zero(vals[k])) f.mul(acc, vals[k], acc);
    }
    uint64_t inv[4];
    f.inv(acc, inv);
    for (size_t k = n; k-- > 0;) {
        if (f.is_zero(vals[k])) continue;
        uint64_t t[4];
        f.mul(inv, &pre[4 * k], t);
        f.mul(inv, vals[k], inv);
        std::memcpy(vals[k], t, 32);
    }
}

// ---------------------------------------------------------------- VRF transcript (primitives.py:26-55)
struct VrfSuite {
    Bytes suite_id;
    bool xof;                 // SHAKE128 suite; otherwise SHA-512 counter mode
    uint8_t generator[64], blinding_base[64];
    const TeCurveHost* cv = te_curve(0);
};
// squeeze `size` bytes of the stream defined by everything absorbed
inline void vrf_squeeze(bool xof, const uint8_t* absorbed, size_t len, uint8_t* out, size_t size) {
    if (xof) {
        Shake128 s;
        s.update(absorbed, len);
        s.digest(out, size);
        return;
    }
    uint8_t seed[72], blk[64];
    Sha512::hash(absorbed, len, seed);
    for (size_t off = 0, ctr = 0; off < size; off += 64, ctr++) {
        for (int i = 0; i < 8; i++) seed[64 + i] = (uint8_t)((uint64_t)ctr >> (8 * i));
        Sha512::hash(seed, 72, blk);
        std::memcpy(out + off, blk, std::min<size_t>(64, size - off));
    }
}
// primitives.py:61-79: nonce from a secret scalar and a copy of the transcript; false if the result is zero
inline bool vrf_nonce(const VrfSuite& su, const Bytes& transcript, const uint64_t secret[4], uint64_t out[4]) {
    Bytes t = transcript;
    put8(t, 0x10);                                   // NONCE_EXPAND
    uint8_t sk[32], exp[64], raw[48];
    store_le32(secret, sk);
    put(t, sk, 32);
    vrf_squeeze(su.xof, t.data(), t.size(), exp, 64);
    t = transcript;
    put8(t, 0x11);                                   // NONCE
    put(t, exp, 64);
    vrf_squeeze(su.xof, t.data(), t.size(), raw, 48);   // ceil((scalar_bits + 128) / 8) = 48 for 253 and for 252 bits
    su.cv->n.reduce_bytes(raw, 48, false, out);
    return !su.cv->n.is_zero(out);
}
// primitives.py:82-88: 128-bit challenge over the given compressed points
inline void vrf_challenge(const VrfSuite& su, const Bytes& transcript, const uint8_t* enc_points, size_t count, uint64_t out[4]) {
    Bytes t = transcript;
    put8(t, 0x40);                                   // CHALLENGE
    put(t, enc_points, 32 * count);
    uint8_t raw[16];
    vrf_squeeze(su.xof, t.data(), t.size(), raw, 16);
    su.cv->n.reduce_bytes(raw, 16, false, out);
}

// try-and-increment hash-to-curve, host half (dot_ring/curve/point.py:252-296): candidate `counter` of a message is the
// first 32 squeezed bytes of suite_id || 0x60 || LE64(len) || data || counter — read by the device as a compressed point
inline void tai_candidate(const VrfSuite& su, const uint8_t* data, size_t len, unsigned counter, uint8_t out[32]) {
    Bytes t = su.suite_id;
    put8(t, 0x60);                                   // HASH_TO_CURVE
    put_le64(t, len);
    put(t, data, len);
    put8(t, (uint8_t)counter);
    vrf_squeeze(su.xof, t.data(), t.size(), out, 32);
}

// curve.py:110-185 hash_to_field(msg, 2): two field elements, 32-byte little-endian each
inline void hash_to_field2(const VrfSuite& su, const uint8_t* msg, size_t len, uint8_t out[64]) {
    Bytes dst = su.suite_id;
    put8(dst, 0x60);                                 // HASH_TO_CURVE
    put8(dst, (uint8_t)dst.size());                  // DST_prime = DST || len(DST)
    const size_t L = 96;
    uint8_t raw[128];
    if (su.xof) {
        Shake128 s;
        s.update(msg, len);
        const uint8_t lb[2] = {0, (uint8_t)L};
        s.update(lb, 2);
        s.update(dst.data(), dst.size());
        s.digest(raw, L);
    } else {
        uint8_t b0[64], zpad[48] = {0};
        Sha512 h;
        h.update(zpad, 48);                          // this suite's Z_pad is 48 bytes, not SHA-512's block size
        h.update(msg, len);
        const uint8_t lb[3] = {0, (uint8_t)L, 0};
        h.update(lb, 3);
        h.update(dst.data(), dst.size());
        h.final(b0);
        uint8_t prev[64];
        for (int i = 1; i <= 2; i++) {
            Sha512 g;
            uint8_t x[64];
            for (int j = 0; j < 64; j++) x[j] = i == 1 ? b0[j] : (uint8_t)(b0[j] ^ prev[j]);
            g.update(x, 64);
            uint8_t ib = (uint8_t)i;
            g.update(&ib, 1);
            g.update(dst.data(), dst.size());
            g.final(prev);
            std::memcpy(raw + 64 * (i - 1), prev, 64);
        }
    }
    for (int k = 0; k < 2; k++) {
        uint64_t v[4];
        mod_p().reduce_bytes(raw + 48 * k, 48, true, v);
        store_le32(v, out + 32 * k);
    }
}

// ---------------------------------------------------------------- ring-proof Fiat-Shamir transcript (transcript.py:21-136)
struct FsTranscript {
    Shake128 sh;
    static void be32(uint32_t v, uint8_t o[4]) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; }
    void framed(const char* label) {
        size_t n = std::strlen(label);
        uint8_t l[4];
        be32((uint32_t)n, l);
        sh.update(label, n);
        sh.update(l, 4);
    }
    void absorb_labeled(const char* label, const uint8_t* data, size_t len) {
        framed(label);
        uint8_t l[4];
        be32((uint32_t)len, l);
        sh.update(data, len);
        sh.update(l, 4);
    }
    // n challenges: 48 squeezed bytes, big-endian, mod p; out = n little-endian 32-byte values
    void challenges(const char* label, int n, uint8_t* out) {
        static const uint8_t footer[4] = {0, 0, 0, 9};
        for (int i = 0; i < n; i++) {
            if (i > 0) sh.update(footer, 4);
            framed(label);
            sh.update("challenge", 9);
            uint8_t raw[48];
            sh.digest(raw, 48);
            uint64_t v[4];
            mod_p().reduce_bytes(raw, 48, true, v);
            store_le32(v, out + 32 * i);
        }
        sh.update(footer, 4);
    }
};

// The transcripts of FOUR proofs side by side (Shake4): the ring proof's absorbs have the same lengths for every proof, so the
// four sponges stay in lockstep.  Same call sequence as FsTranscript, data and outputs as arrays of four pointers.
struct FsTranscript4 {
    Shake128x4 sh;
    void framed(const char* label) {
        size_t n = std::strlen(label);
        uint8_t l[4];
        FsTranscript::be32((uint32_t)n, l);
        sh.update_same(label, n);
        sh.update_same(l, 4);
    }
    void absorb_labeled(const char* label, const uint8_t* const data[4], size_t len) {
        framed(label);
        uint8_t l[4];
        FsTranscript::be32((uint32_t)len, l);
        sh.update(data, len);
        sh.update_same(l, 4);
    }
    // n challenges per proof; out[k] = n little-endian 32-byte values of proof k
    void challenges(const char* label, int n, uint8_t* const out[4]) {
        static const uint8_t footer[4] = {0, 0, 0, 9};
        for (int i = 0; i < n; i++) {
            if (i > 0) sh.update_same(footer, 4);
            framed(label);
            sh.update_same("challenge", 9);
            uint8_t raw[4][48];
            uint8_t* const rp[4] = {raw[0], raw[1], raw[2], raw[3]};
            sh.digest(rp, 48);
            for (int k = 0; k < 4; k++) {
                uint64_t v[4];
                mod_p().reduce_bytes(raw[k], 48, true, v);
                store_le32(v, out[k] + 32 * i);
            }
        }
        sh.update_same(footer, 4);
    }
};

// ---------------------------------------------------------------- ring-proof verifier scalar pass (verify.py:51-210)
struct RingVerifierDomain {
    unsigned log2n;
    uint64_t omega[4], w_nm1[4], w_nm2[4], w_nm3[4], w_nm4[4], inv_n[4];     // standard form mod p
    uint64_t seed_x[4], seed_y[4];
    void init(unsigned log2n_, const uint8_t omega_le[32], const uint8_t seed_xy[64]) {
        const Mod256& f = mod_p();
        log2n = log2n_;
        load_le32(omega_le, omega);
        uint64_t wi[4], nn[4];
        f.inv(omega, wi);
        std::memcpy(w_nm1, wi, 32);                 // w^(n-1) = w^-1
        f.mul(w_nm1, wi, w_nm2);
        f.mul(w_nm2, wi, w_nm3);
        f.mul(w_nm3, wi, w_nm4);
        f.set_u64((uint64_t)1 << log2n, nn);
        f.inv(nn, inv_n);
        load_le32(seed_xy, seed_x);
        load_le32(seed_xy + 32, seed_y);
    }
};
struct RingClaimScalars {      // what one proof contributes to the folded pairing equation
    uint64_t nus[8][4];        // commitment scalars of the aggregated opening at zeta (px, py, s, b, accip, accx, accy, q)
    uint64_t k_ip[4], k_x[4], k_y[4];
    uint64_t zeta[4], zeta_omega[4], agg_zeta[4], l_zw[4];
};
// In two halves around the one inversion (batch verification inverts the denominators of many proofs together, batch_inv).
struct RingTermsPending { uint64_t z1[4], d4[4], zn1[4], a[4], b[4], prod[4]; };
inline bool ring_verifier_terms_prep(const RingVerifierDomain& dm, const uint8_t* zeta_le, RingTermsPending& pd) {
    const Mod256& f = mod_p();
    uint64_t zeta[4], one[4], t[4];
    load_le32(zeta_le, zeta);
    f.set_u64(1, one);
    f.sub(zeta, one, pd.z1);
    f.sub(zeta, dm.w_nm4, pd.d4);
    std::memcpy(t, zeta, 32);
    for (unsigned i = 0; i < dm.log2n; i++) f.mul(t, t, t);
    f.sub(t, one, pd.zn1);
    if (f.is_zero(pd.zn1)) return false;
    // one inversion for the three denominators (zeros map to zero)
    std::memcpy(pd.a, f.is_zero(pd.z1) ? one : pd.z1, 32);
    std::memcpy(pd.b, f.is_zero(pd.d4) ? one : pd.d4, 32);
    f.mul(pd.a, pd.b, pd.prod); f.mul(pd.prod, pd.zn1, pd.prod);
    return true;
}
inline void ring_verifier_terms_finish(const TeCurveHost& cv, const RingVerifierDomain& dm, const uint8_t* alphas, const uint8_t* nus, const uint8_t* zeta_le,
                                       const uint8_t* evals, const uint8_t* l_zw_le, const uint8_t result_seed[64], const RingTermsPending& pd,
                                       const uint64_t inv[4] /* 1 / pd.prod */, RingClaimScalars& out) {
    const Mod256& f = mod_p();
    uint64_t al[7][4], ev[7][4], zeta[4], lzw[4], rsx[4], rsy[4], one[4];
    const uint64_t* five = cv.neg_a;        // -a of the curve
    for (int i = 0; i < 7; i++) { load_le32(alphas + 32 * i, al[i]); load_le32(evals + 32 * i, ev[i]); }
    for (int i = 0; i < 8; i++) load_le32(nus + 32 * i, out.nus[i]);
    load_le32(zeta_le, zeta); load_le32(l_zw_le, lzw);
    load_le32(result_seed, rsx); load_le32(result_seed + 32, rsy);
    f.set_u64(1, one);
    const uint64_t *pxz = ev[0], *pyz = ev[1], *sz = ev[2], *bz = ev[3], *ipz = ev[4], *axz = ev[5], *ayz = ev[6];
    const uint64_t *z1 = pd.z1, *d4 = pd.d4, *zn1 = pd.zn1, *a = pd.a, *b = pd.b;
    uint64_t t[4], u[4];
    uint64_t iz1[4] = {0, 0, 0, 0}, id4[4] = {0, 0, 0, 0}, izn1[4];
    {
        f.mul(a, b, t); f.mul(inv, t, izn1);
        f.mul(inv, zn1, t);                  // 1/(a b)
        if (!f.is_zero(z1)) f.mul(t, b, iz1);
        if (!f.is_zero(d4)) f.mul(t, a, id4);
    }
    uint64_t l0[4], ln[4];
    if (f.is_zero(z1)) std::memcpy(l0, one, 32); else { f.mul(dm.inv_n, zn1, l0); f.mul(l0, iz1, l0); }
    if (f.is_zero(d4)) std::memcpy(ln, one, 32); else { f.mul(dm.w_nm4, dm.inv_n, ln); f.mul(ln, zn1, ln); f.mul(ln, id4, ln); }
    uint64_t one_b[4], c[7][4], axay[4], pxpy[4];
    f.sub(one, bz, one_b);
    f.mul(axz, ayz, axay);
    f.mul(pxz, pyz, pxpy);
    // c1 = -(ip + b s) d4
    f.mul(bz, sz, t); f.add(ipz, t, t); f.neg(t, t); f.mul(t, d4, c[0]);
    // c2 = (b * -(ax ay + px py) + (1-b) * -ax) d4
    f.add(axay, pxpy, t); f.neg(t, t); f.mul(bz, t, t); f.neg(axz, u); f.mul(one_b, u, u); f.add(t, u, t); f.mul(t, d4, c[1]);
    // c3 = (b * -(ax ay - px py) + (1-b) * -ay) d4
    f.sub(axay, pxpy, t); f.neg(t, t); f.mul(bz, t, t); f.neg(ayz, u); f.mul(one_b, u, u); f.add(t, u, t); f.mul(t, d4, c[2]);
    // c4 = b (1-b)
    f.mul(bz, one_b, c[3]);
    // c5 = (ax - seed.x) L0 + (ax - result_seed.x) Ln ; c6 likewise in y
    f.sub(axz, dm.seed_x, t); f.mul(t, l0, t); f.sub(axz, rsx, u); f.mul(u, ln, u); f.add(t, u, c[4]);
    f.sub(ayz, dm.seed_y, t); f.mul(t, l0, t); f.sub(ayz, rsy, u); f.mul(u, ln, u); f.add(t, u, c[5]);
    // c7 = ip L0 + (ip - 1) Ln
    f.mul(ipz, l0, t); f.sub(ipz, one, u); f.mul(u, ln, u); f.add(t, u, c[6]);
    uint64_t lin[4] = {0, 0, 0, 0};
    for (int i = 0; i < 7; i++) { f.mul(al[i], c[i], t); f.add(lin, t, lin); }
    uint64_t tail[4], qz[4];
    f.sub(zeta, dm.w_nm1, tail); f.sub(zeta, dm.w_nm2, t); f.mul(tail, t, tail); f.sub(zeta, dm.w_nm3, t); f.mul(tail, t, tail);
    f.add(lin, lzw, qz); f.mul(qz, tail, qz); f.mul(qz, izn1, qz);
    uint64_t agg[4] = {0, 0, 0, 0};
    for (int i = 0; i < 7; i++) { f.mul(out.nus[i], ev[i], t); f.add(agg, t, agg); }
    f.mul(out.nus[7], qz, t); f.add(agg, t, agg);
    // fx = b (ay py + a ax px) + (1-b),  fy = b (ax py - px ay) + (1-b)
    uint64_t fx[4], fy[4];
    f.mul(ayz, pyz, t); f.mul(axz, pxz, u); f.mul(u, five, u); f.sub(t, u, t); f.mul(bz, t, t); f.add(t, one_b, fx);
    f.mul(axz, pyz, t); f.mul(pxz, ayz, u); f.sub(t, u, t); f.mul(bz, t, t); f.add(t, one_b, fy);
    f.mul(al[0], d4, out.k_ip);
    f.mul(al[1], fx, t); f.mul(t, d4, out.k_x);
    f.mul(al[2], fy, t); f.mul(t, d4, out.k_y);
    std::memcpy(out.zeta, zeta, 32);
    f.mul(zeta, dm.omega, out.zeta_omega);
    std::memcpy(out.agg_zeta, agg, 32);
    std::memcpy(out.l_zw, lzw, 32);
}
// alphas[7], nus[8], zeta, evals[7] (px py s b accip accx accy), l_zw: 32-byte LE canonical values; result_seed = seed + relation.
// false when zeta lies in the domain (verify.py raises there).
inline bool ring_verifier_terms(const TeCurveHost& cv, const RingVerifierDomain& dm, const uint8_t* alphas, const uint8_t* nus, const uint8_t* zeta_le,
                                const uint8_t* evals, const uint8_t* l_zw_le, const uint8_t result_seed[64], RingClaimScalars& out) {
    RingTermsPending pd;
    if (!ring_verifier_terms_prep(dm, zeta_le, pd)) return false;
    uint64_t inv[4];
    mod_p().inv(pd.prod, inv);
    ring_verifier_terms_finish(cv, dm, alphas, nus, zeta_le, evals, l_zw_le, result_seed, pd, inv, out);
    return true;
}

// 96-byte BE x||y record (or infinity) -> serialize() form: infinity = 0x40 || zeros  (kzg.py:133)
inline void g1_serialized(const uint8_t rec[96], int is_inf, uint8_t out[96]) {
    if (is_inf) { std::memset(out, 0, 96); out[0] = 0x40; } else std::memcpy(out, rec, 96);
}

}  // namespace drh
