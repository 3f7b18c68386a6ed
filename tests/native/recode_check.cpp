// Host check of dot_ring_amd/csrc/msm_recode.hip.h — the very functions the sort kernels of the G1 Pippenger run.
// for_each_wnaf_digit(k, w): digits odd, |d| < 2^(w-1), at most one per slot, non-overlapping (a digit of a bits ends before the next
// one starts), rows < 256, sum d 2^row = k, ceil(256 / w) slots, and — over uniformly random scalars — no bucket of a set collects more
// than five times the average list (the top digits share the remaining bits evenly); for_each_digit: sum d 2^start = k, |d| <= 2^(c-1).
// Scalars: edges of the field, runs of ones, alternating bits, single bits, 2^w - 1 at every offset, and random values.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "msm_recode.hip.h"

typedef unsigned __int128 u128;

struct Big {                    // 320-bit two's complement accumulator, enough for sums of +-digit * 2^pos with pos < 260
    uint32_t w[10];
    Big() { std::memset(w, 0, sizeof w); }
    void add_shifted(int64_t d, unsigned pos) {
        // add d * 2^pos
        uint32_t t[10];
        std::memset(t, 0, sizeof t);
        const bool neg = d < 0;
        uint64_t mag = neg ? (uint64_t)(-d) : (uint64_t)d;
        unsigned word = pos / 32, sh = pos % 32;
        u128 v = (u128)mag << sh;
        for (int i = 0; i < 4 && word + i < 10; i++) t[word + i] = (uint32_t)(v >> (32 * i));
        uint64_t carry = 0;
        if (!neg) {
            for (int i = 0; i < 10; i++) { uint64_t s = (uint64_t)w[i] + t[i] + carry; w[i] = (uint32_t)s; carry = s >> 32; }
        } else {
            uint64_t borrow = 0;
            for (int i = 0; i < 10; i++) { uint64_t s = (uint64_t)w[i] - t[i] - borrow; w[i] = (uint32_t)s; borrow = (s >> 32) & 1; }
        }
    }
    bool equals(const uint32_t (&k)[9]) const {
        for (int i = 0; i < 8; i++) if (w[i] != k[i]) return false;
        return w[8] == 0 && w[9] == 0;
    }
};

static const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};

static bool below_r(const uint32_t (&k)[9]) {
    for (int i = 7; i >= 0; i--) if (k[i] != R[i]) return k[i] < R[i];
    return false;
}

static int failures = 0;
static void fail(const char* what, int w, const uint32_t (&k)[9]) {
    if (failures++ < 10) {
        std::fprintf(stderr, "FAIL %s (w = %d) k =", what, w);
        for (int i = 7; i >= 0; i--) std::fprintf(stderr, " %08x", k[i]);
        std::fprintf(stderr, "\n");
    }
}

static dr::WindowTable naf_table(int w) {         // capi_msm.hip: msm_device, tiling mode 2
    dr::WindowTable wt{};
    wt.W = (256 + w - 1) / w;
    wt.cmax = w;
    const int shift = wt.W * w - 256;
    for (int j = 0; j < wt.W; j++) { wt.start[j] = (uint8_t)(w * j); wt.row[j] = (uint8_t)(j ? w * j - shift : 0); wt.width[j] = (uint8_t)w; }
    wt.odd = 2;
    return wt;
}
static dr::WindowTable window_table(int c) {     // capi_msm.hip: make_window_table
    dr::WindowTable wt{};
    wt.W = (256 + c - 1) / c;
    int base = 256 / wt.W, rem = 256 % wt.W, bit = 0;
    wt.cmax = base + (rem ? 1 : 0);
    for (int w = 0; w < wt.W; w++) {
        int width = base + (w >= wt.W - rem ? 1 : 0);
        wt.start[w] = (uint8_t)bit; wt.width[w] = (uint8_t)width; wt.row[w] = (uint8_t)w;
        bit += width;
    }
    return wt;
}

static unsigned long long total_digits[16], total_scalars[16];
static bool stats = false;                        // set for the uniformly random scalars: bucket loads are meaningful only there
static std::vector<unsigned long long> bin_load[16];

static void check(const uint32_t (&k)[9]) {
    if (!below_r(k)) return;
    for (int w = 9; w <= 13; w++) {
        const dr::WindowTable wt = naf_table(w);
        Big sum;
        int last_end = -1, last_slot = -1, count = 0, slots_seen = 0;
        bool ok = true;
        dr::for_each_wnaf_digit<true>(k, wt, [&](int j, uint32_t o, int32_t d) {
            slots_seen++;
            if (j != last_slot + 1) ok = false;                        // WITH_ZEROS visits every slot once, in order
            last_slot = j;
            if (d == 0) return;
            const int pos = (int)wt.row[j] + (int)o;                   // table row = bit position of the digit
            if (!(d & 1) || d >= (1 << (w - 1)) || d <= -(1 << (w - 1))) ok = false;
            if (o > 14u || pos > 255 || pos <= last_end) ok = false;   // offset fits its 4-bit field; digits do not overlap
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            if ((mag - 1u) >> 1 >= (1u << (w - 2))) ok = false;
            last_end = pos + (32 - __builtin_clz(mag)) - 1;            // top bit of |d| 2^pos
            count++;
            sum.add_shifted(d, (unsigned)pos);
            if (stats) bin_load[w][(mag - 1u) >> 1]++;
        });
        if (!ok || slots_seen != wt.W || !sum.equals(k)) fail("non-adjacent form", w, k);
        // the same digits without the empty slots
        int count2 = 0;
        dr::for_each_wnaf_digit(k, wt, [&](int, uint32_t, int32_t d) { count2 += d != 0; });
        if (count2 != count) fail("non-adjacent form, WITH_ZEROS = false", w, k);
        if (stats) { total_digits[w] += (unsigned long long)count; total_scalars[w]++; }
    }
    for (int c = 7; c <= 16; c++) {
        const dr::WindowTable wt = window_table(c);
        Big sum;
        bool ok = true;
        dr::for_each_digit(k, wt, 0, wt.W, [&](int w, int32_t d) {
            if (d > (1 << (wt.width[w] - 1)) || d < -(1 << (wt.width[w] - 1)) || d == 0) ok = false;
            sum.add_shifted(d, wt.start[w]);
        });
        if (!ok || !sum.equals(k)) fail("window digits", c, k);
    }
}

static void set_bits(uint32_t (&k)[9], int lo, int hi) {       // bits [lo, hi) set
    for (int b = lo; b < hi; b++) k[b / 32] |= 1u << (b % 32);
}

int main() {
    std::mt19937_64 rng(20261005);
    uint32_t k[9];
    auto clear = [&] { std::memset(k, 0, sizeof k); };
    // zero, one, r - 1, (r - 1) / 2 and neighbours
    clear(); check(k);
    clear(); k[0] = 1; check(k);
    for (int delta = 1; delta <= 40; delta++) {
        clear();
        uint64_t borrow = (uint64_t)delta;
        for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)R[i] - borrow; k[i] = (uint32_t)s; borrow = (s >> 32) & 1; }
        check(k);
        uint32_t h[9];
        std::memcpy(h, k, sizeof h);
        for (int i = 0; i < 8; i++) k[i] = (h[i] >> 1) | (i < 7 ? h[i + 1] << 31 : 0);
        check(k);
    }
    // single bits, runs of ones from every start to every end (the carry crosses every slot boundary), minus small values
    for (int b = 0; b < 255; b++) { clear(); k[b / 32] = 1u << (b % 32); check(k); }
    for (int lo = 0; lo < 255; lo += 1)
        for (int hi = lo + 1; hi <= 255; hi += (hi - lo < 30 ? 1 : 7)) { clear(); set_bits(k, lo, hi); check(k); }
    // alternating patterns at every shift
    for (int sh = 0; sh < 32; sh++)
        for (uint32_t pat : {0x55555555u, 0xaaaaaaaau, 0x33333333u, 0x0f0f0f0fu, 0xfffe0001u, 0x80000001u, 0x7fffffffu}) {
            clear();
            for (int i = 0; i < 8; i++) k[i] = pat;
            k[7] &= 0x3fffffffu;
            uint32_t t[9];
            std::memcpy(t, k, sizeof t);
            for (int i = 0; i < 8; i++) k[i] = (t[i] >> sh) | (sh && i < 7 ? t[i + 1] << (32 - sh) : 0);
            check(k);
        }
    // 2^w - 1, 2^(w-1) and 2^(w-1) +- 1 at every position
    for (int w = 9; w <= 14; w++)
        for (int pos = 0; pos + w < 254; pos++)
            for (int v = 0; v < 4; v++) {
                clear();
                uint64_t val = v == 0 ? (1ull << w) - 1 : v == 1 ? (1ull << (w - 1)) : v == 2 ? (1ull << (w - 1)) + 1 : (1ull << (w - 1)) - 1;
                u128 sv = (u128)val << (pos % 32);
                for (int i = 0; i < 3 && pos / 32 + i < 8; i++) k[pos / 32 + i] = (uint32_t)(sv >> (32 * i));
                check(k);
            }
    // random values: sparse, dense, and random with the top cleared
    for (int it = 0; it < 60000; it++) {
        clear();
        const int mode = it % 3;
        for (int i = 0; i < 8; i++) {
            uint32_t v = (uint32_t)rng();
            if (mode == 0) v &= (uint32_t)rng();
            if (mode == 1) v |= (uint32_t)rng();
            if (mode == 2) v &= (uint32_t)rng() & (uint32_t)rng();
            k[i] = v;
        }
        k[7] &= 0x7fffffffu;
        check(k);
    }
    // uniformly random scalars below r: digit count and bucket loads
    for (int w = 9; w <= 13; w++) bin_load[w].assign((size_t)1 << (w - 2), 0);
    stats = true;
    for (int it = 0; it < 200000; it++) {
        clear();
        for (int i = 0; i < 8; i++) k[i] = (uint32_t)rng();
        k[7] &= 0x7fffffffu;
        check(k);                                                      // (values >= r are skipped)
    }
    stats = false;
    for (int w = 9; w <= 13; w++) {
        const double avg = (double)total_digits[w] / (double)bin_load[w].size();
        unsigned long long most = 0;
        for (unsigned long long v : bin_load[w]) most = v > most ? v : most;
        std::printf("w = %d: fullest bucket holds %.2f x the average list\n", w, (double)most / avg);
        if ((double)most > 5.0 * avg) { std::fprintf(stderr, "FAIL: skewed buckets at w = %d\n", w); failures++; }
        if ((double)total_digits[w] / (double)total_scalars[w] > 256.0 / (w + 1) + 0.8) { std::fprintf(stderr, "FAIL: too many digits at w = %d\n", w); failures++; }
    }
    if (failures) {
        std::fprintf(stderr, "%d failures\n", failures);
        return 1;
    }
    for (int w = 9; w <= 13; w++)
        std::printf("w = %d: %.3f digits per scalar on average over %llu scalars (256 / (w + 1) = %.3f)\n", w,
                    (double)total_digits[w] / (double)total_scalars[w], total_scalars[w], 256.0 / (w + 1));
    std::printf("recoding ok\n");
    return 0;
}
