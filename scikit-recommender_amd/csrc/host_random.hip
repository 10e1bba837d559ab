// Host-side replay of numpy's legacy generator for the ONE host draw on the training path: the epoch permutation.
// The reference shuffles every epoch with np.random.permutation(n) from numpy's global MT19937
// (skrec/io/batch_iterator.py:61-63); at 50 M interactions that is ~1 s of single-threaded numpy per epoch, under the
// GIL.  This file restates numpy's algorithm (numpy/random/_mt19937 + legacy shuffle: arange(n), then for i = n-1 .. 1:
// j = random_interval(i) by masked rejection on 32-bit outputs, swap) on the generator state handed in, so the result
// and the state afterwards are numpy's, bit for bit -- but it runs outside the GIL (ctypes releases it), works on
// int32, and looks a few hundred swaps ahead to prefetch their targets (the draws do not depend on the data).
// No GPU involved; it lives in this library because the library is the path's native layer.
#include "skr_common.h"

#include <cstdint>
#include <cstring>

namespace {

constexpr int MT_LEN = 624, MT_M = 397;
constexpr uint32_t MT_MATRIX_A = 0x9908b0dfu, MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu;

struct Mt19937 {
    uint32_t key[MT_LEN];
    int pos;
    void regenerate() {
        int kk = 0;
        uint32_t y;
        for (; kk < MT_LEN - MT_M; ++kk) {
            y = (key[kk] & MT_UPPER) | (key[kk + 1] & MT_LOWER);
            key[kk] = key[kk + MT_M] ^ (y >> 1) ^ (-(y & 1u) & MT_MATRIX_A);
        }
        for (; kk < MT_LEN - 1; ++kk) {
            y = (key[kk] & MT_UPPER) | (key[kk + 1] & MT_LOWER);
            key[kk] = key[kk + (MT_M - MT_LEN)] ^ (y >> 1) ^ (-(y & 1u) & MT_MATRIX_A);
        }
        y = (key[MT_LEN - 1] & MT_UPPER) | (key[0] & MT_LOWER);
        key[MT_LEN - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(y & 1u) & MT_MATRIX_A);
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos >= MT_LEN) regenerate();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

inline uint32_t smear(uint32_t m) {
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    return m;
}

}  // namespace

extern "C" int skr_host_permutation(uint32_t* key624, int* pos, int64_t n, int32_t* out) {
    SKR_REQUIRE(key624 && pos && (out || n == 0), "skr_host_permutation: NULL argument");
    SKR_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "skr_host_permutation: n must be in [0, 2^31)");
    SKR_REQUIRE(*pos >= 0 && *pos <= MT_LEN, "skr_host_permutation: generator position out of range");
    Mt19937 g;
    std::memcpy(g.key, key624, sizeof(g.key));
    g.pos = *pos;
    for (int64_t i = 0; i < n; ++i) out[i] = static_cast<int32_t>(i);
    constexpr int AHEAD = 256;
    uint32_t js[AHEAD];
    int64_t i = n - 1;
    while (i >= 1) {
        const int m = static_cast<int>(i < AHEAD ? i : AHEAD);   // swaps i, i-1, ..., i-m+1
        for (int b = 0; b < m; ++b) {
            const uint32_t hi = static_cast<uint32_t>(i - b), mask = smear(hi);
            uint32_t v;
            do v = g.next32() & mask; while (v > hi);              // numpy legacy random_interval(max <= 0xffffffff)
            js[b] = v;
            __builtin_prefetch(&out[v], 1, 0);
        }
        for (int b = 0; b < m; ++b) {
            const int64_t ii = i - b;
            const int32_t t = out[js[b]];
            out[js[b]] = out[ii];
            out[ii] = t;
        }
        i -= m;
    }
    std::memcpy(key624, g.key, sizeof(g.key));
    *pos = g.pos;
    return SKR_OK;
}
