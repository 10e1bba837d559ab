// shuffle.hip -- SURVEY 8(f-1): the epoch shuffle + batch assembly between the sampler and the train step.
//
// Replaces the per-element Python batching of the reference's BatchIterator
//   utils/py/batch_iterator.py:48-66,132-155   (np.random.permutation(n), then [data[i] for i in idx] per column and batch)
//   io/data_iterator.py:226-234                (users / pos / neg columns zipped into batches)
// with ONE launch per epoch: row r of every output column = row src(r) of the matching input column, where
//   * src = the permutation handed in (numpy-compatible contract: the values of np.random.permutation(n), computed by
//     skr_host_permutation and uploaded once), or
//   * src = pi_seed(r), a keyed bijection of [0, n) evaluated in registers (device shuffle: no permutation array, no
//     sort, equal to the reference in law only).
// HBM-bound integer work: per output row one random 4*width-byte read per column and one coalesced write.
#include "skr_common.h"

namespace {

constexpr int SH_MAX_COLS = 4;

struct ShuffleCols {
    const uint32_t* src[SH_MAX_COLS];
    uint32_t* dst[SH_MAX_COLS];
    int width[SH_MAX_COLS];   // 32-bit words per row
    int n_cols;
};

// A bijection of [0, 2^bits): every step is one (odd multiply, key add, xor with a right shift, bit reversal -- all mod
// 2^bits).  The multiply/add carry information upwards only, the shift and the reversal bring it back down.
struct Bijection {
    uint32_t mask, key[6];
    int bits, shift;
};

__host__ __device__ __forceinline__ uint32_t bij_round(uint32_t x, uint32_t k, const Bijection& b) {
    x = (x * 0x9E3779B1u + k) & b.mask;          // odd multiplier: invertible mod 2^bits
    x ^= x >> b.shift;                           // invertible (shift >= 1)
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(x) >> (32 - b.bits);              // the low bits, which only ever saw low bits, become the high ones
#else
    uint32_t r = 0;
    for (int i = 0; i < b.bits; ++i) r |= ((x >> i) & 1u) << (b.bits - 1 - i);
    x = r;
#endif
    return x;
}

__host__ __device__ __forceinline__ uint32_t bij_apply(uint32_t x, const Bijection& b) {
#pragma unroll
    for (int r = 0; r < 6; ++r) x = bij_round(x, b.key[r], b);
    return x;
}

// cycle walking: pi(r) = first element of r's orbit under the bijection that falls below n (2^bits < 2n: <= 2 steps expected)
__host__ __device__ __forceinline__ uint32_t perm_of(uint32_t r, uint32_t n, const Bijection& b) {
    uint32_t x = bij_apply(r, b);
    while (x >= n) x = bij_apply(x, b);
    return x;
}

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline Bijection make_bijection(uint64_t seed, int64_t n) {
    Bijection b;
    int bits = 1;
    while ((int64_t{1} << bits) < n) ++bits;
    b.bits = bits;
    b.mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    b.shift = bits > 1 ? (bits + 1) / 2 : 1;
    uint64_t s = seed ^ 0x5DEECE66Dull;
    for (int r = 0; r < 6; ++r) b.key[r] = static_cast<uint32_t>(splitmix64(s) >> 16);
    return b;
}

template <bool HAVE_PERM>
__global__ __launch_bounds__(256) void shuffle_gather_kernel(const int32_t* __restrict__ perm, Bijection bij, uint32_t n_src,
                                                             int64_t n_out, ShuffleCols c) {
    const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (r >= n_out) return;
    const uint32_t s = HAVE_PERM ? static_cast<uint32_t>(perm[r]) : perm_of(static_cast<uint32_t>(r), n_src, bij);
#pragma unroll
    for (int k = 0; k < SH_MAX_COLS; ++k) {
        if (k < c.n_cols) {
            const int w = c.width[k];
            if (w == 1) {
                c.dst[k][r] = c.src[k][s];
            } else {
                for (int j = 0; j < w; ++j) c.dst[k][r * w + j] = c.src[k][static_cast<int64_t>(s) * w + j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void perm_values_kernel(Bijection bij, uint32_t n, int64_t n_out, int32_t* __restrict__ out) {
    const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (r < n_out) out[r] = static_cast<int32_t>(perm_of(static_cast<uint32_t>(r), n, bij));
}

}  // namespace

extern "C" {

int skr_shuffle_gather(const int32_t* d_perm, uint64_t seed, int64_t n_src, int64_t n_out, int n_cols,
                       const void* const* d_cols, const int* widths, void* const* d_outs, void* stream) {
    SKR_REQUIRE(n_src >= 0 && n_out >= 0 && n_out <= n_src, "skr_shuffle_gather: need 0 <= n_out <= n_src");
    SKR_REQUIRE(n_src < (int64_t{1} << 31), "skr_shuffle_gather: n_src must fit int32 (ids are int32 everywhere)");
    SKR_REQUIRE(n_cols >= 1 && n_cols <= SH_MAX_COLS, "skr_shuffle_gather: 1..%d columns per call (got %d)", SH_MAX_COLS, n_cols);
    SKR_REQUIRE(d_cols && widths && d_outs, "skr_shuffle_gather: NULL argument");
    ShuffleCols c{};
    c.n_cols = n_cols;
    for (int k = 0; k < n_cols; ++k) {
        SKR_REQUIRE(widths[k] >= 1, "skr_shuffle_gather: column %d has width %d", k, widths[k]);
        SKR_REQUIRE(n_out == 0 || (d_cols[k] && d_outs[k]), "skr_shuffle_gather: column %d is NULL", k);
        SKR_REQUIRE(d_cols[k] != d_outs[k] || n_out == 0, "skr_shuffle_gather: in-place shuffling is not supported");
        c.src[k] = static_cast<const uint32_t*>(d_cols[k]);
        c.dst[k] = static_cast<uint32_t*>(d_outs[k]);
        c.width[k] = widths[k];
    }
    if (n_out == 0) return SKR_OK;
    const Bijection b = make_bijection(seed, n_src);
    const dim3 grid(static_cast<unsigned>((n_out + 255) / 256)), block(256);
    hipStream_t st = skr::as_stream(stream);
    if (d_perm)
        hipLaunchKernelGGL(shuffle_gather_kernel<true>, grid, block, 0, st, d_perm, b, static_cast<uint32_t>(n_src), n_out, c);
    else
        hipLaunchKernelGGL(shuffle_gather_kernel<false>, grid, block, 0, st, d_perm, b, static_cast<uint32_t>(n_src), n_out, c);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_shuffle_permutation(uint64_t seed, int64_t n, int64_t n_out, int32_t* d_out, void* stream) {
    SKR_REQUIRE(n >= 0 && n_out >= 0 && n_out <= n && n < (int64_t{1} << 31), "skr_shuffle_permutation: need 0 <= n_out <= n < 2^31");
    if (n_out == 0) return SKR_OK;
    SKR_REQUIRE(d_out, "skr_shuffle_permutation: NULL output");
    const Bijection b = make_bijection(seed, n);
    hipLaunchKernelGGL(perm_values_kernel, dim3(static_cast<unsigned>((n_out + 255) / 256)), dim3(256), 0, skr::as_stream(stream), b,
                       static_cast<uint32_t>(n), n_out, d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

// the same bijection on the host (no GPU): lets callers and tests reason about a device shuffle without running it
int skr_shuffle_permutation_host(uint64_t seed, int64_t n, int64_t n_out, int32_t* out) {
    SKR_REQUIRE(n >= 0 && n_out >= 0 && n_out <= n && n < (int64_t{1} << 31), "skr_shuffle_permutation_host: need 0 <= n_out <= n < 2^31");
    SKR_REQUIRE(out || n_out == 0, "skr_shuffle_permutation_host: NULL output");
    const Bijection b = make_bijection(seed, n);
    for (int64_t r = 0; r < n_out; ++r) out[r] = static_cast<int32_t>(perm_of(static_cast<uint32_t>(r), static_cast<uint32_t>(n), b));
    return SKR_OK;
}

}  // extern "C"
