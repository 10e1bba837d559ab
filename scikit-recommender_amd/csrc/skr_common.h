// skr_common.h -- shared host/device helpers of libskrec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/skrec_hip.h"

#define SKR_WAVE 64

namespace skr {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define SKR_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return skr::fail(SKR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define SKR_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return skr::fail(SKR_EINVAL, __VA_ARGS__); \
    } while (0)

#define SKR_LAUNCH_CHECK() SKR_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (SKR_WAVE - 1); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = SKR_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SKR_WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = SKR_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SKR_WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = SKR_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SKR_WAVE);
    return v;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < SKR_WAVE; o <<= 1) {
        int t = __shfl_up(v, o, SKR_WAVE);
        if (l >= o) v += t;
    }
    return v;
}

// sorted-ascending membership test over [beg, end) of a global/LDS int array
template <typename P>
__device__ __forceinline__ bool contains_sorted(P a, int64_t beg, int64_t end, int v) {
    int64_t lo = beg, hi = end;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo < end && a[lo] == v;
}

// Total order used everywhere a top-K is taken: score descending, then item id ascending.
// Packed so that a plain unsigned 64-bit "greater" implements it.  (-0.0 is folded into +0.0, as the
// reference's float comparison sees them; NaN scores are not supported -- the reference's comparator is
// undefined on them as well.)
__device__ __forceinline__ uint64_t rank_key(float score, int id) {
    uint32_t u = __float_as_uint(score == 0.0f ? 0.0f : score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (static_cast<uint64_t>(u) << 32) | static_cast<uint32_t>(~static_cast<uint32_t>(id));
}
__device__ __forceinline__ float key_score(uint64_t k) {
    uint32_t u = static_cast<uint32_t>(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_id(uint64_t k) { return static_cast<int>(~static_cast<uint32_t>(k)); }
// the smallest possible key: below every real (score, id) pair, used as padding
#define SKR_KEY_MIN 0ull

}  // namespace skr
