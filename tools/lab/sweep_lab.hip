// sweep_lab.hip -- experiment (round 3): the item side of the propagation for rows below the long-row threshold as ONE sweep
// over the column blocks with the partial rows in LDS instead of HBM.  A workgroup owns a group of <= 128 rows (accumulators:
// 32 KB of LDS), each of its four wavefronts owns a quarter of them (so that a row is only ever added to by one wavefront, in
// program order: no timing dependence); the group's entries are stored wave by wave in (column block, row) order, and every
// workgroup walks the blocks in the same order at about the same pace -- the X slice being gathered from sits in the XCD's
// L2 because the other workgroups of the XCD are reading it too.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int R = 128;       // rows per group
constexpr int NF = 4;        // gather instructions in flight

template <int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(const int64_t* __restrict__ seg_ptr, const int32_t* __restrict__ scol,
                                                    const float* __restrict__ sval, const uint8_t* __restrict__ slrow,
                                                    const int32_t* __restrict__ group_rows, const float* __restrict__ X,
                                                    float* __restrict__ Y) {
    __shared__ float acc[R * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    for (int i = threadIdx.x; i < R * 64; i += 256) acc[i] = 0.0f;
    __syncthreads();
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4 dummy = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t b = seg_ptr[blockIdx.x * 4 + wv], e_end = seg_ptr[blockIdx.x * 4 + wv + 1];
    for (int64_t e = b; e < e_end; e += 64) {
        const int m = static_cast<int>(e_end - e < 64 ? e_end - e : 64);
        int cl = scol[b];
        float vl = 0.0f;
        int lr = wv;
        if (lane < m) { cl = scol[e + lane]; vl = sval[e + lane]; lr = slrow[e + lane]; }
        for (int k = 0; k < m; k += 4 * NF) {
            float4 x[NF];
            float v[NF];
            int l[NF];
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const int src = (k + 4 * q + grp) & 63;
                x[q] = X4[static_cast<int64_t>(__shfl(cl, src)) * 16 + sub];
                v[q] = (k + 4 * q + grp < 64) ? __shfl(vl, src) : 0.0f;
                l[q] = __shfl(lr, src);
            }
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                float p0 = v[q] * x[q].x, p1 = v[q] * x[q].y, p2 = v[q] * x[q].z, p3 = v[q] * x[q].w;
                if (MODE == 1) { dummy.x += p0; dummy.y += p1; dummy.z += p2; dummy.w += p3 + l[q]; continue; }
                // component (j + grp) & 3 goes out in instruction j: the four 16-lane groups of an instruction hit disjoint banks
                if (grp & 1) { const float t = p0; p0 = p1; p1 = p2; p2 = p3; p3 = t; }
                if (grp & 2) { float t = p0; p0 = p2; p2 = t; t = p1; p1 = p3; p3 = t; }
                float* row = acc + l[q] * 64 + sub * 4;
                atomicAdd(row + ((0 + grp) & 3), p0);
                atomicAdd(row + ((1 + grp) & 3), p1);
                atomicAdd(row + ((2 + grp) & 3), p2);
                atomicAdd(row + ((3 + grp) & 3), p3);
            }
        }
    }
    if (MODE == 1) acc[threadIdx.x] = dummy.x + dummy.y + dummy.z + dummy.w;
    __syncthreads();
    for (int lr = wv; lr < R; lr += 4) {
        const int r = group_rows[blockIdx.x * R + lr];
        if (r >= 0) Y[static_cast<int64_t>(r) * 64 + lane] = acc[lr * 64 + lane];
    }
}
}  // namespace

extern "C" int lab_sweep(int mode, int n_groups, const int64_t* seg_ptr, const int32_t* scol, const float* sval, const uint8_t* slrow,
                         const int32_t* group_rows, const float* X, float* Y, void* stream) {
    if (mode == 1)
        hipLaunchKernelGGL(sweep_kernel<1>, dim3(n_groups), dim3(256), 0, static_cast<hipStream_t>(stream), seg_ptr, scol, sval, slrow,
                           group_rows, X, Y);
    else
        hipLaunchKernelGGL(sweep_kernel<0>, dim3(n_groups), dim3(256), 0, static_cast<hipStream_t>(stream), seg_ptr, scol, sval, slrow,
                           group_rows, X, Y);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
