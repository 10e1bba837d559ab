// spmm_lab.hip -- experiment kernels for the CSR propagation (NOT part of the product library; built into
// tools/lab/libspmm_lab.so by tools/lab/Makefile and driven by tools/spmm_lab.py).  Variants of "one wavefront per
// row" with different gather shapes and an optional LDS cache of the hottest columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int D = 64;

// entries of a row: [rowptr[r], rowptr[r] + hotcnt[r]) name LDS slots (hot columns), the rest global columns
template <int NF, bool HOT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rows_dword_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ hotcnt,
                                                                const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                const float* __restrict__ X, const int32_t* __restrict__ hot_cols,
                                                                int n_hot, float* __restrict__ Y) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (HOT) {
        for (int i = threadIdx.x; i < n_hot * D; i += WAVES * 64) lds[i] = X[static_cast<int64_t>(hot_cols[i >> 6]) * D + (i & 63)];
        __syncthreads();
    }
    for (int64_t r = blockIdx.x * WAVES + wv; r < n_rows; r += static_cast<int64_t>(gridDim.x) * WAVES) {
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        int64_t e = rb;
        float acc = 0.0f;
        if (HOT) {
            const int64_t he = rb + hotcnt[r];
            for (; e < he; e += 64) {
                const int m = static_cast<int>(he - e < 64 ? he - e : 64);
                int cl = 0;
                float vl = 0.0f;
                if (lane < m) { cl = col[e + lane]; vl = val[e + lane]; }
                int k = 0;
                for (; k + 4 <= m; k += 4) {
                    const float x0 = lds[__shfl(cl, k) * D + lane], x1 = lds[__shfl(cl, k + 1) * D + lane];
                    const float x2 = lds[__shfl(cl, k + 2) * D + lane], x3 = lds[__shfl(cl, k + 3) * D + lane];
                    acc = fmaf(__shfl(vl, k), x0, acc); acc = fmaf(__shfl(vl, k + 1), x1, acc);
                    acc = fmaf(__shfl(vl, k + 2), x2, acc); acc = fmaf(__shfl(vl, k + 3), x3, acc);
                }
                for (; k < m; ++k) acc = fmaf(__shfl(vl, k), lds[__shfl(cl, k) * D + lane], acc);
            }
            e = he;
        }
        for (; e < re; e += 64) {
            const int m = static_cast<int>(re - e < 64 ? re - e : 64);
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) { cl = col[e + lane]; vl = val[e + lane]; }
            int k = 0;
            for (; k + NF <= m; k += NF) {
                float x[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) x[q] = X[static_cast<int64_t>(__shfl(cl, k + q)) * D + lane];
#pragma unroll
                for (int q = 0; q < NF; ++q) acc = fmaf(__shfl(vl, k + q), x[q], acc);
            }
            for (; k < m; ++k) acc = fmaf(__shfl(vl, k), X[static_cast<int64_t>(__shfl(cl, k)) * D + lane], acc);
        }
        Y[r * D + lane] = acc;
    }
}

// 16 bytes per lane: a 16-lane group fetches one 256-B row, a wave-instruction fetches 4 rows
template <int NF, bool HOT, int WAVES, int LSTRIDE = 64, bool NT = false>
__global__ __launch_bounds__(WAVES * 64) void rows_x4_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ hotcnt, const int32_t* __restrict__ col,
                                                             const float* __restrict__ val, const float* __restrict__ X,
                                                             const int32_t* __restrict__ hot_cols, int n_hot,
                                                             float* __restrict__ Y) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;   // group g handles entries k*4+g; lane covers dims sub*4 .. sub*4+3
    if (HOT) {
        for (int i = threadIdx.x; i < n_hot * D; i += WAVES * 64)
            lds[(i >> 6) * LSTRIDE + (i & 63)] = X[static_cast<int64_t>(hot_cols[i >> 6]) * D + (i & 63)];
        __syncthreads();
    }
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const float4* L4 = reinterpret_cast<const float4*>(lds);
    auto ldi = [](const int32_t* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
    auto ldf = [](const float* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
    for (int64_t r = blockIdx.x * WAVES + wv; r < n_rows; r += static_cast<int64_t>(gridDim.x) * WAVES) {
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        int64_t e = rb;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HOT) {
            const int64_t he = rb + hotcnt[r];
            for (; e < he; e += 64) {
                const int m = static_cast<int>(he - e < 64 ? he - e : 64);
                int cl = 0;
                float vl = 0.0f;     // lanes >= m keep val 0: their (valid slot 0) product adds nothing
                if (lane < m) { cl = ldi(&col[e + lane]); vl = ldf(&val[e + lane]); }
                for (int k = 0; k < m; k += 4) {
                    const int c = __shfl(cl, k + grp);
                    const float v = __shfl(vl, k + grp);
                    const float4 x = L4[c * (LSTRIDE / 4) + sub];
                    acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
                }
            }
            e = he;
        }
        for (; e < re; e += 64) {
            const int m = static_cast<int>(re - e < 64 ? re - e : 64);
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) { cl = ldi(&col[e + lane]); vl = ldf(&val[e + lane]); }
            int k = 0;
            for (; k + 4 * NF <= m; k += 4 * NF) {
                float4 x[NF];
                float v[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    x[q] = X4[static_cast<int64_t>(__shfl(cl, k + 4 * q + grp)) * 16 + sub];
                    v[q] = __shfl(vl, k + 4 * q + grp);
                }
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y);
                    acc.z = fmaf(v[q], x[q].z, acc.z); acc.w = fmaf(v[q], x[q].w, acc.w);
                }
            }
            for (; k < m; k += 4) {      // tail: lanes past m carry col 0 / val 0 -- a valid address, a zero product
                const float4 x = X4[static_cast<int64_t>(__shfl(cl, k + grp)) * 16 + sub];
                const float v = __shfl(vl, k + grp);
                acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
            }
        }
        // sum the four groups' partial rows (lanes l, l^16, l^32, l^48 hold the same dims)
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if (grp == 0) {
            if (NT) {
                float* y = Y + r * 64 + sub * 4;
                __builtin_nontemporal_store(acc.x, y); __builtin_nontemporal_store(acc.y, y + 1);
                __builtin_nontemporal_store(acc.z, y + 2); __builtin_nontemporal_store(acc.w, y + 3);
            } else {
                reinterpret_cast<float4*>(Y)[r * 16 + sub] = acc;
            }
        }
    }
}


// variant 2: rows_x4 with the NEXT row's bounds and first 64 entries requested before the current row's gathers
template <int NF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rows_x4_pf_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                const float* __restrict__ X, float* __restrict__ Y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * WAVES;
    int64_t r = blockIdx.x * WAVES + wv;
    if (r >= n_rows) return;
    int64_t rb = rowptr[r], re = rowptr[r + 1];
    int cl = 0;
    float vl = 0.0f;
    if (rb + lane < re) { cl = col[rb + lane]; vl = val[rb + lane]; }
    while (true) {
        // request the next row's bounds now; its entries as soon as the bounds are here (before this row's gathers)
        const int64_t rn = r + stride;
        int64_t rbn = 0, ren = 0;
        if (rn < n_rows) { rbn = rowptr[rn]; ren = rowptr[rn + 1]; }
        int cln = 0;
        float vln = 0.0f;
        if (rbn + lane < ren) { cln = col[rbn + lane]; vln = val[rbn + lane]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t e = rb;
        while (true) {
            const int m = static_cast<int>(re - e < 64 ? re - e : 64);
            int k = 0;
            for (; k + 4 * NF <= m; k += 4 * NF) {
                float4 x[NF];
                float v[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    x[q] = X4[static_cast<int64_t>(__shfl(cl, k + 4 * q + grp)) * 16 + sub];
                    v[q] = __shfl(vl, k + 4 * q + grp);
                }
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y);
                    acc.z = fmaf(v[q], x[q].z, acc.z); acc.w = fmaf(v[q], x[q].w, acc.w);
                }
            }
            for (; k < m; k += 4) {
                const float4 x = X4[static_cast<int64_t>(__shfl(cl, k + grp)) * 16 + sub];
                const float v = __shfl(vl, k + grp);
                acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
            }
            e += 64;
            if (e >= re) break;
            cl = 0; vl = 0.0f;
            if (e + lane < re) { cl = col[e + lane]; vl = val[e + lane]; }
        }
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if (grp == 0) reinterpret_cast<float4*>(Y)[r * 16 + sub] = acc;
        if (rn >= n_rows) break;
        r = rn; rb = rbn; re = ren; cl = cln; vl = vln;
    }
}

// variant 3: column-blocked tasks.  A task = up to 256 consecutive entries of ONE row whose columns lie in ONE column
// block; tasks are sorted by block; launch `group` lets the workgroups with blockIdx % 8 == x (one XCD under round-robin
// placement) work through block 8*group + x, so that the block's slice of X stays in that XCD's L2.
template <int NF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void blocked_tasks_kernel(const int64_t* __restrict__ tptr, int n_blocks, int group,
                                                                   const int64_t* __restrict__ task_beg,
                                                                   const int32_t* __restrict__ task_len,
                                                                   const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                   const float* __restrict__ X, float* __restrict__ part) {
    const int b = group * 8 + (blockIdx.x & 7);
    if (b >= n_blocks) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int64_t n_w = static_cast<int64_t>(gridDim.x >> 3) * WAVES;
    for (int64_t t = tptr[b] + (blockIdx.x >> 3) * WAVES + wv; t < tptr[b + 1]; t += n_w) {
        const int64_t eb = task_beg[t];
        const int len = task_len[t];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e0 = 0; e0 < len; e0 += 64) {
            const int m = len - e0 < 64 ? len - e0 : 64;
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) { cl = col[eb + e0 + lane]; vl = val[eb + e0 + lane]; }
            int k = 0;
            for (; k + 4 * NF <= m; k += 4 * NF) {
                float4 x[NF];
                float v[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    x[q] = X4[static_cast<int64_t>(__shfl(cl, k + 4 * q + grp)) * 16 + sub];
                    v[q] = __shfl(vl, k + 4 * q + grp);
                }
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y);
                    acc.z = fmaf(v[q], x[q].z, acc.z); acc.w = fmaf(v[q], x[q].w, acc.w);
                }
            }
            for (; k < m; k += 4) {
                // lanes past m hold col 0 / val 0; keep the address inside the block: use the task's first column instead
                // (both shuffles run with every lane active: a bpermute reads 0 from a lane that is masked off)
                const bool in = k + grp < m;
                const int idx = in ? k + grp : 0;
                const int c = __shfl(cl, idx);
                const float vs = __shfl(vl, idx);
                const float v = in ? vs : 0.0f;
                const float4 x = X4[static_cast<int64_t>(c) * 16 + sub];
                acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
            }
        }
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if (grp == 0) reinterpret_cast<float4*>(part)[t * 16 + sub] = acc;
    }
}


// ---- variant 5: every quad of a 64-entry block requested before the first one is used (up to 16 KB per wave in flight) ----
// one 64-entry block held in (cl, vl): accumulate its m entries
__device__ __forceinline__ void block64_deep(const float4* __restrict__ X4, int cl, float vl, int m, int grp, int sub, float4& acc) {
    float4 x[16];
    float v[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 * 16 < m) {          // wave-uniform: four quads at a time
#pragma unroll
            for (int q = g4 * 4; q < g4 * 4 + 4; ++q) {
                // lanes past m hold col 0 / val 0 (a valid row, a zero product)
                x[q] = X4[static_cast<int64_t>(__shfl(cl, 4 * q + grp)) * 16 + sub];
                v[q] = __shfl(vl, 4 * q + grp);
            }
        }
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 * 16 < m) {
#pragma unroll
            for (int q = g4 * 4; q < g4 * 4 + 4; ++q) {
                acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y);
                acc.z = fmaf(v[q], x[q].z, acc.z); acc.w = fmaf(v[q], x[q].w, acc.w);
            }
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rows_x4_deep_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                  const float* __restrict__ X, float* __restrict__ Y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * WAVES;
    int64_t r = blockIdx.x * WAVES + wv;
    if (r >= n_rows) return;
    int64_t rb = rowptr[r], re = rowptr[r + 1];
    int cl = 0;
    float vl = 0.0f;
    if (rb + lane < re) { cl = col[rb + lane]; vl = val[rb + lane]; }
    while (true) {
        const int64_t rn = r + stride;
        int64_t rbn = 0, ren = 0;
        if (rn < n_rows) { rbn = rowptr[rn]; ren = rowptr[rn + 1]; }
        int cln = 0;
        float vln = 0.0f;
        if (rbn + lane < ren) { cln = col[rbn + lane]; vln = val[rbn + lane]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t e = rb;
        while (true) {
            const int m = static_cast<int>(re - e < 64 ? re - e : 64);
            block64_deep(X4, cl, vl, m, grp, sub, acc);
            e += 64;
            if (e >= re) break;
            cl = 0; vl = 0.0f;
            if (e + lane < re) { cl = col[e + lane]; vl = val[e + lane]; }
        }
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if (grp == 0) reinterpret_cast<float4*>(Y)[r * 16 + sub] = acc;
        if (rn >= n_rows) break;
        r = rn; rb = rbn; re = ren; cl = cln; vl = vln;
    }
}

// blocked tasks, deep: a task's entries are padded so that lanes past its length read (col = the task's first column, val 0)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void blocked_deep_kernel(const int64_t* __restrict__ tptr, int n_blocks, int group,
                                                                  const int64_t* __restrict__ task_beg, const int32_t* __restrict__ task_len,
                                                                  const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                  const float* __restrict__ X, float* __restrict__ part) {
    const int b = group * 8 + (blockIdx.x & 7);
    if (b >= n_blocks) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int64_t n_w = static_cast<int64_t>(gridDim.x >> 3) * WAVES;
    const int64_t t_end = tptr[b + 1];
    int64_t t = tptr[b] + (blockIdx.x >> 3) * WAVES + wv;
    if (t >= t_end) return;
    int64_t eb = task_beg[t];
    int len = task_len[t];
    int cl = col[eb];                // the task's first column: what the padding lanes gather (inside the block)
    float vl = 0.0f;
    if (lane < len) { cl = col[eb + lane]; vl = val[eb + lane]; }
    while (true) {
        const int64_t tn = t + n_w;
        int64_t ebn = 0;
        int lenn = 0;
        if (tn < t_end) { ebn = task_beg[tn]; lenn = task_len[tn]; }
        int cln = 0;
        float vln = 0.0f;
        if (tn < t_end) {
            cln = col[ebn];
            if (lane < lenn) { cln = col[ebn + lane]; vln = val[ebn + lane]; }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e0 = 0;
        while (true) {
            const int m = len - e0 < 64 ? len - e0 : 64;
            block64_deep(X4, cl, vl, m, grp, sub, acc);
            e0 += 64;
            if (e0 >= len) break;
            cl = col[eb]; vl = 0.0f;
            if (e0 + lane < len) { cl = col[eb + e0 + lane]; vl = val[eb + e0 + lane]; }
        }
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if (grp == 0) reinterpret_cast<float4*>(part)[t * 16 + sub] = acc;
        if (tn >= t_end) break;
        t = tn; eb = ebn; len = lenn; cl = cln; vl = vln;
    }
}

// ordered sum of a long row's partial rows: one workgroup of 4 waves per row, each wave a contiguous quarter of the row's
// tasks with 8 loads in flight, the four sums combined in wave order
__global__ __launch_bounds__(256) void reduce_parts4_kernel(int n_long, const int32_t* __restrict__ long_rows, const int64_t* __restrict__ rt_ptr,
                                                            const int32_t* __restrict__ rt_ids, const float* __restrict__ part,
                                                            float* __restrict__ Y) {
    __shared__ float s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.x;
    const int64_t b = rt_ptr[k], n = rt_ptr[k + 1] - b;
    const int64_t q0 = b + n * wv / 4, q1 = b + n * (wv + 1) / 4;
    float acc = 0.0f;
    int64_t t = q0;
    for (; t + 8 <= q1; t += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = part[static_cast<int64_t>(rt_ids[t + j]) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[j];
    }
    for (; t < q1; ++t) acc += part[static_cast<int64_t>(rt_ids[t]) * 64 + lane];
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) Y[static_cast<int64_t>(long_rows[k]) * 64 + lane] = ((s[0][lane] + s[1][lane]) + s[2][lane]) + s[3][lane];
}

template <typename K>
int launch(K kern, int grid, int threads, size_t lds_bytes, hipStream_t st, int n_rows, const int64_t* rowptr, const int32_t* hotcnt,
           const int32_t* col, const float* val, const float* X, const int32_t* hot_cols, int n_hot, float* Y) {
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds_bytes));
        if (e != hipSuccess) return -10;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, st, n_rows, rowptr, hotcnt, col, val, X, hot_cols, n_hot, Y);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
}  // namespace

#define CASE(KERN, NFV, WV)                                                                                              \
    if (nf == NFV && waves == WV) {                                                                                      \
        return hot ? launch(KERN<NFV, true, WV>, grid, WV * 64, lds_bytes, st, n_rows, rowptr, hotcnt, col, val, X, hot_cols, n_hot, Y) \
                   : launch(KERN<NFV, false, WV>, grid, WV * 64, 0, st, n_rows, rowptr, hotcnt, col, val, X, hot_cols, 0, Y);           \
    }

extern "C" int lab_spmm_rows(int variant, int nf, int waves, int grid, int n_rows, const int64_t* rowptr, const int32_t* hotcnt,
                             const int32_t* col, const float* val, const float* X, const int32_t* hot_cols, int n_hot, float* Y,
                             void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool hot = n_hot > 0;
    const size_t lds_bytes = static_cast<size_t>(n_hot) * D * 4;
    if (variant == 0) {
        CASE(rows_dword_kernel, 4, 4) CASE(rows_dword_kernel, 8, 4) CASE(rows_dword_kernel, 16, 4)
        CASE(rows_dword_kernel, 4, 16) CASE(rows_dword_kernel, 8, 16) CASE(rows_dword_kernel, 16, 16)
        CASE(rows_dword_kernel, 8, 8) CASE(rows_dword_kernel, 16, 8)
    } else {
        CASE(rows_x4_kernel, 1, 4) CASE(rows_x4_kernel, 2, 4) CASE(rows_x4_kernel, 4, 4) CASE(rows_x4_kernel, 8, 4)
        CASE(rows_x4_kernel, 2, 16) CASE(rows_x4_kernel, 4, 16) CASE(rows_x4_kernel, 8, 16)
        CASE(rows_x4_kernel, 2, 8) CASE(rows_x4_kernel, 4, 8)
    }
    return -1;
}

extern "C" int lab_spmm_rows_pf(int nf, int waves, int grid, int n_rows, const int64_t* rowptr, const int32_t* col, const float* val,
                                const float* X, float* Y, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define PF(NFV, WV) if (nf == NFV && waves == WV) { hipLaunchKernelGGL((rows_x4_pf_kernel<NFV, WV>), dim3(grid), dim3(WV * 64), 0, st, n_rows, rowptr, col, val, X, Y); return hipGetLastError() == hipSuccess ? 0 : -11; }
    PF(2, 4) PF(4, 4) PF(8, 4) PF(2, 8) PF(4, 8) PF(4, 16)
    return -1;
}

extern "C" int lab_spmm_blocked(int nf, int waves, int wgs_per_xcd, const int64_t* tptr, int n_blocks, const int64_t* task_beg,
                                const int32_t* task_len, const int32_t* col, const float* val, const float* X, float* part,
                                void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int groups = (n_blocks + 7) / 8;
    for (int g = 0; g < groups; ++g) {
#define BT(NFV, WV) if (nf == NFV && waves == WV) hipLaunchKernelGGL((blocked_tasks_kernel<NFV, WV>), dim3(8 * wgs_per_xcd), dim3(WV * 64), 0, st, tptr, n_blocks, g, task_beg, task_len, col, val, X, part);
        BT(2, 4) BT(4, 4) BT(8, 4) BT(4, 8) BT(4, 16)
    }
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

// variant 4: rows_x4 with padded LDS rows (stride 68 floats) and/or non-temporal streams
extern "C" int lab_spmm_rows_v4(int nf, int waves, int grid, int lstride, int nt, int n_rows, const int64_t* rowptr, const int32_t* hotcnt,
                                const int32_t* col, const float* val, const float* X, const int32_t* hot_cols, int n_hot, float* Y,
                                void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds_bytes = static_cast<size_t>(n_hot) * lstride * 4;
#define V4(NFV, WV, HOTV, LS, NTV) if (nf == NFV && waves == WV && (n_hot > 0) == HOTV && lstride == LS && (nt != 0) == NTV) \
        return launch(rows_x4_kernel<NFV, HOTV, WV, LS, NTV>, grid, WV * 64, lds_bytes, st, n_rows, rowptr, hotcnt, col, val, X, hot_cols, n_hot, Y);
    V4(4, 4, false, 64, true) V4(4, 8, false, 64, true) V4(2, 4, false, 64, true)
    V4(4, 16, true, 68, false) V4(4, 16, true, 68, true) V4(2, 16, true, 68, false) V4(4, 16, true, 64, true)
    V4(4, 8, true, 68, false) V4(4, 8, true, 68, true)
    return -1;
}

// reduce the partial rows of the blocked tasks: one wavefront per long row, tasks listed per row (rt_ptr / rt_ids)
__global__ __launch_bounds__(256) void reduce_parts_kernel(int n_long, const int32_t* __restrict__ long_rows, const int64_t* __restrict__ rt_ptr,
                                                           const int32_t* __restrict__ rt_ids, const float* __restrict__ part,
                                                           float* __restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (k >= n_long) return;
    float acc = 0.0f;
    for (int64_t t = rt_ptr[k]; t < rt_ptr[k + 1]; ++t) acc += part[static_cast<int64_t>(rt_ids[t]) * 64 + lane];
    Y[static_cast<int64_t>(long_rows[k]) * 64 + lane] = acc;
}
extern "C" int lab_reduce_parts(int n_long, const int32_t* long_rows, const int64_t* rt_ptr, const int32_t* rt_ids, const float* part, float* Y,
                                void* stream) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((n_long * 64 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n_long,
                       long_rows, rt_ptr, rt_ids, part, Y);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

extern "C" int lab_spmm_rows_deep(int waves, int grid, int n_rows, const int64_t* rowptr, const int32_t* col, const float* val,
                                  const float* X, float* Y, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define DP(WV) if (waves == WV) { hipLaunchKernelGGL((rows_x4_deep_kernel<WV>), dim3(grid), dim3(WV * 64), 0, st, n_rows, rowptr, col, val, X, Y); return hipGetLastError() == hipSuccess ? 0 : -11; }
    DP(4) DP(8) DP(2)
    return -1;
}
extern "C" int lab_spmm_blocked_deep(int waves, int wgs_per_xcd, const int64_t* tptr, int n_blocks, const int64_t* task_beg,
                                     const int32_t* task_len, const int32_t* col, const float* val, const float* X, float* part,
                                     void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int groups = (n_blocks + 7) / 8;
    for (int g = 0; g < groups; ++g) {
        if (waves == 4) hipLaunchKernelGGL((blocked_deep_kernel<4>), dim3(8 * wgs_per_xcd), dim3(256), 0, st, tptr, n_blocks, g, task_beg, task_len, col, val, X, part);
        else if (waves == 8) hipLaunchKernelGGL((blocked_deep_kernel<8>), dim3(8 * wgs_per_xcd), dim3(512), 0, st, tptr, n_blocks, g, task_beg, task_len, col, val, X, part);
        else return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
extern "C" int lab_reduce_parts4(int n_long, const int32_t* long_rows, const int64_t* rt_ptr, const int32_t* rt_ids, const float* part, float* Y,
                                 void* stream) {
    hipLaunchKernelGGL(reduce_parts4_kernel, dim3(n_long), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n_long, long_rows, rt_ptr,
                       rt_ids, part, Y);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
