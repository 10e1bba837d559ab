// eval_select.hip -- E rows: drop-in for cpp_evaluate_matrix (utils/py/cython/include/evaluate.h:57)
// on a score matrix that already sits in HBM, plus the stand-alone metric kernel.
//
//   topk_rows_kernel   one workgroup per user row: stream the row once with 16-byte loads, keep only
//                      the scores that beat the running K-th best (threshold filter into an LDS
//                      candidate buffer, bitonic compaction when it fills), then rank + metrics.
//                      HBM-bound: n_items*4 bytes read per user, nothing else of size.
//   rank_metrics_kernel  metric.h:19-109 from arg-top-K lists (one lane per (user, metric)).
//   colsum_kernel        fp64 column sums of the per-user rows (evaluator.py:207-208 does an fp32
//                        np.mean; fp64 sums are exposed so the host can choose).
#include "eval_common.h"

#include <cmath>

namespace skr {
// 1 / log2(i + 2) for i < SKR_MAX_TOPK_SCORES, evaluated on the HOST with the libm the reference's metric.h uses, kept in a
// device buffer per GPU (512 doubles are too many for a kernel argument)
const double* inv_log2_table_device() {
    static const double* tables[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!tables[dev]) {
        double h[SKR_MAX_TOPK_SCORES];
        for (int i = 0; i < SKR_MAX_TOPK_SCORES; ++i) h[i] = 1.0 / std::log2(static_cast<double>(static_cast<unsigned>(i + 2)));
        double* d = nullptr;
        if (hipMalloc(&d, sizeof(h)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        tables[dev] = d;
    }
    return tables[dev];
}
}  // namespace skr

namespace {

constexpr int TK_T = 256;      // threads per row
constexpr int TK_CAP = 2048;   // LDS candidate capacity (16 KB of keys)
constexpr int TK_TILE = TK_T * 4;

struct RowOut {
    float* rows;       // [n_users, n_metric*top_k] or NULL
    int32_t* ids;      // [n_users, top_k] or NULL
};

__global__ __launch_bounds__(TK_T) void topk_rows_kernel(const float* __restrict__ scores, int n_items, int64_t ld,
                                                         const int64_t* __restrict__ test_rowptr,
                                                         const int32_t* __restrict__ test_items, skr::MetricArgs margs,
                                                         const double* __restrict__ tbl, int top_k, RowOut o) {
    __shared__ uint64_t keys[TK_CAP];
    __shared__ int s_cnt;
    __shared__ int s_over;     // index of the last tile whose appends took s_cnt past the compaction mark
    __shared__ uint64_t s_thr;
    __shared__ int s_rank[SKR_MAX_TOPK_SCORES];
    __shared__ double s_inv[SKR_MAX_TOPK_SCORES];
    __shared__ int s_wsum[TK_T / 64];
    __shared__ int s_tie;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float* r = scores + row * ld;
    if (tid == 0) {
        s_cnt = 0;
        s_over = -2;
        s_thr = SKR_KEY_MIN;
    }
    for (int i = tid; i < SKR_MAX_TOPK_SCORES; i += TK_T) s_inv[i] = tbl[i];
    __syncthreads();

    auto compact = [&]() {  // all threads; keeps the best top_k keys at the front, updates threshold
        const int cnt = s_cnt < TK_CAP ? s_cnt : TK_CAP;
        for (int i = cnt + tid; i < TK_CAP; i += TK_T) keys[i] = SKR_KEY_MIN;
        __syncthreads();
        skr::bitonic_sort_desc_lds<TK_CAP, TK_T>(keys);
        if (tid == 0) {   // one key beyond the K-th is kept: the tie test below needs the true (K+1)-th best
            const int keep = cnt < top_k + 1 ? cnt : top_k + 1;
            s_cnt = keep;
            s_thr = keep == top_k + 1 ? keys[top_k] : SKR_KEY_MIN;
        }
        __syncthreads();
    };

    const bool vec_ok = ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
    // The decision to compact must be the same in every thread.  Reading s_cnt at the top of a tile is NOT:
    // a fast thread may already be appending to it (this tile) while a slow one still evaluates the test, the
    // two then disagree and meet different barriers -- measured as a lost candidate in ~1 of 10^5 rows at
    // top-100 (tools/debug_dense_race.py).  Instead the thread whose append crosses the mark records the TILE
    // INDEX; the test at the top of tile `it` looks for `it - 1`, a value nobody can write during tile `it`.
    int it = 0;
    for (int base = 0; base < n_items; base += TK_TILE, ++it) {
        if (s_over == it - 1) compact();
        const uint64_t thr = s_thr;
        const int i0 = base + tid * 4;
        float v[4];
        int nv = 0;
        if (i0 + 3 < n_items && vec_ok) {
            const float4 q = *reinterpret_cast<const float4*>(r + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            nv = 4;
        } else {
            for (int e = 0; e < 4; ++e)
                if (i0 + e < n_items) { v[e] = r[i0 + e]; nv = e + 1; }
        }
        for (int e = 0; e < nv; ++e) {
            const uint64_t key = skr::rank_key(v[e], i0 + e);
            if (key > thr) {
                const int p = atomicAdd(&s_cnt, 1);
                keys[p] = key;  // p < TK_CAP: at most TK_TILE appends per tile and s_cnt <= CAP-TILE before
                if (p + 1 > TK_CAP - TK_TILE) s_over = it;
            }
        }
        __syncthreads();
    }
    compact();
    // rank list (top_k <= n_items guarantees s_cnt >= top_k here)
    if (tid == 0) s_tie = 0;
    __syncthreads();
    {   // equal scores among the best K+1?  Then the reference's order is heap-defined (see eval_common.h)
        const int n_chk = s_cnt - 1 < top_k ? s_cnt - 1 : top_k;
        for (int i = tid; i < n_chk; i += TK_T)
            if ((keys[i] >> 32) == (keys[i + 1] >> 32)) s_tie = 1;
    }
    __syncthreads();
    if (!s_tie) {
        for (int i = tid; i < top_k; i += TK_T) s_rank[i] = skr::key_id(keys[i]);
    } else {
        // second pass over the row, libstdc++'s partial_sort_copy step for step.  All threads scan a tile for
        // elements above the heap's current top (a superset of those that will enter: the top only rises),
        // compact them in index order, then one lane feeds them to the heap.
        const int sort_len = 2 * top_k < n_items ? 2 * top_k : n_items;   // evaluate.h:39
        // the key buffer is free from here on: its 16 KB hold the heap and the per-tile candidate list, so that
        // the tie path costs the common (tie-free) case no LDS and no occupancy
        int* s_hid = reinterpret_cast<int*>(keys);
        float* s_hval = reinterpret_cast<float*>(s_hid + 2 * SKR_MAX_TOPK_SCORES);
        int* s_cand = s_hid + 4 * SKR_MAX_TOPK_SCORES;
        static_assert((4 * SKR_MAX_TOPK_SCORES + TK_TILE) * 4 <= TK_CAP * 8, "tie-path scratch must fit the key buffer");
        skr::RefHeap h{s_hid, s_hval, sort_len};
        for (int i = tid; i < sort_len; i += TK_T) {
            s_hid[i] = i;
            s_hval[i] = r[i];
        }
        __syncthreads();
        if (tid == 0) skr::ref_make_heap(h);
        __syncthreads();
        const int lane = tid & 63, wv = tid >> 6;
        for (int base = sort_len; base < n_items; base += TK_TILE) {
            const float topv = s_hval[0];
            const int i0 = base + tid * 4;
            float v[4];
            int flags = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (i0 + e < n_items) ? r[i0 + e] : 0.0f;
                if (i0 + e < n_items && v[e] > topv) flags |= 1 << e;
            }
            const int mine = __popc(flags);
            const int incl = skr::wave_incl_scan(mine);
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            int off = incl - mine, total = 0;
            for (int w = 0; w < TK_T / 64; ++w) {
                if (w < wv) off += s_wsum[w];
                total += s_wsum[w];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (flags & (1 << e)) s_cand[off++] = i0 + e;
            __syncthreads();
            if (tid == 0) {
                for (int c = 0; c < total; ++c) {
                    const int x = s_cand[c];
                    const float xv = r[x];
                    if (xv > s_hval[0]) skr::ref_adjust_heap(h, 0, sort_len, x, xv);
                }
            }
            __syncthreads();
        }
        if (tid == 0) skr::ref_sort_heap(h);
        __syncthreads();
        for (int i = tid; i < top_k; i += TK_T) s_rank[i] = s_hid[i];
    }
    __syncthreads();
    if (o.ids)
        for (int i = tid; i < top_k; i += TK_T) o.ids[row * top_k + i] = s_rank[i];
    if (o.rows && tid < margs.n_metric) {
        const int64_t tb = test_rowptr[row], te = test_rowptr[row + 1];
        float* out = o.rows + (row * margs.n_metric + tid) * top_k;
        skr::metric_row(margs.ids[tid], s_rank, top_k, test_items, tb, te, s_inv, out);
    }
}

__global__ void rank_metrics_kernel(const int32_t* __restrict__ topk_ids, int B, int top_k,
                                    const int32_t* __restrict__ truth_rows, const int64_t* __restrict__ test_rowptr,
                                    const int32_t* __restrict__ test_items, skr::MetricArgs margs, const double* __restrict__ tbl,
                                    float* __restrict__ rows) {
    __shared__ double s_inv[SKR_MAX_TOPK_SCORES];
    for (int i = threadIdx.x; i < SKR_MAX_TOPK_SCORES; i += blockDim.x) s_inv[i] = tbl[i];
    __syncthreads();
    const int64_t g = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (g >= static_cast<int64_t>(B) * margs.n_metric) return;
    const int b = static_cast<int>(g / margs.n_metric), m = static_cast<int>(g % margs.n_metric);
    const int64_t tr = truth_rows ? truth_rows[b] : b;
    skr::metric_row(margs.ids[m], topk_ids + static_cast<int64_t>(b) * top_k, top_k, test_items, test_rowptr[tr],
                    test_rowptr[tr + 1], s_inv, rows + (static_cast<int64_t>(b) * margs.n_metric + m) * top_k);
}

// sums[c] += sum_r rows[r, c] in fp64; each block takes a slab of rows, one atomic per (block, col)
__global__ void colsum_kernel(const float* __restrict__ rows, int64_t n_rows, int n_cols, double* __restrict__ sums) {
    const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * per;
    int64_t r1 = r0 + per;
    if (r1 > n_rows) r1 = n_rows;
    for (int c = threadIdx.x; c < n_cols; c += blockDim.x) {
        double acc = 0.0;
        for (int64_t r = r0; r < r1; ++r) acc += static_cast<double>(rows[r * n_cols + c]);
        if (r1 > r0) atomicAdd(&sums[c], acc);
    }
}

// evaluator.py:197-200: one wavefront per (user row), lanes stride over the user's train items
__global__ void mask_train_kernel(float* __restrict__ scores, int B, int n_items, int64_t ld,
                                  const int32_t* __restrict__ users, const int64_t* __restrict__ rowptr,
                                  const int32_t* __restrict__ items) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (b >= B) return;
    const int64_t u = users[b];
    for (int64_t p = rowptr[u] + lane; p < rowptr[u + 1]; p += 64) {
        const int it = items[p];
        if (it >= 0 && it < n_items) scores[b * ld + it] = -INFINITY;
    }
}

// dense scores[b, i] = <user_table[users[b]], item_table[i]> (+ bias[i]) for the predict() API surface
// (BPRMF.py:84-88).  One wavefront per (user, 64-item strip): the user row is broadcast from LDS, every
// lane owns one item row (64 sequential fp32 fmas, k ascending, bias added last like the reference).
__global__ __launch_bounds__(256) void score_matrix_kernel(const float* __restrict__ user_table,
                                                           const int32_t* __restrict__ users, int B,
                                                           const float* __restrict__ item_table,
                                                           const float* __restrict__ bias, int n_items, int dim,
                                                           float* __restrict__ scores, int64_t ld) {
    __shared__ float s_u[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y * 4 + wv;
    const int item = blockIdx.x * 64 + lane;
    if (b < B)
        for (int c = lane; c < dim; c += 64) s_u[wv][c] = user_table[static_cast<int64_t>(users[b]) * dim + c];
    __syncthreads();
    if (b >= B || item >= n_items) return;
    const float4* row = reinterpret_cast<const float4*>(item_table + static_cast<int64_t>(item) * dim);
    float acc = 0.0f;
    for (int q = 0; q < dim / 4; ++q) {     // dim = 64: 16 trips, the same sums in the same order as before
        const float4 v = row[q];
        acc = fmaf(s_u[wv][4 * q + 0], v.x, acc);
        acc = fmaf(s_u[wv][4 * q + 1], v.y, acc);
        acc = fmaf(s_u[wv][4 * q + 2], v.z, acc);
        acc = fmaf(s_u[wv][4 * q + 3], v.w, acc);
    }
    if (bias) acc += bias[item];
    scores[static_cast<int64_t>(b) * ld + item] = acc;
}

int make_metric_args(const int* metric, int n_metric, skr::MetricArgs* out) {
    SKR_REQUIRE(metric && n_metric >= 1 && n_metric <= 8, "n_metric must be in [1, 8]");
    out->n_metric = n_metric;
    for (int m = 0; m < n_metric; ++m) {
        SKR_REQUIRE(metric[m] >= 1 && metric[m] <= 5, "unknown metric id %d (valid: 1..5, evaluator.py:57)", metric[m]);
        out->ids[m] = metric[m];
    }
    return SKR_OK;
}

}  // namespace

namespace skr {
int launch_colsum(const float* d_rows, int64_t n_rows, int n_cols, double* d_sums, hipStream_t st) {
    int blocks = static_cast<int>((n_rows + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, st, d_rows, n_rows, n_cols, d_sums);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}
}  // namespace skr

extern "C" {

int skr_eval_scores(const float* d_scores, int n_users, int n_items, int64_t ld, const int64_t* d_test_rowptr,
                    const int32_t* d_test_items, const int* metric, int n_metric, int top_k, float* d_rows,
                    int32_t* d_topk_ids, double* d_sums, void* stream) {
    SKR_REQUIRE(d_scores, "skr_eval_scores: d_scores is NULL");
    SKR_REQUIRE(n_users >= 0 && n_items > 0 && ld >= n_items, "skr_eval_scores: bad shape");
    SKR_REQUIRE(top_k >= 1 && top_k <= SKR_MAX_TOPK_SCORES, "top_k %d outside [1, %d]", top_k, SKR_MAX_TOPK_SCORES);
    SKR_REQUIRE(top_k <= n_items, "top_k %d larger than the catalogue (%d items)", top_k, n_items);
    SKR_REQUIRE(!d_sums || d_rows, "skr_eval_scores: d_sums needs d_rows");
    skr::MetricArgs margs{};
    if (d_rows) {
        SKR_REQUIRE(d_test_rowptr && d_test_items, "skr_eval_scores: test CSR is NULL");
        int rc = make_metric_args(metric, n_metric, &margs);
        if (rc) return rc;
    }
    if (n_users == 0) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    RowOut o{d_rows, d_topk_ids};
    const double* inv_tbl = skr::inv_log2_table_device();
    SKR_REQUIRE(inv_tbl, "skr_eval_scores: no device memory for the 1 / log2 table");
    hipLaunchKernelGGL(topk_rows_kernel, dim3(n_users), dim3(TK_T), 0, st, d_scores, n_items, ld, d_test_rowptr,
                       d_test_items, margs, inv_tbl, top_k, o);
    SKR_LAUNCH_CHECK();
    if (d_sums) return skr::launch_colsum(d_rows, n_users, margs.n_metric * top_k, d_sums, st);
    return SKR_OK;
}

int skr_score_matrix(const float* d_user_table, const int32_t* d_users, int B, const float* d_item_table,
                     const float* d_item_bias, int n_items, int dim, float* d_scores, int64_t ld, void* stream) {
    SKR_REQUIRE(d_user_table && d_users && d_item_table && d_scores, "skr_score_matrix: NULL argument");
    SKR_REQUIRE(dim >= 4 && dim <= 256 && dim % 4 == 0, "skr_score_matrix: dim must be a multiple of 4 in [4, 256] (got %d)", dim);
    SKR_REQUIRE(B >= 0 && n_items > 0 && ld >= n_items, "skr_score_matrix: bad shape");
    if (B == 0) return SKR_OK;
    dim3 grid(static_cast<unsigned>((n_items + 63) / 64), static_cast<unsigned>((B + 3) / 4));
    SKR_REQUIRE(grid.y <= 65535, "skr_score_matrix: at most 262140 users per call");
    hipLaunchKernelGGL(score_matrix_kernel, grid, dim3(256), 0, skr::as_stream(stream), d_user_table, d_users, B,
                       d_item_table, d_item_bias, n_items, dim, d_scores, ld);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_mask_train(float* d_scores, int B, int n_items, int64_t ld, const int32_t* d_users,
                   const int64_t* d_train_rowptr, const int32_t* d_train_items, void* stream) {
    SKR_REQUIRE(d_scores && d_users && d_train_rowptr && d_train_items, "skr_mask_train: NULL argument");
    SKR_REQUIRE(B >= 0 && n_items > 0 && ld >= n_items, "skr_mask_train: bad shape");
    if (B == 0) return SKR_OK;
    hipLaunchKernelGGL(mask_train_kernel, dim3(static_cast<unsigned>((static_cast<int64_t>(B) * 64 + 255) / 256)), dim3(256),
                       0, skr::as_stream(stream), d_scores, B, n_items, ld, d_users, d_train_rowptr, d_train_items);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_rank_metrics(const int32_t* d_topk_ids, int B, int top_k, const int32_t* d_truth_rows,
                     const int64_t* d_test_rowptr, const int32_t* d_test_items, const int* metric, int n_metric,
                     float* d_rows, double* d_sums, void* stream) {
    SKR_REQUIRE(d_topk_ids && d_test_rowptr && d_test_items && d_rows, "skr_rank_metrics: NULL argument");
    SKR_REQUIRE(B >= 0, "skr_rank_metrics: negative B");
    SKR_REQUIRE(top_k >= 1 && top_k <= SKR_MAX_TOPK_SCORES, "top_k %d outside [1, %d]", top_k, SKR_MAX_TOPK_SCORES);
    skr::MetricArgs margs{};
    int rc = make_metric_args(metric, n_metric, &margs);
    if (rc) return rc;
    if (B == 0) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    const int64_t work = static_cast<int64_t>(B) * n_metric;
    const double* inv_tbl = skr::inv_log2_table_device();
    SKR_REQUIRE(inv_tbl, "skr_rank_metrics: no device memory for the 1 / log2 table");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(static_cast<unsigned>((work + 127) / 128)), dim3(128), 0, st,
                       d_topk_ids, B, top_k, d_truth_rows, d_test_rowptr, d_test_items, margs, inv_tbl,
                       d_rows);
    SKR_LAUNCH_CHECK();
    if (d_sums) return skr::launch_colsum(d_rows, B, n_metric * top_k, d_sums, st);
    return SKR_OK;
}

}  // extern "C"
