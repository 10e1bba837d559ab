// spmm.hip -- T7 / T10: the sparse propagation  Y = A X  (d = 64) with a per-matrix PLAN.
//
// Replaces torch.sparse.mm of the reference, called once per layer and direction on the full graph for every mini-batch:
//   recommender/LightGCN.py:89-100   (_forward_gcn: K x torch.sparse.mm + the autograd backward of each)
//   recommender/LayerGCN.py:207-220  (forward: sparse.mm per layer)
//
// A bipartite recommendation graph has two very different kinds of rows.  User rows are short (tens of entries) and
// gather from the small, popularity-skewed item table, which mostly sits in the XCDs' L2.  Item rows are long (hundreds
// to ~10^6 entries) and gather from the user table (256 MB at 1 M users), far larger than any cache.  Measured on the
// 1 M x 100 k x 48 M graph (tools/spmm_lab.py, profiles/r02_spmm_lab.txt), one dword per lane and four row gathers in
// flight (the round-1 kernel) moved rows at 8.6 / 7.3 TB/s (user / item side).  This file:
//   * short rows (< SPMM_LONG entries): one wavefront per row, 16 BYTES per lane -- a 16-lane group fetches one 256-byte
//     row, a wave instruction four rows -- with four such instructions in flight: 15.3 TB/s on the user side;
//   * long rows: cut into TASKS of <= 256 consecutive entries whose columns lie in one COLUMN BLOCK of 16 384 columns
//     (a 4 MB slice of X).  Tasks are ordered by block, and launch g lets the workgroups with blockIdx % 8 == x -- which
//     share an XCD, hence an L2, under the round-robin placement (a speed assumption only) -- work through block 8 g + x:
//     the slice is fetched into that L2 once and the rows' entries gather from it.  11 TB/s instead of 7 on the long
//     rows of the item side.  A task leaves a partial row in a scratch buffer; a third kernel adds a row's partial rows
//     in a fixed order, so results do not depend on timing (the round-1 kernel combined split rows with float atomics).
// Entries stay where they are in the caller's CSR: a plan is a task list (16 bytes per task), not a copy of the matrix.
#include "skr_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

constexpr int D = 64;
constexpr int SPMM_LONG = 512;       // rows with at least this many entries are cut into column-blocked tasks
constexpr int SPMM_CBLK = 16384;     // columns per block of matrices too narrow to choose (see plan_default_cblk)
// Columns per block of a plan.  The task launches walk the blocks in groups of 8 (launch g: the workgroups that share XCD x
// take block 8g + x), so a block count that is NOT a multiple of 8 leaves XCDs idle in the last launch, and blocks of
// ~30 k columns (7.5 MB of X per XCD and launch) beat the 16 384 of round 2 now that the densest rows no longer go this
// way.  Measured on the item side of the 1 M-user graph, LightGCN step / item side (tools/r3_spmm_params.sh, same box):
// 15 680 (64 blocks) 10.82 / 1.309 ms, 16 384 (62) 10.80 / 1.303, 20 864 (48) 10.58 / 1.255, 25 024 (40) 10.43 / 1.217,
// 31 296 (32) 10.36 / 1.210, 32 768 (31) 10.52 / 1.25, 36 864 (28) 10.85 / 1.339, 41 728 (24) 10.65 / 1.278, 65 536 (16)
// 11.02 / 1.377.  Hence: 8 * round(n_cols / (8 * 28 672)) blocks, at least 8, of equal width (a multiple of 64).
inline int plan_default_cblk(int n_cols) {
    if (n_cols < 8 * 256) return SPMM_CBLK;
    const int64_t groups = std::max<int64_t>(1, (static_cast<int64_t>(n_cols) + 4 * 28672) / (8 * 28672));
    const int64_t nb = 8 * groups;
    const int64_t w = ((static_cast<int64_t>(n_cols) + nb - 1) / nb + 63) / 64 * 64;
    return static_cast<int>(std::max<int64_t>(w, 2048));     // (narrow matrices: fewer, not thinner, blocks)
}
constexpr int SPMM_TASK = 256;       // entries per task at most
constexpr int ROW_WAVES = 4;         // wavefronts per workgroup, both kernels
constexpr int ROW_NF = 4;            // 16-byte gathers in flight per lane
// workgroups per XCD of a task launch, at most.  Round 2 launched 256 (each wavefront striding through ~7 tasks of 1 .. 256
// entries); with 2 048 a wavefront has one or two tasks and the hardware dispatcher does the balancing: LightGCN step
// 10.25 -> 9.83 ms together with 16 384 instead of 8 192 workgroups of the short-row kernel (tools/r3_spmm_params.sh, same
// box: 128 / 256 / 512 / 1 024 / 2 048 / 4 096 / 8 192 per XCD = 10.70 / 10.25 / 10.16 / 9.92 / 9.83 / 9.88 / 10.00 ms; rows
// kernel 4 096 / 8 192 / 16 384 / 65 536 / 262 144 workgroups: user side 0.885 / 0.870 / 0.841 / 0.838 / 0.907 ms)
constexpr int BLK_WGS_PER_XCD = 2048;
constexpr int ROWS_WGS_MAX = 16384;
// the DENSEST rows (round 3): a row that names a sizeable share of ALL columns does not gather -- X is streamed through LDS
// in blocks of HOT_UB rows and the row's entries read it there
constexpr int HOT_UB = 128;          // rows of X per LDS block (32 KB)
constexpr int HOT_V = 128;           // virtual hot rows at most: 8 wavefronts x 4 slots x 4 sixteen-lane groups
constexpr int HOT_WAVES = 8;         // wavefronts per workgroup
constexpr int HOT_WGS = 512;         // workgroups (two per CU: 64 KB of LDS each)

struct Task {
    int64_t beg;         // first entry (index into the caller's col / val)
    uint32_t len_slot;   // (length - 1) in the low 8 bits, the long row's slot (index into long_rows) above them
    int32_t part;        // slot of its partial row (row-major over the long rows)
};
static_assert(SPMM_TASK <= 256, "a task's length - 1 is packed into 8 bits");

// accumulate entries [e, e + m) (m <= 64, held one per lane in cl / vl; lanes >= m hold a valid column and value 0)
__device__ __forceinline__ void gather_block(const float4* __restrict__ X4, int cl, float vl, int m, int grp, int sub, float4& acc, int ld4 = 16) {
    int k = 0;
    for (; k + 4 * ROW_NF <= m; k += 4 * ROW_NF) {
        float4 x[ROW_NF];
        float v[ROW_NF];
#pragma unroll
        for (int q = 0; q < ROW_NF; ++q) {
            x[q] = X4[static_cast<int64_t>(__shfl(cl, k + 4 * q + grp)) * ld4 + sub];
            v[q] = __shfl(vl, k + 4 * q + grp);
        }
#pragma unroll
        for (int q = 0; q < ROW_NF; ++q) {
            acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y);
            acc.z = fmaf(v[q], x[q].z, acc.z); acc.w = fmaf(v[q], x[q].w, acc.w);
        }
    }
    for (; k < m; k += 4) {   // every lane takes part in both shuffles (a bpermute reads 0 from a masked-off lane)
        const float4 x = X4[static_cast<int64_t>(__shfl(cl, k + grp)) * ld4 + sub];
        const float v = __shfl(vl, k + grp);
        acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
    }
}

// the four 16-lane groups hold partial sums of the same four dims: add them up (afterwards every group holds the total)
__device__ __forceinline__ void sum_groups(float4& acc) {
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
}

// keep only the entries whose column is marked in `col_mask` (the others would multiply rows of X that are known to be
// zero): kept entries move to the front of the block in their order, the others behind them with value 0; returns how
// many were kept.  Every lane takes part in the two permutes.
// (tried in round 3: the mask as bits in LDS instead of a byte per entry out of L2 -- LightGCN 10.92-10.94 vs 10.96 ms per step:
//  the mask look-ups are not what the column-masked products wait for)
__device__ __forceinline__ int keep_marked(const uint8_t* __restrict__ col_mask, int lane, int m, int& cl, float& vl) {
    const bool keep = lane < m && col_mask[cl] != 0;
    const unsigned long long b = __ballot(keep);
    const int kept = __popcll(b);
    const int below = __popcll(b & ((1ull << lane) - 1ull));
    const int dst = keep ? below : kept + (lane - below);
    cl = __builtin_amdgcn_ds_permute(dst << 2, cl);
    vl = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(keep ? vl : 0.0f)));
    return kept;
}

// ---- what happens to a finished row (skr_spmm_epilogue, include/skrec_hip.h) ---------------------------------------------
// The row is held V floats per lane: V = 4 in the row kernel (lane `idx` = sub of a 16-lane group owns floats 4 idx .. 4 idx + 3;
// all four groups hold the same totals, `writer` = group 0 stores), V = 1 in the reduce kernel (64 lanes, one float each).
// The LayerGCN refinements need sums over the row: V = 4 adds its four floats, then 16 lanes by xor-shuffles.
constexpr float COS_EPS = 1e-8f;     // F.cosine_similarity default (LayerGCN.py:214)

template <int V>
__device__ __forceinline__ float row_total(const float (&p)[V]) {
    float s = p[0];
#pragma unroll
    for (int j = 1; j < V; ++j) s += p[j];
    if (V == 4) {
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    } else {
        s = skr::wave_sum(s);
    }
    return s;
}
// (ld: the row stride of every dense operand in floats -- 64 for a [n, 64] table, 64 C for one 64-column slice of a
//  [n, 64 C] table: a wider embedding is multiplied slice by slice, the product being separable in the columns)
template <int V>
__device__ __forceinline__ void row_load(const float* __restrict__ t, int64_t r, int idx, float (&o)[V], int ld = D) {
    if (V == 4) {
        const float4 a = reinterpret_cast<const float4*>(t + r * ld)[idx];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    } else {
        o[0] = t[r * ld + idx];
    }
}
template <int V>
__device__ __forceinline__ void row_store(float* __restrict__ t, int64_t r, int idx, const float (&o)[V], int ld = D) {
    if (V == 4) reinterpret_cast<float4*>(t + r * ld)[idx] = make_float4(o[0], o[1], o[2], o[3]);
    else t[r * ld + idx] = o[0];
}

template <int V>
__device__ __forceinline__ void epilogue(const skr_spmm_epilogue& ep, float (&y)[V], int64_t r, int idx, bool writer) {
    const int ld = ep.ld ? ep.ld : D;        // the refinements (whole-row reductions) are only run with ld == 64
    if (ep.addend && !(ep.addend_mask && !ep.addend_mask[r])) {     // a row of the addend known to be zero is not read
        float a[V];
        row_load<V>(ep.addend, r, idx, a, ld);
#pragma unroll
        for (int j = 0; j < V; ++j) y[j] += a[j];
    }
    const bool accumulate = ep.accum && !(ep.accum_mask && !ep.accum_mask[r]);   // only the rows somebody will read
    if (ep.mode == SKR_EPI_PLAIN) {
        if (writer && ep.Y) row_store<V>(ep.Y, r, idx, y, ld);
        if (accumulate) {
            float c[V];
            if (ep.accum_base) {              // accum = scale * base + scale * y: the layer mean's first term folded in
                row_load<V>(ep.accum_base, r, idx, c, ld);
#pragma unroll
                for (int j = 0; j < V; ++j) c[j] = ep.accum_scale * c[j];
            } else if (ep.accum_init) {
#pragma unroll
                for (int j = 0; j < V; ++j) c[j] = 0.0f;
            } else {
                row_load<V>(ep.accum, r, idx, c, ld);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) c[j] += ep.accum_scale * y[j];
            if (writer) row_store<V>(ep.accum, r, idx, c, ld);
        }
        return;
    }
    float e[V], p[V];
    row_load<V>(ep.E, r, idx, e);
    if (ep.mode == SKR_EPI_REFINE_FWD) {
        // LayerGCN.py:214-216: w = cos(y, e0); z = w * y feeds the next layer and the sum of the layers
#pragma unroll
        for (int j = 0; j < V; ++j) p[j] = y[j] * y[j];
        const float ny = fmaxf(sqrtf(row_total<V>(p)), COS_EPS);
#pragma unroll
        for (int j = 0; j < V; ++j) p[j] = e[j] * e[j];
        const float ne = fmaxf(sqrtf(row_total<V>(p)), COS_EPS);
#pragma unroll
        for (int j = 0; j < V; ++j) p[j] = (y[j] / ny) * (e[j] / ne);
        const float w = row_total<V>(p);
        float z[V];
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] = w * y[j];
        if (writer) {
            if (ep.Y) row_store<V>(ep.Y, r, idx, y);
            row_store<V>(ep.Z, r, idx, z);
            if (idx == 0) ep.w[r] = w;
        }
        if (accumulate) {
            float c[V];
            if (ep.accum_init) {
#pragma unroll
                for (int j = 0; j < V; ++j) c[j] = z[j];
            } else {
                row_load<V>(ep.accum, r, idx, c);
#pragma unroll
                for (int j = 0; j < V; ++j) c[j] += z[j];
            }
            if (writer) row_store<V>(ep.accum, r, idx, c);
        }
        return;
    }
    // SKR_EPI_REFINE_BWD: the finished row is dZ of the layer below; dY = backward of that layer's refinement, dE += its part
    float yr[V];
    row_load<V>(ep.rawY, r, idx, yr);
    const float w = ep.w[r];
#pragma unroll
    for (int j = 0; j < V; ++j) p[j] = yr[j] * yr[j];
    const float nyr = sqrtf(row_total<V>(p));
#pragma unroll
    for (int j = 0; j < V; ++j) p[j] = e[j] * e[j];
    const float ner = sqrtf(row_total<V>(p));
    const float ny = fmaxf(nyr, COS_EPS), ne = fmaxf(ner, COS_EPS);
#pragma unroll
    for (int j = 0; j < V; ++j) p[j] = y[j] * yr[j];
    const float dw = row_total<V>(p);
    float dy[V], de[V];
    row_load<V>(ep.dE, r, idx, de);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const float yh = yr[j] / ny, eh = e[j] / ne;
        // d(yh)/dy = (I - yh yh^T)/ny when the norm is not clamped, I/eps when it is (clamp_min has zero slope)
        const float gy = (nyr > COS_EPS) ? (eh - w * yh) / ny : eh / ny;
        const float ge = (ner > COS_EPS) ? (yh - w * eh) / ne : yh / ne;
        dy[j] = w * y[j] + dw * gy;
        de[j] += dw * ge;
    }
    if (writer) {
        row_store<V>(ep.Y, r, idx, dy);
        row_store<V>(ep.dE, r, idx, de);
    }
}

// rows shorter than `long_thr`: one wavefront per row (empty rows included: Y = addend)
template <bool COLMASK>
__global__ __launch_bounds__(ROW_WAVES * 64) void spmm_rows_kernel(int n_rows, int long_thr, const int64_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                   const float* __restrict__ X, const skr_spmm_epilogue ep,
                                                                   const uint8_t* __restrict__ row_mask,
                                                                   const uint8_t* __restrict__ col_mask,
                                                                   const int64_t* __restrict__ split, int n_win, int win) {
    // split / n_win / win: the columns are cut into n_win WINDOWS (X is far larger than the Infinity Cache: each launch
    // gathers from one window of it); split[r * (n_win - 1) + w] = first entry of row r in window w + 1.  Launch `win`
    // takes the row's entries of its window; the first writes Y, the later ones add to it, the last applies the epilogue.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int ld = ep.ld ? ep.ld : D, ld4 = ld >> 2;
    for (int64_t r = blockIdx.x * ROW_WAVES + wv; r < n_rows; r += static_cast<int64_t>(gridDim.x) * ROW_WAVES) {
        if (row_mask && !row_mask[r]) continue;
        int64_t rb = rowptr[r], re = rowptr[r + 1];
        if (re - rb >= long_thr) continue;
        if (n_win > 1) {
            const int64_t* sp = split + r * (n_win - 1);
            if (win > 0) rb = sp[win - 1];
            if (win < n_win - 1) re = sp[win];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n_win > 1 && win > 0 && grp == 0) acc = reinterpret_cast<const float4*>(ep.Y + r * ld)[sub];   // the earlier windows' sum
        for (int64_t e = rb; e < re; e += 64) {
            int m = static_cast<int>(re - e < 64 ? re - e : 64);
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) { cl = col[e + lane]; vl = val[e + lane]; }
            if (COLMASK) m = keep_marked(col_mask, lane, m, cl, vl);
            gather_block(X4, cl, vl, m, grp, sub, acc, ld4);
        }
        sum_groups(acc);
        if (win == n_win - 1) {
            float y[4] = {acc.x, acc.y, acc.z, acc.w};
            epilogue<4>(ep, y, r, sub, grp == 0);
        } else if (grp == 0) {
            reinterpret_cast<float4*>(ep.Y + r * ld)[sub] = acc;
        }
    }
}

// first entry of every short row in each column window after the first (plan construction)
__global__ void window_split_kernel(int n_rows, int long_thr, int n_win, int64_t win_cols, const int64_t* __restrict__ rowptr,
                                    const int32_t* __restrict__ col, int64_t* __restrict__ split) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<int64_t>(n_rows) * (n_win - 1)) return;
    const int64_t r = i / (n_win - 1);
    const int w = static_cast<int>(i - r * (n_win - 1));
    const int64_t rb = rowptr[r], re = rowptr[r + 1];
    int64_t lo = rb, hi = re;
    const int64_t c = static_cast<int64_t>(w + 1) * win_cols;
    if (re - rb < long_thr)
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col[mid] < c) lo = mid + 1; else hi = mid; }
    split[i] = lo;
}

// long rows: the tasks of the column blocks 8 * g + (blockIdx % 8), g = group .. group + span - 1, one block after the other
template <bool COLMASK>
__global__ __launch_bounds__(ROW_WAVES * 64) void spmm_tasks_kernel(const int64_t* __restrict__ first_task, int n_long, int n_blocks, int group, int span,
                                                                    const Task* __restrict__ tasks, const int32_t* __restrict__ col,
                                                                    const float* __restrict__ val, const float* __restrict__ X,
                                                                    float* __restrict__ part, const int32_t* __restrict__ long_rows,
                                                                    const uint8_t* __restrict__ row_mask,
                                                                    const uint8_t* __restrict__ col_mask, int ld4,
                                                                    const uint8_t* __restrict__ hot_flag) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int64_t n_w = static_cast<int64_t>(gridDim.x >> 3) * ROW_WAVES;
    for (int g = group; g < group + span; ++g) {
    const int b = g * 8 + (blockIdx.x & 7);
    if (b >= n_blocks) return;
    const int64_t t_end = first_task[static_cast<int64_t>(b + 1) * n_long];
    for (int64_t t = first_task[static_cast<int64_t>(b) * n_long] + (blockIdx.x >> 3) * ROW_WAVES + wv; t < t_end; t += n_w) {
        const Task tk = tasks[t];
        if (hot_flag && hot_flag[tk.len_slot >> 8]) continue;          // the row goes through the LDS-streamed form in this call
        if (row_mask && !row_mask[long_rows[tk.len_slot >> 8]]) continue;
        const int len = static_cast<int>(tk.len_slot & 255u) + 1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e0 = 0; e0 < len; e0 += 64) {
            int m = len - e0 < 64 ? len - e0 : 64;
            int cl = col[tk.beg];          // padding lanes gather the task's first column: an address inside the block
            float vl = 0.0f;
            if (lane < m) { cl = col[tk.beg + e0 + lane]; vl = val[tk.beg + e0 + lane]; }
            if (COLMASK) m = keep_marked(col_mask, lane, m, cl, vl);
            gather_block(X4, cl, vl, m, grp, sub, acc, ld4);
        }
        sum_groups(acc);
        if (grp == 0) reinterpret_cast<float4*>(part)[static_cast<int64_t>(tk.part) * 16 + sub] = acc;
    }
    }
}

// a long row = the sum of its partial rows, added in slot order by one workgroup: every wavefront sums a contiguous
// quarter of the row's slots (8 loads in flight), the four sums are combined in wave order
__global__ __launch_bounds__(256) void spmm_reduce_kernel(const int32_t* __restrict__ long_rows, const int64_t* __restrict__ part_ptr,
                                                          const float* __restrict__ part, const skr_spmm_epilogue ep,
                                                          const uint8_t* __restrict__ row_mask, const uint8_t* __restrict__ hot_flag) {
    __shared__ float s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (hot_flag && hot_flag[blockIdx.x]) return;                  // finished by spmm_hot_reduce_kernel in this call
    if (row_mask && !row_mask[long_rows[blockIdx.x]]) return;      // the whole workgroup leaves together
    const int64_t b = part_ptr[blockIdx.x], n = part_ptr[blockIdx.x + 1] - b;
    const int64_t q0 = b + n * wv / 4, q1 = b + n * (wv + 1) / 4;
    float acc = 0.0f;
    int64_t t = q0;
    for (; t + 8 <= q1; t += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = part[(t + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[j];
    }
    for (; t < q1; ++t) acc += part[t * 64 + lane];
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) {
        const int64_t r = long_rows[blockIdx.x];
        float y[1] = {((s[0][lane] + s[1][lane]) + s[2][lane]) + s[3][lane]};
        epilogue<1>(ep, y, r, lane, true);
    }
}

// ---- the densest rows: X streamed through LDS -------------------------------------------------------------------------
// A row of the item side that names 3 .. 80 % of all users gathers most of X anyway -- row by row, 256 bytes at a time, out
// of L2 at best.  Those rows (at most HOT_V "virtual" rows: a row far heavier than the others is cut into P pieces, piece p
// taking every P-th of the row's entries inside each X block) are multiplied the other way round: a workgroup walks a
// contiguous range of X blocks of HOT_UB rows, brings each into LDS with coalesced loads (X is read exactly once, in
// order), and the entries read their X rows THERE.  Work mapping: a sixteen-lane group multiplies one virtual row at a
// time (a lane owns four of the 64 dims: ds_read_b128, four fmas, no cross-lane sum), the four groups of a wavefront
// take four rows of similar length side by side (the plan sorts them so), a wavefront has four such slots, a workgroup
// eight wavefronts: 8 x 4 x 4 = 128 rows, each with its own float4 accumulator register per lane.  The entries are re-packed
// at plan time block-major (hot_meta = column inside the block, hot_val; hot_list_ptr per (block, virtual row)), the first
// sixteen of every list are fetched one block ahead, and a wavefront hands its 64 fetched entries to its groups through
// 512 bytes of LDS (a group reads its j-th entry with one broadcast ds_read_b64).  At the end the workgroup's 128 partial
// rows go to hot_part, and spmm_hot_reduce_kernel adds the workgroups' (and pieces') partial rows in a fixed order and runs
// the epilogue.  Every order of addition is fixed by the plan: results do not depend on timing.
constexpr int HOT_SLOTS = 4;         // rows a group handles, one after the other, per block

template <int HOT_PF>     // rounds of sixteen entries fetched ahead per list
__global__ __launch_bounds__(HOT_WAVES * 64) void spmm_hot_kernel(const int64_t* __restrict__ list_ptr, const int32_t* __restrict__ meta,
                                                                  const float* __restrict__ hval, const float* __restrict__ X, int ld4,
                                                                  int n_cols, int n_ub, float* __restrict__ part) {
    __shared__ float xb[2][HOT_UB * D];
    __shared__ int2 stage[HOT_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int per = (n_ub + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int ub0 = static_cast<int>(blockIdx.x) * per, ub1 = ub0 + per < n_ub ? ub0 + per : n_ub;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4 r4[4];
    auto load_block = [&](int ub) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + HOT_WAVES * 64 * q;                    // float4 index inside the block: 128 rows x 16
            const int64_t c = static_cast<int64_t>(ub) * HOT_UB + (idx >> 4);
            r4[q] = c < n_cols ? X4[c * ld4 + (idx & 15)] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_block = [&](int b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(xb[b])[tid + HOT_WAVES * 64 * q] = r4[q];
    };
    float4 acc[HOT_SLOTS];
#pragma unroll
    for (int s_ = 0; s_ < HOT_SLOTS; ++s_) acc[s_] = make_float4(0.f, 0.f, 0.f, 0.f);
    // virtual row of (wavefront, slot, group) = wv * 16 + slot * 4 + grp; its list of a block: bounds and the first sixteen
    // entries (a lane of the group each) are fetched one block ahead
    int64_t l0c[HOT_SLOTS], l0n[HOT_SLOTS];
    int nc[HOT_SLOTS], nn[HOT_SLOTS];
    int mc[HOT_SLOTS][HOT_PF], mn[HOT_SLOTS][HOT_PF];
    float vc[HOT_SLOTS][HOT_PF], vn[HOT_SLOTS][HOT_PF];
    auto fetch = [&](int ub, int64_t (&l0)[HOT_SLOTS], int (&n)[HOT_SLOTS], int (&m)[HOT_SLOTS][HOT_PF], float (&v)[HOT_SLOTS][HOT_PF]) {
#pragma unroll
        for (int s_ = 0; s_ < HOT_SLOTS; ++s_) {
            l0[s_] = 0; n[s_] = 0;
#pragma unroll
            for (int t = 0; t < HOT_PF; ++t) { m[s_][t] = 0; v[s_][t] = 0.0f; }
            if (ub < ub1) {
                const int64_t* lp = list_ptr + static_cast<int64_t>(ub) * HOT_V + wv * 16 + s_ * 4 + grp;
                l0[s_] = lp[0];
                n[s_] = static_cast<int>(lp[1] - lp[0]);
#pragma unroll
                for (int t = 0; t < HOT_PF; ++t)
                    if (t * 16 + sub < n[s_]) { m[s_][t] = meta[l0[s_] + t * 16 + sub]; v[s_][t] = hval[l0[s_] + t * 16 + sub]; }
            }
        }
    };
    if (ub0 < ub1) {
        load_block(ub0);
        store_block(0);
    }
    fetch(ub0, l0c, nc, mc, vc);
    __syncthreads();
    for (int ub = ub0; ub < ub1; ++ub) {
        const int b = (ub - ub0) & 1;
        if (ub + 1 < ub1) load_block(ub + 1);                           // X of the next block: in flight while this one is multiplied
        fetch(ub + 1, l0n, nn, mn, vn);                                  // ... and the next block's entries
        const float4* xs = reinterpret_cast<const float4*>(xb[b]) + sub;
        const int2* mine = &stage[wv][grp * 16];
#pragma unroll
        for (int s_ = 0; s_ < HOT_SLOTS; ++s_) {
            const int n = nc[s_];
            int nmax = __builtin_amdgcn_readlane(n, 0);
            nmax = max(nmax, __builtin_amdgcn_readlane(n, 16));
            nmax = max(nmax, __builtin_amdgcn_readlane(n, 32));
            nmax = max(nmax, __builtin_amdgcn_readlane(n, 48));
            for (int i0 = 0; i0 < nmax; i0 += 16) {
                int m_ = mc[s_][0];
                float v_ = vc[s_][0];
                if (HOT_PF > 1 && i0 == 16) { m_ = mc[s_][HOT_PF - 1]; v_ = vc[s_][HOT_PF - 1]; }
                if (i0 >= 16 * HOT_PF) {                                 // a longer list: fetched as needed
                    m_ = 0; v_ = 0.0f;
                    if (i0 + sub < n) { m_ = meta[l0c[s_] + i0 + sub]; v_ = hval[l0c[s_] + i0 + sub]; }
                }
                stage[wv][lane] = make_int2(m_, __float_as_int(v_));     // (lanes behind a list's end: column 0, value 0)
                const int jn = nmax - i0 < 16 ? nmax - i0 : 16;
                int j = 0;
                for (; j + 4 <= jn; j += 4) {
                    int2 e[4];
                    float4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        e[u] = mine[j + u];
                        x[u] = xs[e[u].x * 16];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float vj = __int_as_float(e[u].y);
                        acc[s_].x = fmaf(vj, x[u].x, acc[s_].x); acc[s_].y = fmaf(vj, x[u].y, acc[s_].y);
                        acc[s_].z = fmaf(vj, x[u].z, acc[s_].z); acc[s_].w = fmaf(vj, x[u].w, acc[s_].w);
                    }
                }
                for (; j < jn; ++j) {
                    const int2 e = mine[j];
                    const float4 x = xs[e.x * 16];
                    const float vj = __int_as_float(e.y);
                    acc[s_].x = fmaf(vj, x.x, acc[s_].x); acc[s_].y = fmaf(vj, x.y, acc[s_].y);
                    acc[s_].z = fmaf(vj, x.z, acc[s_].z); acc[s_].w = fmaf(vj, x.w, acc[s_].w);
                }
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < HOT_SLOTS; ++s_) {
            l0c[s_] = l0n[s_]; nc[s_] = nn[s_];
#pragma unroll
            for (int t = 0; t < HOT_PF; ++t) { mc[s_][t] = mn[s_][t]; vc[s_][t] = vn[s_][t]; }
        }
        if (ub + 1 < ub1) store_block(b ^ 1);
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(part + (static_cast<int64_t>(blockIdx.x) * HOT_V + wv * 16 + grp) * D) + sub;
#pragma unroll
    for (int s_ = 0; s_ < HOT_SLOTS; ++s_) out[s_ * 4 * 16] = acc[s_];
}

// a virtual row = the sum of the workgroups' partial rows, in workgroup order: 16 wavefronts take contiguous shares of the
// workgroups, their sums are combined in wave order (one workgroup per virtual row: the heaviest real row has 8 of them)
__global__ __launch_bounds__(1024) void spmm_hot_vsum_kernel(const int32_t* __restrict__ virt_row, const float* __restrict__ part,
                                                              int n_wgs, float* __restrict__ vsum) {
    __shared__ float s[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v = blockIdx.x;
    if (virt_row[v] < 0) return;
    const int q0 = n_wgs * wv / 16, q1 = n_wgs * (wv + 1) / 16;
    float sum = 0.0f;
    int q = q0;
    for (; q + 16 <= q1; q += 16) {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = part[(static_cast<int64_t>(q + j) * HOT_V + v) * D + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += x[j];
    }
    for (; q < q1; ++q) sum += part[(static_cast<int64_t>(q) * HOT_V + v) * D + lane];
    s[wv][lane] = sum;
    __syncthreads();
    if (wv == 0) {
        float t = s[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += s[w][lane];
        vsum[v * D + lane] = t;
    }
}

// a hot row = the sum of its pieces (virtual rows), in piece order, then the epilogue: one wavefront per row
__global__ __launch_bounds__(256) void spmm_hot_reduce_kernel(int n_hot, const int32_t* __restrict__ hot_rows,
                                                              const int32_t* __restrict__ hot_vptr, const int32_t* __restrict__ hot_vidx,
                                                              const float* __restrict__ vsum, const skr_spmm_epilogue ep,
                                                              const uint8_t* __restrict__ row_mask) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_hot) return;
    const int64_t row = hot_rows[k];
    if (row_mask && !row_mask[row]) return;
    float t = 0.0f;
    for (int q = hot_vptr[k]; q < hot_vptr[k + 1]; ++q) t += vsum[hot_vidx[q] * D + lane];
    float y[1] = {t};
    epilogue<1>(ep, y, row, lane, true);
}

// ---- plan construction --------------------------------------------------------------------------------------------
__global__ void long_deg_kernel(int n_long, const int32_t* __restrict__ long_rows, const int64_t* __restrict__ rowptr,
                                int64_t* __restrict__ deg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_long) deg[s] = rowptr[long_rows[s] + 1] - rowptr[long_rows[s]];
}

__global__ void flag_long_kernel(int n_rows, int thr, const int64_t* __restrict__ rowptr, int64_t* __restrict__ flag) {
    const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (r < n_rows) flag[r] = (rowptr[r + 1] - rowptr[r] >= thr) ? 1 : 0;
    if (r == n_rows) flag[r] = 0;
}

// in-place exclusive scan of n int64 values by ONE workgroup (plan construction only: n is a few 10^5 .. 10^6)
__global__ __launch_bounds__(1024) void scan_kernel(int64_t* __restrict__ a, int64_t n) {
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n ? a[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        int64_t before = s_carry;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        if (i < n) a[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
}

__global__ void compact_long_kernel(int n_rows, const int64_t* __restrict__ rowptr, int thr, const int64_t* __restrict__ slot_of,
                                    int32_t* __restrict__ long_rows) {
    const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (r < n_rows && rowptr[r + 1] - rowptr[r] >= thr) long_rows[slot_of[r]] = static_cast<int32_t>(r);
}

// first entry of long row `s` whose column is >= c  (columns ascend within a row)
__device__ __forceinline__ int64_t lower_col(const int32_t* __restrict__ col, int64_t lo, int64_t hi, int64_t c) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// entries of virtual hot row v in X block ub (0 when the piece does not take that block)
__global__ void hot_count_kernel(int n_ub, const int32_t* __restrict__ virt_row, const int32_t* __restrict__ virt_piece,
                                 const int32_t* __restrict__ virt_pieces, const int64_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ col, int64_t* __restrict__ cnt) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<int64_t>(n_ub) * HOT_V) return;
    const int ub = static_cast<int>(i / HOT_V), v = static_cast<int>(i % HOT_V);
    int64_t n = 0;
    const int r = virt_row[v];
    if (r >= 0) {      // piece p of P takes the entries p, p + P, ... of the row's segment in this block
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        const int64_t e0 = lower_col(col, rb, re, static_cast<int64_t>(ub) * HOT_UB);
        const int64_t len = lower_col(col, e0, re, static_cast<int64_t>(ub + 1) * HOT_UB) - e0;
        const int P = virt_pieces[v], q = virt_piece[v];
        n = len > q ? (len - q + P - 1) / P : 0;
    }
    cnt[i] = n;
}

// one wavefront per (block, virtual row): its entries, in column order, to their place in the block-major lists
__global__ __launch_bounds__(256) void hot_fill_kernel(int n_ub, const int32_t* __restrict__ virt_row,
                                                       const int32_t* __restrict__ virt_piece, const int32_t* __restrict__ virt_pieces,
                                                       const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, const int64_t* __restrict__ list_ptr,
                                                       int32_t* __restrict__ meta, float* __restrict__ hval) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (i >= static_cast<int64_t>(n_ub) * HOT_V) return;
    const int64_t o = list_ptr[i], n = list_ptr[i + 1] - o;
    if (n == 0) return;
    const int ub = static_cast<int>(i / HOT_V), v = static_cast<int>(i % HOT_V);
    const int r = virt_row[v];
    const int64_t e0 = lower_col(col, rowptr[r], rowptr[r + 1], static_cast<int64_t>(ub) * HOT_UB);
    const int P = virt_pieces[v], q = virt_piece[v];
    for (int64_t k = lane; k < n; k += 64) {
        meta[o + k] = col[e0 + q + k * P] - ub * HOT_UB;
        hval[o + k] = val[e0 + q + k * P];
    }
}

// tasks per (block, long row) segment, written twice: block-major (the order the tasks run in) and row-major (the order
// the partial rows are stored and added in)
__global__ void count_tasks_kernel(int n_long, int n_blocks, int cblk, const int32_t* __restrict__ long_rows,
                                   const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   int64_t* __restrict__ cnt_block_major, int64_t* __restrict__ cnt_row_major) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<int64_t>(n_long) * n_blocks) return;
    const int s = static_cast<int>(i / n_blocks), b = static_cast<int>(i % n_blocks);
    const int64_t r = long_rows[s], rb = rowptr[r], re = rowptr[r + 1];
    const int64_t e0 = lower_col(col, rb, re, static_cast<int64_t>(b) * cblk);
    const int64_t e1 = lower_col(col, e0, re, static_cast<int64_t>(b + 1) * cblk);
    const int64_t nt = (e1 - e0 + SPMM_TASK - 1) / SPMM_TASK;
    cnt_block_major[static_cast<int64_t>(b) * n_long + s] = nt;
    cnt_row_major[i] = nt;
}

__global__ void fill_tasks_kernel(int n_long, int n_blocks, int cblk, const int32_t* __restrict__ long_rows,
                                  const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                  const int64_t* __restrict__ first_block_major, const int64_t* __restrict__ first_row_major,
                                  Task* __restrict__ tasks) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<int64_t>(n_long) * n_blocks) return;
    const int s = static_cast<int>(i / n_blocks), b = static_cast<int>(i % n_blocks);
    const int64_t r = long_rows[s], rb = rowptr[r], re = rowptr[r + 1];
    const int64_t e0 = lower_col(col, rb, re, static_cast<int64_t>(b) * cblk);
    const int64_t e1 = lower_col(col, e0, re, static_cast<int64_t>(b + 1) * cblk);
    int64_t t = first_block_major[static_cast<int64_t>(b) * n_long + s];
    int64_t p = first_row_major[i];
    for (int64_t e = e0; e < e1; e += SPMM_TASK, ++t, ++p) {
        Task tk;
        tk.beg = e;
        tk.len_slot = static_cast<uint32_t>((e1 - e < SPMM_TASK ? e1 - e : SPMM_TASK) - 1) | (static_cast<uint32_t>(s) << 8);
        tk.part = static_cast<int32_t>(p);
        tasks[t] = tk;
    }
}

__global__ void part_ptr_kernel(int n_long, int n_blocks, const int64_t* __restrict__ first_row_major, int64_t* __restrict__ part_ptr) {
    const int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (s <= n_long) part_ptr[s] = first_row_major[s * n_blocks];
}

}  // namespace

struct skr_spmm_plan {
    int n_rows = 0, n_cols = 0;
    int64_t nnz = 0;
    const int64_t* rowptr = nullptr;     // the caller's CSR (device memory, must outlive the plan)
    const int32_t* col = nullptr;
    const float* val = nullptr;
    int long_thr = SPMM_LONG, cblk = SPMM_CBLK, n_blocks = 0, n_long = 0;
    int64_t n_tasks = 0;
    int32_t* long_rows = nullptr;        // [n_long] row ids, ascending
    int64_t* first_task = nullptr;       // [n_blocks * n_long + 1] block-major
    int64_t* part_ptr = nullptr;         // [n_long + 1] a row's partial-row slots
    int n_win = 1;                       // column windows of the short-row kernel (X beyond the Infinity Cache)
    int64_t* split = nullptr;            // [n_rows * (n_win - 1)]
    Task* tasks = nullptr;               // [n_tasks] block-major
    float* part = nullptr;               // [n_tasks, 64]
    // the densest rows (spmm_hot_kernel)
    int n_hot = 0, n_virt = 0, n_ub = 0, hot_wgs = 0;
    int64_t hot_nnz = 0;
    int32_t* hot_rows = nullptr;         // [n_hot] row ids
    int32_t* hot_vptr = nullptr;         // [n_hot + 1] a row's virtual rows ...
    int32_t* hot_vidx = nullptr;         // [n_virt]    ... as indices 0 .. HOT_V - 1
    int32_t* virt_tab = nullptr;         // [3][HOT_V]: row (-1: unused), piece, pieces
    uint8_t* hot_flag = nullptr;         // [n_long] 1 = the long row is a hot row
    int64_t* hot_list_ptr = nullptr;     // [n_ub * HOT_V + 1]
    int32_t* hot_meta = nullptr;         // [hot_nnz]
    float* hot_val = nullptr;            // [hot_nnz]
    float* hot_part = nullptr;           // [hot_wgs, HOT_V, 64]
    float* hot_vsum = nullptr;           // [HOT_V, 64]
};

namespace {
void free_plan(skr_spmm_plan* p) {
    if (!p) return;
    (void)hipFree(p->long_rows);
    (void)hipFree(p->first_task);
    (void)hipFree(p->part_ptr);
    (void)hipFree(p->tasks);
    (void)hipFree(p->part);
    (void)hipFree(p->split);
    (void)hipFree(p->hot_rows); (void)hipFree(p->hot_vptr); (void)hipFree(p->hot_vidx); (void)hipFree(p->virt_tab);
    (void)hipFree(p->hot_flag); (void)hipFree(p->hot_list_ptr); (void)hipFree(p->hot_meta); (void)hipFree(p->hot_val);
    (void)hipFree(p->hot_part); (void)hipFree(p->hot_vsum);
    delete p;
}
}  // namespace

extern "C" {

int skr_spmm_plan_create(int n_rows, int n_cols, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t nnz,
                         int long_rows_from, skr_spmm_plan** out, void* stream) {
    SKR_REQUIRE(out, "skr_spmm_plan_create: NULL output");
    *out = nullptr;
    SKR_REQUIRE(d_rowptr && d_col && d_val, "skr_spmm_plan_create: NULL argument");
    SKR_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "skr_spmm_plan_create: negative size");
    SKR_REQUIRE(long_rows_from == 0 || long_rows_from >= 2, "skr_spmm_plan_create: long_rows_from must be 0 (default) or >= 2");
    hipStream_t st = skr::as_stream(stream);
    skr_spmm_plan* p = new skr_spmm_plan();
    p->n_rows = n_rows; p->n_cols = n_cols; p->nnz = nnz;
    p->rowptr = d_rowptr; p->col = d_col; p->val = d_val;
    if (long_rows_from) p->long_thr = long_rows_from;
    p->cblk = plan_default_cblk(n_cols);
    if (const char* e = getenv("SKR_SPMM_CBLK")) {       // tuning switch: columns per block (default: plan_default_cblk)
        const int v = atoi(e);
        if (v >= 256) p->cblk = v;
    }
    p->n_blocks = (n_cols + p->cblk - 1) / p->cblk;
    *out = p;
    if (n_rows == 0 || nnz == 0 || p->n_blocks == 0) return SKR_OK;
    // 0. column windows for the short rows (SKR_SPMM_WINDOWS=n; default 1 = off).  The idea: an X of more than ~160 MB does
    //    not stay in the 256 MB Infinity Cache beside the kernel's streams, windows of <= 128 MB would.  Measured on the item
    //    side of the 1 M-user graph (X = 256 MB): 1.388 ms with one window, 1.385 with two, 1.392 with three -- the short
    //    rows' gathers are not HBM-limited -- so the plan does not cut by itself; kept as a switch for other shapes.
    {
        const char* e = getenv("SKR_SPMM_WINDOWS");
        int n_win = e ? atoi(e) : 1;
        if (n_win < 1) n_win = 1;
        if (n_win > 16) n_win = 16;
        p->n_win = n_win;
        if (n_win > 1) {
            const int64_t n_sp = static_cast<int64_t>(n_rows) * (n_win - 1);
            hipError_t e1 = hipMalloc(&p->split, sizeof(int64_t) * n_sp);
            if (e1 != hipSuccess) { free_plan(p); *out = nullptr; return skr::fail(SKR_EHIP, "hipMalloc failed: %s", hipGetErrorString(e1)); }
            const int64_t win_cols = (static_cast<int64_t>(n_cols) + n_win - 1) / n_win;
            hipLaunchKernelGGL(window_split_kernel, dim3(static_cast<unsigned>((n_sp + 255) / 256)), dim3(256), 0, st, n_rows, p->long_thr, n_win,
                               win_cols, d_rowptr, d_col, p->split);
        }
    }
    // 1. the long rows, in ascending order
    int64_t* slot_of = nullptr;
#define PLAN_HIP(call)                                                                                     \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            (void)hipFree(slot_of); (void)hipFree(cnt_rm);                                                 \
            free_plan(p); *out = nullptr;                                                                  \
            return skr::fail(SKR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
        }                                                                                                  \
    } while (0)
    int64_t* cnt_rm = nullptr;
    PLAN_HIP(hipMalloc(&slot_of, sizeof(int64_t) * (static_cast<size_t>(n_rows) + 1)));
    hipLaunchKernelGGL(flag_long_kernel, dim3(static_cast<unsigned>((n_rows + 1 + 255) / 256)), dim3(256), 0, st, n_rows, p->long_thr,
                       d_rowptr, slot_of);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, slot_of, static_cast<int64_t>(n_rows) + 1);
    int64_t n_long = 0;
    PLAN_HIP(hipMemcpyAsync(&n_long, slot_of + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PLAN_HIP(hipStreamSynchronize(st));
    p->n_long = static_cast<int>(n_long);
    if (n_long >= (int64_t{1} << 24)) {
        (void)hipFree(slot_of);
        free_plan(p); *out = nullptr;
        return skr::fail(SKR_EINVAL, "skr_spmm_plan_create: %lld long rows (at most 2^24 - 1)", static_cast<long long>(n_long));
    }
    if (n_long > 0) {
        PLAN_HIP(hipMalloc(&p->long_rows, sizeof(int32_t) * n_long));
        hipLaunchKernelGGL(compact_long_kernel, dim3(static_cast<unsigned>((n_rows + 255) / 256)), dim3(256), 0, st, n_rows, d_rowptr,
                           p->long_thr, slot_of, p->long_rows);
        // 2. tasks per (block, row) segment, both orders, and their prefix sums
        const int64_t n_seg = n_long * p->n_blocks;
        PLAN_HIP(hipMalloc(&p->first_task, sizeof(int64_t) * (n_seg + 1)));
        PLAN_HIP(hipMalloc(&cnt_rm, sizeof(int64_t) * (n_seg + 1)));
        PLAN_HIP(hipMemsetAsync(p->first_task + n_seg, 0, sizeof(int64_t), st));
        PLAN_HIP(hipMemsetAsync(cnt_rm + n_seg, 0, sizeof(int64_t), st));
        const dim3 sgrid(static_cast<unsigned>((n_seg + 255) / 256));
        hipLaunchKernelGGL(count_tasks_kernel, sgrid, dim3(256), 0, st, p->n_long, p->n_blocks, p->cblk, p->long_rows, d_rowptr, d_col,
                           p->first_task, cnt_rm);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, p->first_task, n_seg + 1);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, cnt_rm, n_seg + 1);
        PLAN_HIP(hipMemcpyAsync(&p->n_tasks, p->first_task + n_seg, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PLAN_HIP(hipStreamSynchronize(st));
        if (p->n_tasks >= (int64_t{1} << 31)) {
            (void)hipFree(slot_of); (void)hipFree(cnt_rm);
            free_plan(p); *out = nullptr;
            return skr::fail(SKR_EINVAL, "skr_spmm_plan_create: too many tasks (%lld)", static_cast<long long>(p->n_tasks));
        }
        PLAN_HIP(hipMalloc(&p->tasks, sizeof(Task) * std::max<int64_t>(p->n_tasks, 1)));
        PLAN_HIP(hipMalloc(&p->part, sizeof(float) * D * std::max<int64_t>(p->n_tasks, 1)));
        PLAN_HIP(hipMalloc(&p->part_ptr, sizeof(int64_t) * (n_long + 1)));
        hipLaunchKernelGGL(fill_tasks_kernel, sgrid, dim3(256), 0, st, p->n_long, p->n_blocks, p->cblk, p->long_rows, d_rowptr, d_col,
                           p->first_task, cnt_rm, p->tasks);
        hipLaunchKernelGGL(part_ptr_kernel, dim3(static_cast<unsigned>((n_long + 1 + 255) / 256)), dim3(256), 0, st, p->n_long, p->n_blocks,
                           cnt_rm, p->part_ptr);
        PLAN_HIP(hipStreamSynchronize(st));
        // 3. the densest rows: at least `dens` entries per HOT_UB columns on average (SKR_SPMM_HOT_DENSITY, default 4; 0: none).
        //    Chosen on the host from the long rows' lengths (a few thousand numbers): the densest first, a row far heavier than a
        //    wavefront's fair share cut into pieces, pieces dealt to the 8 wavefronts' 16 slots by load (largest first).
        static const double dens = [] { const char* e = getenv("SKR_SPMM_HOT_DENSITY"); return e ? atof(e) : 4.0; }();
        if (dens > 0.0 && n_cols >= HOT_UB) {
            int64_t* d_deg = nullptr;
            PLAN_HIP(hipMalloc(&d_deg, sizeof(int64_t) * n_long));
            hipLaunchKernelGGL(long_deg_kernel, dim3(static_cast<unsigned>((n_long + 255) / 256)), dim3(256), 0, st, p->n_long, p->long_rows,
                               d_rowptr, d_deg);
            std::vector<int64_t> deg(n_long);
            std::vector<int32_t> lrows(n_long);
            hipError_t e1 = hipMemcpyAsync(deg.data(), d_deg, sizeof(int64_t) * n_long, hipMemcpyDeviceToHost, st);
            hipError_t e2 = hipMemcpyAsync(lrows.data(), p->long_rows, sizeof(int32_t) * n_long, hipMemcpyDeviceToHost, st);
            hipError_t e3 = hipStreamSynchronize(st);
            (void)hipFree(d_deg);
            PLAN_HIP(e1); PLAN_HIP(e2); PLAN_HIP(e3);
            const double thr = dens * static_cast<double>(n_cols) / HOT_UB;
            std::vector<int> cand;
            for (int s_ = 0; s_ < static_cast<int>(n_long); ++s_)
                if (static_cast<double>(deg[s_]) >= thr) cand.push_back(s_);
            std::sort(cand.begin(), cand.end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] > deg[y] : x < y; });
            if (cand.size() > 96) cand.resize(96);           // room for the pieces of the heaviest ones among HOT_V virtual rows
            double total = 0.0;
            for (int s_ : cand) total += static_cast<double>(deg[s_]);
            struct Virt { int slot, piece, pieces; double load; };
            std::vector<Virt> virt;
            std::vector<int> kept;
            for (int s_ : cand) {
                int pieces = static_cast<int>(std::ceil(static_cast<double>(deg[s_]) / (total / 64.0)));
                if (pieces < 1) pieces = 1;
                if (pieces > 8) pieces = 8;
                if (static_cast<int>(virt.size()) + pieces > HOT_V) break;
                for (int q = 0; q < pieces; ++q) virt.push_back(Virt{s_, q, pieces, static_cast<double>(deg[s_]) / pieces});
                kept.push_back(s_);
            }
            if (!virt.empty()) {
                // The four groups of a wavefront run side by side, each on one row of a slot, for as many trips as the longest
                // of the four needs: the virtual rows, sorted by length, are taken four at a time (a CHUNK = rows of similar
                // length), and the chunks are dealt to the 8 wavefronts' 4 slots by load, largest first.
                std::vector<int> order(virt.size());
                for (size_t q = 0; q < virt.size(); ++q) order[q] = static_cast<int>(q);
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return virt[x].load > virt[y].load; });
                double wload[HOT_WAVES] = {0};
                int wcnt[HOT_WAVES] = {0};
                std::vector<int32_t> tab(3 * HOT_V, 0);
                for (int v = 0; v < HOT_V; ++v) { tab[v] = -1; tab[2 * HOT_V + v] = 1; }
                std::vector<int> v_of(virt.size());
                for (size_t c0 = 0; c0 < order.size(); c0 += 4) {
                    int best = -1;
                    for (int w_ = 0; w_ < HOT_WAVES; ++w_)
                        if (wcnt[w_] < HOT_SLOTS && (best < 0 || wload[w_] < wload[best])) best = w_;
                    const int slot = wcnt[best]++;
                    for (size_t c = c0; c < c0 + 4 && c < order.size(); ++c) {
                        const int q = order[c];
                        const int v = best * 16 + slot * 4 + static_cast<int>(c - c0);
                        wload[best] += virt[q].load;
                        v_of[q] = v;
                        tab[v] = lrows[virt[q].slot];
                        tab[HOT_V + v] = virt[q].piece;
                        tab[2 * HOT_V + v] = virt[q].pieces;
                    }
                }
                p->n_hot = static_cast<int>(kept.size());
                p->n_virt = static_cast<int>(virt.size());
                p->n_ub = (n_cols + HOT_UB - 1) / HOT_UB;
                static const int wgs_max = [] { const char* e = getenv("SKR_SPMM_HOT_WGS"); const int v = e ? atoi(e) : HOT_WGS; return v < 1 ? 1 : (v > 2048 ? 2048 : v); }();
                p->hot_wgs = p->n_ub < wgs_max ? p->n_ub : wgs_max;
                std::vector<int32_t> h_rows(p->n_hot), h_vptr(p->n_hot + 1, 0), h_vidx;
                std::vector<uint8_t> h_flag(n_long, 0);
                for (int k_ = 0; k_ < p->n_hot; ++k_) {
                    h_rows[k_] = lrows[kept[k_]];
                    h_flag[kept[k_]] = 1;
                    for (size_t q = 0; q < virt.size(); ++q)
                        if (virt[q].slot == kept[k_]) h_vidx.push_back(v_of[q]);          // pieces in order 0 .. P - 1
                    h_vptr[k_ + 1] = static_cast<int32_t>(h_vidx.size());
                }
                const int64_t n_seg = static_cast<int64_t>(p->n_ub) * HOT_V;
                PLAN_HIP(hipMalloc(&p->hot_rows, sizeof(int32_t) * p->n_hot));
                PLAN_HIP(hipMalloc(&p->hot_vptr, sizeof(int32_t) * (p->n_hot + 1)));
                PLAN_HIP(hipMalloc(&p->hot_vidx, sizeof(int32_t) * p->n_virt));
                PLAN_HIP(hipMalloc(&p->virt_tab, sizeof(int32_t) * 3 * HOT_V));
                PLAN_HIP(hipMalloc(&p->hot_flag, n_long));
                PLAN_HIP(hipMalloc(&p->hot_list_ptr, sizeof(int64_t) * (n_seg + 1)));
                PLAN_HIP(hipMalloc(&p->hot_part, sizeof(float) * static_cast<size_t>(p->hot_wgs) * HOT_V * D));
                PLAN_HIP(hipMalloc(&p->hot_vsum, sizeof(float) * HOT_V * D));
                PLAN_HIP(hipMemcpyAsync(p->hot_rows, h_rows.data(), sizeof(int32_t) * p->n_hot, hipMemcpyHostToDevice, st));
                PLAN_HIP(hipMemcpyAsync(p->hot_vptr, h_vptr.data(), sizeof(int32_t) * (p->n_hot + 1), hipMemcpyHostToDevice, st));
                PLAN_HIP(hipMemcpyAsync(p->hot_vidx, h_vidx.data(), sizeof(int32_t) * p->n_virt, hipMemcpyHostToDevice, st));
                PLAN_HIP(hipMemcpyAsync(p->virt_tab, tab.data(), sizeof(int32_t) * 3 * HOT_V, hipMemcpyHostToDevice, st));
                PLAN_HIP(hipMemcpyAsync(p->hot_flag, h_flag.data(), n_long, hipMemcpyHostToDevice, st));
                PLAN_HIP(hipMemsetAsync(p->hot_list_ptr + n_seg, 0, sizeof(int64_t), st));
                hipLaunchKernelGGL(hot_count_kernel, dim3(static_cast<unsigned>((n_seg + 255) / 256)), dim3(256), 0, st, p->n_ub, p->virt_tab,
                                   p->virt_tab + HOT_V, p->virt_tab + 2 * HOT_V, d_rowptr, d_col, p->hot_list_ptr);
                hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, p->hot_list_ptr, n_seg + 1);
                PLAN_HIP(hipMemcpyAsync(&p->hot_nnz, p->hot_list_ptr + n_seg, sizeof(int64_t), hipMemcpyDeviceToHost, st));
                PLAN_HIP(hipStreamSynchronize(st));                       // (the host vectors above are done with, too)
                PLAN_HIP(hipMalloc(&p->hot_meta, sizeof(int32_t) * std::max<int64_t>(p->hot_nnz, 1)));
                PLAN_HIP(hipMalloc(&p->hot_val, sizeof(float) * std::max<int64_t>(p->hot_nnz, 1)));
                hipLaunchKernelGGL(hot_fill_kernel, dim3(static_cast<unsigned>((n_seg + 3) / 4)), dim3(256), 0, st, p->n_ub, p->virt_tab,
                                   p->virt_tab + HOT_V, p->virt_tab + 2 * HOT_V, d_rowptr, d_col, d_val, p->hot_list_ptr, p->hot_meta,
                                   p->hot_val);
                PLAN_HIP(hipStreamSynchronize(st));
            }
        }
    }
    PLAN_HIP(hipGetLastError());
    (void)hipFree(slot_of);
    (void)hipFree(cnt_rm);
#undef PLAN_HIP
    return SKR_OK;
}

int skr_spmm_plan_destroy(skr_spmm_plan* plan) {
    free_plan(plan);
    return SKR_OK;
}

int skr_spmm_plan_info(const skr_spmm_plan* plan, int64_t* h_info4) {
    SKR_REQUIRE(plan && h_info4, "skr_spmm_plan_info: NULL argument");
    h_info4[0] = plan->n_long + (static_cast<int64_t>(plan->n_hot) << 32);      // hot rows (LDS-streamed) in the high half
    h_info4[1] = plan->n_tasks;
    h_info4[2] = plan->n_blocks;
    h_info4[3] = plan->long_thr + (static_cast<int64_t>(plan->n_win) << 32);     // column windows in the high half
    return SKR_OK;
}

int skr_spmm_plan_run_ex(const skr_spmm_plan* plan, const float* d_X, int dim, const skr_spmm_epilogue* epi, const uint8_t* d_row_mask,
                         const uint8_t* d_col_mask, void* stream) {
    SKR_REQUIRE(plan && d_X && epi, "skr_spmm_plan_run: NULL argument");
    SKR_REQUIRE(dim == D, "skr_spmm_plan_run: dim must be 64 (got %d)", dim);
    const skr_spmm_epilogue ep = *epi;
    const int ld = ep.ld ? ep.ld : D;
    SKR_REQUIRE(ld >= D && ld % 4 == 0, "skr_spmm_plan_run: the row stride must be a multiple of 4 floats and at least 64 (got %d)", ld);
    SKR_REQUIRE(ep.mode == SKR_EPI_PLAIN || ld == D, "skr_spmm_plan_run: the refinements need whole 64-float rows (row stride 64)");
    SKR_REQUIRE(ep.mode == SKR_EPI_PLAIN || ep.mode == SKR_EPI_REFINE_FWD || ep.mode == SKR_EPI_REFINE_BWD, "skr_spmm_plan_run: unknown epilogue mode %d", ep.mode);
    SKR_REQUIRE(ep.Y != d_X, "skr_spmm_plan_run: in-place propagation is not supported");
    SKR_REQUIRE(ep.mode != SKR_EPI_PLAIN || ep.Y || ep.accum, "skr_spmm_plan_run: the plain epilogue needs Y or accum");
    SKR_REQUIRE(ep.mode != SKR_EPI_REFINE_FWD || (ep.E && ep.w && ep.Z && ep.Z != d_X), "skr_spmm_plan_run: the forward refinement needs E, w and Z");
    SKR_REQUIRE(ep.mode != SKR_EPI_REFINE_BWD || (ep.E && ep.w && ep.rawY && ep.dE && ep.Y), "skr_spmm_plan_run: the backward refinement needs E, w, rawY, dE and Y");
    SKR_REQUIRE(plan->n_win == 1 || ep.Y, "skr_spmm_plan_run: column windows need Y");
    if (plan->n_rows == 0) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    int64_t wgs = (static_cast<int64_t>(plan->n_rows) + ROW_WAVES - 1) / ROW_WAVES;
    // tuning switches: SKR_SPMM_ROWS_WGS (workgroups of the short-row kernel at most), SKR_SPMM_TASK_WGS (per XCD and task launch)
    static const int rows_cap = [] { const char* e = getenv("SKR_SPMM_ROWS_WGS"); const int v = e ? atoi(e) : ROWS_WGS_MAX; return v < 256 ? 256 : v; }();
    static const int task_cap = [] { const char* e = getenv("SKR_SPMM_TASK_WGS"); const int v = e ? atoi(e) : BLK_WGS_PER_XCD; return v < 8 ? 8 : v; }();
    if (wgs > rows_cap) wgs = rows_cap;
    // a task launch covers 8 blocks; no more workgroups than ~1.25 x the average block has tasks for (a matrix with a handful
    // of long rows launches a handful of workgroups)
    int64_t task_wgs = plan->n_blocks > 0 ? (5 * plan->n_tasks / (4 * static_cast<int64_t>(plan->n_blocks)) + ROW_WAVES - 1) / ROW_WAVES : 1;
    task_wgs = std::min<int64_t>(std::max<int64_t>(task_wgs, 8), task_cap);
    const dim3 rgrid(static_cast<unsigned>(wgs)), blk(ROW_WAVES * 64), tgrid(static_cast<unsigned>(8 * task_wgs));
    for (int w = 0; w < plan->n_win; ++w) {
        if (d_col_mask)
            hipLaunchKernelGGL(spmm_rows_kernel<true>, rgrid, blk, 0, st, plan->n_rows, plan->long_thr, plan->rowptr, plan->col, plan->val,
                               d_X, ep, d_row_mask, d_col_mask, plan->split, plan->n_win, w);
        else
            hipLaunchKernelGGL(spmm_rows_kernel<false>, rgrid, blk, 0, st, plan->n_rows, plan->long_thr, plan->rowptr, plan->col, plan->val,
                               d_X, ep, d_row_mask, d_col_mask, plan->split, plan->n_win, w);
    }
    SKR_LAUNCH_CHECK();
    // the densest rows through LDS -- unless the call says that most of X is zero (col_mask: the first backward hop), where
    // streaming all of X would be the waste; SKR_SPMM_HOT=0 keeps them on the task path
    static const bool hot_on = [] { const char* e = getenv("SKR_SPMM_HOT"); return !(e && atoi(e) == 0); }();
    const bool hot = hot_on && plan->n_hot > 0 && !d_col_mask;
    const uint8_t* hot_flag = hot ? plan->hot_flag : nullptr;
    if (hot) {
        static const int pf = [] { const char* e = getenv("SKR_SPMM_HOT_PF"); return e ? atoi(e) : 1; }();
        if (pf >= 2)
            hipLaunchKernelGGL(spmm_hot_kernel<2>, dim3(plan->hot_wgs), dim3(HOT_WAVES * 64), 0, st, plan->hot_list_ptr, plan->hot_meta,
                               plan->hot_val, d_X, ld >> 2, plan->n_cols, plan->n_ub, plan->hot_part);
        else
            hipLaunchKernelGGL(spmm_hot_kernel<1>, dim3(plan->hot_wgs), dim3(HOT_WAVES * 64), 0, st, plan->hot_list_ptr, plan->hot_meta,
                               plan->hot_val, d_X, ld >> 2, plan->n_cols, plan->n_ub, plan->hot_part);
        hipLaunchKernelGGL(spmm_hot_vsum_kernel, dim3(HOT_V), dim3(1024), 0, st, plan->virt_tab, plan->hot_part, plan->hot_wgs,
                           plan->hot_vsum);
        hipLaunchKernelGGL(spmm_hot_reduce_kernel, dim3((plan->n_hot + 3) / 4), dim3(256), 0, st, plan->n_hot, plan->hot_rows,
                           plan->hot_vptr, plan->hot_vidx, plan->hot_vsum, ep, d_row_mask);
        SKR_LAUNCH_CHECK();
    }
    if (plan->n_long > 0) {
        if (plan->n_tasks > 0) {
            const int groups = (plan->n_blocks + 7) / 8;
            // (tried: the next task's descriptor and entries requested before this task's gathers -- 1.389 vs 1.394 ms on the
            //  item side: the gather path, not the task's fixed cost, is the limit.  profiles/r02_spmm_lab.txt, run 7)
            // SKR_SPMM_TASK_SPAN: groups of 8 blocks per launch (experiment; default 1 = a launch per group)
            static const int span_cfg = [] { const char* e = getenv("SKR_SPMM_TASK_SPAN"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
            for (int g = 0; g < groups; g += span_cfg) {
                const int span = std::min(span_cfg, groups - g);
                if (d_col_mask)
                    hipLaunchKernelGGL(spmm_tasks_kernel<true>, tgrid, blk, 0, st, plan->first_task, plan->n_long, plan->n_blocks, g, span, plan->tasks,
                                       plan->col, plan->val, d_X, plan->part, plan->long_rows, d_row_mask, d_col_mask, ld >> 2, hot_flag);
                else
                    hipLaunchKernelGGL(spmm_tasks_kernel<false>, tgrid, blk, 0, st, plan->first_task, plan->n_long, plan->n_blocks, g, span, plan->tasks,
                                       plan->col, plan->val, d_X, plan->part, plan->long_rows, d_row_mask, d_col_mask, ld >> 2, hot_flag);
            }
            SKR_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(spmm_reduce_kernel, dim3(plan->n_long), dim3(256), 0, st, plan->long_rows, plan->part_ptr, plan->part, ep,
                           d_row_mask, hot_flag);
        SKR_LAUNCH_CHECK();
    }
    return SKR_OK;
}

int skr_spmm_plan_run_masked(const skr_spmm_plan* plan, const float* d_X, int dim, const float* d_addend, float* d_Y, float* d_accum,
                             float accum_scale, const uint8_t* d_row_mask, const uint8_t* d_col_mask, void* stream) {
    SKR_REQUIRE(d_Y, "skr_spmm_plan_run: NULL argument");
    skr_spmm_epilogue ep = {};
    ep.mode = SKR_EPI_PLAIN;
    ep.addend = d_addend; ep.Y = d_Y; ep.accum = d_accum; ep.accum_scale = accum_scale;
    return skr_spmm_plan_run_ex(plan, d_X, dim, &ep, d_row_mask, d_col_mask, stream);
}

int skr_spmm_plan_run(const skr_spmm_plan* plan, const float* d_X, int dim, const float* d_addend, float* d_Y, float* d_accum,
                      float accum_scale, void* stream) {
    return skr_spmm_plan_run_masked(plan, d_X, dim, d_addend, d_Y, d_accum, accum_scale, nullptr, nullptr, stream);
}

// Y[col] += val * X[r] for every entry (col, val) of every MARKED row r of a CSR: the product with the TRANSPOSE of the
// matrix where X is zero outside the marked rows.  One wavefront looks at 64 rows' marks, then walks the marked rows'
// entries: a 256-byte row atomic each.
__global__ __launch_bounds__(256) void spmm_scatter_marked_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                  const uint8_t* __restrict__ row_mask, const float* __restrict__ X,
                                                                  float* __restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = (blockIdx.x * 4ll + (threadIdx.x >> 6)) * 64;
    if (r0 >= n_rows) return;
    unsigned long long b = __ballot(r0 + lane < n_rows && row_mask[r0 + lane] != 0);
    while (b) {
        const int64_t r = r0 + __ffsll(static_cast<long long>(b)) - 1;
        b &= b - 1;
        const float x = X[r * D + lane];
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        for (int64_t e = rb; e < re; e += 64) {
            const int m = static_cast<int>(re - e < 64 ? re - e : 64);
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) { cl = col[e + lane]; vl = val[e + lane]; }
            for (int k = 0; k < m; ++k)
                atomicAdd(&Y[static_cast<int64_t>(__shfl(cl, k)) * D + lane], __shfl(vl, k) * x);
        }
    }
}

// d_mask[ids[k]] = 1 for every k (ids < 0 are skipped); the caller clears the mask (hipMemsetAsync) before
__global__ void mark_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int64_t offset, uint8_t* __restrict__ mask) {
    const int64_t k = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (k < n && ids[k] >= 0) mask[offset + ids[k]] = 1;
}

int skr_csr_scatter_marked_rows(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val,
                                const uint8_t* d_row_mask, const float* d_X, int dim, float* d_Y, void* stream) {
    SKR_REQUIRE(d_rowptr && d_col && d_val && d_row_mask && d_X && d_Y, "skr_csr_scatter_marked_rows: NULL argument");
    SKR_REQUIRE(dim == D, "skr_csr_scatter_marked_rows: dim must be 64 (got %d)", dim);
    SKR_REQUIRE(n_rows >= 0, "skr_csr_scatter_marked_rows: negative size");
    if (n_rows == 0) return SKR_OK;
    const int64_t waves = (static_cast<int64_t>(n_rows) + 63) / 64;
    hipLaunchKernelGGL(spmm_scatter_marked_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, skr::as_stream(stream), n_rows,
                       d_rowptr, d_col, d_val, d_row_mask, d_X, d_Y);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_mark_ids(const int32_t* d_ids, int64_t n, int64_t offset, uint8_t* d_mask, void* stream) {
    SKR_REQUIRE(n >= 0 && offset >= 0, "skr_mark_ids: negative size");
    if (n == 0) return SKR_OK;
    SKR_REQUIRE(d_ids && d_mask, "skr_mark_ids: NULL argument");
    hipLaunchKernelGGL(mark_ids_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, skr::as_stream(stream), d_ids, n, offset,
                       d_mask);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

}  // extern "C"
