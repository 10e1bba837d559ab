// eval_common.h -- pieces shared by the evaluation kernels (E rows):
//   * the five ranking metrics of utils/py/cython/include/metric.h:19-109, op for op
//   * an in-LDS bitonic sort on 64-bit rank keys (score desc, id asc)
#pragma once
#include "skr_common.h"

namespace skr {

struct MetricArgs {
    int n_metric;
    int ids[8];  // metric ids 1..5 (evaluator.py:57), at most 8 per call
};

// 1.0/log2(i+2) for i < SKR_MAX_TOPK_SCORES, evaluated on the HOST with the same libm the reference's
// C++ uses (metric.h:76,80), so that the device never depends on a device log2 (eval_select.hip keeps the table in HBM).
const double* inv_log2_table_device();

// One metric for one user; `rank` = arg-top-K list (any address space), truth sorted ascending.
// Accumulators are `float`, the `+= 1.0/log2(i+2)` and `1.0/(i+1)` terms are double and rounded to
// float at every step -- exactly the implicit conversions of metric.h.
template <typename RankP>
__device__ inline void metric_row(int metric_id, RankP rank, int k, const int32_t* __restrict__ truth, int64_t tb,
                                  int64_t te, const double* __restrict__ invlog2, float* __restrict__ out) {
    const int nt = static_cast<int>(te - tb);
    const int truth_len = nt > 1 ? nt : 1;
    switch (metric_id) {
        case SKR_PRECISION: {
            float hits = 0.0f;
            for (int i = 0; i < k; ++i) {
                if (contains_sorted(truth, tb, te, rank[i])) hits = static_cast<float>(static_cast<double>(hits) + 1.0);
                out[i] = hits / static_cast<float>(static_cast<unsigned>(i + 1));
            }
        } break;
        case SKR_RECALL: {
            float hits = 0.0f;
            const float tl = static_cast<float>(truth_len);
            for (int i = 0; i < k; ++i) {
                if (contains_sorted(truth, tb, te, rank[i])) hits = static_cast<float>(static_cast<double>(hits) + 1.0);
                out[i] = hits / tl;
            }
        } break;
        case SKR_MAP: {
            float hits = 0.0f, sum_pre = 0.0f;
            for (int i = 0; i < k; ++i) {
                if (contains_sorted(truth, tb, te, rank[i])) {
                    hits = static_cast<float>(static_cast<double>(hits) + 1.0);
                    sum_pre += hits / static_cast<float>(static_cast<unsigned>(i + 1));
                }
                const float den = static_cast<float>(truth_len < i + 1 ? truth_len : i + 1);
                out[i] = sum_pre / den;
            }
        } break;
        case SKR_NDCG: {
            float idcg = 0.0f, dcg = 0.0f;
            for (int i = 0; i < k; ++i) {
                if (contains_sorted(truth, tb, te, rank[i])) dcg = static_cast<float>(static_cast<double>(dcg) + invlog2[i]);
                if (i < truth_len) idcg = static_cast<float>(static_cast<double>(idcg) + invlog2[i]);
                out[i] = dcg / idcg;
            }
        } break;
        case SKR_MRR: {
            float rr = 0.0f;
            bool found = false;
            for (int i = 0; i < k; ++i) {
                if (!found && contains_sorted(truth, tb, te, rank[i])) {
                    rr = static_cast<float>(1.0 / static_cast<double>(static_cast<unsigned>(i + 1)));
                    found = true;
                }
                out[i] = rr;
            }
        } break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------
// The reference's order among EQUAL scores.  eval_one_user (evaluate.h:24-45) ranks with
// std::partial_sort_copy(index.begin(), index.end(), topk.begin(), topk.begin() + min(2K, I), ratings[a] > ratings[b]):
// libstdc++ copies the first 2K indices, make_heap, then for every later index x with
// ratings[x] > ratings[heap top] replaces the top (__adjust_heap), finally sort_heap.  The order of equal
// scores is whatever these heap moves produce, so rows with ties inside their top-(K+1) are re-ranked by
// running exactly that algorithm (published libstdc++ semantics: bits/stl_algo.h __partial_sort_copy,
// bits/stl_heap.h __adjust_heap / __push_heap / __make_heap / __sort_heap).  `val` caches the rating of
// every heap slot.  One lane executes these; heaps of at most 2 * SKR_MAX_TOPK entries live in LDS.
// ------------------------------------------------------------------------------------------------
struct RefHeap {
    int* id;      // [len]
    float* val;   // [len]
    int len;
};

// comp(a, b) = ratings[a] > ratings[b]: the heap's top is the SMALLEST rating kept
__device__ inline void ref_push_heap(RefHeap& h, int hole, int top, int v_id, float v_val) {
    int parent = (hole - 1) / 2;
    while (hole > top && h.val[parent] > v_val) {
        h.id[hole] = h.id[parent];
        h.val[hole] = h.val[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h.id[hole] = v_id;
    h.val[hole] = v_val;
}

__device__ inline void ref_adjust_heap(RefHeap& h, int hole, int len, int v_id, float v_val) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (h.val[child] > h.val[child - 1]) child--;
        h.id[hole] = h.id[child];
        h.val[hole] = h.val[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.id[hole] = h.id[child - 1];
        h.val[hole] = h.val[child - 1];
        hole = child - 1;
    }
    ref_push_heap(h, hole, top, v_id, v_val);
}

__device__ inline void ref_make_heap(RefHeap& h) {
    if (h.len < 2) return;
    int parent = (h.len - 2) / 2;
    for (;;) {
        const int v_id = h.id[parent];
        const float v_val = h.val[parent];
        ref_adjust_heap(h, parent, h.len, v_id, v_val);
        if (parent == 0) return;
        parent--;
    }
}

__device__ inline void ref_sort_heap(RefHeap& h) {
    int last = h.len;
    while (last > 1) {
        --last;
        const int v_id = h.id[last];
        const float v_val = h.val[last];
        h.id[last] = h.id[0];
        h.val[last] = h.val[0];
        ref_adjust_heap(h, 0, last, v_id, v_val);
    }
}

// Descending bitonic sort of N (power of two) 64-bit keys in LDS by a workgroup of T threads.
template <int N, int T>
__device__ inline void bitonic_sort_desc_lds(uint64_t* keys) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += T) {
                // pair (i, i^j) with i having bit j clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool desc = ((i & k) == 0);  // this run sorts descending
                const bool swap = desc ? (a < b) : (a > b);
                if (swap) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace skr
