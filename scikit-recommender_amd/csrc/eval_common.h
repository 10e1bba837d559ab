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

// 1.0/log2(i+2) for i < SKR_MAX_TOPK, evaluated on the HOST with the same libm the reference's
// C++ uses (metric.h:76,80), so that the device never depends on a device log2.
struct InvLog2Table {
    double v[SKR_MAX_TOPK];
};
const InvLog2Table& inv_log2_table();

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
