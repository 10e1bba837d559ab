// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// A thin extern "C" shim of OUR OWN around the reference's header-only C++
// (skrec/utils/py/cython/include/{randint,evaluate,metric,thread_pool}.h).  The headers are
// compiled WHERE THEY LIE under /root/reference (see oracle/Makefile: -I$(REF_INC));
// nothing of the reference is copied into this repository.  The result,
// oracle/_ref/libskrec_ref.so, is git-ignored, travels to the GPU box with gpurun,
// and serves two purposes only:
//   1. pinning oracle/skr_oracle.c (our C restatement) bit-for-bit, and
//   2. the "reference" CPU baseline leg of bench.py.
//
// The shim restates only what the reference's Cython glue does around the C++ calls:
//   pyx_random.pyx:59-72      ndarray exclusion -> std::unordered_set<int>, call c_randint_choice
//   data_iterator.py:81-94    one c_randint_choice call per user, ascending user id
//   pyx_eval_matrix.pyx:22-37 list[ndarray] -> vector<unordered_set<int>>, call cpp_evaluate_matrix
#include <cstdint>
#include <vector>
#include <unordered_set>

#include "randint.h"   // c_randint_choice, c_batch_randint_choice, global _gen   (randint.h:20,75,90)
#include "evaluate.h"  // cpp_evaluate_matrix, eval_one_user, metric_dict          (evaluate.h:24,57)

extern "C" {

// Re-seed the reference's process-global generator (randint.h:20 seeds it with 2020 at load
// time; the reference itself offers no API for this -- tests use it to replay "fresh process").
void ref_reseed(unsigned int seed) { _gen.seed(seed); }

// pyx_randint_choice's native part (pyx_random.pyx:59-72).
int ref_randint_choice(int high, int size, int replace, const float* prob,
                       const int* exclusion, int n_exclusion, int has_exclusion, int* result) {
    if (has_exclusion) {
        int_set excl;
        for (int i = 0; i < n_exclusion; ++i) excl.insert(exclusion[i]);
        return c_randint_choice(high, size, replace != 0, prob, &excl, result);
    }
    return c_randint_choice(high, size, replace != 0, prob, nullptr, result);
}

// pyx_batch_randint_choice's native part (pyx_random.pyx:124-146), exclusion given as CSR.
int ref_batch_randint_choice(int high, const int* sizes, int batch_num, int replace, const float* prob,
                             const int64_t* excl_rowptr, const int* excl_items, int has_exclusion,
                             int n_threads, int* result) {
    std::vector<int_set> excl;
    if (has_exclusion) {
        excl.resize(batch_num);
        for (int b = 0; b < batch_num; ++b)
            for (int64_t p = excl_rowptr[b]; p < excl_rowptr[b + 1]; ++p) excl[b].insert(excl_items[p]);
    }
    return c_batch_randint_choice(high, sizes, batch_num, replace != 0, prob,
                                  has_exclusion ? excl.data() : nullptr, n_threads, result);
}

// _sampling_negative_items (data_iterator.py:81-94) at the native level: for every user with
// positives, in ascending row order, one c_randint_choice(num_items, n_pos*num_neg, exclusion=train
// positives).  Users with an empty row are skipped (they never appear in user_n_pos).
int ref_sample_epoch(int num_items, int n_users, const int64_t* rowptr, const int* pos_items,
                     int num_neg, int* out) {
    int64_t off = 0;
    for (int u = 0; u < n_users; ++u) {
        const int64_t beg = rowptr[u], end = rowptr[u + 1];
        if (end == beg) continue;
        int_set excl;  // Cython builds a fresh set per call (pyx_random.pyx:62)
        for (int64_t p = beg; p < end; ++p) excl.insert(pos_items[p]);
        const int size = static_cast<int>((end - beg) * num_neg);
        c_randint_choice(num_items, size, true, nullptr, &excl, out + off);
        off += size;
    }
    return 0;
}

// eval_score_matrix's native part (pyx_eval_matrix.pyx:26-35).
void ref_evaluate_matrix(float* ratings, int n_users, int rating_len,
                         const int64_t* test_rowptr, const int* test_items,
                         const int* metric, int n_metric, int top_k, int thread_num, float* out) {
    std::vector<std::unordered_set<int>> truth(n_users);
    for (int u = 0; u < n_users; ++u)
        for (int64_t p = test_rowptr[u]; p < test_rowptr[u + 1]; ++p) truth[u].insert(test_items[p]);
    std::vector<int> m(metric, metric + n_metric);
    cpp_evaluate_matrix(ratings, rating_len, truth, m, top_k, thread_num, out);
}

}  // extern "C"
