/* oracle/skr_oracle.c -- TEST INFRASTRUCTURE (parity oracle).  Never linked into, imported by or
 * called from the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * A plain-C, single-threaded CPU restatement of the native half of scikit-recommender's hot path:
 *
 *   S4  negative sampling      skrec/utils/py/cython/include/randint.h:20-88
 *   S2  per-user epoch loop    skrec/io/data_iterator.py:81-94
 *   E5  arg-top-K (2K rule)    skrec/utils/py/cython/include/evaluate.h:24-54
 *   E6  ranking metrics        skrec/utils/py/cython/include/metric.h:19-118
 *
 * The arithmetic of those files lives partly in a third-party dependency that is not under
 * /root/reference: libstdc++ (GCC 11.4, the compiler the survey built the reference with).  The
 * pieces restated from its published algorithms are
 *   std::mt19937                          (bits/random.tcc  mersenne_twister_engine)
 *   std::uniform_int_distribution<int>    (bits/uniform_int_dist.h, Lemire "nearly divisionless"
 *                                          path taken when the engine range is exactly 2^32)
 *   std::discrete_distribution<int>       (bits/random.tcc: normalise, partial_sum, lower_bound on
 *                                          generate_canonical<double,53>)
 *   std::partial_sort_copy                (bits/stl_algo.h + bits/stl_heap.h: make_heap,
 *                                          __adjust_heap, __push_heap, sort_heap)
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares every function below bit-for-bit
 * with oracle/_ref/libskrec_ref.so (the reference's own headers compiled where they lie), and
 * tests/golden/ holds vectors produced by the reference itself (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* MT19937 (std::mt19937): randint.h:20 `std::mt19937 _gen(2020);`                             */
/* ------------------------------------------------------------------------------------------ */
#define ORC_MT_N 624
#define ORC_MT_M 397

typedef struct orc_sampler {
    uint32_t mt[ORC_MT_N];
    int p; /* next word to hand out; ORC_MT_N means "twist first" */
    uint64_t draws; /* raw 32-bit words consumed so far (diagnostics) */
} orc_sampler;

static void mt_seed(orc_sampler* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < ORC_MT_N; ++i) {
        uint32_t x = g->mt[i - 1];
        g->mt[i] = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
    }
    g->p = ORC_MT_N;
    g->draws = 0;
}

static void mt_twist(orc_sampler* g) {
    uint32_t* mt = g->mt;
    for (int k = 0; k < ORC_MT_N; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % ORC_MT_N] & 0x7fffffffu);
        mt[k] = mt[(k + ORC_MT_M) % ORC_MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->p = 0;
}

static inline uint32_t mt_next(orc_sampler* g) {
    if (g->p >= ORC_MT_N) mt_twist(g);
    uint32_t z = g->mt[g->p++];
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    g->draws++;
    return z;
}

orc_sampler* orc_sampler_new(uint32_t seed) {
    orc_sampler* g = (orc_sampler*)malloc(sizeof(orc_sampler));
    if (g) mt_seed(g, seed);
    return g;
}
void orc_sampler_free(orc_sampler* g) { free(g); }
void orc_sampler_reseed(orc_sampler* g, uint32_t seed) { mt_seed(g, seed); }
uint64_t orc_sampler_draws(const orc_sampler* g) { return g->draws; }
uint32_t orc_sampler_next_u32(orc_sampler* g) { return mt_next(g); }
/* Export / import the raw state (624 untempered words + position) so the HIP exact-stream
 * sampler can be started from, and compared with, any point of the stream. */
void orc_sampler_get_state(const orc_sampler* g, uint32_t* words624, int* pos) {
    memcpy(words624, g->mt, sizeof(g->mt));
    *pos = g->p;
}
void orc_sampler_set_state(orc_sampler* g, const uint32_t* words624, int pos) {
    memcpy(g->mt, words624, sizeof(g->mt));
    g->p = pos;
}

/* std::uniform_int_distribution<int>(0, high-1)(mt19937): engine range 2^32-1 > high-1, and the
 * engine range is exactly 32 bits, so libstdc++-11 takes _S_nd<uint64_t> (Lemire). */
static inline int uniform_below(orc_sampler* g, uint32_t range) {
    uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (int)(product >> 32);
}

/* std::generate_canonical<double, 53>(mt19937): two 32-bit words, low word first. */
static inline double canonical53(orc_sampler* g) {
    const double R = 4294967296.0;
    double sum = (double)mt_next(g);
    sum += (double)mt_next(g) * R;
    double ret = sum / (R * R);
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* sorted-array membership (the reference uses unordered_set<int>::count, randint.h:43) */
static int cmp_int(const void* a, const void* b) {
    int x = *(const int*)a, y = *(const int*)b;
    return (x > y) - (x < y);
}
static inline int contains_sorted(const int* a, int64_t n, int v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == v;
}

typedef struct { /* growable sorted set for replace=False (randint.h:53-70) */
    int* v; int64_t n, cap;
} grow_set;
static void gs_insert(grow_set* s, int x) {
    int64_t lo = 0, hi = s->n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (s->v[mid] < x) lo = mid + 1; else hi = mid; }
    if (lo < s->n && s->v[lo] == x) return;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->v = (int*)realloc(s->v, (size_t)s->cap * sizeof(int)); }
    memmove(s->v + lo + 1, s->v + lo, (size_t)(s->n - lo) * sizeof(int));
    s->v[lo] = x; s->n++;
}

/* c_randint_choice (randint.h:75-88) + _random_int (randint.h:23-72).
 * exclusion: any order, n_exclusion entries, ignored when has_exclusion == 0.
 * prob: NULL -> uniform (randint.h:84), else discrete_distribution over prob[0..high) (:79).
 * Returns 0 like the reference.  Like the reference it spins forever on impossible requests;
 * callers validate first (pyx_random.pyx:34-54). */
int orc_randint_choice(orc_sampler* g, int high, int size, int replace, const float* prob,
                       const int* exclusion, int n_exclusion, int has_exclusion, int* result) {
    double* cp = NULL; /* cumulative probabilities of discrete_distribution */
    int ncp = 0;
    if (prob) {
        if (high >= 2) {
            double sum = 0.0;
            for (int i = 0; i < high; ++i) sum += (double)prob[i];
            cp = (double*)malloc((size_t)high * sizeof(double));
            double acc = 0.0;
            for (int i = 0; i < high; ++i) { acc += (double)prob[i] / sum; cp[i] = acc; }
            cp[high - 1] = 1.0;
            ncp = high;
        }
    }
    grow_set set = {NULL, 0, 0};
    if (has_exclusion && n_exclusion > 0) {
        set.v = (int*)malloc((size_t)n_exclusion * sizeof(int));
        memcpy(set.v, exclusion, (size_t)n_exclusion * sizeof(int));
        qsort(set.v, (size_t)n_exclusion, sizeof(int), cmp_int);
        int64_t w = 0; /* dedupe */
        for (int64_t r = 0; r < n_exclusion; ++r) if (w == 0 || set.v[w - 1] != set.v[r]) set.v[w++] = set.v[r];
        set.n = w; set.cap = n_exclusion;
    }
    int i = 0;
    while (i < size) {
        int s;
        if (prob) {
            if (ncp == 0) s = 0;
            else {
                double p = canonical53(g);
                int lo = 0, hi = ncp; /* std::lower_bound */
                while (lo < hi) { int mid = (lo + hi) >> 1; if (cp[mid] < p) lo = mid + 1; else hi = mid; }
                s = lo;
            }
        } else {
            s = uniform_below(g, (uint32_t)high);
        }
        if (replace) {
            if (!has_exclusion || !contains_sorted(set.v, set.n, s)) result[i++] = s;
        } else {
            if (!contains_sorted(set.v, set.n, s)) { result[i++] = s; gs_insert(&set, s); }
        }
    }
    free(set.v);
    free(cp);
    return 0;
}

/* _sampling_negative_items (data_iterator.py:81-94) over a CSR of train positives:
 * users ascending, rows with no positives skipped, n_pos*num_neg draws per user, one global stream
 * that is never reset (so epoch 2 continues where epoch 1 stopped). */
int orc_sample_epoch(orc_sampler* g, int num_items, int n_users, const int64_t* rowptr,
                     const int* pos_items, int num_neg, int* out) {
    int64_t off = 0;
    for (int u = 0; u < n_users; ++u) {
        int64_t beg = rowptr[u], end = rowptr[u + 1];
        if (end == beg) continue;
        int size = (int)((end - beg) * num_neg);
        orc_randint_choice(g, num_items, size, 1, NULL, pos_items + beg, (int)(end - beg), 1, out + off);
        off += size;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* E5: arg-top-K with the heap order of libstdc++'s partial_sort_copy (evaluate.h:27-45)       */
/* ------------------------------------------------------------------------------------------ */
/* comp(a, b) := ratings[a] > ratings[b]   (evaluate.h:43) */
#define COMP(a, b) (r[(a)] > r[(b)])

static void push_heap_(int* first, int64_t hole, int64_t top, int value, const float* r) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && COMP(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void adjust_heap_(int* first, int64_t hole, int64_t len, int value, const float* r) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (COMP(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, r);
}
/* ids_out must hold min(2*top_k, rating_len) ints; the first top_k are the reference's list. */
int orc_partial_sort_ids(const float* r, int rating_len, int top_k, int* ids_out) {
    int64_t sort_len = (int64_t)top_k * 2 < rating_len ? (int64_t)top_k * 2 : rating_len; /* evaluate.h:39 */
    if (sort_len <= 0) return 0;
    int64_t i = 0;
    for (; i < sort_len; ++i) ids_out[i] = (int)i;
    if (sort_len >= 2) { /* make_heap */
        int64_t parent = (sort_len - 2) / 2;
        for (;;) {
            int v = ids_out[parent];
            adjust_heap_(ids_out, parent, sort_len, v, r);
            if (parent == 0) break;
            parent--;
        }
    }
    for (; i < rating_len; ++i)
        if (COMP((int)i, ids_out[0])) adjust_heap_(ids_out, 0, sort_len, (int)i, r);
    for (int64_t last = sort_len; last > 1;) { /* sort_heap */
        --last;
        int v = ids_out[last];
        ids_out[last] = ids_out[0];
        adjust_heap_(ids_out, 0, last, v, r);
    }
    return (int)sort_len;
}
#undef COMP

/* ------------------------------------------------------------------------------------------ */
/* E6: metrics (metric.h:19-109).  float accumulators; the `+= 1.0/log2(i+2)` and `1.0/(i+1)`   */
/* terms are evaluated in double and rounded to float at each step, exactly as the C++ does.    */
/* ------------------------------------------------------------------------------------------ */
static void m_precision(const int* rank, int k, const int* truth, int64_t nt, float* out) {
    float hits = 0.0f;
    for (int i = 0; i < k; ++i) {
        if (contains_sorted(truth, nt, rank[i])) hits = (float)((double)hits + 1.0);
        out[i] = hits / (float)(unsigned)(i + 1);
    }
}
static void m_recall(const int* rank, int k, const int* truth, int64_t nt, float* out) {
    float hits = 0.0f;
    float truth_len = (float)(nt > 1 ? (int)nt : 1);
    for (int i = 0; i < k; ++i) {
        if (contains_sorted(truth, nt, rank[i])) hits = (float)((double)hits + 1.0);
        out[i] = hits / truth_len;
    }
}
static void m_ap(const int* rank, int k, const int* truth, int64_t nt, float* out) {
    float hits = 0.0f, pre = 0.0f, sum_pre = 0.0f, denominator = 1.0f;
    int truth_len = nt > 1 ? (int)nt : 1;
    for (int i = 0; i < k; ++i) {
        if (contains_sorted(truth, nt, rank[i])) {
            hits = (float)((double)hits + 1.0);
            pre = hits / (float)(unsigned)(i + 1);
            sum_pre += pre;
        }
        denominator = (float)(truth_len < i + 1 ? truth_len : i + 1);
        out[i] = sum_pre / denominator;
    }
}
static void m_ndcg(const int* rank, int k, const int* truth, int64_t nt, float* out) {
    float iDCG = 0.0f, DCG = 0.0f;
    unsigned truth_len = (unsigned)(nt > 1 ? (int)nt : 1);
    for (unsigned i = 0; i < (unsigned)k; ++i) {
        if (contains_sorted(truth, nt, rank[i])) DCG = (float)((double)DCG + 1.0 / log2((double)(i + 2)));
        if (i < truth_len) iDCG = (float)((double)iDCG + 1.0 / log2((double)(i + 2)));
        out[i] = DCG / iDCG;
    }
}
static void m_mrr(const int* rank, int k, const int* truth, int64_t nt, float* out) {
    for (int i = 0; i < k; ++i) {
        if (contains_sorted(truth, nt, rank[i])) {
            float rr = (float)(1.0 / (double)(unsigned)(i + 1));
            for (int j = i; j < k; ++j) out[j] = rr;
            return;
        }
        out[i] = 0.0f;
    }
}

/* cpp_evaluate_matrix + eval_one_user (evaluate.h:24-76), test items as CSR (any order per row).
 * ratings [n_users, rating_len] row-major fp32, already train-masked; out [n_users, n_metric*top_k]
 * metric-major, zero-initialised by the caller like results_pt (pyx_eval_matrix.pyx:32).
 * ids_out (optional, may be NULL): [n_users, top_k] the arg-top-K lists.
 * Returns 0, or -1 on an unknown metric id / top_k > rating_len (the reference would crash). */
int orc_evaluate_matrix(const float* ratings, int n_users, int rating_len,
                        const int64_t* test_rowptr, const int* test_items,
                        const int* metric, int n_metric, int top_k, float* out, int* ids_out) {
    if (top_k <= 0 || top_k > rating_len) return -1;
    for (int m = 0; m < n_metric; ++m) if (metric[m] < 1 || metric[m] > 5) return -1;
    int* rank = (int*)malloc((size_t)(2 * (int64_t)top_k) * sizeof(int));
    int* tbuf = NULL; int64_t tcap = 0;
    for (int u = 0; u < n_users; ++u) {
        const float* r = ratings + (int64_t)u * rating_len;
        orc_partial_sort_ids(r, rating_len, top_k, rank);
        int64_t nt = test_rowptr[u + 1] - test_rowptr[u];
        if (nt > tcap) { tcap = nt * 2; tbuf = (int*)realloc(tbuf, (size_t)tcap * sizeof(int)); }
        if (nt > 0) {
            memcpy(tbuf, test_items + test_rowptr[u], (size_t)nt * sizeof(int));
            qsort(tbuf, (size_t)nt, sizeof(int), cmp_int);
            int64_t w = 0; /* set semantics: duplicates collapse (unordered_set) */
            for (int64_t q = 0; q < nt; ++q) if (w == 0 || tbuf[w - 1] != tbuf[q]) tbuf[w++] = tbuf[q];
            nt = w;
        }
        float* o = out + (int64_t)u * n_metric * top_k;
        for (int m = 0; m < n_metric; ++m) {
            float* om = o + (int64_t)m * top_k;
            switch (metric[m]) {
                case 1: m_precision(rank, top_k, tbuf, nt, om); break;
                case 2: m_recall(rank, top_k, tbuf, nt, om); break;
                case 3: m_ap(rank, top_k, tbuf, nt, om); break;
                case 4: m_ndcg(rank, top_k, tbuf, nt, om); break;
                case 5: m_mrr(rank, top_k, tbuf, nt, om); break;
            }
        }
        if (ids_out) memcpy(ids_out + (int64_t)u * top_k, rank, (size_t)top_k * sizeof(int));
    }
    free(rank);
    free(tbuf);
    return 0;
}

/* Deterministic-tie variant used as the checker for the HIP path's documented tie rule
 * (score descending, then item id ascending).  On tie-free rows it equals orc_partial_sort_ids. */
typedef struct { float s; int id; } sid_t;
static int cmp_sid(const void* a, const void* b) {
    const sid_t* x = (const sid_t*)a; const sid_t* y = (const sid_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->id > y->id) - (x->id < y->id);
}
int orc_topk_ids_lowid(const float* r, int rating_len, int top_k, int* ids_out) {
    sid_t* a = (sid_t*)malloc((size_t)rating_len * sizeof(sid_t));
    for (int i = 0; i < rating_len; ++i) { a[i].s = r[i]; a[i].id = i; }
    qsort(a, (size_t)rating_len, sizeof(sid_t), cmp_sid);
    for (int i = 0; i < top_k && i < rating_len; ++i) ids_out[i] = a[i].id;
    free(a);
    return 0;
}
