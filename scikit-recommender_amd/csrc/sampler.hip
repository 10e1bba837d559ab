// sampler.hip -- S rows of the hot path: negative sampling on gfx950.
//
// Reference being replaced (paths under skrec/):
//   utils/py/cython/include/randint.h:20      std::mt19937 _gen(2020)   (process-global stream)
//   utils/py/cython/include/randint.h:23-88   _random_int / c_randint_choice
//   io/data_iterator.py:81-94                 _sampling_negative_items  (per-user Python loop)
//
// Three device paths:
//   1. randint_serial_kernel   the reference's serial algorithm, one lane (API-surface calls)
//   2. exact epoch             mt_generate_kernel -> exact_assign_kernel -> mt_commit_kernel:
//                              the MT19937 word stream is produced in bulk, then ONE workgroup
//                              resolves "which draw fills which slot" chunk by chunk with a
//                              fixed-point iteration on the (rare) rejections; bit-exact with the
//                              reference stream, latency-bound by construction (a serial chain).
//   3. fast epoch              sample_fast_kernel: slot-keyed xoshiro128++ + rejection against
//                              LDS-staged CSR positives; HBM-bound (8 B per sampled negative).
#include "skr_common.h"

#include <cstring>
#include <vector>

namespace {

constexpr int MT_N = 624;
constexpr int MT_M = 397;

__host__ __device__ __forceinline__ uint32_t mt_temper(uint32_t z) {
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
}
__host__ __device__ __forceinline__ uint32_t mt_untemper(uint32_t y) {
    y ^= (y >> 18);
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y;  // invert y ^= (y << 7) & 0x9d2c5680
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    y = t;
    t = y ^ (y >> 11);  // invert y ^= (y >> 11)
    y = y ^ (t >> 11);
    return y;
}
__host__ __device__ __forceinline__ uint32_t mt_mix(uint32_t hi, uint32_t lo) {
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// ------------------------------------------------------------------------------------------------
// 1. serial path: c_randint_choice on one lane
// ------------------------------------------------------------------------------------------------
struct MtRef {  // MT19937 living in global memory, advanced by a single lane
    uint32_t* mt;
    int p;
    unsigned long long n;
    __device__ void twist() {
        for (int k = 0; k < MT_N; ++k)
            mt[k] = mt[(k + MT_M) % MT_N] ^ mt_mix(mt[k], mt[(k + 1) % MT_N]);
        p = 0;
    }
    __device__ uint32_t next() {
        if (p >= MT_N) twist();
        ++n;
        return mt_temper(mt[p++]);
    }
    // std::uniform_int_distribution<int>(0, range-1), libstdc++-11 Lemire path
    __device__ int below(uint32_t range) {
        uint64_t prod = static_cast<uint64_t>(next()) * range;
        uint32_t low = static_cast<uint32_t>(prod);
        if (low < range) {
            const uint32_t thr = (0u - range) % range;
            while (low < thr) {
                prod = static_cast<uint64_t>(next()) * range;
                low = static_cast<uint32_t>(prod);
            }
        }
        return static_cast<int>(prod >> 32);
    }
    __device__ double canonical() {  // std::generate_canonical<double, 53>
        const double R = 4294967296.0;
        double sum = static_cast<double>(next());
        sum += static_cast<double>(next()) * R;
        double ret = sum / (R * R);
        if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;
        return ret;
    }
};

__global__ void randint_serial_kernel(uint32_t* state, int* pos, unsigned long long* draws, int high, int size,
                                      int replace, const float* prob, double* cp, const int32_t* excl, int n_excl,
                                      int32_t* result) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    MtRef g{state, *pos, 0ull};
    if (prob) {  // std::discrete_distribution: normalise, partial sums, last = 1 (bits/random.tcc)
        double sum = 0.0;
        for (int i = 0; i < high; ++i) sum += static_cast<double>(prob[i]);
        double acc = 0.0;
        for (int i = 0; i < high; ++i) {
            acc += static_cast<double>(prob[i]) / sum;
            cp[i] = acc;
        }
        cp[high - 1] = 1.0;
    }
    int i = 0;
    while (i < size) {
        int s;
        if (prob) {
            const double p = g.canonical();
            int lo = 0, hi = high;  // std::lower_bound
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (cp[mid] < p) lo = mid + 1; else hi = mid;
            }
            s = lo;
        } else {
            s = g.below(static_cast<uint32_t>(high));
        }
        bool rejected = excl && skr::contains_sorted(excl, 0, n_excl, s);
        if (!rejected && !replace) {  // randint.h:53-70: accepted values join the exclusion set
            for (int k = 0; k < i; ++k)
                if (result[k] == s) { rejected = true; break; }
        }
        if (!rejected) result[i++] = s;
    }
    *pos = g.p;
    *draws += g.n;
}

// ------------------------------------------------------------------------------------------------
// 2a. bulk MT19937 word generation (one workgroup, LDS-resident state, 3-phase parallel twist)
// ------------------------------------------------------------------------------------------------
constexpr int GEN_T = 256;

// n >= 0: generate exactly n words.  n < 0: generate at least -n words and stop on a block boundary of the generator
// (the host does not know the stream position without a read-back); the count goes to *n_out.
// `piece` / `piece_words`: the stretch may be produced in pieces (piece p = words [p * piece_words, (p + 1) * piece_words) of
// it, on a stream of their own beside the kernels that consume them): piece 0 starts from the sampler's state and fixes
// the stretch's length, every piece leaves the generator's state in `carry` (624 words + position) for the next one.
__global__ __launch_bounds__(GEN_T) void mt_generate_kernel(const uint32_t* __restrict__ state,
                                                            const int* __restrict__ pos_p,
                                                            uint32_t* __restrict__ raw, int64_t n, int64_t* __restrict__ n_out,
                                                            uint32_t* __restrict__ carry = nullptr, int piece = 0,
                                                            int64_t piece_words = 0) {
    // two copies of the state: a twist reads one and writes the other, so a phase needs ONE workgroup barrier (before the next
    // phase reads what it wrote) instead of two (reads done / writes done): three barriers per 624 words instead of eight
    __shared__ uint32_t mt2[2][MT_N];
    const int tid = threadIdx.x;
    const bool resume = carry && piece > 0;
    int cur = 0;
    for (int i = tid; i < MT_N; i += GEN_T) mt2[0][i] = resume ? carry[i] : state[i];
    __syncthreads();
    int p = resume ? static_cast<int>(carry[MT_N]) : *pos_p;
    if (n < 0) {
        if (resume) {
            n = *n_out;                    // fixed by piece 0
        } else {
            const int64_t want = -n, r0 = MT_N - p;
            n = want <= r0 ? r0 : r0 + ((want - r0 + MT_N - 1) / MT_N) * MT_N;
            if (tid == 0 && n_out) *n_out = n;
        }
    }
    if (carry) {                           // this piece's share of the stretch
        const int64_t beg = static_cast<int64_t>(piece) * piece_words;
        raw += beg;
        n = n - beg < piece_words ? n - beg : piece_words;
        if (n < 0) n = 0;
    }
    int64_t k = 0;
    while (k < n) {
        if (p >= MT_N) {
            const uint32_t* __restrict__ old = mt2[cur];
            uint32_t* __restrict__ nw = mt2[cur ^ 1];
            // new[i] = old[i+397] ^ mix(old[i], old[i+1])                     i in [0, 227)
            if (tid < MT_N - MT_M) nw[tid] = old[tid + MT_M] ^ mt_mix(old[tid], old[tid + 1]);
            __syncthreads();
            // new[i] = new[i-227] ^ mix(old[i], old[i+1])                     i in [227, 454)
            if (tid < MT_N - MT_M) nw[tid + 227] = nw[tid] ^ mt_mix(old[tid + 227], old[tid + 228]);
            __syncthreads();
            // new[i] = new[i-227] ^ mix(old[i], old[i+1])                     i in [454, 623);  new[623] = new[396] ^ mix(old[623], new[0])
            if (tid < 169) nw[tid + 454] = nw[tid + 227] ^ mt_mix(old[tid + 454], old[tid + 455]);
            if (tid == 169) nw[623] = nw[396] ^ mt_mix(old[623], nw[0]);
            __syncthreads();
            cur ^= 1;
            p = 0;
        }
        const uint32_t* __restrict__ mt = mt2[cur];
        const int64_t left = n - k;
        const int m = static_cast<int>(left < (MT_N - p) ? left : (MT_N - p));
        for (int t = tid; t < m; t += GEN_T) raw[k + t] = mt_temper(mt[p + t]);
        k += m;
        p += m;
        // no barrier here: the next twist writes the OTHER copy, whose last readers passed the three barriers above
    }
    __syncthreads();
    if (carry) {
        for (int i = tid; i < MT_N; i += GEN_T) carry[i] = mt2[cur][i];
        if (tid == 0) carry[MT_N] = static_cast<uint32_t>(p);
    }
}

// 2c. advance the stored state by `consumed` words of the buffer generated above.
// The tempered words of a whole block ARE that block's state (tempering is a bijection).
__global__ void mt_commit_kernel(uint32_t* state, int* pos_p, unsigned long long* draws,
                                 const uint32_t* __restrict__ raw, int64_t n_raw, const int64_t* ctl,
                                 const int64_t* __restrict__ n_raw_dev) {
    if (n_raw_dev) n_raw = *n_raw_dev;
    const int64_t consumed = ctl[1];
    const int p0 = *pos_p;
    const int64_t r0 = MT_N - p0;  // words of the current block that were still unread
    __shared__ int64_t kb_s;
    __shared__ int newpos_s;
    if (threadIdx.x == 0) {
        if (consumed < r0) {
            kb_s = -1;
            newpos_s = p0 + static_cast<int>(consumed);
        } else {
            int64_t b = (consumed - r0) / MT_N;
            int64_t kb = r0 + b * MT_N;
            if (kb + MT_N <= n_raw) {
                kb_s = kb;
                newpos_s = static_cast<int>(consumed - kb);
            } else if (b >= 1) {  // consumed == n_raw exactly at a block boundary
                kb_s = kb - MT_N;
                newpos_s = MT_N;
            } else {
                kb_s = -1;
                newpos_s = MT_N;
            }
        }
        *draws += static_cast<unsigned long long>(consumed);
    }
    __syncthreads();
    if (kb_s >= 0)
        for (int i = threadIdx.x; i < MT_N; i += blockDim.x) state[i] = mt_untemper(raw[kb_s + i]);
    if (threadIdx.x == 0) *pos_p = newpos_s;
}

// ------------------------------------------------------------------------------------------------
// 2b. exact assignment: which draw fills which slot
// ------------------------------------------------------------------------------------------------
// words of the control block d_ctl (int64 each)
constexpr int SL_S = 0;         // slots filled so far
constexpr int SL_D = 1;         // words of the generated stretch consumed so far
constexpr int SL_SCRATCH = 2;   // max_row_len scratch (two ints)
constexpr int SL_SUMSQ = 3;     // sum of squared row lengths (dataset statistic)
constexpr int SL_FALLBACK = 4;  // the slab path met a case it leaves to the serial kernel
constexpr int SL_NRAW = 5;      // words generated
constexpr int SL_SLAB_S = 6;    // S at the start of the slab being scattered
constexpr int SL_SLAB_N = 7;    // draws of that slab that were consumed (0: nothing to scatter)
constexpr int SL_STATUS = 8;    // 1 = the last exact epoch ended with every slot filled

constexpr int AS_T = 1024;
constexpr int AS_PER = 16;
constexpr int AS_C = AS_T * AS_PER;

// largest u in [lo, hi] with rowptr[u] <= q   (rows with no positives can never own q)
__device__ __forceinline__ int owner_of(const int64_t* __restrict__ rowptr, int lo, int hi, int64_t q) {
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= q) lo = mid; else hi = mid - 1;
    }
    return lo;
}

constexpr int AS_ROW_CAP = 8192;    // row offsets staged in LDS per chunk (32 KB)
constexpr int AS_POS_CAP = 24576;   // positives staged in LDS per chunk (96 KB)
constexpr int AS_POS_CAP_SEP = 16384;   // ... when a second offset window shares the LDS (64 KB)

// largest idx in [lo, hi] with a[idx] <= q (a ascending, in LDS)
__device__ __forceinline__ int lds_owner(const int32_t* a, int lo, int hi, int32_t q) {
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (a[mid] <= q) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One workgroup walks the draw stream in chunks of AS_C.  Per chunk the part of the CSR the chunk can
// touch -- row offsets of the users from the current one on, and their positives -- is staged in LDS
// with two coalesced reads, so that the fixed-point rounds (owner search + membership search per
// draw) run on LDS latency instead of a chain of ~14 dependent HBM misses per draw and round
// (measured before staging: 0.45 s per 48 M-slot epoch, 8.3 ns per slot).
//
// SEP = false: user u owns rowptr[u+1]-rowptr[u] positives and num_neg slots per positive (the pairwise /
// pointwise iterators).  SEP = true: the slots of user u are drawptr[u] .. drawptr[u+1] while the CSR row
// is only the exclusion set (sequential and knowledge-graph iterators: the number of training instances
// of a user is not the size of its exclusion set); num_neg is 1 then.
template <bool SEP>
__global__ __launch_bounds__(AS_T) void exact_assign_kernel(
    const uint32_t* __restrict__ raw, int64_t n_raw, uint32_t high, const int64_t* __restrict__ rowptr, int n_users,
    const int32_t* __restrict__ pos_sorted, int num_neg, int64_t n_slots, int64_t slot_start, int32_t* __restrict__ out,
    int64_t* __restrict__ ctl, const int64_t* __restrict__ drawptr) {
    constexpr int POS_CAP = SEP ? AS_POS_CAP_SEP : AS_POS_CAP;
    __shared__ int32_t l_row[AS_ROW_CAP + 1];
    __shared__ int32_t l_draw[SEP ? AS_ROW_CAP + 1 : 1];
    __shared__ int32_t l_pos[POS_CAP];
    const int32_t* l_own = SEP ? l_draw : l_row;          // cumulative slot counts: who owns a slot
    const int64_t* __restrict__ optr = SEP ? drawptr : rowptr;
    __shared__ int wave_tot[AS_T / SKR_WAVE];
    __shared__ int s_w0, s_uhi_global;
    __shared__ int s_last_idx;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t lemire_thr = (0u - high) % high;

    int64_t draw_origin = 0;
    if (slot_start < 0) {
        // FALLBACK of the slab path (run_exact_epoch_slabs): continue from the point it gave up at, if it did
        if (ctl[SL_FALLBACK] == 0 || ctl[SL_S] >= n_slots) return;
        slot_start = ctl[SL_S];
        draw_origin = ctl[SL_D];
        n_raw = ctl[SL_NRAW] - draw_origin;
        raw += draw_origin;
    }
    int64_t slot_base = slot_start;
    int64_t draw_base = 0;
    int w0 = 0;  // first user of the staged window; the owner of slot_base never moves backwards
#ifdef SKR_SAMPLER_STAMPS
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = clock64(), n_iters = 0;
#define STAMP(k) { if (tid == 0) { long long tn_ = clock64(); tph[k] += tn_ - tprev; tprev = tn_; } }
#else
#define STAMP(k)
#endif
    while (slot_base < n_slots && draw_base < n_raw) {
        const int64_t left = n_raw - draw_base;
        const int clen = static_cast<int>(left < AS_C ? left : AS_C);
        const int64_t q0 = slot_base / num_neg;
        int64_t last_slot = slot_base + clen - 1;
        if (last_slot > n_slots - 1) last_slot = n_slots - 1;
        const int64_t q1 = last_slot / num_neg;
        // ---- window of row offsets starting at a user <= owner(q0) ---------------------------------
        if (tid == 0) {
            const int wend = (w0 + AS_ROW_CAP < n_users) ? w0 + AS_ROW_CAP : n_users;
            int nw0 = w0;
            if (!(optr[wend] > q0)) nw0 = owner_of(optr, w0, n_users - 1, q0);  // window would miss q0: re-anchor
            s_w0 = nw0;
            s_last_idx = -1;
        }
        __syncthreads();
        STAMP(0)
        w0 = s_w0;
        const int wlen = (n_users - w0 < AS_ROW_CAP) ? n_users - w0 : AS_ROW_CAP;  // rows w0 .. w0+wlen-1
        const int64_t pbeg0 = rowptr[w0];
        const int64_t obeg0 = optr[w0];
        for (int i = tid; i <= wlen; i += AS_T) {
            const int64_t d = rowptr[w0 + i] - pbeg0;
            l_row[i] = d > 0x7fffffff ? 0x7fffffff : static_cast<int32_t>(d);
            if (SEP) {
                const int64_t dd = drawptr[w0 + i] - obeg0;
                l_draw[i] = dd > 0x7fffffff ? 0x7fffffff : static_cast<int32_t>(dd);
            }
        }
        __syncthreads();
        const bool q_fits = (q1 - obeg0) < 0x7fffffff;
        const bool rows_staged = q_fits && (static_cast<int64_t>(l_own[wlen]) > q1 - obeg0);  // owner(q1) inside the window
        const int ulo_i = lds_owner(l_own, 0, wlen - 1, static_cast<int32_t>(q0 - obeg0));  // q0 is inside by construction
        int uhi_i;
        if (rows_staged) {
            uhi_i = lds_owner(l_own, ulo_i, wlen - 1, static_cast<int32_t>(q1 - obeg0));
        } else {
            if (tid == 0) s_uhi_global = owner_of(optr, w0 + ulo_i, n_users - 1, q1);
            __syncthreads();
            uhi_i = s_uhi_global - w0;
        }
        const int ulo = w0 + ulo_i, uhi = w0 + uhi_i;
        STAMP(1)
        // ---- positives of users ulo..uhi ---------------------------------------------------------------
        const int64_t pb = rows_staged ? pbeg0 + l_row[ulo_i] : rowptr[ulo];
        const int64_t pe = rows_staged ? pbeg0 + l_row[uhi_i + 1] : rowptr[uhi + 1];
        const bool pos_staged = rows_staged && (pe - pb <= POS_CAP);
        if (pos_staged)
            for (int i = tid; i < static_cast<int>(pe - pb); i += AS_T) l_pos[i] = pos_sorted[pb + i];
        const int32_t pb_rel = static_cast<int32_t>(pb - pbeg0);  // only used when rows_staged
        // ---- this lane's draws: value, Lemire rejection -------------------------------------------------
        int val[AS_PER];
        uint32_t lem = 0, inr = 0, rej = 0;  // bit masks over the AS_PER draws
#pragma unroll
        for (int e = 0; e < AS_PER; ++e) {
            const int idx = tid * AS_PER + e;
            uint32_t x = 0;
            if (idx < clen) {
                x = raw[draw_base + idx];
                inr |= 1u << e;
            }
            const uint64_t prod = static_cast<uint64_t>(x) * high;
            val[e] = static_cast<int>(prod >> 32);
            if (idx < clen && static_cast<uint32_t>(prod) < lemire_thr) lem |= 1u << e;
        }
        rej = lem;
        __syncthreads();  // l_pos complete
        STAMP(2)
        int my_off = 0, total = 0;
        const bool nn1 = (num_neg == 1);
        const uint32_t nn = static_cast<uint32_t>(num_neg);
        const uint32_t rb = static_cast<uint32_t>(slot_base % num_neg);
        const int32_t qb_rel = static_cast<int32_t>(q0 - obeg0);
        const int64_t sl64 = n_slots - slot_base;
        const int slots_left = sl64 > AS_C ? AS_C + 1 : static_cast<int>(sl64);   // draws beyond it are never consumed
        // Per draw the last (owner, membership) answer is kept: a later round shifts slots by the number of
        // newly found rejections, and a draw whose slot stays inside the same user's row needs no new
        // search.  Within a lane slots are consecutive, so the owner only ever moves forward.
        int16_t own[AS_PER];       // window index of the owner the cached answer belongs to (-1: none)
        uint32_t hitbits = 0;      // cached membership answers
#pragma unroll
        for (int e = 0; e < AS_PER; ++e) own[e] = -1;
        // fixed point: rej -> slot of every draw -> owner -> membership -> rej
        for (int iter = 0; iter < AS_C + 2; ++iter) {
            const int acc_cnt = __popc(inr & ~rej);
            int incl = skr::wave_incl_scan(acc_cnt);
            if (lane == 63) wave_tot[wv] = incl;
            __syncthreads();
            int wbase = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < AS_T / SKR_WAVE; ++w) {
                const int t = wave_tot[w];
                if (w < wv) wbase += t;
                total += t;
            }
            my_off = wbase + incl - acc_cnt;
            uint32_t nrej = lem;
            int run = my_off;
            if (pos_staged) {
                // pass 1: owners (forward walk from the previous draw's owner), decide who needs a search
                int ui = -1;
                uint32_t need = 0;
                int32_t lo[AS_PER], hi[AS_PER];
#pragma unroll
                for (int e = 0; e < AS_PER; ++e) {
                    const uint32_t bit = 1u << e;
                    lo[e] = hi[e] = 0;
                    if (!(inr & bit)) continue;
                    const int run_e = run;
                    if (!(rej & bit)) ++run;
                    if ((lem & bit) || run_e >= slots_left) continue;
                    // 32-bit arithmetic only: a 64-bit division per draw and round dominated this kernel
                    const int32_t qr = qb_rel + static_cast<int32_t>(nn1 ? static_cast<uint32_t>(run_e)
                                                                         : (rb + static_cast<uint32_t>(run_e)) / nn);
                    if (ui < 0) ui = lds_owner(l_own, ulo_i, uhi_i, qr);
                    else while (l_own[ui + 1] <= qr) ++ui;      // users that own no slot are skipped too
                    if (own[e] == ui) {                         // same user as last round: answer stands
                        if (hitbits & bit) nrej |= bit;
                    } else {
                        own[e] = static_cast<int16_t>(ui);
                        need |= bit;
                        lo[e] = l_row[ui] - pb_rel;
                        hi[e] = l_row[ui + 1] - pb_rel;
                    }
                }
                // pass 2: the membership searches of this lane in lock step (independent LDS reads per step)
                if (__any(need != 0)) {
                    bool go = true;
                    while (go) {
                        go = false;
#pragma unroll
                        for (int e = 0; e < AS_PER; ++e) {
                            if ((need >> e) & 1u) {
                                if (lo[e] < hi[e]) {
                                    const int32_t mid = (lo[e] + hi[e]) >> 1;
                                    if (l_pos[mid] < val[e]) lo[e] = mid + 1; else hi[e] = mid;
                                    go = true;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < AS_PER; ++e) {
                        const uint32_t bit = 1u << e;
                        if (need & bit) {
                            const bool h = lo[e] < l_row[own[e] + 1] - pb_rel && l_pos[lo[e]] == val[e];
                            hitbits = h ? (hitbits | bit) : (hitbits & ~bit);
                            if (h) nrej |= bit;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < AS_PER; ++e) {
                    const uint32_t bit = 1u << e;
                    if (!(inr & bit)) continue;
                    const int run_e = run;
                    if (!(rej & bit)) ++run;
                    if (lem & bit) continue;
                    if (run_e >= slots_left) continue;  // never consumed: leave un-rejected, ignored below
                    const int64_t q = q0 + (nn1 ? static_cast<uint32_t>(run_e) : (rb + static_cast<uint32_t>(run_e)) / nn);
                    bool hit;
                    if (rows_staged) {
                        const int ui = lds_owner(l_own, ulo_i, uhi_i, static_cast<int32_t>(q - obeg0));
                        hit = skr::contains_sorted(pos_sorted, pbeg0 + l_row[ui], pbeg0 + l_row[ui + 1], val[e]);
                    } else {
                        const int u = owner_of(optr, ulo, uhi, q);
                        hit = skr::contains_sorted(pos_sorted, rowptr[u], rowptr[u + 1], val[e]);
                    }
                    if (hit) nrej |= bit;
                }
            }
            const int changed = (nrej != rej);
            rej = nrej;
#ifdef SKR_SAMPLER_STAMPS
            if (tid == 0) tph[5] += 1;
#endif
            if (!__syncthreads_or(changed)) break;  // also orders wave_tot for the next round
        }
        STAMP(3)
        // `rej` is the fixed point and my_off/total were computed from it in the last round.
        {
            int run = my_off;
#pragma unroll
            for (int e = 0; e < AS_PER; ++e) {
                const uint32_t bit = 1u << e;
                if (!(inr & bit) || (rej & bit)) continue;
                const int64_t slot = slot_base + run;
                ++run;
                if (slot < n_slots) {
                    out[slot] = val[e];
                    if (slot == n_slots - 1) s_last_idx = tid * AS_PER + e;
                }
            }
        }
        __syncthreads();
        int64_t filled = slot_base + total;
        int consumed = clen;
        if (filled >= n_slots) {
            filled = n_slots;
            consumed = s_last_idx + 1;
        }
        __syncthreads();
        slot_base = filled;
        draw_base += consumed;
        w0 = ulo;  // next chunk's first owner is >= this chunk's
        STAMP(4)
    }
    if (tid == 0) {
        ctl[0] = slot_base;
        ctl[1] = draw_origin + draw_base;
#ifdef SKR_SAMPLER_STAMPS
        for (int k = 0; k < 6; ++k) ctl[4 + k] = tph[k];
#endif
    }
#undef STAMP
}

// ------------------------------------------------------------------------------------------------
// 2d. the same assignment on the whole chip, for sparse data: SLABS of draws, three parallel-friendly kernels per slab
// ------------------------------------------------------------------------------------------------
// Which slot a draw fills depends on every rejection before it, so the stream is a serial chain -- but rejections are
// RARE on recommendation data (a draw is rejected when it hits one of its user's positives: ~0.05 % at 48 positives
// per user and 100 k items).  Per slab of SLAB_B draws, with the exact number S of slots filled before the slab known:
//   detect   (every CU)  draw j can only land in slots S + j - W(j) .. S + j, where W(j) bounds the rejections inside the
//            slab before it.  Those slots belong to a handful of users; the draw's value is searched in the positives of
//            each of them.  A hit anywhere, or a Lemire rejection, makes the draw an EVENT (one bit per draw + a mask of
//            the users it would be rejected for).  Everything here is independent of the rejections inside the slab.
//   resolve  (one workgroup) walks the slab's few hundred events in order with the running rejection count r: the
//            event's slot S + j - r names its owner among the candidate users, the mask says whether the draw is
//            rejected there.  This is the serial chain, shrunk from every draw to the events.
//   scatter  (every CU)  out[S + j - (rejections before j)] = value for the accepted draws.
// Anything outside the assumptions -- more rejections than W(j), more events or candidate users than the tables hold, a
// window of row offsets that does not fit -- raises SL_FALLBACK: the remaining stream is then assigned by
// exact_assign_kernel (the serial kernel above) from the point the slabs stopped at.  Same stream, same slots, bit for bit.
constexpr int SLAB_B = 1 << 17;        // draws per slab at most (sparse data); denser data gets shorter slabs, see run_exact_epoch_slabs
constexpr int SLAB_T = 256;            // threads per workgroup (detect / scatter)
constexpr int SLAB_ROWS = 1024;        // row offsets staged per detect workgroup
constexpr int SLAB_EV = 1024;          // events per slab the resolver holds
constexpr int SLAB_BND = 8;            // candidate owners per event the resolver looks at
constexpr int32_t EV_LEMIRE = -2, EV_UNKNOWN = -1;

struct SlabArgs {
    const uint32_t* raw;
    uint32_t high;
    const int64_t* rowptr;       // membership rows (CSR of the exclusion sets)
    const int64_t* optr;         // ownership: cumulative q-units per user (rowptr itself, or drawptr)
    int n_users;
    const int32_t* pos_sorted;
    int num_neg;
    int64_t n_slots;
    int64_t* ctl;
    int64_t slab_off;            // first draw of this slab in raw
    int slab_b;                  // draws per slab in this call (a multiple of 256, <= SLAB_B)
    int win0;                    // W(j) = win0 + (j * win_rate) >> 16
    uint32_t win_rate;
    unsigned long long* ev_bits; // [SLAB_B / 64]
    uint32_t* ev_mask;           // [SLAB_B]
    int32_t* ev_uhi;             // [SLAB_B]
    unsigned long long* rej_bits;// [SLAB_B / 64]
    int32_t* out;
};

__device__ __forceinline__ int slab_window(const SlabArgs& a, int j) {
    const int w = a.win0 + static_cast<int>((static_cast<uint64_t>(j) * a.win_rate) >> 16);
    return w < j ? w : j;    // there cannot be more rejections before draw j than draws
}

// draws of this slab that exist (the generated stretch may end inside it); 0 when the slab has nothing to do
__device__ __forceinline__ int slab_len(const SlabArgs& a) {
    if (a.ctl[SL_FALLBACK] != 0 || a.ctl[SL_D] != a.slab_off || a.ctl[SL_S] >= a.n_slots) return 0;
    const int64_t left = a.ctl[SL_NRAW] - a.slab_off;
    return left <= 0 ? 0 : static_cast<int>(left < a.slab_b ? left : a.slab_b);
}

__global__ __launch_bounds__(SLAB_T) void slab_detect_kernel(SlabArgs a) {
    __shared__ int64_t l_own[SLAB_ROWS + 1];
    __shared__ int s_ubase, s_cover;
    const int tid = threadIdx.x, lane = tid & 63;
    const int len = slab_len(a);
    const int j0 = blockIdx.x * SLAB_T;
    if (j0 >= len) return;
    const int64_t S = a.ctl[SL_S];
    const int j = j0 + tid;
    const int jl = (j0 + SLAB_T - 1 < len ? j0 + SLAB_T - 1 : len - 1);      // last draw of this workgroup
    // slots this workgroup's draws can land in
    int64_t slot_lo = S + j0 - slab_window(a, jl);
    if (slot_lo < S) slot_lo = S;
    int64_t slot_hi = S + jl;
    if (slot_hi > a.n_slots - 1) slot_hi = a.n_slots - 1;
    const bool beyond = slot_lo > a.n_slots - 1;        // every draw here comes after the last slot is filled
    const int64_t q_lo = slot_lo / a.num_neg, q_hi = slot_hi / a.num_neg;
    if (tid == 0) s_ubase = beyond ? 0 : owner_of(a.optr, 0, a.n_users - 1, q_lo);
    __syncthreads();
    const int ubase = s_ubase;
    const int nrow = a.n_users - ubase < SLAB_ROWS ? a.n_users - ubase : SLAB_ROWS;   // users ubase .. ubase + nrow - 1
    for (int i = tid; i <= nrow; i += SLAB_T) l_own[i] = a.optr[ubase + i];
    __syncthreads();
    if (tid == 0) s_cover = (l_own[nrow] > q_hi) ? 1 : 0;      // the owner of q_hi is inside the staged window
    __syncthreads();
    const bool covered = s_cover != 0;
    bool is_ev = false;
    uint32_t mask = 0;
    int32_t uhi = EV_UNKNOWN;
    if (j < len && !beyond) {
        const uint64_t prod = static_cast<uint64_t>(a.raw[a.slab_off + j]) * a.high;
        const int val = static_cast<int>(prod >> 32);
        const uint32_t thr = (0u - a.high) % a.high;
        int64_t smin = S + j - slab_window(a, j);
        if (smin < S) smin = S;
        if (static_cast<uint32_t>(prod) < thr) {
            is_ev = true;
            uhi = EV_LEMIRE;
        } else if (smin <= a.n_slots - 1) {
            if (!covered) {
                is_ev = true;                      // the resolver hands the slab to the serial kernel
            } else {
                int64_t smax = S + j;
                if (smax > a.n_slots - 1) smax = a.n_slots - 1;
                const int64_t qmin = smin / a.num_neg, qmax = smax / a.num_neg;
                // largest index with l_own[idx] <= q
                int lo = 0, hi = nrow - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (l_own[mid] <= qmax) lo = mid; else hi = mid - 1; }
                const int ihi = lo;
                lo = 0; hi = ihi;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (l_own[mid] <= qmin) lo = mid; else hi = mid - 1; }
                const int ilo = lo;
                if (ihi - ilo >= 32) {
                    is_ev = true;                  // more candidate users than a mask holds
                } else {
                    for (int i = ihi; i >= ilo; --i) {
                        if (l_own[i + 1] == l_own[i]) continue;           // owns no slot: can never be the owner
                        const int u = ubase + i;
                        if (skr::contains_sorted(a.pos_sorted, a.rowptr[u], a.rowptr[u + 1], val)) mask |= 1u << (ihi - i);
                    }
                    if (mask) {
                        is_ev = true;
                        uhi = ubase + ihi;
                    }
                }
            }
        }
    }
    const unsigned long long b = __ballot(is_ev);
    if (lane == 0) a.ev_bits[j >> 6] = b;          // (j is a multiple of 64 on lane 0)
    if (is_ev) {
        a.ev_mask[j] = mask;
        a.ev_uhi[j] = uhi;
    }
}

__global__ __launch_bounds__(256) void slab_resolve_kernel(SlabArgs a) {
    __shared__ int32_t l_j[SLAB_EV], l_uhi[SLAB_EV];      // l_uhi[e] becomes EV_REJECTED once event e is known to be rejected
    __shared__ uint32_t l_mask[SLAB_EV];
    __shared__ int64_t l_q0[SLAB_EV];                     // optr[uhi]
    __shared__ int32_t l_bnd[SLAB_EV][SLAB_BND];          // optr[uhi] - optr[uhi - i]  (>= 0; INT32_MAX: no such user)
    __shared__ int l_cnt[256];
    __shared__ int s_total;
    constexpr int32_t EV_REJECTED = -3;
    const int tid = threadIdx.x;
    const int len = slab_len(a);
    if (tid == 0) a.ctl[SL_SLAB_N] = 0;
    if (len == 0) return;
    const int n_words = (len + 63) >> 6;
    const int WPT = ((a.slab_b >> 6) + 255) / 256;      // bitmap words per thread
    // events, in draw order
    int cnt = 0;
    for (int w = tid * WPT; w < (tid + 1) * WPT && w < n_words; ++w) {
        a.rej_bits[w] = 0ull;
        cnt += __popcll(a.ev_bits[w]);
    }
    l_cnt[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int c = l_cnt[t]; l_cnt[t] = run; run += c; }
        s_total = run;
    }
    __syncthreads();
    const int E = s_total;
    const bool too_many = E > SLAB_EV;
    if (!too_many) {
        int e = l_cnt[tid];
        for (int w = tid * WPT; w < (tid + 1) * WPT && w < n_words; ++w) {
            unsigned long long bits = a.ev_bits[w];
            while (bits) {
                const int bpos = __ffsll(static_cast<long long>(bits)) - 1;
                bits &= bits - 1;
                l_j[e++] = (w << 6) + bpos;
            }
        }
    }
    __syncthreads();
    if (!too_many) {
        for (int e = tid; e < E; e += 256) {
            const int j = l_j[e];
            const int32_t uhi = a.ev_uhi[j];
            l_uhi[e] = uhi;
            l_mask[e] = a.ev_mask[j];
            if (uhi >= 0) {
                const int64_t q0 = a.optr[uhi];
                l_q0[e] = q0;
#pragma unroll
                for (int i = 0; i < SLAB_BND; ++i) {
                    int64_t d = (uhi - i >= 0) ? q0 - a.optr[uhi - i] : 0x7fffffff;
                    l_bnd[e][i] = d > 0x7fffffff ? 0x7fffffff : static_cast<int32_t>(d);
                }
            }
        }
    }
    __syncthreads();
    // the serial chain: one lane, everything it reads is in LDS
    if (tid == 0) {
        const int64_t S = a.ctl[SL_S];
        const int64_t remaining = a.n_slots - S;
        int r = 0;
        bool give_up = too_many;
        for (int e = 0; e < E && !give_up; ++e) {
            const int j = l_j[e];
            if (j - r >= remaining) break;                       // every slot is filled before this draw is reached
            if (r > slab_window(a, j)) { give_up = true; break; }  // more rejections than the detector allowed for
            const int32_t uhi = l_uhi[e];
            bool rejected;
            if (uhi == EV_LEMIRE) {
                rejected = true;
            } else if (uhi < 0) {
                give_up = true;
                break;
            } else {
                // owner = the largest user u <= uhi with optr[u] <= q, i.e. the smallest i with optr[uhi] - optr[uhi - i] >= optr[uhi] - q
                const int64_t need = l_q0[e] - (S + j - r) / a.num_neg;
                int i = 0;
                while (i < SLAB_BND && static_cast<int64_t>(l_bnd[e][i]) < need) ++i;
                if (i == SLAB_BND) { give_up = true; break; }
                rejected = (l_mask[e] >> i) & 1u;
            }
            if (rejected) {
                ++r;
                l_uhi[e] = EV_REJECTED;
            }
        }
        if (give_up) {
            a.ctl[SL_FALLBACK] = 1;                              // S and D stay at the slab's start
            s_total = 0;
        } else {
            // r counts the rejections among the CONSUMED draws: the walk stops at the first event that is not reached
            int64_t consumed = remaining + r;
            if (consumed > len) consumed = len;
            a.ctl[SL_SLAB_S] = S;
            a.ctl[SL_SLAB_N] = consumed;
            a.ctl[SL_S] = S + (consumed - r);
            a.ctl[SL_D] = a.slab_off + consumed;
        }
    }
    __syncthreads();
    for (int e = tid; e < s_total; e += 256)
        if (l_uhi[e] == EV_REJECTED) atomicOr(&a.rej_bits[l_j[e] >> 6], 1ull << (l_j[e] & 63));
}

__global__ __launch_bounds__(SLAB_T) void slab_scatter_kernel(SlabArgs a) {
    __shared__ int l_part[SLAB_T];
    const int tid = threadIdx.x;
    const int64_t consumed = a.ctl[SL_SLAB_N];
    // the resolver has moved SL_D past this slab (or left it, or given up): SL_SLAB_N > 0 only for the slab just resolved
    if (consumed <= 0 || a.ctl[SL_D] != a.slab_off + consumed) return;
    const int j0 = blockIdx.x * SLAB_T;
    if (j0 >= consumed) return;
    const int64_t S = a.ctl[SL_SLAB_S];
    // rejections before this workgroup's first draw: the bitmap words below j0 / 64
    const int w0 = j0 >> 6;
    int part = 0;
    for (int w = tid; w < w0; w += SLAB_T) part += __popcll(a.rej_bits[w]);
    l_part[tid] = part;
    __syncthreads();
    for (int o = SLAB_T / 2; o > 0; o >>= 1) {
        if (tid < o) l_part[tid] += l_part[tid + o];
        __syncthreads();
    }
    const int before_wg = l_part[0];
    const int j = j0 + tid;
    if (j >= consumed) return;
    int before = before_wg;
    for (int w = w0; w < (j >> 6); ++w) before += __popcll(a.rej_bits[w]);
    const unsigned long long word = a.rej_bits[j >> 6];
    if ((word >> (j & 63)) & 1ull) return;                       // rejected
    before += __popcll(word & ((1ull << (j & 63)) - 1ull));
    const uint64_t prod = static_cast<uint64_t>(a.raw[a.slab_off + j]) * a.high;
    a.out[S + j - before] = static_cast<int>(prod >> 32);
}

__global__ void slab_begin_kernel(int64_t* ctl) {
    ctl[SL_S] = 0; ctl[SL_D] = 0; ctl[SL_FALLBACK] = 0; ctl[SL_NRAW] = 0; ctl[SL_SLAB_S] = 0; ctl[SL_SLAB_N] = 0; ctl[SL_STATUS] = 0;
}
__global__ void slab_end_kernel(int64_t* ctl, int64_t n_slots) { ctl[SL_STATUS] = (ctl[SL_S] >= n_slots) ? 1 : 2; }

// ------------------------------------------------------------------------------------------------
// 3. fast path: slot-keyed xoshiro128++ with rejection against LDS-staged positives
// ------------------------------------------------------------------------------------------------
constexpr int FS_T = 256;
constexpr int FS_PER = 8;
constexpr int FS_SLOTS = FS_T * FS_PER;  // slots per workgroup
constexpr int FS_POS_CAP = 6144;         // positives staged in LDS (24 KB)
constexpr int FS_ROW_CAP = 2048;         // row offsets staged in LDS (8 KB)

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t& x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

struct Xoshiro128pp {
    uint32_t s0, s1, s2, s3;
    __host__ __device__ void seed(uint64_t seed, uint64_t epoch, uint64_t slot) {
        uint64_t x = seed;
        uint64_t k = splitmix64(x) ^ (epoch * 0xD1B54A32D192ED03ull);
        x = k;
        k = splitmix64(x) ^ slot;
        x = k;
        const uint64_t a = splitmix64(x), b = splitmix64(x);
        s0 = static_cast<uint32_t>(a); s1 = static_cast<uint32_t>(a >> 32);
        s2 = static_cast<uint32_t>(b); s3 = static_cast<uint32_t>(b >> 32);
        if ((s0 | s1 | s2 | s3) == 0) s0 = 1;
    }
    __host__ __device__ uint32_t next() {
        const uint32_t r = rotl32(s0 + s3, 7) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
        s3 = rotl32(s3, 11);
        return r;
    }
};

__global__ __launch_bounds__(FS_T) void sample_fast_kernel(uint64_t seed, uint64_t epoch, int64_t slot_offset,
                                                           uint32_t high, int n_users,
                                                           const int64_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ pos_sorted, int num_neg,
                                                           int64_t n_slots, int32_t* __restrict__ out) {
    __shared__ int32_t l_pos[FS_POS_CAP];
    __shared__ int32_t l_row[FS_ROW_CAP + 1];
    __shared__ int s_ulo, s_uhi;
    const int tid = threadIdx.x;
    const int64_t a0 = static_cast<int64_t>(blockIdx.x) * FS_SLOTS;
    if (a0 >= n_slots) return;
    int64_t a1 = a0 + FS_SLOTS;
    if (a1 > n_slots) a1 = n_slots;
    if (tid == 0) {
        s_ulo = owner_of(rowptr, 0, n_users - 1, a0 / num_neg);
        s_uhi = owner_of(rowptr, 0, n_users - 1, (a1 - 1) / num_neg);
    }
    __syncthreads();
    const int ulo = s_ulo, uhi = s_uhi;
    const int64_t pbeg = rowptr[ulo], pend = rowptr[uhi + 1];
    const bool staged = (uhi - ulo + 1 <= FS_ROW_CAP) && (pend - pbeg <= FS_POS_CAP);  // block-uniform
    if (staged) {
        for (int i = tid; i <= uhi - ulo + 1; i += FS_T) l_row[i] = static_cast<int32_t>(rowptr[ulo + i] - pbeg);
        for (int64_t i = tid; i < pend - pbeg; i += FS_T) l_pos[i] = pos_sorted[pbeg + i];
    }
    __syncthreads();
    const uint32_t lemire_thr = (0u - high) % high;
#pragma unroll 1
    for (int e = 0; e < FS_PER; ++e) {
        const int64_t a = a0 + e * FS_T + tid;  // coalesced stores
        if (a >= a1) break;
        const int64_t q = a / num_neg;
        int64_t rb, re;
        if (staged) {
            const int32_t ql = static_cast<int32_t>(q - pbeg);
            int lo = 0, hi = uhi - ulo;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (l_row[mid] <= ql) lo = mid; else hi = mid - 1;
            }
            rb = l_row[lo];
            re = l_row[lo + 1];
        } else {
            const int u = owner_of(rowptr, ulo, uhi, q);
            rb = rowptr[u];
            re = rowptr[u + 1];
        }
        Xoshiro128pp g;
        g.seed(seed, epoch, static_cast<uint64_t>(slot_offset + a));
        // Bounded so that an (invalid) row covering the whole catalogue cannot hang the GPU; the
        // host mirror rejects such rows up front exactly like pyx_random.pyx:49.  With one free item
        // the bound is missed with probability < e^-64.
        int cand = -1;
        for (uint64_t tries = 0, cap = 64ull * high + 1024; tries < cap; ++tries) {
            const uint64_t prod = static_cast<uint64_t>(g.next()) * high;
            if (static_cast<uint32_t>(prod) < lemire_thr) continue;
            const int c = static_cast<int>(prod >> 32);
            const bool hit = staged ? skr::contains_sorted(l_pos, rb, re, c)
                                    : skr::contains_sorted(pos_sorted, rb, re, c);
            if (!hit) { cand = c; break; }
        }
        out[a] = cand;
    }
}

__global__ void max_row_len_kernel(const int64_t* __restrict__ rowptr, int n_rows, int* __restrict__ out_max,
                                   unsigned long long* __restrict__ out_sumsq) {
    int m = 0;
    unsigned long long sq = 0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n_rows;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int len = static_cast<int>(rowptr[i + 1] - rowptr[i]);
        m = len > m ? len : m;
        sq += static_cast<unsigned long long>(len) * static_cast<unsigned long long>(len);
    }
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
        sq += __shfl_xor(sq, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out_max, m);
        if (out_sumsq) atomicAdd(out_sumsq, sq);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct skr_sampler {
    uint32_t* d_state = nullptr;            // 624 words
    int* d_pos = nullptr;                   // next word (0..624)
    unsigned long long* d_draws = nullptr;  // words consumed
    int64_t* d_ctl = nullptr;               // [0] slots filled, [1] words consumed, [2] scratch (max row len)
    uint32_t* d_raw = nullptr;
    size_t raw_cap = 0;  // in words
    // scratch of the slab path (one slab's worth)
    uint32_t* d_carry = nullptr;            // generator state between the pieces of a stretch (624 words + position)
    hipStream_t gen_stream = nullptr;       // the pieces are generated here, beside the slabs that consume them
    std::vector<hipEvent_t> gen_events;
    hipEvent_t start_event = nullptr;
    int64_t* h_ctl = nullptr;               // pinned: the one read-back per exact-epoch call lands here
    int64_t* h_status = nullptr;            // pinned: the status word of the LAST epoch, copied behind it (see status_event)
    hipEvent_t status_event = nullptr;      // recorded behind that copy
    bool status_pending = false;
    unsigned long long* d_ev_bits = nullptr;
    unsigned long long* d_rej_bits = nullptr;
    uint32_t* d_ev_mask = nullptr;
    int32_t* d_ev_uhi = nullptr;
};

namespace skr {
int max_row_len(const int64_t* d_rowptr, int n_rows, int* d_scratch, hipStream_t st, int* out) {
    SKR_HIP(hipMemsetAsync(d_scratch, 0, sizeof(int), st));
    int blocks = (n_rows + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(max_row_len_kernel, dim3(blocks), dim3(256), 0, st, d_rowptr, n_rows, d_scratch,
                       static_cast<unsigned long long*>(nullptr));
    SKR_LAUNCH_CHECK();
    SKR_HIP(hipMemcpyAsync(out, d_scratch, sizeof(int), hipMemcpyDeviceToHost, st));
    SKR_HIP(hipStreamSynchronize(st));
    return SKR_OK;
}
}  // namespace skr

extern "C" {

int skr_sampler_create(uint32_t seed, skr_sampler** out) {
    SKR_REQUIRE(out != nullptr, "skr_sampler_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return skr::fail(SKR_ENODEV, "no HIP device visible");
    skr_sampler* s = new skr_sampler();
    SKR_HIP(hipMalloc(&s->d_state, MT_N * sizeof(uint32_t)));
    SKR_HIP(hipMalloc(&s->d_pos, sizeof(int)));
    SKR_HIP(hipMalloc(&s->d_draws, sizeof(unsigned long long)));
    SKR_HIP(hipMalloc(&s->d_ctl, 32 * sizeof(int64_t)));
    SKR_HIP(hipMemset(s->d_ctl, 0, 32 * sizeof(int64_t)));
    std::vector<uint32_t> mt(MT_N);
    mt[0] = seed;  // std::mt19937::seed(value)
    for (int i = 1; i < MT_N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
    *out = s;
    return skr_sampler_set_state(s, mt.data(), MT_N);
}

int skr_sampler_destroy(skr_sampler* s) {
    if (!s) return SKR_OK;
    (void)hipFree(s->d_state);
    (void)hipFree(s->d_pos);
    (void)hipFree(s->d_draws);
    (void)hipFree(s->d_ctl);
    (void)hipFree(s->d_raw);
    (void)hipFree(s->d_carry);
    if (s->gen_stream) (void)hipStreamDestroy(s->gen_stream);
    for (hipEvent_t e : s->gen_events) (void)hipEventDestroy(e);
    if (s->start_event) (void)hipEventDestroy(s->start_event);
    if (s->h_ctl) (void)hipHostFree(s->h_ctl);
    if (s->h_status) (void)hipHostFree(s->h_status);
    if (s->status_event) (void)hipEventDestroy(s->status_event);
    (void)hipFree(s->d_ev_bits);
    (void)hipFree(s->d_rej_bits);
    (void)hipFree(s->d_ev_mask);
    (void)hipFree(s->d_ev_uhi);
    delete s;
    return SKR_OK;
}

int skr_sampler_get_state(skr_sampler* s, uint32_t* words624, int* pos) {
    SKR_REQUIRE(s && words624 && pos, "skr_sampler_get_state: NULL argument");
    SKR_HIP(hipDeviceSynchronize());
    SKR_HIP(hipMemcpy(words624, s->d_state, MT_N * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SKR_HIP(hipMemcpy(pos, s->d_pos, sizeof(int), hipMemcpyDeviceToHost));
    return SKR_OK;
}

int skr_sampler_set_state(skr_sampler* s, const uint32_t* words624, int pos) {
    SKR_REQUIRE(s && words624, "skr_sampler_set_state: NULL argument");
    SKR_REQUIRE(pos >= 0 && pos <= MT_N, "skr_sampler_set_state: pos %d outside [0, 624]", pos);
    SKR_HIP(hipDeviceSynchronize());
    SKR_HIP(hipMemcpy(s->d_state, words624, MT_N * sizeof(uint32_t), hipMemcpyHostToDevice));
    SKR_HIP(hipMemcpy(s->d_pos, &pos, sizeof(int), hipMemcpyHostToDevice));
    SKR_HIP(hipMemset(s->d_draws, 0, sizeof(unsigned long long)));
    return SKR_OK;
}

int skr_sampler_draws(skr_sampler* s, uint64_t* n) {
    SKR_REQUIRE(s && n, "skr_sampler_draws: NULL argument");
    SKR_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    SKR_HIP(hipMemcpy(&v, s->d_draws, sizeof(v), hipMemcpyDeviceToHost));
    *n = v;
    return SKR_OK;
}

int skr_sampler_last_epoch(skr_sampler* s, int64_t* h_info4) {
    SKR_REQUIRE(s && h_info4, "skr_sampler_last_epoch: NULL argument");
    SKR_HIP(hipDeviceSynchronize());
    int64_t ctl[SL_STATUS + 1];
    SKR_HIP(hipMemcpy(ctl, s->d_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
    h_info4[0] = ctl[SL_STATUS];      // 0: serial path (or no epoch yet), 1: slab path, every slot filled, 2: slab path, incomplete
    h_info4[1] = ctl[SL_FALLBACK];    // 1: the slab path handed the rest of the stream to the serial kernel
    h_info4[2] = ctl[SL_S];           // slots filled
    h_info4[3] = ctl[SL_D];           // words of the generated stretch consumed
    return SKR_OK;
}

int skr_randint_choice(skr_sampler* s, int high, int size, int replace, const float* d_prob,
                       const int32_t* d_exclusion, int n_exclusion, int32_t* d_result, void* stream) {
    SKR_REQUIRE(s && d_result, "skr_randint_choice: NULL argument");
    SKR_REQUIRE(high > 1, "'high' must be larger than 1.");                        // pyx_random.pyx:34
    SKR_REQUIRE(size > 0, "'size' must be a positive integer.");                   // :36
    SKR_REQUIRE(n_exclusion >= 0 && (n_exclusion == 0 || d_exclusion), "exclusion pointer/length mismatch");
    SKR_REQUIRE(n_exclusion < high, "The length of 'exclusion' must be smaller than 'high'.");  // :49
    SKR_REQUIRE(replace || high - n_exclusion > size, "There is not enough integers to be sampled.");  // :53
    hipStream_t st = skr::as_stream(stream);
    double* d_cp = nullptr;
    if (d_prob) SKR_HIP(hipMalloc(&d_cp, static_cast<size_t>(high) * sizeof(double)));
    hipLaunchKernelGGL(randint_serial_kernel, dim3(1), dim3(64), 0, st, s->d_state, s->d_pos, s->d_draws, high, size,
                       replace, d_prob, d_cp, n_exclusion ? d_exclusion : nullptr, n_exclusion, d_result);
    SKR_LAUNCH_CHECK();
    if (d_cp) {
        SKR_HIP(hipStreamSynchronize(st));
        SKR_HIP(hipFree(d_cp));
    }
    return SKR_OK;
}

// the exact epoch on ONE workgroup: generate a stretch of the word stream, assign, commit, repeat (dense data, tiny calls,
// SKR_EXACT_PATH=serial)
static int run_exact_epoch_serial(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                           const int32_t* d_pos_sorted, const int64_t* d_drawptr, int num_neg, int64_t n_slots,
                           int32_t* d_out, hipStream_t st) {
    int64_t filled = 0;
    int pos = 0;
    while (filled < n_slots) {
        SKR_HIP(hipMemcpyAsync(&pos, s->d_pos, sizeof(int), hipMemcpyDeviceToHost, st));
        SKR_HIP(hipStreamSynchronize(st));
        const int64_t need = n_slots - filled;
        const int64_t want = need + need / 8 + 4096;  // head-room for rejected draws
        const int64_t r0 = MT_N - pos;
        int64_t n_gen = r0 + ((want - r0 + MT_N - 1) / MT_N) * MT_N;  // ends on a block boundary
        if (n_gen < r0) n_gen = r0;
        if (static_cast<size_t>(n_gen) > s->raw_cap) {
            if (s->d_raw) SKR_HIP(hipFree(s->d_raw));
            s->d_raw = nullptr;
            s->raw_cap = 0;
            SKR_HIP(hipMalloc(&s->d_raw, static_cast<size_t>(n_gen) * sizeof(uint32_t)));
            s->raw_cap = static_cast<size_t>(n_gen);
        }
        hipLaunchKernelGGL(mt_generate_kernel, dim3(1), dim3(GEN_T), 0, st, s->d_state, s->d_pos, s->d_raw, n_gen,
                           static_cast<int64_t*>(nullptr));
        SKR_LAUNCH_CHECK();
        if (d_drawptr)
            hipLaunchKernelGGL(exact_assign_kernel<true>, dim3(1), dim3(AS_T), 0, st, s->d_raw, n_gen,
                               static_cast<uint32_t>(num_items), d_rowptr, n_users, d_pos_sorted, 1, n_slots, filled,
                               d_out, s->d_ctl, d_drawptr);
        else
            hipLaunchKernelGGL(exact_assign_kernel<false>, dim3(1), dim3(AS_T), 0, st, s->d_raw, n_gen,
                               static_cast<uint32_t>(num_items), d_rowptr, n_users, d_pos_sorted, num_neg, n_slots, filled,
                               d_out, s->d_ctl, static_cast<const int64_t*>(nullptr));
        SKR_LAUNCH_CHECK();
        hipLaunchKernelGGL(mt_commit_kernel, dim3(1), dim3(256), 0, st, s->d_state, s->d_pos, s->d_draws, s->d_raw,
                           n_gen, s->d_ctl, static_cast<const int64_t*>(nullptr));
        SKR_LAUNCH_CHECK();
        int64_t ctl[2] = {0, 0};
        SKR_HIP(hipMemcpyAsync(ctl, s->d_ctl, sizeof(ctl), hipMemcpyDeviceToHost, st));
        SKR_HIP(hipStreamSynchronize(st));
        if (ctl[0] <= filled && ctl[1] == 0) return skr::fail(SKR_EHIP, "exact sampler made no progress");
#ifdef SKR_SAMPLER_STAMPS
        {
            int64_t st_[10];
            SKR_HIP(hipMemcpy(st_, s->d_ctl + 4, sizeof(st_), hipMemcpyDeviceToHost));
            fprintf(stderr, "[exact_assign stamps] anchor %lld window %lld pos+draws %lld fixedpoint %lld finish %lld (cycles) rounds %lld\n",
                    (long long)st_[0], (long long)st_[1], (long long)st_[2], (long long)st_[3], (long long)st_[4], (long long)st_[5]);
        }
#endif
        filled = ctl[0];
    }
    return SKR_OK;
}

// The exact epoch in slabs (kernels 2d): nothing here waits for the device.  The stretch of the word stream is generated
// with head-room for the rejections (the caller has checked that no row covers more than 1/16 of the catalogue, so the
// 12.5 % + 4096 words always suffice); how many words were used is committed on the device (mt_commit_kernel), and
// whether every slot was filled is left in d_ctl[SL_STATUS] for the next call to look at.
static int run_exact_epoch_slabs(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                                 const int32_t* d_pos_sorted, const int64_t* d_drawptr, int num_neg, int64_t n_slots,
                                 int32_t* d_out, double reject_rate, hipStream_t st) {
    const int64_t want = n_slots + n_slots / 8 + 4096;
    const size_t cap = static_cast<size_t>(want) + 2 * MT_N;
    if (cap > s->raw_cap) {
        if (s->d_raw) SKR_HIP(hipFree(s->d_raw));
        s->d_raw = nullptr;
        s->raw_cap = 0;
        SKR_HIP(hipMalloc(&s->d_raw, cap * sizeof(uint32_t)));
        s->raw_cap = cap;
    }
    if (!s->d_ev_bits) {
        SKR_HIP(hipMalloc(&s->d_ev_bits, (SLAB_B / 64) * sizeof(unsigned long long)));
        SKR_HIP(hipMalloc(&s->d_rej_bits, (SLAB_B / 64) * sizeof(unsigned long long)));
        SKR_HIP(hipMalloc(&s->d_ev_mask, SLAB_B * sizeof(uint32_t)));
        SKR_HIP(hipMalloc(&s->d_ev_uhi, SLAB_B * sizeof(int32_t)));
    }
    // the word stream is a serial chain on one workgroup (~0.8 G words/s): it is produced in pieces on a stream of the
    // sampler's own, and the slabs of the caller's stream start as soon as the piece they read is there
    constexpr int64_t PIECE = int64_t{1} << 21;
    const int n_pieces = static_cast<int>((want + MT_N + PIECE - 1) / PIECE);
    if (!s->gen_stream) {
        SKR_HIP(hipStreamCreateWithFlags(&s->gen_stream, hipStreamNonBlocking));
        SKR_HIP(hipEventCreateWithFlags(&s->start_event, hipEventDisableTiming));
        SKR_HIP(hipMalloc(&s->d_carry, (MT_N + 1) * sizeof(uint32_t)));
    }
    while (static_cast<int>(s->gen_events.size()) < n_pieces) {
        hipEvent_t e;
        SKR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->gen_events.push_back(e);
    }
    hipLaunchKernelGGL(slab_begin_kernel, dim3(1), dim3(1), 0, st, s->d_ctl);
    // SKR_SAMPLER_ONE_STREAM=1: everything on the caller's stream (several processes sharing ONE GPU -- rehearsals of the
    // multi-rank paths -- oversubscribe the hardware queues, and cross-queue waits then cost milliseconds)
    static const bool env_one_stream = getenv("SKR_SAMPLER_ONE_STREAM") && !strcmp(getenv("SKR_SAMPLER_ONE_STREAM"), "1");
    // a call of one piece has nothing to generate BESIDE: the first slab waits for the whole stretch either way, and the two
    // stream hand-offs (~25 us) would be all the side stream adds
    const bool one_stream = env_one_stream || n_pieces == 1;
    hipStream_t gst = one_stream ? st : s->gen_stream;
    if (!one_stream) {
        SKR_HIP(hipEventRecord(s->start_event, st));
        SKR_HIP(hipStreamWaitEvent(gst, s->start_event, 0));
    }
    for (int pc = 0; pc < n_pieces; ++pc) {
        hipLaunchKernelGGL(mt_generate_kernel, dim3(1), dim3(GEN_T), 0, gst, s->d_state, s->d_pos, s->d_raw, -want,
                           s->d_ctl + SL_NRAW, s->d_carry, pc, PIECE);
        if (!one_stream) SKR_HIP(hipEventRecord(s->gen_events[pc], gst));
    }
    SKR_LAUNCH_CHECK();
    int pieces_awaited = one_stream ? n_pieces : 0;
    SlabArgs a{};
    a.raw = s->d_raw;
    a.high = static_cast<uint32_t>(num_items);
    a.rowptr = d_rowptr;
    a.optr = d_drawptr ? d_drawptr : d_rowptr;
    a.n_users = n_users;
    a.pos_sorted = d_pos_sorted;
    a.num_neg = num_neg;
    a.n_slots = n_slots;
    a.ctl = s->d_ctl;
    // W(j) = 24 + 2 * (expected rejections among j draws): a slab's rejection count is Poisson-like, so this is many
    // standard deviations of head-room; if it is ever exceeded the resolver hands over to the serial kernel
    a.win0 = 24;
    const double rate = 2.0 * reject_rate * 65536.0 + 1.0;
    a.win_rate = rate > 4.0e9 ? 4000000000u : static_cast<uint32_t>(rate);
    a.ev_bits = s->d_ev_bits;
    a.rej_bits = s->d_rej_bits;
    a.ev_mask = s->d_ev_mask;
    a.ev_uhi = s->d_ev_uhi;
    a.out = d_out;
    // slab length: about 256 expected rejections per slab (each shows up as a few events, the resolver holds SLAB_EV)
    int slab_b = SLAB_B;
    while (slab_b > 4096 && reject_rate * slab_b > 256.0) slab_b >>= 1;
    a.slab_b = slab_b;
    const int64_t n_slabs = (want + MT_N + slab_b - 1) / slab_b;      // the generator stops at most MT_N words past `want`
    for (int64_t k = 0; k < n_slabs; ++k) {
        a.slab_off = k * slab_b;
        int last_piece = static_cast<int>((a.slab_off + slab_b - 1) / PIECE);     // the piece holding this slab's last word
        if (last_piece > n_pieces - 1) last_piece = n_pieces - 1;
        for (; pieces_awaited <= last_piece; ++pieces_awaited) SKR_HIP(hipStreamWaitEvent(st, s->gen_events[pieces_awaited], 0));
        hipLaunchKernelGGL(slab_detect_kernel, dim3(slab_b / SLAB_T), dim3(SLAB_T), 0, st, a);
        hipLaunchKernelGGL(slab_resolve_kernel, dim3(1), dim3(256), 0, st, a);
        hipLaunchKernelGGL(slab_scatter_kernel, dim3(slab_b / SLAB_T), dim3(SLAB_T), 0, st, a);
    }
    SKR_LAUNCH_CHECK();
    for (; pieces_awaited < n_pieces; ++pieces_awaited) SKR_HIP(hipStreamWaitEvent(st, s->gen_events[pieces_awaited], 0));
    // whatever the slabs left (normally nothing: the kernel returns at once)
    if (d_drawptr)
        hipLaunchKernelGGL(exact_assign_kernel<true>, dim3(1), dim3(AS_T), 0, st, s->d_raw, 0, static_cast<uint32_t>(num_items),
                           d_rowptr, n_users, d_pos_sorted, 1, n_slots, static_cast<int64_t>(-1), d_out, s->d_ctl, d_drawptr);
    else
        hipLaunchKernelGGL(exact_assign_kernel<false>, dim3(1), dim3(AS_T), 0, st, s->d_raw, 0, static_cast<uint32_t>(num_items),
                           d_rowptr, n_users, d_pos_sorted, num_neg, n_slots, static_cast<int64_t>(-1), d_out, s->d_ctl,
                           static_cast<const int64_t*>(nullptr));
    hipLaunchKernelGGL(slab_end_kernel, dim3(1), dim3(1), 0, st, s->d_ctl, n_slots);
    hipLaunchKernelGGL(mt_commit_kernel, dim3(1), dim3(256), 0, st, s->d_state, s->d_pos, s->d_draws, s->d_raw,
                       static_cast<int64_t>(0), s->d_ctl, s->d_ctl + SL_NRAW);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

// One read-back per call, at its START (the stream is normally idle then): the longest row (the reference's argument check,
// pyx_random.pyx:49), the sum of squared row lengths (how often a draw hits a positive), and whether the PREVIOUS exact
// epoch filled every slot.  Then the path: slabs for sparse data, the serial kernel otherwise.
static int run_exact_epoch(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr, const int32_t* d_pos_sorted,
                           const int64_t* d_drawptr, int num_neg, int64_t n_slots, int64_t nnz, int32_t* d_out, hipStream_t st,
                           const char* who, const int64_t* h_stats = nullptr) {
    int max_len;
    unsigned long long sumsq;
    bool prev_failed = false;
    if (h_stats) {
        // the caller took the row statistics once (skr_csr_row_stats: the CSR does not change between epochs): nothing to read
        // back here, the call only queues work.  The previous epoch's status word was copied behind that epoch: it has
        // normally arrived long ago.
        max_len = static_cast<int>(h_stats[0]);
        sumsq = static_cast<unsigned long long>(h_stats[1]);
        if (s->status_pending) {
            SKR_HIP(hipEventSynchronize(s->status_event));
            prev_failed = s->h_status[0] == 2;
        }
    } else {
        SKR_HIP(hipMemsetAsync(s->d_ctl + SL_SCRATCH, 0, 2 * sizeof(int64_t), st));
        int blocks = (n_users + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(max_row_len_kernel, dim3(blocks), dim3(256), 0, st, d_rowptr, n_users, reinterpret_cast<int*>(s->d_ctl + SL_SCRATCH),
                           reinterpret_cast<unsigned long long*>(s->d_ctl + SL_SUMSQ));
        SKR_LAUNCH_CHECK();
        if (!s->h_ctl) SKR_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_ctl), (SL_STATUS + 1) * sizeof(int64_t), hipHostMallocDefault));
        int64_t* host = s->h_ctl;
        SKR_HIP(hipMemcpyAsync(host, s->d_ctl, (SL_STATUS + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        SKR_HIP(hipStreamSynchronize(st));
        max_len = static_cast<int>(host[SL_SCRATCH] & 0xffffffff);
        sumsq = static_cast<unsigned long long>(host[SL_SUMSQ]);
        prev_failed = host[SL_STATUS] == 2;
    }
    s->status_pending = false;
    if (prev_failed) {
        SKR_HIP(hipMemsetAsync(s->d_ctl + SL_STATUS, 0, sizeof(int64_t), st));
        return skr::fail(SKR_EOVERFLOW, "%s: the previous exact epoch ran out of generated words before every slot was filled", who);
    }
    SKR_REQUIRE(max_len < num_items, "The length of 'exclusion' must be smaller than 'high'.");  // pyx_random.pyx:49
    const char* path = getenv("SKR_EXACT_PATH");
    const bool force_slab = path && !strcmp(path, "slab"), force_serial = path && !strcmp(path, "serial");
    const bool sparse = static_cast<int64_t>(max_len) * 16 <= num_items;
    int rc;
    if (!force_serial && sparse && (force_slab || n_slots >= 4096)) {
        const double lemire = static_cast<double>((0u - static_cast<uint32_t>(num_items)) % static_cast<uint32_t>(num_items)) / 4294967296.0;
        const double rate = (nnz > 0 ? static_cast<double>(sumsq) / (static_cast<double>(nnz) * num_items) : 0.0) + lemire;
        rc = run_exact_epoch_slabs(s, num_items, n_users, d_rowptr, d_pos_sorted, d_drawptr, num_neg, n_slots, d_out, rate, st);
    } else {
        rc = run_exact_epoch_serial(s, num_items, n_users, d_rowptr, d_pos_sorted, d_drawptr, num_neg, n_slots, d_out, st);
    }
    if (rc != SKR_OK) return rc;
    // this epoch's status word, copied behind it: the next call looks at it without waiting for anything
    if (!s->h_status) SKR_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_status), sizeof(int64_t), hipHostMallocDefault));
    if (!s->status_event) SKR_HIP(hipEventCreateWithFlags(&s->status_event, hipEventDisableTiming));
    SKR_HIP(hipMemcpyAsync(s->h_status, s->d_ctl + SL_STATUS, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SKR_HIP(hipEventRecord(s->status_event, st));
    s->status_pending = true;
    return SKR_OK;
}

int skr_sample_epoch_exact(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                           const int32_t* d_pos_sorted, int64_t nnz, int num_neg, int32_t* d_out, void* stream) {
    SKR_REQUIRE(s && d_rowptr && d_pos_sorted && d_out, "skr_sample_epoch_exact: NULL argument");
    SKR_REQUIRE(num_items > 1, "'high' must be larger than 1.");
    SKR_REQUIRE(n_users > 0 && num_neg > 0, "skr_sample_epoch_exact: n_users and num_neg must be positive");
    SKR_REQUIRE(nnz >= 0, "skr_sample_epoch_exact: negative nnz");
    hipStream_t st = skr::as_stream(stream);
    if (nnz == 0) return SKR_OK;
    const int64_t n_slots = nnz * num_neg;
    SKR_REQUIRE(n_slots < (int64_t(1) << 31), "more than 2^31-1 samples per call (the reference's int limit)");
    return run_exact_epoch(s, num_items, n_users, d_rowptr, d_pos_sorted, nullptr, num_neg, n_slots, nnz, d_out, st,
                           "skr_sample_epoch_exact");
}

int skr_csr_row_stats(const int64_t* d_rowptr, int n_rows, int64_t* h_stats2, void* stream) {
    SKR_REQUIRE(d_rowptr && h_stats2, "skr_csr_row_stats: NULL argument");
    SKR_REQUIRE(n_rows >= 0, "skr_csr_row_stats: negative size");
    hipStream_t st = skr::as_stream(stream);
    int64_t* d = nullptr;
    SKR_HIP(hipMalloc(&d, 2 * sizeof(int64_t)));
    SKR_HIP(hipMemsetAsync(d, 0, 2 * sizeof(int64_t), st));
    if (n_rows > 0) {
        int blocks = (n_rows + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(max_row_len_kernel, dim3(blocks), dim3(256), 0, st, d_rowptr, n_rows, reinterpret_cast<int*>(d),
                           reinterpret_cast<unsigned long long*>(d + 1));
    }
    int64_t h[2] = {0, 0};
    const hipError_t e1 = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st), e2 = hipStreamSynchronize(st);
    (void)hipFree(d);
    SKR_HIP(e1);
    SKR_HIP(e2);
    h_stats2[0] = h[0] & 0xffffffff;
    h_stats2[1] = h[1];
    return SKR_OK;
}

int skr_sample_epoch_exact_stats(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr, const int32_t* d_pos_sorted,
                                 int64_t nnz, int num_neg, int32_t* d_out, const int64_t* h_stats2, void* stream) {
    SKR_REQUIRE(s && d_rowptr && d_pos_sorted && d_out && h_stats2, "skr_sample_epoch_exact_stats: NULL argument");
    SKR_REQUIRE(num_items > 1, "'high' must be larger than 1.");
    SKR_REQUIRE(n_users > 0 && num_neg > 0, "skr_sample_epoch_exact_stats: n_users and num_neg must be positive");
    SKR_REQUIRE(nnz >= 0 && h_stats2[0] >= 0 && h_stats2[1] >= 0, "skr_sample_epoch_exact_stats: negative size");
    hipStream_t st = skr::as_stream(stream);
    if (nnz == 0) return SKR_OK;
    const int64_t n_slots = nnz * num_neg;
    SKR_REQUIRE(n_slots < (int64_t(1) << 31), "more than 2^31-1 samples per call (the reference's int limit)");
    return run_exact_epoch(s, num_items, n_users, d_rowptr, d_pos_sorted, nullptr, num_neg, n_slots, nnz, d_out, st,
                           "skr_sample_epoch_exact_stats", h_stats2);
}

int skr_sample_epoch_exact_counts(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                                  const int32_t* d_excl_sorted, int64_t nnz, const int64_t* d_drawptr, int64_t n_draws,
                                  int32_t* d_out, void* stream) {
    SKR_REQUIRE(s && d_rowptr && d_drawptr && d_out, "skr_sample_epoch_exact_counts: NULL argument");
    SKR_REQUIRE(nnz == 0 || d_excl_sorted, "skr_sample_epoch_exact_counts: NULL exclusion array");
    SKR_REQUIRE(num_items > 1, "'high' must be larger than 1.");
    SKR_REQUIRE(n_users > 0, "skr_sample_epoch_exact_counts: n_users must be positive");
    SKR_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31), "skr_sample_epoch_exact_counts: bad exclusion size");
    SKR_REQUIRE(n_draws >= 0 && n_draws < (int64_t(1) << 31), "more than 2^31-1 samples per call (the reference's int limit)");
    hipStream_t st = skr::as_stream(stream);
    if (n_draws == 0) return SKR_OK;
    return run_exact_epoch(s, num_items, n_users, d_rowptr, d_excl_sorted, d_drawptr, 1, n_draws, nnz, d_out, st,
                           "skr_sample_epoch_exact_counts");
}

int skr_sample_epoch_fast(uint64_t seed, uint64_t epoch, int64_t slot_offset, int num_items, int n_users,
                          const int64_t* d_rowptr, const int32_t* d_pos_sorted, int64_t nnz, int num_neg,
                          int32_t* d_out, void* stream) {
    SKR_REQUIRE(d_rowptr && d_pos_sorted && d_out, "skr_sample_epoch_fast: NULL argument");
    SKR_REQUIRE(num_items > 1, "'high' must be larger than 1.");
    SKR_REQUIRE(n_users > 0 && num_neg > 0, "skr_sample_epoch_fast: n_users and num_neg must be positive");
    SKR_REQUIRE(nnz >= 0, "skr_sample_epoch_fast: negative nnz");
    hipStream_t st = skr::as_stream(stream);
    const int64_t n_slots = nnz * num_neg;
    if (n_slots == 0) return SKR_OK;
    const int64_t blocks = (n_slots + FS_SLOTS - 1) / FS_SLOTS;
    SKR_REQUIRE(blocks < (int64_t(1) << 31), "too many slots for one launch");
    hipLaunchKernelGGL(sample_fast_kernel, dim3(static_cast<unsigned>(blocks)), dim3(FS_T), 0, st, seed, epoch,
                       slot_offset, static_cast<uint32_t>(num_items), n_users, d_rowptr, d_pos_sorted, num_neg, n_slots,
                       d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

}  // extern "C"
