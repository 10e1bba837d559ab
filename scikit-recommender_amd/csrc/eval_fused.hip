// eval_fused.hip -- K4: user x item scoring on the matrix cores fused with train masking and top-K.
// Two arithmetic modes: exact fp32 FMA chains on the FP32 MFMA (described first), and the default bf16x3
// split on the bf16 MFMA (second half of the file).
//
// Replaces, in one launch and without ever writing the [B, I] score matrix to HBM:
//   recommender/BPRMF.py:84-88 / LightGCN.py:102-107   U[b] @ V.T (+ bias)         (torch.matmul)
//   utils/py/evaluator.py:197-200                      scores[train items] = -inf  (numpy loop)
//   utils/py/cython/include/evaluate.h:27-45           iota + partial_sort_copy    (arg-top-K)
//
// Mapping (gfx950, wave64, v_mfma_f32_32x32x2_f32 = exact fp32 fma chains):
//   * one wavefront owns 64 users (two 32-column B fragments, 64 VGPRs, loaded once) and sweeps the
//     whole catalogue in tiles of 32 items; a 32x32 MFMA tile has the ITEM on the row and the USER
//     on the column, so every lane holds 16 scores of ONE user -> the running top-K threshold of
//     that user is a single register compare per score;
//   * item tiles (32 x 256 B) go global -> LDS by LDS-DMA (global_load_lds_dwordx4, full 256-byte
//     lines, one tile ahead) into a wave-private, XOR-swizzled image that the A fragments are read
//     from with conflict-free ds_read_b128; waves share nothing, so there are no barriers and a
//     wave that stops to compact its candidate lists never stalls its neighbours;
//   * scores above the user's threshold (rare once it has warmed up) are appended to a per-user
//     candidate list in HBM scratch (LDS counters, one atomic per lane and tile); once a list holds
//     more than top_k + 32 entries the wave sorts it (register bitonic over 64/128/256 keys), drops
//     train items (binary search in the CSR), keeps the K best and raises the threshold;
//   * k order inside a dot product: the chain starts from the item bias (if any) and adds dims
//     {t, 32+t} at MFMA step t, a fixed order (results are deterministic run to run, and differ from
//     a CPU sgemm + bias add only in summation order).
// Bound: MFMA (2*B*I*64 flop); the 25.6 MB item table streams from L2 / Infinity Cache.
#include "eval_common.h"

#include <algorithm>
#include <cstdlib>
#include <string>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FE_WAVES = 4;   // waves per workgroup (independent of each other)
constexpr int FE_UW = 64;     // users per wave
constexpr int FE_TI = 32;     // items per tile
constexpr int FE_CAP = 256;   // candidate capacity per user (keys of 8 B); 512 was measured and bought nothing
constexpr int FE_D = 64;

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const int lo = __shfl_xor(static_cast<int>(static_cast<uint32_t>(v)), m, 64);
    const int hi = __shfl_xor(static_cast<int>(static_cast<uint32_t>(v >> 32)), m, 64);
    return (static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo);
}

// Descending sort of 64*NR keys held NR per lane; element index = e*64 + lane.
template <int NR>
__device__ __forceinline__ void wave_sort_desc(uint64_t (&k)[NR], int lane) {
    constexpr int N = 64 * NR;
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int de = j >> 6;
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    if ((e & de) == 0) {
                        const int e2 = e | de;
                        const bool desc = (((e * 64) & kk) == 0);
                        const uint64_t a = k[e], b = k[e2];
                        const bool sw = desc ? (a < b) : (a > b);
                        k[e] = sw ? b : a;
                        k[e2] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const bool desc = (((e * 64 + lane) & kk) == 0);
                    const bool low = ((lane & j) == 0);
                    const uint64_t a = k[e];
                    const uint64_t o = shfl_xor_u64(a, j);
                    const bool keep_max = (low == desc);
                    k[e] = keep_max ? (a > o ? a : o) : (a < o ? a : o);
                }
            }
        }
    }
}

struct FusedArgs {
    const float* user_table;
    const int32_t* users;
    int B;
    const float* item_table;
    const float* item_bias;
    int n_items;
    const int64_t* train_rowptr;
    const int32_t* train_items;
    int top_k;
    uint64_t* cand;  // [ceil(B/64)*64][cap]
    int32_t* out_ids;
    float* out_scores;
    int ablate;      // timing experiments only (SKR_FUSED_ABLATE): 1 = thresholds +inf (pure GEMM sweep)
    int trigger;     // a list is compacted once it holds more than this many candidates (K <= trigger <= cap-32)
    int cap;         // list capacity per user (FE_CAP)
    // --- the fp16x2 sweep and its fall-back (fused_topk_kernel_v7 / the bf16x3 kernel over the rows it flagged) -------------
    const int32_t* row_map;     // nullable: logical row r of this launch is row row_map[r] of users / outputs
    const int32_t* n_rows_dev;  // nullable: number of logical rows, read on the device (<= B)
    const float* guard_s_inv;   // device: 1 / (s_u s_v); a user whose smallest returned |score| is below 2^19 of it is flagged
    int32_t* flag_list;
    int32_t* flag_count;
};

__device__ __forceinline__ int rows_of(const FusedArgs& a) { return a.n_rows_dev ? *a.n_rows_dev : a.B; }
__device__ __forceinline__ int64_t src_row(const FusedArgs& a, int64_t row) { return a.row_map ? a.row_map[row] : row; }

constexpr int FE_ROWBUF = 256;   // train rows up to this length are staged in LDS for the membership searches

struct WaveCtx {
    int lane, c, h;
    int64_t ubase;
    int* cnt;            // LDS: list length per user of this wave
    uint64_t* my_cand;   // HBM: the wave's 64 candidate lists
    int64_t* row_beg;    // LDS: start of each user's train row
    int* row_len;        // LDS: its length (0 when no train mask was given)
    int* rowbuf;         // LDS: scratch for one staged row
    int rowbuf_len;      // its capacity in ints (a multiple of 64, at most FE_ROWBUF)
};

// The user's train row, fetched into registers when it is short enough to be staged in LDS: issued TOGETHER with the loads
// of the candidate list (one memory round trip per compaction instead of two: a compaction holds the three other
// wavefronts of its workgroup at the next barrier for as long as it takes).
struct RowRegs {
    int v[FE_ROWBUF / 64];
};
__device__ __forceinline__ RowRegs prefetch_row(const FusedArgs& a, const WaveCtx& w, int ul) {
    RowRegs r;
    const int len = w.row_len[ul];   // wave-uniform
    const int32_t* __restrict__ row = a.train_items + w.row_beg[ul];
#pragma unroll
    for (int e = 0; e < FE_ROWBUF / 64; ++e) {
        const int idx = e * 64 + w.lane;
        r.v[e] = (len <= w.rowbuf_len && idx < len) ? row[idx] : 0;
    }
    return r;
}

// Drop the user's train items from the NR keys each lane holds: NR binary searches per lane advance in
// lock-step, so a compaction pays ceil(log2(row length)) dependent load latencies, not NR times that.
template <int NR>
__device__ __forceinline__ void mask_train_lockstep(const FusedArgs& a, const WaveCtx& w, int ul, uint64_t (&k)[NR],
                                                    const RowRegs& rr) {
    const int len = w.row_len[ul];   // wave-uniform
    if (len <= 0) return;
    const int32_t* __restrict__ row = a.train_items + w.row_beg[ul];
    int lo[NR], hi[NR], v[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        lo[e] = 0;
        hi[e] = (k[e] != SKR_KEY_MIN) ? len : 0;
        v[e] = skr::key_id(k[e]);
    }
    const int steps = 32 - __clz(len);
    if (len <= w.rowbuf_len) {
        // one coalesced read of the row into LDS, then the searches run at LDS latency
        int* __restrict__ buf = w.rowbuf;
#pragma unroll
        for (int e = 0; e < FE_ROWBUF / 64; ++e) {
            const int idx = e * 64 + w.lane;
            if (idx < len) buf[idx] = rr.v[e];    // len <= rowbuf_len bounds the writes
        }
        __threadfence_block();   // the wave's own LDS writes, ordered before its reads
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int mid = (lo[e] + hi[e]) >> 1;
                const int x = buf[mid < len ? mid : len - 1];
                if (lo[e] < hi[e]) {
                    if (x < v[e]) lo[e] = mid + 1; else hi[e] = mid;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            const bool in_range = (k[e] != SKR_KEY_MIN) && lo[e] < len;
            if (in_range && buf[lo[e] < len ? lo[e] : 0] == v[e]) k[e] = SKR_KEY_MIN;
        }
        __threadfence_block();   // reads done before the next compaction overwrites the buffer
        return;
    }
    for (int s = 0; s < steps; ++s) {
        int x[NR], mid[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            mid[e] = (lo[e] + hi[e]) >> 1;
            x[e] = row[mid[e] < len ? mid[e] : len - 1];
        }
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            if (lo[e] < hi[e]) {
                if (x[e] < v[e]) lo[e] = mid[e] + 1; else hi[e] = mid[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const bool in_range = (k[e] != SKR_KEY_MIN) && lo[e] < len;
        const int x = row[in_range ? lo[e] : 0];
        if (in_range && x == v[e]) k[e] = SKR_KEY_MIN;
    }
}

// End of the sweep: sort user `ul`'s candidate list (at most 64*NR entries, already in registers together with the
// train row), drop train items, write the best top_k ids / scores to the outputs.
template <int NR>
__device__ __forceinline__ void final_user(const FusedArgs& a, const WaveCtx& w, int ul, uint64_t (&k)[NR], const RowRegs& rr,
                                           int64_t out_row) {
    const int lane = w.lane;
    mask_train_lockstep<NR>(a, w, ul, k, rr);
    wave_sort_desc<NR>(k, lane);
    int valid = 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) valid += __popcll(__ballot(k[e] != SKR_KEY_MIN));
    const int K = a.top_k;
    const int keep = valid < K ? valid : K;
#pragma unroll
    for (int e = 0; e < 2 && e < NR; ++e) {  // top_k <= 128: the survivors live in elements 0..127
        const int idx = e * 64 + lane;
        if (idx < keep) {
            if (a.out_ids) a.out_ids[out_row * K + idx] = skr::key_id(k[e]);
            if (a.out_scores) a.out_scores[out_row * K + idx] = skr::key_score(k[e]);
        }
    }
    if (a.flag_count && keep > 0) {
        // the fp16x2 sweep's guard (see fused_topk_kernel_v7): its absolute error floor must lie below the fp32 rounding noise
        // of the scores that decide the list, i.e. of the smallest one kept
        const int idx = keep - 1, src = idx & 63;
        uint64_t ke = k[0];
        if (NR > 1 && (idx >> 6) == 1) ke = k[NR > 1 ? 1 : 0];
        const int lo = __shfl(static_cast<int>(static_cast<uint32_t>(ke)), src, 64);
        const int hi = __shfl(static_cast<int>(static_cast<uint32_t>(ke >> 32)), src, 64);
        const float t = skr::key_score((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo));
        if (!(fabsf(t) >= 524288.0f * *a.guard_s_inv) && lane == 0) a.flag_list[atomicAdd(a.flag_count, 1)] = static_cast<int32_t>(out_row);
    }
}

// The final compactions of a wavefront's users, software-pipelined: the loads of user ul + 1 (candidate list and train
// row) are in flight while user ul is masked and sorted -- run one after the other a user costs two dependent memory
// round trips, 64 times per wavefront, at a point where every wavefront of the chip is doing the same and no MFMA runs.
template <int NR>
__device__ __forceinline__ void final_compactions_n(const FusedArgs& a, const WaveCtx& w) {
    const int64_t left = static_cast<int64_t>(rows_of(a)) - w.ubase;
    const int n_users = left < FE_UW ? static_cast<int>(left) : FE_UW;
    if (n_users <= 0) return;
    __threadfence_block();   // the wave's appends, re-read below by other lanes of the same wave
    uint64_t kn[NR];
    RowRegs rn;
    int nn = 0;
    auto fetch = [&](int ul) {
        const int n = w.cnt[ul];   // wave-uniform
        nn = n;
        const uint64_t* __restrict__ list = w.my_cand + static_cast<int64_t>(ul) * a.cap;
        rn = prefetch_row(a, w, ul);
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            const int idx = e * 64 + w.lane;
            kn[e] = (idx < n) ? list[idx] : SKR_KEY_MIN;
        }
    };
    fetch(0);
    for (int ul = 0; ul < n_users; ++ul) {
        uint64_t k[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) k[e] = kn[e];
        const RowRegs rr = rn;
        const int n = nn;
        if (ul + 1 < n_users) fetch(ul + 1);
        if (n <= 64 && a.top_k <= 64) {   // the common case at top-10: a 64-key sort instead of a 128-key one
            uint64_t k1[1] = {k[0]};
            final_user<1>(a, w, ul, k1, rr, src_row(a, w.ubase + ul));
        } else {
            final_user<NR>(a, w, ul, k, rr, src_row(a, w.ubase + ul));
        }
    }
}
__device__ __forceinline__ void final_compactions(const FusedArgs& a, const WaveCtx& w) {
    // a list never holds more than trigger + 32 entries (the compaction rule)
    if (a.trigger + FE_TI <= 128) final_compactions_n<2>(a, w);
    else final_compactions_n<4>(a, w);
}

// smallest real key among the NR*64 held by the wave (the K-th best when exactly K are real)
template <int NR>
__device__ __forceinline__ uint64_t kth_min_key(const uint64_t (&k)[NR], int lane) {
    (void)lane;
    uint64_t m = ~0ull;
#pragma unroll
    for (int e = 0; e < NR; ++e)
        if (k[e] != SKR_KEY_MIN && k[e] < m) m = k[e];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = shfl_xor_u64(m, d);
        m = o < m ? o : m;
    }
    return m;
}

// Mid-sweep compaction: only the K best entries and the K-th best score are needed, not their order, so
// this SELECTS instead of sorting.  The K-th largest key is built bit by bit from the top: a candidate
// value is kept if at least K keys are >= it.  The counts are v_cmp + s_bcnt1 on the wave's mask, i.e.
// NR vector compares per step and the rest on the scalar unit (the sort costs ~10 VALU instructions per
// register and stage, 36 stages for 256 keys).  Keys of one user are distinct (an item is scored once),
// so exactly K keys are >= the result.  Survivors are written back unordered; the final compaction sorts.
template <int NR>
__device__ __forceinline__ float compact_select_n(const FusedArgs& a, const WaveCtx& w, int ul, int n) {
    const int lane = w.lane;
    uint64_t* __restrict__ list = w.my_cand + static_cast<int64_t>(ul) * a.cap;
    int* cnt_p = &w.cnt[ul];
    uint64_t k[NR];
    const RowRegs rr = prefetch_row(a, w, ul);
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const int idx = e * 64 + lane;
        k[e] = (idx < n) ? list[idx] : SKR_KEY_MIN;
    }
    mask_train_lockstep<NR>(a, w, ul, k, rr);
    if (a.ablate == 6) mask_train_lockstep<NR>(a, w, ul, k, rr);
    // every load of this compaction has been consumed: a wait the compiler can SEE costs nothing here and tells its waitcnt pass
    // that only stores are in flight from now on (without it hipcc's waitcnt pass carries the loads around the sweep loop and drains the DMA ring in
    // front of every tile); fused_topk_kernel_v7 relies on this one and does not wait for the survivors' stores to be acknowledged
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), gfx9 encoding
    const int K = a.top_k;
    int valid = 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) valid += __popcll(__ballot(k[e] != SKR_KEY_MIN));
    uint64_t T = 1;   // keep every real key
    if (valid > K) {
        // score word first (32 steps on 32-bit compares) ...
        uint32_t hi[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) hi[e] = static_cast<uint32_t>(k[e] >> 32);
        uint32_t th = 0;
        for (int rep = (a.ablate == 5 ? 2 : 1); rep > 0; --rep) {
            th = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = th | (1u << bit);
                int c = 0;
#pragma unroll
                for (int e = 0; e < NR; ++e) c += __popcll(__ballot(hi[e] >= cand));
                if (c >= K) th = cand;
            }
        }
        int c_ge = 0, c_gt = 0;
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            c_ge += __popcll(__ballot(hi[e] >= th));
            c_gt += __popcll(__ballot(hi[e] > th));
        }
        uint32_t tl = 0;
        if (c_ge > K) {   // ... then, only if equal scores straddle the K-th place, the id word among them
            const int need = K - c_gt;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = tl | (1u << bit);
                int c = 0;
#pragma unroll
                for (int e = 0; e < NR; ++e) c += __popcll(__ballot(hi[e] == th && static_cast<uint32_t>(k[e]) >= cand));
                if (c >= need) tl = cand;
            }
        }
        T = (static_cast<uint64_t>(th) << 32) | tl;
    }
    int base = 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const bool p = (k[e] >= T);
        const uint64_t m = __ballot(p);
        if (p) list[base + __popcll(m & ((1ull << lane) - 1ull))] = k[e];
        base += __popcll(m);
    }
    if (lane == 0) *cnt_p = base;
    return (base == K && valid >= K) ? skr::key_score(valid > K ? T : kth_min_key<NR>(k, lane)) : -INFINITY;
}

// mid-sweep compaction of one list; returns the user's new threshold
__device__ __forceinline__ float compact_user(const FusedArgs& a, const WaveCtx& w, int ul, int64_t /*unused: mid-sweep only*/) {
    if (a.ablate != 20) __threadfence_block();  // this wave's earlier appends are re-read below by other lanes of the same wave (same CU, same L1: in order); 20: timing experiment without it
    const int n = w.cnt[ul];   // wave-uniform
    float thr;
    if (n <= 64) thr = compact_select_n<1>(a, w, ul, n);
    else if (n <= 128) thr = compact_select_n<2>(a, w, ul, n);
    else thr = compact_select_n<4>(a, w, ul, n);
    __threadfence_block();
    return thr;
}

// ------------------------------------------------------------------------------------------------
// Building blocks of the sweep kernel
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA issued from inline asm: the compiler's waitcnt pass must not see it, otherwise it treats every
// later ds_read as a possible reader of the DMA's destination and drains vmcnt(0) in front of it
// (measured: the wave waited for the tile it had just requested, ~2000 cycles per tile).  The wait for
// the DMA is placed by hand (FE2_WAIT) half a tile later.  M0 carries the wave-uniform LDS address;
// the hardware adds lane * size.
__device__ __forceinline__ void glds_b128(const void* gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ void glds_b32(const void* gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(lds_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(reinterpret_cast<uintptr_t>((lds_ptr_t)p))));
}


// candidate path for one finished tile: append every score above the user's threshold, then compact
// the lists that came within one tile of their capacity
__device__ __forceinline__ void tile_candidates(const FusedArgs& a, const WaveCtx& w, const f32x16& acc0,
                                                const f32x16& acc1, int tile_base, float (&thr)[2], const int (&uid)[2]) {
    // which of this lane's 16 rows are real items (only the last, partial tile has holes)
    uint32_t valid = 0xffffu;
    if (tile_base + FE_TI > a.n_items) {
        valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (tile_base + (r & 3) + 8 * (r >> 2) + 4 * w.h < a.n_items) valid |= 1u << r;
    }
    // one LDS atomic per lane and fragment reserves the slots of all its passing scores at once
    uint32_t m[2];
    int base[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const f32x16& acc = f ? acc1 : acc0;
        uint32_t mm = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mm |= (acc[r] > thr[f]) ? (1u << r) : 0u;
        m[f] = mm & valid;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = __popc(m[f]);
        base[f] = n ? atomicAdd(&w.cnt[32 * f + w.c], n) : 0;  // < cap by the compaction rule
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const f32x16& acc = f ? acc1 : acc0;
        uint64_t* list = w.my_cand + static_cast<int64_t>(32 * f + w.c) * a.cap + base[f];
        if (m[f]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (m[f] & (1u << r)) {
                    const int item = tile_base + (r & 3) + 8 * (r >> 2) + 4 * w.h;
                    list[__popc(m[f] & ((1u << r) - 1u))] = skr::rank_key(acc[r], item);
                }
            }
        }
    }
    uint64_t need = __ballot(w.cnt[w.lane] > a.trigger);
    while (need) {
        const int ul = __ffsll(static_cast<long long>(need)) - 1;
        need &= need - 1;
        const int f = ul >> 5;
        const float nt = compact_user(a, w, ul, -1);
        if (w.c == (ul & 31)) {
            if (f) thr[1] = nt; else thr[0] = nt;
        }
    }
}

// ================================================================================================
// The sweep kernel.  History (profiles/r01_eval_history.txt): v1 staged tiles through registers
// (50 % of FP32-MFMA peak: hipcc hoisted the ds_writes and their vmcnt waits into the MFMA chain);
// v2 switched to LDS-DMA and a second accumulator set; this version pins every auxiliary instruction
// into an MFMA gap.
//   One tile = two chains of 16 "slots"; a slot is the two MFMAs (one per user fragment) of one k-step
//   and owns a few independent instructions that issue while the matrix pipe is busy:
//     chain 1 (k 0..15, operands avLo):  slot j < 8: LDS-DMA j of tile t+1 (scalar base + per-lane constant
//                                        offset: no address arithmetic), slot 8: bias DMA, every slot: the
//                                        threshold test of one accumulator row of tile t-1;
//     chain 2 (k 16..31, operands avHi): slot 11: wait for the DMA, read avLo of tile t+1.
//   avHi of tile t is read at the top of the step; the bias seeds the accumulators as the C operand of the
//   first MFMA pair.  Only tiles whose successor is a full tile take this path; the last one or two tiles
//   of a sweep (and catalogues smaller than 64 items) go through the generic step below.
// ================================================================================================
__device__ __forceinline__ void glds_b128_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ void glds_b32_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
}

#define FE2_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define FE3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)
#define FE3_PIN() __builtin_amdgcn_sched_barrier(0)

template <bool HAS_BIAS>
__global__ __launch_bounds__(FE_WAVES * 64, 2) void fused_topk_kernel_v3(FusedArgs a) {
    __shared__ float4 s_tile0[FE_WAVES][FE_TI * FE_D / 4];  // wave-private, 8 KB per wave and buffer
    __shared__ float4 s_tile1[FE_WAVES][FE_TI * FE_D / 4];
    __shared__ float4 s_bias0[FE_WAVES][16];
    __shared__ float4 s_bias1[FE_WAVES][16];
    __shared__ int s_cnt[FE_WAVES][FE_UW];
    __shared__ int64_t s_row_beg[FE_WAVES][FE_UW];
    __shared__ int s_row_len[FE_WAVES][FE_UW];
    __shared__ int s_rowbuf[FE_WAVES][FE_ROWBUF];
    WaveCtx w;
    w.lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    w.c = w.lane & 31;
    w.h = w.lane >> 5;
    const int lane = w.lane, c = w.c, h = w.h;
    w.ubase = (static_cast<int64_t>(blockIdx.x) * FE_WAVES + wv) * FE_UW;
    if (w.ubase >= a.B) return;
    w.cnt = s_cnt[wv];
    w.cnt[lane] = 0;
    w.my_cand = a.cand + w.ubase * a.cap;
    w.row_beg = s_row_beg[wv];
    w.row_len = s_row_len[wv];
    w.rowbuf = s_rowbuf[wv];
    w.rowbuf_len = FE_ROWBUF;
    {   // lane l caches the train-row extent of the wave's l-th user
        const int64_t row = w.ubase + lane;
        int64_t rb = 0;
        int len = 0;
        if (a.train_rowptr && row < a.B) {
            const int u = a.users[row];
            rb = a.train_rowptr[u];
            len = static_cast<int>(a.train_rowptr[u + 1] - rb);
        }
        w.row_beg[lane] = rb;
        w.row_len[lane] = len;
    }

    float bf[2][32];
    float thr[2];
    int uid[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t row = w.ubase + 32 * f + c;
        const bool ok = row < a.B;
        uid[f] = a.users[ok ? row : (a.B - 1)];
        thr[f] = (ok && a.ablate != 1) ? -INFINITY : INFINITY;
        const float4* up = reinterpret_cast<const float4*>(a.user_table + static_cast<int64_t>(uid[f]) * FE_D + 32 * h);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = up[q];
            bf[f][4 * q + 0] = v.x; bf[f][4 * q + 1] = v.y; bf[f][4 * q + 2] = v.z; bf[f][4 * q + 3] = v.w;
        }
    }
    // The B-fragment loads must be retired HERE: hipcc cannot see the hand-issued LDS-DMA below, so any
    // counted `s_waitcnt vmcnt(N)` it would otherwise place inside the loop for these loads ends up
    // waiting on the DMA instead (measured in the ISA: vmcnt(6..0) in the middle of the chains).
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 32; ++q) asm volatile("" : "+v"(bf[f][q]));
    const int n_tiles = (a.n_items + FE_TI - 1) / FE_TI;
    const int n_full = a.n_items / FE_TI;

    // per-lane byte offset of DMA instruction j inside a tile (constant over the sweep) and LDS targets
    uint32_t doff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rl = 4 * j + (lane >> 4);
        doff[j] = static_cast<uint32_t>((rl * FE_D + 4 * ((lane & 15) ^ (rl & 15))) * 4);
    }
    const uint32_t boff = static_cast<uint32_t>((lane & 31) * 4);
    const uint32_t lt0 = lds_addr_of(&s_tile0[wv][0]), lt1 = lds_addr_of(&s_tile1[wv][0]);
    const uint32_t lb0 = lds_addr_of(&s_bias0[wv][0]), lb1 = lds_addr_of(&s_bias1[wv][0]);

    // generic (clamped, per-lane pointer) DMA of tile T into LDS addresses LT / LB
    auto issue_generic = [&](int T, uint32_t LT, uint32_t LB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = 4 * j + (lane >> 4);
            int item = T * FE_TI + rl;
            item = item < a.n_items ? item : a.n_items - 1;
            const int piece = (lane & 15) ^ (rl & 15);
            glds_b128(a.item_table + static_cast<int64_t>(item) * FE_D + 4 * piece, LT + j * 1024);
        }
        if (HAS_BIAS) {
            int bi = T * FE_TI + (lane & 31);
            bi = bi < a.n_items ? bi : a.n_items - 1;
            glds_b32(a.item_bias + bi, LB);
        }
    };
#define FE3_READ_HALF(DST, TILEPTR, HALF)                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
        const float4 v_ = (TILEPTR)[c * 16 + ((8 * h + 4 * (HALF) + q) ^ (c & 15))];            \
        DST[4 * q + 0] = v_.x; DST[4 * q + 1] = v_.y; DST[4 * q + 2] = v_.z; DST[4 * q + 3] = v_.w; \
    }
#define FE3_SEED(SEED, BIASPTR)                                                                 \
    if (HAS_BIAS) {                                                                             \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                         \
            const float4 b4_ = (BIASPTR)[2 * g + h];                                            \
            SEED[4 * g + 0] = b4_.x; SEED[4 * g + 1] = b4_.y; SEED[4 * g + 2] = b4_.z; SEED[4 * g + 3] = b4_.w; \
        }                                                                                       \
    } else {                                                                                    \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) SEED[r] = 0.0f;                          \
    }

    // fast step: tile T (operands in CT/CB, avLo preloaded), successor T+1 is a FULL tile -> NT/NB
#define FE3_FAST_STEP(C0, C1, P0, P1, T, CT, CB, NT, NB, LNT, LNB)                                       \
    {                                                                                                    \
        const int t_ = (T);                                                                              \
        const char* nb_ = reinterpret_cast<const char*>(a.item_table) + static_cast<int64_t>(t_ + 1) * (FE_TI * FE_D * 4); \
        const char* nbb_ = reinterpret_cast<const char*>(a.item_bias) + static_cast<int64_t>(t_ + 1) * (FE_TI * 4); \
        f32x16 seed_;                                                                                    \
        FE3_SEED(seed_, CB[wv])                                                                          \
        FE3_READ_HALF(avHi, CT[wv], 1)                                                                   \
        bool any_ = false;                                                                               \
        FE3_PIN();                                                                                       \
        _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                              \
            if (s_ == 0) {                                                                               \
                C0 = FE3_MFMA(avLo[0], bf[0][0], seed_);                                                 \
                C1 = FE3_MFMA(avLo[0], bf[1][0], seed_);                                                 \
            } else {                                                                                     \
                C0 = FE3_MFMA(avLo[s_], bf[0][s_], C0);                                                  \
                C1 = FE3_MFMA(avLo[s_], bf[1][s_], C1);                                                  \
            }                                                                                            \
            if (s_ < 8) glds_b128_s(doff[s_], nb_, (LNT) + s_ * 1024);                                   \
            if (HAS_BIAS && s_ == 8) glds_b32_s(boff, nbb_, (LNB));                                      \
            any_ |= (P0[s_] > thr[0]) | (P1[s_] > thr[1]);                                               \
            FE3_PIN();                                                                                   \
        }                                                                                                \
        _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                              \
            C0 = FE3_MFMA(avHi[s_], bf[0][16 + s_], C0);                                                 \
            C1 = FE3_MFMA(avHi[s_], bf[1][16 + s_], C1);                                                 \
            if (s_ == 11) {                                                                              \
                FE2_WAIT();                                                                              \
                FE3_READ_HALF(avLo, NT[wv], 0)                                                           \
            }                                                                                            \
            FE3_PIN();                                                                                   \
        }                                                                                                \
        if (__any(any_)) tile_candidates(a, w, P0, P1, (t_ - 1) * FE_TI, thr, uid);                      \
    }

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = -INFINITY;  // "tile -1": nothing passes
    f32x16 accA1 = accB, accB1 = accB;
    float avLo[16], avHi[16];
    issue_generic(0, lt0, lb0);
    FE2_WAIT();
    int t = 0;
    const int n_fast = n_full - 1;  // tiles 0 .. n_fast-1 have a full successor
    if (n_fast >= 2) {
        FE3_READ_HALF(avLo, s_tile0[wv], 0)
        for (; t + 2 <= n_fast; t += 2) {  // even tiles: s_tile0 / accA, odd tiles: s_tile1 / accB
            FE3_FAST_STEP(accA, accA1, accB, accB1, t, s_tile0, s_bias0, s_tile1, s_bias1, lt1, lb1)
            FE3_FAST_STEP(accB, accB1, accA, accA1, t + 1, s_tile1, s_bias1, s_tile0, s_bias0, lt0, lb0)
        }
    }
    // generic tail (t is even here; tile t sits in buffer t & 1; previous tile's scores are in accB)
    for (; t < n_tiles; ++t) {
        const float4* ct = (t & 1) ? s_tile1[wv] : s_tile0[wv];
        const float4* cb = (t & 1) ? s_bias1[wv] : s_bias0[wv];
        if (t + 1 < n_tiles) issue_generic(t + 1, (t & 1) ? lt0 : lt1, (t & 1) ? lb0 : lb1);
        f32x16 seed;
        FE3_SEED(seed, cb)
        FE3_READ_HALF(avLo, ct, 0)
        FE3_READ_HALF(avHi, ct, 1)
        accA = seed;
        accA1 = seed;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            accA = FE3_MFMA(avLo[s], bf[0][s], accA);
            accA1 = FE3_MFMA(avLo[s], bf[1][s], accA1);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            accA = FE3_MFMA(avHi[s], bf[0][16 + s], accA);
            accA1 = FE3_MFMA(avHi[s], bf[1][16 + s], accA1);
        }
        if (t > 0) tile_candidates(a, w, accB, accB1, (t - 1) * FE_TI, thr, uid);
        FE2_WAIT();
        accB = accA;
        accB1 = accA1;
    }
    tile_candidates(a, w, accB, accB1, (n_tiles - 1) * FE_TI, thr, uid);
#undef FE3_FAST_STEP
#undef FE3_SEED
#undef FE3_READ_HALF
    final_compactions(a, w);
}

// ================================================================================================
// The split arithmetics (SKR_FUSED_MODE=bf16x3 / f16x2; fp32 selects the kernel above): the same sweep with every fp32 operand
// split into 16-bit pieces and a product formed from half-precision MFMA products with fp32 accumulation.
// bf16x3: x = hi + mid + lo in bf16 (exact: 3 x 8 significand bits), the six piece products of weight >= 2^-16 (hi*hi, hi*mid,
// mid*hi, mid*mid, hi*lo, lo*hi; the dropped mid*lo, lo*mid, lo*lo are <= 2^-23 relative): six bf16 MFMAs of 16x the fp32 MFMA
// rate replace one fp32 MFMA chain, at fp32-level accuracy for ANY operands (products are exact in fp32; the error is the dropped
// terms plus fp32 accumulation: 24 roundings per dot product, the fp32 chain has 64).  f16x2 (fused_topk_kernel_v7 below): two fp16
// pieces of the operands scaled by a power of two per table, three products, behind a guard.  Measured against float64
// (tools/fused_accuracy.py, error / sum|u_i v_i| over the returned top-50 scores of 512 users x 20 000 items, factor scales
// 1e-3 .. 30): f16x2 max 2.1e-7 / mean 2.8e-8, bf16x3 2.6e-7 / 3.1e-8, the FP32-MFMA kernel 3.5e-7 / 5.1e-8.
// Common mechanics: a split kernel writes the item table once per call in FRAGMENT order -- 1 KB blocks in which lane l's 16
// bytes are the 8 pieces the MFMA wants from it -- so LDS-DMA copies blocks verbatim and ds_read_b128 at lane*16 is
// conflict-free without a swizzle; the user fragments are split in registers at kernel start.
// History (profiles/r01_eval_history.txt): rounds 1 and 2 ran bf16x3 on v_mfma_f32_32x32x16_bf16 -- fused_topk_kernel_v4
// (a ring of half tiles per wavefront, no barriers: 18.1 ms per 262 144 users at top-10) and fused_topk_kernel_v5 (one ring per
// workgroup, a barrier per half-tile step: 17.3 ms); fused_topk_kernel_v6 (16x16x32, 16.3 ms) is faster at every top_k, and the
// two were taken out in round 3 (git history has them).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rne(float x) {   // round-to-nearest-even on the bits (finite input)
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = bf16_rne(x);
    const float r1 = x - __uint_as_float(hi << 16);
    mid = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(mid << 16);
    lo = bf16_rne(r2);
}
// eight consecutive floats -> three uint4 of packed bf16 (element j in bits [16 (j&1), +16) of word j>>1)
__device__ __forceinline__ void split3x8(const float* v, uint4& hi, uint4& mid, uint4& lo) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    mid = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

constexpr int F4_FRAGS = 12;                 // 4 chunks of 16 dims x 3 pieces
constexpr int F4_TILE_U4 = F4_FRAGS * 64;    // uint4 per tile (12 KB): fragments q*3+p, i.e. two 6 KB halves

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4& u) {
    union { uint4 u; bf16x8 b; } c;
    c.u = u;
    return c.b;
}
constexpr int F4_HALF_U4 = 6 * 64;      // uint4 per half tile (two 16-dim chunks x 3 pieces = 6 KB)
constexpr int F4_ROWBUF = 128;

constexpr int F5_RING = 3;              // item groups resident per WORKGROUP (the counted waits and the bias rows assume 3)

// ================================================================================================
// fused_topk_kernel_v6 (round 3): the bf16x3 sweep on v_mfma_f32_16x16x32_bf16, one step = a GROUP of 16 items over all 64
// dims.  Why another shape: under bf16 MFMA load the chip holds its clock down (1.7 GHz in the 32x32x16 kernels' loop, MFMA pipe busy 75 %
// of it), so cycles are not what the wall time is made of; MI355X_MICROARCH.md ("DVFS give-back", item 7) measures the
// 16x16x32 form at 1.12-1.15 x the FLOP/s of the 32x32x16 form at equal cycles on random data.  The shape also turns the
// schedule round:
//   * an output tile is 16 items x 16 users with FOUR accumulator registers; a wavefront's 64 users are four user groups,
//     so a step (16 items x 64 users x 64 dims = 2 k-steps x 6 piece products x 4 user groups = 48 MFMAs of 16 cycles, the
//     same matrix-pipe time as a 32-item half tile of the 32x32x16 form) FINISHES 16 accumulator registers, and two such sets
//     alternate in the 32 registers one 32-item tile needs.  The threshold tests of step hs-1 therefore issue between the MFMAs of step hs
//     (one v_cmp + one scalar OR per accumulator register, pinned two per slot), with no second accumulator pair: what
//     VERDICT round 2, item 4 asked for without the 32 VGPRs it was priced at;
//   * lane l holds, per user group g, the scores of user 16 g + (l & 15) against items 4 (l >> 4) + i, i < 4: a user's
//     16 scores of a step sit in four lanes; the list lengths live in LDS and a passing lane reserves its slot with one
//     LDS atomic (group_candidates_v6; a first form carried the lengths in registers and crossed the four lanes' counts
//     by v_permlane16_swap / v_permlane32_swap: correct, ~4x the vector instructions per event);
//   * around the arithmetic: ONE ring of three 6 KB blocks per workgroup (a block = one item group:
//     2 k-steps x 3 pieces), brought by the four wavefronts together, one workgroup barrier per step, candidate lists in
//     HBM scratch, the same compaction code.
// split_items_kernel_v6 writes the table in this kernel's fragment order: group (16 items) x k-step (32 dims) x piece ->
// 1 KB blocks in which lane l's 16 bytes are A[row l & 15][k = 8 (l >> 4) + j].
// ================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define F6_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(A), as_bf16x8(B), C, 0, 0, 0)
constexpr int F6_GI = 16;               // items per step

__global__ __launch_bounds__(256) void split_items_kernel_v6(const float* __restrict__ table, int n_items, int n_tiles,
                                                             uint4* __restrict__ frags) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;   // (group, k-step, lane)
    if (g >= static_cast<int64_t>(n_tiles) * 2 * 2 * 64) return;
    const int lane = static_cast<int>(g & 63), ks = static_cast<int>((g >> 6) & 1);
    const int64_t G = g >> 7;
    int64_t item = G * F6_GI + (lane & 15);
    if (item >= n_items) item = n_items - 1;
    const float4* src = reinterpret_cast<const float4*>(table + item * FE_D + ks * 32 + 8 * (lane >> 4));
    const float4 v0 = src[0], v1 = src[1];
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint4 hi, mid, lo;
    split3x8(v, hi, mid, lo);
    uint4* dst = frags + G * F4_HALF_U4 + (ks * 3) * 64 + lane;
    dst[0] = hi;
    dst[64] = mid;
    dst[128] = lo;
}

// Candidate path of one finished item group, taken only when some score of the wavefront passed its threshold.  The
// sixteen threshold tests were made between the MFMAs (one lane mask per accumulator register); here every register
// with a passing lane is handled under that mask: the lane reserves a slot of its user's list with ONE LDS atomic on the
// list length (up to four lanes hold scores of the same user) and stores the key.  Rows of the last tile that are not
// items carry -inf (the kernel seeds them so) and never pass.  No per-lane bit masks, no counts
// carried between lanes, no loops: ~10 vector instructions per register that has an event, none for the others.
// (Tried and dropped: RAW list entries -- score bits and item id stored as they are, rank_key applied on the load side of a
// compaction where 64 lanes are active instead of here where one or two are: slower at every top_k, 1.023 vs 0.990 of the
// older kernel's time at top-100.)
// (Tried and dropped: counting the key stores left in flight and adding them to the counted vmcnt waits of the tile DMA,
// which otherwise also wait for those younger stores -- no gain, the extra branches cost what the shorter waits saved.)
__device__ __forceinline__ void group_candidates_v6(const FusedArgs& a, const WaveCtx& w, const f32x4 (&acc)[4],
                                                    const bool (&pass)[4][4], int base, float (&thr)[4]) {
    const int qd = w.lane >> 4, c16 = w.lane & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int* cnt_p = &w.cnt[16 * g + c16];
        uint64_t* list = w.my_cand + static_cast<int64_t>(16 * g + c16) * a.cap;
        unsigned pos[4];   // unsigned: see group_candidates_v7
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (pass[g][i]) pos[i] = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(cnt_p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (pass[g][i]) list[pos[i]] = skr::rank_key(acc[g][i], base + 4 * qd + i);   // pos < trigger + 16 <= cap by the compaction rule
    }
    // the wavefront's LDS operations are executed in order: the lengths read here include every reservation above
    uint64_t need = __ballot(w.cnt[w.lane] > a.trigger);
    if (need) {
        while (need) {
            const int ul = __ffsll(static_cast<long long>(need)) - 1;
            need &= need - 1;
            const float nt = compact_user(a, w, ul, -1);
            const int ug = ul >> 4;   // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g == ug && c16 == (ul & 15)) thr[g] = nt;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to the compiler (see compact_select_n)
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(FE_WAVES * 64, 2) void fused_topk_kernel_v6(FusedArgs a, const uint4* __restrict__ frags) {
    __shared__ uint4 s_tile[F5_RING * F4_HALF_U4];             // ONE ring of F5_RING item groups for the workgroup's four wavefronts
    __shared__ float4 s_bias[3][16];                           // a tile's bias row (the DMA writes 4 B for each of the 64 lanes: 32 items, twice);
                                                               // three rows: tile t+2's arrives while t's is read in both of t's steps
    __shared__ int s_cnt[FE_WAVES][FE_UW];
    __shared__ int64_t s_row_beg[FE_WAVES][FE_UW];
    __shared__ int s_row_len[FE_WAVES][FE_UW];
    __shared__ int s_rowbuf[FE_WAVES][F4_ROWBUF];
    WaveCtx w;
    w.lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: it selects DMA blocks (scalar operands)
    w.c = w.lane & 31;
    w.h = w.lane >> 5;
    const int lane = w.lane, c16 = w.lane & 15, qd = w.lane >> 4;
    w.ubase = (static_cast<int64_t>(blockIdx.x) * FE_WAVES + wv) * FE_UW;
    const int nB = rows_of(a);   // a.B, or (the fall-back launch over the rows the fp16x2 sweep flagged) a count read here
    if (static_cast<int64_t>(blockIdx.x) * FE_WAVES * FE_UW >= nB) return;   // the whole workgroup: nothing to do
    // a wavefront whose users lie beyond B stays: it carries its share of the DMA and of the workgroup barriers
    w.cnt = s_cnt[wv];
    w.cnt[lane] = 0;
    w.my_cand = a.cand + w.ubase * a.cap;
    w.row_beg = s_row_beg[wv];
    w.row_len = s_row_len[wv];
    w.rowbuf = s_rowbuf[wv];
    w.rowbuf_len = F4_ROWBUF;
    {
        const int64_t row = w.ubase + lane;
        int64_t rb = 0;
        int len = 0;
        if (a.train_rowptr && row < nB) {
            const int u = a.users[src_row(a, row)];
            rb = a.train_rowptr[u];
            len = static_cast<int>(a.train_rowptr[u + 1] - rb);
        }
        w.row_beg[lane] = rb;
        w.row_len[lane] = len;
    }
    // user fragments: B[k = 8 qd + j][col c16] of user group g and k-step ks, three pieces each
    uint4 bh[4][2], bm[4][2], bl[4][2];
    float thr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t row = w.ubase + 16 * g + c16;
        const bool ok = row < nB;
        const int uid = a.users[src_row(a, ok ? row : (nB - 1))];
        thr[g] = (ok && a.ablate != 1 && a.ablate != 7 && a.ablate != 11 && a.ablate != 12) ? -INFINITY : INFINITY;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4* up = reinterpret_cast<const float4*>(a.user_table + static_cast<int64_t>(uid) * FE_D + ks * 32 + 8 * qd);
            const float4 v0 = up[0], v1 = up[1];
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            split3x8(v, bh[g][ks], bm[g][ks], bl[g][ks]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // retire the loads here (see the fp32 kernel: hidden DMA vs counted vmcnt)
            asm volatile("" : "+v"(bh[g][ks].x), "+v"(bh[g][ks].y), "+v"(bh[g][ks].z), "+v"(bh[g][ks].w));
            asm volatile("" : "+v"(bm[g][ks].x), "+v"(bm[g][ks].y), "+v"(bm[g][ks].z), "+v"(bm[g][ks].w));
            asm volatile("" : "+v"(bl[g][ks].x), "+v"(bl[g][ks].y), "+v"(bl[g][ks].z), "+v"(bl[g][ks].w));
        }
    const int n_tiles = (a.n_items + FE_TI - 1) / FE_TI;
    const int n_half = 2 * n_tiles;                 // item groups (the second one of the last tile may lie wholly beyond n_items)
    const uint32_t lt = lds_addr_of(&s_tile[0]);
    const uint32_t lb0 = lds_addr_of(&s_bias[0][0]);
    const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
    auto issue_half = [&](int hs, int slot, int brow) {   // group hs -> ring slot; with a tile's first group travels the tile's bias row (-> s_bias[brow])
        const char* sbase = reinterpret_cast<const char*>(frags) + static_cast<int64_t>(hs) * (F4_HALF_U4 * 16);   // wave-uniform
        const uint32_t dst = lt + slot * (F4_HALF_U4 * 16);
        glds_b128_s(lane16, sbase + wv * 1024, dst + wv * 1024);
        if (wv < 2) glds_b128_s(lane16, sbase + (4 + wv) * 1024, dst + (4 + wv) * 1024);
        if (HAS_BIAS && !(hs & 1) && wv == 3) {
            int bi = (hs >> 1) * FE_TI + (lane & 31);
            bi = bi < a.n_items ? bi : a.n_items - 1;
            glds_b32(a.item_bias + bi, lb0 + static_cast<uint32_t>(brow) * 256u);
        }
    };
#pragma unroll
    for (int h0 = 0; h0 < F5_RING; ++h0)
        if (h0 < n_half) issue_half(h0, h0, h0 >> 1);   // n_half >= 2 always
    FE2_WAIT();
    __syncthreads();                                // every wavefront's blocks have landed
    uint4 afA[6], afB[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) afA[i] = s_tile[i * 64 + lane];
    f32x4 accA[4], accB[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) accB[g][i] = -INFINITY;   // "group -1": nothing passes
    int slot = 0;                                   // ring slot of the group held in afA at an even step
    int brow = 0, brow2 = 2;                        // bias rows of tile t and of tile t + 2 (t mod 3)
    // one step = one item group.  Prologue as in v5 (fragments of group hs in registers -> its slot takes the DMA of group
    // hs+3; the fragments of hs+1 are fetched behind a counted wait); then 12 slots of four MFMAs (one piece product on
    // the four user groups), slots 0..7 each carrying two threshold tests of group hs-1.
#define F6_SLOT(S, AF, BP, ACC, PRV)                                                                          \
    {                                                                                                         \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                                      \
            ACC[g_] = F6_MFMA(AF, BP[g_][(S) / 6], (S) == 0 ? seed_ : ACC[g_]);                               \
        if ((S) < 8) {                                                                                        \
            const int i_ = (S) & 3, gp_ = (((S) & 7) >> 2) * 2;                                               \
            pass_[gp_][i_] = PRV[gp_][i_] > thr[gp_];                                                         \
            pass_[gp_ + 1][i_] = PRV[gp_ + 1][i_] > thr[gp_ + 1];                                             \
            any_ |= pass_[gp_][i_] | pass_[gp_ + 1][i_];                                                      \
        }                                                                                                     \
        FE3_PIN();                                                                                            \
    }
#define F6_STEP(CUR, NXT, HS, ACC, PRV, LAST)                                                                     \
    {                                                                                                         \
        const int hs_ = (HS);                                                                                 \
        FE3_PIN();                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* CUR has landed in registers */                \
        FE3_PIN();                                                                                            \
        const bool more3_ = hs_ + F5_RING < n_half && a.ablate != 7 && a.ablate != 11 && a.ablate != 12; \
        const int nslot_ = slot == F5_RING - 1 ? 0 : slot + 1;                                                \
        if (hs_ + 1 < n_half) {                                                                               \
            if (hs_ + 2 >= n_half) FE2_WAIT();                                                                \
            else if (wv < 2 || (HAS_BIAS && !(hs_ & 1) && wv == 3)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); \
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                             \
        }                                                                                                     \
        if (a.ablate != 11 && a.ablate != 19) __syncthreads();   /* 11, 19: timing experiments without the barrier (bare loop; whole kernel, results then unreliable) */ \
        FE3_PIN();                                                                                            \
        if (more3_) issue_half(a.ablate == 8 ? ((hs_ + F5_RING) & 31) : hs_ + F5_RING, slot, brow2);          \
        /* the bias read goes first: LDS answers in order, so the first MFMA waits for it alone (counted) */  \
        f32x4 seed_;                                                                                          \
        if (HAS_BIAS) {                                                                                       \
            const float4 b4 = s_bias[brow][4 * (hs_ & 1) + qd];                                               \
            seed_[0] = b4.x; seed_[1] = b4.y; seed_[2] = b4.z; seed_[3] = b4.w;                               \
        } else {                                                                                              \
            seed_[0] = 0.0f; seed_[1] = 0.0f; seed_[2] = 0.0f; seed_[3] = 0.0f;                               \
        }                                                                                                     \
        FE3_PIN();                                                                                            \
        /* unconditional (behind the last group it fetches a stale slot that nobody uses): a branch here would make  \
           hipcc wait for ALL LDS reads in front of the first MFMA */                                         \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) NXT[i] = s_tile[nslot_ * F4_HALF_U4 + i * 64 + lane];   \
        if (LAST) {   /* the last tile: rows beyond the catalogue start from -inf and stay there */            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
                if (hs_ * F6_GI + 4 * qd + i >= a.n_items) seed_[i] = -INFINITY;                              \
        }                                                                                                     \
        bool any_ = false;                                                                                    \
        bool pass_[4][4] = {};                                                                                \
        FE3_PIN();                                                                                            \
        /* small terms first, per k-step: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi */                     \
        F6_SLOT(0, CUR[2], bh, ACC, PRV)  F6_SLOT(1, CUR[0], bl, ACC, PRV)  F6_SLOT(2, CUR[1], bm, ACC, PRV)  \
        F6_SLOT(3, CUR[1], bh, ACC, PRV)  F6_SLOT(4, CUR[0], bm, ACC, PRV)  F6_SLOT(5, CUR[0], bh, ACC, PRV)  \
        F6_SLOT(6, CUR[5], bh, ACC, PRV)  F6_SLOT(7, CUR[3], bl, ACC, PRV)  F6_SLOT(8, CUR[4], bm, ACC, PRV)  \
        F6_SLOT(9, CUR[4], bh, ACC, PRV)  F6_SLOT(10, CUR[3], bm, ACC, PRV) F6_SLOT(11, CUR[3], bh, ACC, PRV) \
        if (__any(any_)) group_candidates_v6(a, w, PRV, pass_, (hs_ - 1) * F6_GI, thr);                       \
        slot = nslot_;                                                                                        \
    }
    for (int t = 0; t < n_tiles - 1; ++t) {
        F6_STEP(afA, afB, 2 * t, accA, accB, false)
        F6_STEP(afB, afA, 2 * t + 1, accB, accA, false)
        brow = brow == 2 ? 0 : brow + 1;
        brow2 = brow2 == 2 ? 0 : brow2 + 1;
    }
    F6_STEP(afA, afB, 2 * (n_tiles - 1), accA, accB, true)
    F6_STEP(afB, afA, 2 * (n_tiles - 1) + 1, accB, accA, true)
#undef F6_STEP
#undef F6_SLOT
    {
        bool any_ = false;
        bool pass_[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pass_[g][i] = accB[g][i] > thr[g];
                any_ |= pass_[g][i];
            }
        if (__any(any_)) group_candidates_v6(a, w, accB, pass_, (n_half - 1) * F6_GI, thr);
    }
    __threadfence_block();
    if (a.ablate == 12) return;   // timing experiment: bare loop without the final compactions
    final_compactions(a, w);
}

// ================================================================================================
// fused_topk_kernel_v7 (round 3, SKR_FUSED_MODE=f16x2): the same sweep with every fp32 operand split into TWO fp16 pieces.
//   x * s = hi + lo (s a power of two per table, chosen so that the table's largest |x| * s lies in [2^14, 2^15): exact),
//   hi = fp16(x s), lo = fp16(x s - hi); a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32
//   accumulation -- THREE matrix products per fp32 product instead of bf16x3's six, at the same rate per product.
// What it costs in accuracy, and how that is kept honest: hi + lo carries 22 significand bits of x s (fp32: 24) as long as
// lo is a normal fp16 (|x s| >= 2^-3, i.e. |x| within 2^-17 of the table's largest element); below that lo is a
// denormal with ABSOLUTE error <= 2^-25.  Measured against float64 on normal factors of scales 1e-3 .. 30 (numpy
// emulation and tools/fused_accuracy.py): max 1.2e-7 / mean 1.3e-8 of sum |u_i v_i| -- below the fp32 chain's 3.5e-7 /
// 2.0e-8; rows or elements of wildly different magnitude are where it degrades (an absolute floor of 2^-3 / (s_u s_v)
// per score).  So the kernel GUARDS its result: a user is accepted only if the smallest score of the returned list is
// at least 2^19 / (s_u s_v) in magnitude, i.e. the WORST-CASE floor is 2^-22 of that score -- the fp32 chain's own worst case
// is 64 roundings = 2^-18, its measured maximum 2^-21.4; the floor measured with the guard off is 0.02 of the bound
// (tools/f16x2_floor_probe.py), and the error starts to show at scores of 2^14 / (s_u s_v);
// every other user is written to a list on the device and recomputed by the bf16x3 kernel (fused_topk_kernel_v6 with a
// row map) in the same call, stream-ordered, no host round trip -- on factors of one magnitude (bench.py's, a trained
// model's) the list is empty and that launch returns at once.
// Mechanics: v6's, with 4 KB item groups (2 k-steps x 2 pieces; one 1 KB block per wavefront and step), 64 instead of
// 96 user-fragment registers, 24 MFMAs per step in 6 slots, thresholds held in the scaled domain (a compaction's
// threshold is multiplied by s_u s_v once), scores unscaled when an event is stored, the bias row pre-scaled by
// split_items_kernel_v7.
// ================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_f16x8(const uint4& u) {
    union { uint4 u; f16x8 h; } c;
    c.u = u;
    return c.h;
}
#define F7_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(A), as_f16x8(B), C, 0, 0, 0)
constexpr int F7_GROUP_U4 = 4 * 64;     // uint4 per item group (2 k-steps x {hi, lo} = 4 KB)

struct F7Scales {
    uint32_t max_u_bits, max_v_bits;   // bit patterns of the largest |x| of the evaluated user rows / of the item table
    int32_t n_flagged;                 // users the guard sent to the bf16x3 kernel
    int32_t pad;
    float s_u, s_v, S, S_inv;          // the power-of-two scales, their product and its inverse
};

// largest |x| over whole rows of FE_D floats (rows picked by ids when given): a fixed grid strides over the (row, 16-byte
// piece) pairs, one atomic per WORKGROUP (one per wavefront on a single address serialised: 0.29 / 0.75 ms for the two tables)
constexpr int ABSMAX_BLOCKS = 1024;
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                          int64_t n_rows, uint32_t* __restrict__ out_bits) {
    __shared__ uint32_t red[4];
    const int64_t n = n_rows * (FE_D / 4);
    uint32_t m = 0;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < n; t += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t r = t / (FE_D / 4);
        const int q = static_cast<int>(t - r * (FE_D / 4));
        const int64_t row = ids ? ids[r] : r;
        const float4 v = reinterpret_cast<const float4*>(table + row * FE_D)[q];
        const uint32_t a0 = __float_as_uint(v.x) & 0x7fffffffu, a1 = __float_as_uint(v.y) & 0x7fffffffu;
        const uint32_t a2 = __float_as_uint(v.z) & 0x7fffffffu, a3 = __float_as_uint(v.w) & 0x7fffffffu;
        m = max(m, max(max(a0, a1), max(a2, a3)));
    }
    if (m > 0x7f800000u) m = 0x7f800000u;   // NaN counts as inf: the guard will then reject everything
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, static_cast<uint32_t>(__shfl_xor(static_cast<int>(m), d, 64)));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m) atomicMax(out_bits, m);
    }
}

// s = 2^e with max * s in [2^14, 2^15); e clamped so that s_u * s_v stays a finite fp32 (tiny tables then fail the guard)
__device__ __forceinline__ float f7_scale_of(uint32_t max_bits) {
    int ex = static_cast<int>((max_bits >> 23) & 0xffu) - 127;
    if ((max_bits >> 23) == 0) ex = -126;        // zero / denormal maximum
    if (max_bits >= 0x7f800000u) ex = 127;       // inf / NaN
    int e = 14 - ex;
    e = e < -40 ? -40 : (e > 40 ? 40 : e);
    return ldexpf(1.0f, e);
}
__global__ void f7_scales_kernel(F7Scales* sc) {
    if (threadIdx.x == 0) {
        sc->s_u = f7_scale_of(sc->max_u_bits);
        sc->s_v = f7_scale_of(sc->max_v_bits);
        sc->S = sc->s_u * sc->s_v;
        sc->S_inv = 1.0f / sc->S;     // a power of two: exact
        sc->n_flagged = 0;
    }
}

// eight consecutive floats, scaled by a power of two -> hi and lo as packed fp16
__device__ __forceinline__ void split2x8(const float* v, float s, uint4& hi, uint4& lo) {
    union { _Float16 h[8]; uint4 u; } H, L;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * s;
        const _Float16 h = static_cast<_Float16>(xs);                      // round to nearest even
        H.h[j] = h;
        L.h[j] = static_cast<_Float16>(xs - static_cast<float>(h));        // the difference is exact in fp32
    }
    hi = H.u;
    lo = L.u;
}

// item table -> fragment order of this kernel (group of 16 items x k-step of 32 dims x {hi, lo} -> 1 KB blocks), and the
// bias row multiplied by s_u s_v (the accumulators live in the scaled domain)
__global__ __launch_bounds__(256) void split_items_kernel_v7(const float* __restrict__ table, const float* __restrict__ bias,
                                                             int n_items, int n_tiles, const F7Scales* __restrict__ sc,
                                                             uint4* __restrict__ frags, float* __restrict__ bias_s) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;   // (group, k-step, lane)
    if (g >= static_cast<int64_t>(n_tiles) * 2 * 2 * 64) return;
    const float s_v = sc->s_v;
    if (bias && g < static_cast<int64_t>(n_tiles) * FE_TI) bias_s[g] = (g < n_items ? bias[g] : 0.0f) * sc->S;
    const int lane = static_cast<int>(g & 63), ks = static_cast<int>((g >> 6) & 1);
    const int64_t G = g >> 7;
    int64_t item = G * F6_GI + (lane & 15);
    if (item >= n_items) item = n_items - 1;
    const float4* src = reinterpret_cast<const float4*>(table + item * FE_D + ks * 32 + 8 * (lane >> 4));
    const float4 v0 = src[0], v1 = src[1];
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint4 hi, lo;
    split2x8(v, s_v, hi, lo);
    uint4* dst = frags + G * F7_GROUP_U4 + (ks * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

// the bf16x3 fall-back's own split of the item table, skipped when the guard flagged nobody
__global__ __launch_bounds__(256) void split_items_kernel_v6_if(const float* __restrict__ table, int n_items, int n_tiles,
                                                                uint4* __restrict__ frags, const int32_t* __restrict__ n_rows) {
    if (*n_rows <= 0) return;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (g >= static_cast<int64_t>(n_tiles) * 2 * 2 * 64) return;
    const int lane = static_cast<int>(g & 63), ks = static_cast<int>((g >> 6) & 1);
    const int64_t G = g >> 7;
    int64_t item = G * F6_GI + (lane & 15);
    if (item >= n_items) item = n_items - 1;
    const float4* src = reinterpret_cast<const float4*>(table + item * FE_D + ks * 32 + 8 * (lane >> 4));
    const float4 v0 = src[0], v1 = src[1];
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint4 hi, mid, lo;
    split3x8(v, hi, mid, lo);
    uint4* dst = frags + G * F4_HALF_U4 + (ks * 3) * 64 + lane;
    dst[0] = hi;
    dst[64] = mid;
    dst[128] = lo;
}

// candidate path of one finished item group: group_candidates_v6 with the accumulators in the scaled domain
__device__ __forceinline__ void group_candidates_v7(const FusedArgs& a, const WaveCtx& w, const f32x4 (&acc)[4],
                                                    const bool (&pass)[4][4], const uint64_t (&mask)[4][4], int base,
                                                    float (&thr_s)[4], float S, float S_inv) {
    const int qd = w.lane >> 4, c16 = w.lane & 15;
    // All reservations first, then ONE wait, then all stores: one LDS round trip per step with events (this kernel has the
    // 16 registers).  The atomics are issued from inline asm under the lane mask of their test, without a branch: written
    // with the builtin inside `if (pass)`, hipcc copies every result into its merge register in the same masked block and
    // waits for the LDS there -- one round trip per EVENT.  The wait below names every position as an operand, so no use of
    // one can be scheduled in front of it (the compiler does not know these registers are in flight).
    unsigned pos[4][4];
    const unsigned one = 1u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t cnt_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)&w.cnt[16 * g + c16]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pos[g][i] = 0u;
            uint64_t saved;
            asm volatile("s_and_saveexec_b64 %1, %2\n\tds_add_rtn_u32 %0, %3, %4\n\ts_mov_b64 exec, %1"
                         : "+v"(pos[g][i]), "=&s"(saved)
                         : "s"(mask[g][i]), "v"(cnt_addr), "v"(one)
                         : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(pos[0][0]), "+v"(pos[0][1]), "+v"(pos[0][2]), "+v"(pos[0][3]), "+v"(pos[1][0]), "+v"(pos[1][1]),
                   "+v"(pos[1][2]), "+v"(pos[1][3]), "+v"(pos[2][0]), "+v"(pos[2][1]), "+v"(pos[2][2]), "+v"(pos[2][3]),
                   "+v"(pos[3][0]), "+v"(pos[3][1]), "+v"(pos[3][2]), "+v"(pos[3][3])
                 :
                 : "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint64_t* list = w.my_cand + static_cast<int64_t>(16 * g + c16) * a.cap;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (pass[g][i]) list[pos[g][i]] = skr::rank_key(acc[g][i] * S_inv, base + 4 * qd + i);   // the real score: an exact power-of-two scaling
    }
    uint64_t need = __ballot(w.cnt[w.lane] > a.trigger);
    if (need) {
        while (need) {
            const int ul = __ffsll(static_cast<long long>(need)) - 1;
            need &= need - 1;
            const float nt = compact_user(a, w, ul, -1) * S;   // -inf stays -inf
            const int ug = ul >> 4;   // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g == ug && c16 == (ul & 15)) thr_s[g] = nt;
        }
        // (no wait for the survivors' stores here: compact_select_n has told the compiler that its loads are done)
    }
}

// (185 VGPRs: two workgroups per CU.  Squeezed to 168 for three -- a dozen registers spilled, and 1 024 workgroups over 768 places
// leave the second round a third full -- the kernel took 17.1 instead of 11.3 ms.)
template <bool HAS_BIAS>
__global__ __launch_bounds__(FE_WAVES * 64, 2) void fused_topk_kernel_v7(FusedArgs a, const uint4* __restrict__ frags,
                                                                         const float* __restrict__ bias_s,
                                                                         const F7Scales* __restrict__ sc) {
    __shared__ uint4 s_tile[F5_RING * F7_GROUP_U4];            // ONE ring of F5_RING item groups (4 KB each) for the workgroup
    __shared__ float4 s_bias[3][16];                           // three (scaled) bias rows, as in v6
    __shared__ int s_cnt[FE_WAVES][FE_UW];
    __shared__ int64_t s_row_beg[FE_WAVES][FE_UW];
    __shared__ int s_row_len[FE_WAVES][FE_UW];
    __shared__ int s_rowbuf[FE_WAVES][F4_ROWBUF];
    WaveCtx w;
    w.lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.c = w.lane & 31;
    w.h = w.lane >> 5;
    const int lane = w.lane, c16 = w.lane & 15, qd = w.lane >> 4;
    w.ubase = (static_cast<int64_t>(blockIdx.x) * FE_WAVES + wv) * FE_UW;
    w.cnt = s_cnt[wv];
    w.cnt[lane] = 0;
    w.my_cand = a.cand + w.ubase * a.cap;
    w.row_beg = s_row_beg[wv];
    w.row_len = s_row_len[wv];
    w.rowbuf = s_rowbuf[wv];
    w.rowbuf_len = F4_ROWBUF;
    {
        const int64_t row = w.ubase + lane;
        int64_t rb = 0;
        int len = 0;
        if (a.train_rowptr && row < a.B) {
            const int u = a.users[row];
            rb = a.train_rowptr[u];
            len = static_cast<int>(a.train_rowptr[u + 1] - rb);
        }
        w.row_beg[lane] = rb;
        w.row_len[lane] = len;
    }
    const float s_u = sc->s_u, S = sc->S, S_inv = sc->S_inv;
    // user fragments: B[k = 8 qd + j][col c16] of user group g and k-step ks, two pieces each
    uint4 bh[4][2], bl[4][2];
    float thr[4];                                   // thresholds in the scaled domain
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t row = w.ubase + 16 * g + c16;
        const bool ok = row < a.B;
        const int uid = a.users[ok ? row : (a.B - 1)];
        thr[g] = (ok && a.ablate != 1 && a.ablate != 7) ? -INFINITY : INFINITY;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4* up = reinterpret_cast<const float4*>(a.user_table + static_cast<int64_t>(uid) * FE_D + ks * 32 + 8 * qd);
            const float4 v0 = up[0], v1 = up[1];
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            split2x8(v, s_u, bh[g][ks], bl[g][ks]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // retire the loads here (see the fp32 kernel: hidden DMA vs counted vmcnt)
            asm volatile("" : "+v"(bh[g][ks].x), "+v"(bh[g][ks].y), "+v"(bh[g][ks].z), "+v"(bh[g][ks].w));
            asm volatile("" : "+v"(bl[g][ks].x), "+v"(bl[g][ks].y), "+v"(bl[g][ks].z), "+v"(bl[g][ks].w));
        }
    const int n_tiles = (a.n_items + FE_TI - 1) / FE_TI;
    const int n_half = 2 * n_tiles;                 // item groups
    const uint32_t lt = lds_addr_of(&s_tile[0]);
    const uint32_t lb0 = lds_addr_of(&s_bias[0][0]);
    const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
    auto issue_group = [&](int hs, int slot, int brow) {   // group hs -> ring slot: wavefront wv brings block wv; 3 also a tile's bias row
        const char* sbase = reinterpret_cast<const char*>(frags) + static_cast<int64_t>(hs) * (F7_GROUP_U4 * 16);   // wave-uniform
        glds_b128_s(lane16, sbase + wv * 1024, lt + slot * (F7_GROUP_U4 * 16) + wv * 1024);
        if (HAS_BIAS && !(hs & 1) && wv == 3) {
            const int bi = (hs >> 1) * FE_TI + (lane & 31);   // bias_s is padded to whole tiles
            glds_b32(bias_s + bi, lb0 + static_cast<uint32_t>(brow) * 256u);
        }
    };
#pragma unroll
    for (int h0 = 0; h0 < F5_RING; ++h0)
        if (h0 < n_half) issue_group(h0, h0, h0 >> 1);
    FE2_WAIT();
    __syncthreads();
    uint4 afA[4], afB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) afA[i] = s_tile[i * 64 + lane];
    f32x4 accA[4], accB[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) accB[g][i] = -INFINITY;   // "group -1": nothing passes
    int slot = 0;
    int brow = 0, brow2 = 2;
    // one step = one item group: 6 slots of four MFMAs (one piece product on the four user groups); each of the first sixteen
    // MFMAs is followed by one threshold test of group hs-1
#define F7_TEST(Q, PRV)   /* the sixteen lane masks are OR-ed behind the step's last MFMA: a scalar OR right behind its       \
                             v_cmp would make the next MFMA wait for the compare's result to reach the scalar unit */ \
    {                                                                                                         \
        pass_[(Q) >> 2][(Q) & 3] = PRV[(Q) >> 2][(Q) & 3] > thr[(Q) >> 2];                                    \
        mask_[(Q) >> 2][(Q) & 3] = __builtin_amdgcn_ballot_w64(PRV[(Q) >> 2][(Q) & 3] > thr[(Q) >> 2]);      \
    }
#define F7_SLOT(S_, AF, BP, ACC, PRV)                                                                         \
    {                                                                                                         \
        /* ONE threshold test behind each of the step's first sixteen MFMAs: a 16x16x32 MFMA holds the vector issue for 8 of    \
           its 16 cycles, so a single 4-cycle instruction per gap is nearly free and a cluster of three is not               \
           (MI355X_MICROARCH.md, vector-instruction issue cost) */                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                    \
            ACC[g_] = F7_MFMA(AF, BP[g_][(S_) / 3], (S_) == 0 ? seed_ : ACC[g_]);                             \
            if ((S_) < 4) F7_TEST(4 * ((S_) & 3) + g_, PRV)                                                   \
            FE3_PIN();                                                                                        \
        }                                                                                                     \
    }
#define F7_STEP(CUR, NXT, HS, ACC, PRV, LAST)                                                                 \
    {                                                                                                         \
        const int hs_ = (HS);                                                                                 \
        FE3_PIN();                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* CUR has landed in registers */                \
        FE3_PIN();                                                                                            \
        const bool more3_ = hs_ + F5_RING < n_half && a.ablate != 7;                                          \
        const int nslot_ = slot == F5_RING - 1 ? 0 : slot + 1;                                                \
        if (hs_ + 1 < n_half) {                                                                               \
            /* behind my block of group hs+1 only my block of group hs+2 is in flight (and, wavefront 3, its bias row) */ \
            if (hs_ + 2 >= n_half) FE2_WAIT();                                                                \
            else if (HAS_BIAS && !(hs_ & 1) && wv == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      \
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                             \
        }                                                                                                     \
        if (a.ablate != 19) __syncthreads();   /* 19: timing experiment without the barrier (results unreliable) */ \
        FE3_PIN();                                                                                            \
        if (more3_) issue_group(hs_ + F5_RING, slot, brow2);                                                  \
        f32x4 seed_;                                                                                          \
        if (HAS_BIAS) {                                                                                       \
            const float4 b4 = s_bias[brow][4 * (hs_ & 1) + qd];                                               \
            seed_[0] = b4.x; seed_[1] = b4.y; seed_[2] = b4.z; seed_[3] = b4.w;                               \
        } else {                                                                                              \
            seed_[0] = 0.0f; seed_[1] = 0.0f; seed_[2] = 0.0f; seed_[3] = 0.0f;                               \
        }                                                                                                     \
        FE3_PIN();                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) NXT[i] = s_tile[nslot_ * F7_GROUP_U4 + i * 64 + lane];  \
        if (LAST) {   /* the last tile: rows beyond the catalogue start from -inf and stay there */            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
                if (hs_ * F6_GI + 4 * qd + i >= a.n_items) seed_[i] = -INFINITY;                              \
        }                                                                                                     \
        bool any_ = false;                                                                                    \
        bool pass_[4][4] = {};                                                                                \
        uint64_t mask_[4][4] = {};                                                                            \
        FE3_PIN();                                                                                            \
        /* small terms first, per k-step: lo*hi, hi*lo, hi*hi */                                              \
        F7_SLOT(0, CUR[1], bh, ACC, PRV)  F7_SLOT(1, CUR[0], bl, ACC, PRV)  F7_SLOT(2, CUR[0], bh, ACC, PRV)  \
        F7_SLOT(3, CUR[3], bh, ACC, PRV)  F7_SLOT(4, CUR[2], bl, ACC, PRV)  F7_SLOT(5, CUR[2], bh, ACC, PRV)  \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                                      \
            any_ |= (pass_[g_][0] | pass_[g_][1]) | (pass_[g_][2] | pass_[g_][3]);                            \
        if (__any(any_)) group_candidates_v7(a, w, PRV, pass_, mask_, (hs_ - 1) * F6_GI, thr, S, S_inv);      \
        slot = nslot_;                                                                                        \
    }
    for (int t = 0; t < n_tiles - 1; ++t) {
        F7_STEP(afA, afB, 2 * t, accA, accB, false)
        F7_STEP(afB, afA, 2 * t + 1, accB, accA, false)
        brow = brow == 2 ? 0 : brow + 1;
        brow2 = brow2 == 2 ? 0 : brow2 + 1;
    }
    F7_STEP(afA, afB, 2 * (n_tiles - 1), accA, accB, true)
    F7_STEP(afB, afA, 2 * (n_tiles - 1) + 1, accB, accA, true)
#undef F7_STEP
#undef F7_SLOT
#undef F7_TEST
    {
        bool any_ = false;
        bool pass_[4][4];
        uint64_t mask_[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pass_[g][i] = accB[g][i] > thr[g];
                mask_[g][i] = __builtin_amdgcn_ballot_w64(accB[g][i] > thr[g]);
                any_ |= pass_[g][i];
            }
        if (__any(any_)) group_candidates_v7(a, w, accB, pass_, mask_, (n_half - 1) * F6_GI, thr, S, S_inv);
    }
    __threadfence_block();
    final_compactions(a, w);
}

// list capacity per user.  Measured on MI355X (K = 10..100, 262 144 users): 512-entry lists with a
// trigger of 480 were no faster than 256 / 224 once the mid-sweep compaction selects instead of sorting
// (profiles/r01_eval_history.txt), so the smaller scratch footprint stays.
int fused_cap(int top_k) {
    (void)top_k;
    // SKR_FUSED_CAP (64 / 128 / 256; experiments on the scratch lists' footprint -- the trigger must leave 32 free slots)
    static const int cap_env = [] { const char* e = getenv("SKR_FUSED_CAP"); return e ? atoi(e) : 0; }();
    return (cap_env == 64 || cap_env == 128) ? cap_env : FE_CAP;
}

F7Scales* g_f7_scales = nullptr;   // device: scales and guard count of the last call, if it ran in f16x2 (skr_eval_fused_rejected)

}  // namespace

extern "C" {

size_t skr_eval_fused_workspace(int B, int top_k) {
    if (B <= 0) return 0;
    const size_t padded = (static_cast<size_t>(B) + FE_UW - 1) / FE_UW * FE_UW;
    return padded * fused_cap(top_k) * sizeof(uint64_t);
}

int skr_eval_fused_rejected(int32_t* h_count, void* stream) {
    SKR_REQUIRE(h_count, "skr_eval_fused_rejected: NULL argument");
    *h_count = 0;
    if (!g_f7_scales) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    SKR_HIP(hipMemcpyAsync(h_count, &g_f7_scales->n_flagged, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SKR_HIP(hipStreamSynchronize(st));
    return SKR_OK;
}

int skr_eval_fused_topk(const float* d_user_table, const int32_t* d_users, int B, const float* d_item_table,
                        const float* d_item_bias, int n_items, int dim, const int64_t* d_train_rowptr,
                        const int32_t* d_train_items, int top_k, int32_t* d_topk_ids, float* d_topk_scores, void* d_work,
                        size_t work_bytes, void* stream) {
    SKR_REQUIRE(d_user_table && d_users && d_item_table && d_topk_ids, "skr_eval_fused_topk: NULL argument");
    SKR_REQUIRE(dim == FE_D, "skr_eval_fused_topk: dim must be 64 (got %d)", dim);
    SKR_REQUIRE(B >= 0 && n_items > 0, "skr_eval_fused_topk: bad shape");
    SKR_REQUIRE(top_k >= 1 && top_k <= SKR_MAX_TOPK, "top_k %d outside [1, %d]", top_k, SKR_MAX_TOPK);
    SKR_REQUIRE(top_k <= n_items, "top_k %d larger than the catalogue (%d items)", top_k, n_items);
    SKR_REQUIRE((d_train_rowptr == nullptr) == (d_train_items == nullptr), "train CSR: both pointers or neither");
    SKR_REQUIRE(((reinterpret_cast<uintptr_t>(d_user_table) | reinterpret_cast<uintptr_t>(d_item_table) |
                  reinterpret_cast<uintptr_t>(d_item_bias)) & 15) == 0, "tables must be 16-byte aligned");
    g_f7_scales = nullptr;             // set again below if this call runs the fp16x2 sweep
    if (B == 0) return SKR_OK;
    if (work_bytes < skr_eval_fused_workspace(B, top_k) || !d_work)
        return skr::fail(SKR_ENOMEM, "workspace too small: need %zu bytes", skr_eval_fused_workspace(B, top_k));
    FusedArgs a{d_user_table, d_users, B, d_item_table, d_item_bias, n_items, d_train_rowptr, d_train_items,
                top_k, static_cast<uint64_t*>(d_work), d_topk_ids, d_topk_scores, 0, 0, fused_cap(top_k)};
    // compaction trigger: small lists keep the thresholds fresh (fewer candidate events per tile);
    // a tile adds at most 32 entries per user, so trigger + 32 <= cap must hold
    static const int trig_env = [] { const char* e = getenv("SKR_FUSED_TRIGGER"); return e ? atoi(e) : 0; }();
    // measured on MI355X (profiles/r01_eval_history.txt): K=10 is best around K+48, K>=50 at the cap
    a.trigger = trig_env > 0 ? trig_env : 30 + 3 * top_k;
    // a list of at most 128 entries is compacted in two registers per lane, a longer one in four: between top-31 and top-53 the
    // rule above lands just beyond that step and pays for it (top-40 on 262 144 users: 14.7 ms at 150, 13.5 at 112-120)
    if (trig_env <= 0 && a.trigger > 120 && a.trigger < 190) a.trigger = 120;
    if (a.trigger < top_k) a.trigger = top_k;
    static const int ablate = [] { const char* e = getenv("SKR_FUSED_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate;
    const int64_t waves = (static_cast<int64_t>(B) + FE_UW - 1) / FE_UW;
    const unsigned blocks = static_cast<unsigned>((waves + FE_WAVES - 1) / FE_WAVES);
    hipStream_t st = skr::as_stream(stream);
    // arithmetic mode, read per call: "f16x2" (default: two fp16 pieces behind a guard, rejected rows through bf16x3),
    // "bf16x3" (three bf16 pieces, no guard needed) or "fp32" (the FP32-MFMA kernel)
    const char* mode_env = getenv("SKR_FUSED_MODE");
    const std::string mode = mode_env ? mode_env : "f16x2";
    SKR_REQUIRE(mode == "fp32" || mode == "bf16x3" || mode == "f16x2",
                "SKR_FUSED_MODE must be 'f16x2', 'bf16x3' or 'fp32' (got '%s')", mode_env);
    const bool mode_split = mode != "fp32";
    // a list must hold what one step can add on top of the trigger: 32 entries per user in the fp32 kernel (tiles of 32
    // items), 16 in the split kernels (groups of 16) -- at large top_k, where the rule above asks for more than fits, the
    // split kernels compact that much later
    {
        const int step_items = mode_split ? F6_GI : FE_TI;
        if (a.trigger > a.cap - step_items) a.trigger = a.cap - step_items;
    }
    if (mode_split) {
        // library-owned scratch for the split item table (38 MB at 100 k items), grown on demand
        static uint4* frag_buf = nullptr;
        static size_t frag_cap = 0;
        const int n_tiles = (n_items + FE_TI - 1) / FE_TI;
        const size_t need = static_cast<size_t>(n_tiles) * F4_TILE_U4 * sizeof(uint4);
        if (need > frag_cap) {
            if (frag_buf) {
                SKR_HIP(hipStreamSynchronize(st));
                SKR_HIP(hipFree(frag_buf));
            }
            frag_buf = nullptr;
            frag_cap = 0;
            SKR_HIP(hipMalloc(&frag_buf, need));
            frag_cap = need;
        }
        if (mode == "f16x2") {
            // the fp16x2 sweep, its guard, and the bf16x3 kernel over the rows the guard flagged (fused_topk_kernel_v7's header)
            static void* f7_buf = nullptr;       // [scales | fp16 fragments | scaled bias | flagged rows]
            static size_t f7_cap = 0;
            const size_t frag16_bytes = static_cast<size_t>(n_tiles) * 2 * F7_GROUP_U4 * sizeof(uint4);
            const size_t bias_bytes = static_cast<size_t>(n_tiles) * FE_TI * sizeof(float);
            const size_t flag_bytes = (static_cast<size_t>(B) + 63) / 64 * 64 * sizeof(int32_t);
            const size_t need7 = 256 + frag16_bytes + bias_bytes + flag_bytes;
            if (need7 > f7_cap) {
                if (f7_buf) {
                    SKR_HIP(hipStreamSynchronize(st));
                    SKR_HIP(hipFree(f7_buf));
                }
                f7_buf = nullptr;
                f7_cap = 0;
                SKR_HIP(hipMalloc(&f7_buf, need7));
                f7_cap = need7;
            }
            char* base = static_cast<char*>(f7_buf);
            F7Scales* sc = reinterpret_cast<F7Scales*>(base);
            g_f7_scales = sc;
            uint4* frag16 = reinterpret_cast<uint4*>(base + 256);
            float* bias_s = reinterpret_cast<float*>(base + 256 + frag16_bytes);
            int32_t* flags = reinterpret_cast<int32_t*>(base + 256 + frag16_bytes + bias_bytes);
            SKR_HIP(hipMemsetAsync(sc, 0, sizeof(F7Scales), st));
            const int64_t it_thr = static_cast<int64_t>(n_items) * (FE_D / 4), us_thr = static_cast<int64_t>(B) * (FE_D / 4);
            hipLaunchKernelGGL(absmax_rows_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((it_thr + 255) / 256, ABSMAX_BLOCKS))), dim3(256), 0, st, d_item_table,
                               static_cast<const int32_t*>(nullptr), static_cast<int64_t>(n_items), &sc->max_v_bits);
            hipLaunchKernelGGL(absmax_rows_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((us_thr + 255) / 256, ABSMAX_BLOCKS))), dim3(256), 0, st, d_user_table,
                               d_users, static_cast<int64_t>(B), &sc->max_u_bits);
            hipLaunchKernelGGL(f7_scales_kernel, dim3(1), dim3(64), 0, st, sc);
            const int64_t nthr7 = static_cast<int64_t>(n_tiles) * 256;
            hipLaunchKernelGGL(split_items_kernel_v7, dim3(static_cast<unsigned>((nthr7 + 255) / 256)), dim3(256), 0, st, d_item_table,
                               d_item_bias, n_items, n_tiles, sc, frag16, bias_s);
            SKR_LAUNCH_CHECK();
            FusedArgs a7 = a;
            // SKR_F7_GUARD=0 (diagnosis only: tests/test_gpu_eval.py probes the kernel's own error floor with it) accepts every row
            static const bool guard = [] { const char* e = getenv("SKR_F7_GUARD"); return !(e && atoi(e) == 0); }();
            if (guard) {
                a7.guard_s_inv = &sc->S_inv;
                a7.flag_list = flags;
                a7.flag_count = &sc->n_flagged;
            }
            if (d_item_bias)
                hipLaunchKernelGGL(fused_topk_kernel_v7<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a7, frag16, bias_s, sc);
            else
                hipLaunchKernelGGL(fused_topk_kernel_v7<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a7, frag16, bias_s, sc);
            SKR_LAUNCH_CHECK();
            // the rows the guard did not accept: the bf16x3 kernel, on the device's own count (an empty list returns at once)
            hipLaunchKernelGGL(split_items_kernel_v6_if, dim3(static_cast<unsigned>((nthr7 + 255) / 256)), dim3(256), 0, st, d_item_table,
                               n_items, n_tiles, frag_buf, &sc->n_flagged);
            FusedArgs a6 = a;
            a6.row_map = flags;
            a6.n_rows_dev = &sc->n_flagged;
            if (d_item_bias)
                hipLaunchKernelGGL(fused_topk_kernel_v6<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a6, frag_buf);
            else
                hipLaunchKernelGGL(fused_topk_kernel_v6<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a6, frag_buf);
            SKR_LAUNCH_CHECK();
            return SKR_OK;
        }
        // bf16x3: fused_topk_kernel_v6 (16-item steps on v_mfma_f32_16x16x32_bf16)
        const int64_t nthr = static_cast<int64_t>(n_tiles) * 256;
        hipLaunchKernelGGL(split_items_kernel_v6, dim3(static_cast<unsigned>((nthr + 255) / 256)), dim3(256), 0, st, d_item_table,
                           n_items, n_tiles, frag_buf);
        SKR_LAUNCH_CHECK();
        if (d_item_bias)
            hipLaunchKernelGGL(fused_topk_kernel_v6<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a, frag_buf);
        else
            hipLaunchKernelGGL(fused_topk_kernel_v6<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a, frag_buf);
        SKR_LAUNCH_CHECK();
        return SKR_OK;
    }
    if (d_item_bias)
        hipLaunchKernelGGL(fused_topk_kernel_v3<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
    else
        hipLaunchKernelGGL(fused_topk_kernel_v3<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

}  // extern "C"
