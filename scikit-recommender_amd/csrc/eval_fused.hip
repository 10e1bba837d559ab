// eval_fused.hip -- K4: user x item scoring on FP32 MFMA fused with train masking and top-K.
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
//   * item tiles (32 x 256 B) are fetched with full-line coalesced 16-byte loads (prefetched one
//     tile ahead in registers), transposed through a wave-private, XOR-swizzled LDS image into the
//     A-fragment layout (conflict-free ds_read_b128); waves share nothing, so there are no barriers
//     and a wave that stops to compact its candidate lists never stalls its neighbours;
//   * scores above the user's threshold (rare once it has warmed up) are appended to a per-user
//     candidate list in HBM scratch (LDS counters); when a list nears capacity the wave sorts it
//     (register bitonic, 256 keys), drops train items, keeps the K best and raises the threshold;
//   * k order inside a dot product: the chain starts from the item bias (if any) and adds dims
//     {t, 32+t} at MFMA step t, a fixed order (results are deterministic run to run, and differ from
//     a CPU sgemm + bias add only in summation order).
// Bound: MFMA (2*B*I*64 flop); the 25.6 MB item table streams from L2 / Infinity Cache.
#include "eval_common.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FE_WAVES = 4;   // waves per workgroup (independent of each other)
constexpr int FE_UW = 64;     // users per wave
constexpr int FE_TI = 32;     // items per tile
constexpr int FE_CAP = 256;   // candidate capacity per user (keys of 8 B)
constexpr int FE_D = 64;

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const int lo = __shfl_xor(static_cast<int>(static_cast<uint32_t>(v)), m, 64);
    const int hi = __shfl_xor(static_cast<int>(static_cast<uint32_t>(v >> 32)), m, 64);
    return (static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo);
}

// Descending sort of 256 keys held 4 per lane; element index = e*64 + lane.
__device__ __forceinline__ void wave_sort256_desc(uint64_t (&k)[4], int lane) {
#pragma unroll
    for (int kk = 2; kk <= 256; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int de = j >> 6;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
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
                for (int e = 0; e < 4; ++e) {
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
    uint64_t* cand;  // [ceil(B/64)*64][FE_CAP]
    int32_t* out_ids;
    float* out_scores;
};

// Sort user `ul`'s candidate list, drop train items, keep the best top_k at the front.
// Returns the new threshold (score of the K-th best, or -inf while fewer than K are known).
// If out_row >= 0 the final top_k ids / scores are also written to the outputs.
__device__ __forceinline__ float compact_user(const FusedArgs& a, int lane, uint64_t* __restrict__ list, int* cnt_p,
                                              int uid, int64_t out_row) {
    __threadfence_block();  // this wave's earlier appends must have landed before they are re-read
    const int n = *cnt_p;
    uint64_t k[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = e * 64 + lane;
        k[e] = (idx < n) ? list[idx] : SKR_KEY_MIN;
    }
    if (a.train_rowptr) {
        const int64_t rb = a.train_rowptr[uid], re = a.train_rowptr[uid + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k[e] != SKR_KEY_MIN && skr::contains_sorted(a.train_items, rb, re, skr::key_id(k[e]))) k[e] = SKR_KEY_MIN;
    }
    wave_sort256_desc(k, lane);
    int valid = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) valid += __popcll(__ballot(k[e] != SKR_KEY_MIN));
    const int K = a.top_k;
    const int keep = valid < K ? valid : K;
#pragma unroll
    for (int e = 0; e < 2; ++e) {  // top_k <= 128: the survivors live in elements 0..127
        const int idx = e * 64 + lane;
        if (idx < keep) {
            list[idx] = k[e];
            if (out_row >= 0) {
                if (a.out_ids) a.out_ids[out_row * K + idx] = skr::key_id(k[e]);
                if (a.out_scores) a.out_scores[out_row * K + idx] = skr::key_score(k[e]);
            }
        }
    }
    if (lane == 0) *cnt_p = keep;
    float thr = -INFINITY;
    if (keep == K) {
        const int src = (K - 1) & 63;
        const uint64_t ke = ((K - 1) >> 6) ? k[1] : k[0];
        const int lo = __shfl(static_cast<int>(static_cast<uint32_t>(ke)), src, 64);
        const int hi = __shfl(static_cast<int>(static_cast<uint32_t>(ke >> 32)), src, 64);
        thr = skr::key_score((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo));
    }
    __threadfence_block();
    return thr;
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(FE_WAVES * 64, 2) void fused_topk_kernel(FusedArgs a) {
    __shared__ float4 s_tile[FE_WAVES][2][FE_TI * FE_D / 4];  // wave-private double buffer: 2 x 8 KB
    __shared__ int s_cnt[FE_WAVES][FE_UW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * FE_WAVES + wv;
    const int64_t ubase = wave_id * FE_UW;
    if (ubase >= a.B) return;  // whole wave idle (no barriers are used anywhere in this kernel)
    float4* tile0 = s_tile[wv][0];
    float4* tile1 = s_tile[wv][1];
    int* cnt = s_cnt[wv];
    cnt[lane] = 0;

    // ---- B fragments: user (32f + c), dims [32h, 32h+32) -> bf[f][t] = dim 32h + t -----------------
    float bf[2][32];
    float thr[2];
    int uid[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t row = ubase + 32 * f + c;
        const bool ok = row < a.B;
        uid[f] = a.users[ok ? row : (a.B - 1)];
        thr[f] = ok ? -INFINITY : INFINITY;  // padding columns never pass
        const float4* up = reinterpret_cast<const float4*>(a.user_table + static_cast<int64_t>(uid[f]) * FE_D + 32 * h);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = up[q];
            bf[f][4 * q + 0] = v.x; bf[f][4 * q + 1] = v.y; bf[f][4 * q + 2] = v.z; bf[f][4 * q + 3] = v.w;
        }
    }
    uint64_t* my_cand = a.cand + ubase * FE_CAP;

    const int n_tiles = (a.n_items + FE_TI - 1) / FE_TI;
    // global -> register prefetch of one tile: instruction j covers rows 4j..4j+3, lane l reads the
    // 16-byte piece ((l&15) ^ (row&15)) of row 4j + (l>>4): full 256-byte lines, XOR-swizzled image.
    // (macros, not lambdas: the 8 x float4 staging registers must stay in VGPRs)
    float4 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7;  // named scalars: an indexed array ends up in scratch
#define FE_FETCH1(T, J, DST)                                                                              \
    {                                                                                                      \
        const int rl_ = 4 * (J) + (lane >> 4);                                                             \
        int item_ = (T) * FE_TI + rl_;                                                                     \
        item_ = item_ < a.n_items ? item_ : a.n_items - 1; /* clamp (masked in the epilogue) */            \
        const int piece_ = (lane & 15) ^ (rl_ & 15);                                                       \
        DST = *reinterpret_cast<const float4*>(a.item_table + static_cast<int64_t>(item_) * FE_D + 4 * piece_); \
    }
#define FE_FETCH(T)                                                                                    \
    FE_FETCH1(T, 0, pf0) FE_FETCH1(T, 1, pf1) FE_FETCH1(T, 2, pf2) FE_FETCH1(T, 3, pf3) FE_FETCH1(T, 4, pf4) \
        FE_FETCH1(T, 5, pf5) FE_FETCH1(T, 6, pf6) FE_FETCH1(T, 7, pf7)
#define FE_STASH(TILE) /* linear image: row 4j+(l>>4), slot l&15 */                                     \
    (TILE)[0 * 64 + lane] = pf0; (TILE)[1 * 64 + lane] = pf1; (TILE)[2 * 64 + lane] = pf2;               \
    (TILE)[3 * 64 + lane] = pf3; (TILE)[4 * 64 + lane] = pf4; (TILE)[5 * 64 + lane] = pf5;               \
    (TILE)[6 * 64 + lane] = pf6; (TILE)[7 * 64 + lane] = pf7;

    FE_FETCH(0)
    FE_STASH(tile0)

    for (int t = 0; t < n_tiles; ++t) {
        float4* cur = (t & 1) ? tile1 : tile0;
        float4* nxt = (t & 1) ? tile0 : tile1;
        const int tn = (t + 1 < n_tiles) ? t + 1 : t;  // the last iteration re-fetches its own tile (unused)
        FE_FETCH(tn)
        // ---- A fragment: item c, dims [32h, 32h+32): piece g = 8h+q sits in slot g ^ (c&15) ----------
        float av[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = cur[c * 16 + ((8 * h + q) ^ (c & 15))];
            av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        // ---- accumulators start from the item bias (the fma chain's seed), so that the epilogue is a
        //      bare compare: acc[r] = score(item tile_base + (r&3) + 8*(r>>2) + 4h, user 32f + c) -------
        const int tile_base = t * FE_TI;
        const bool full = (tile_base + FE_TI <= a.n_items);
        f32x16 acc0 = {0};
        if (HAS_BIAS) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i0 = tile_base + 8 * g + 4 * h;
                if (full) {
                    const float4 b4 = *reinterpret_cast<const float4*>(a.item_bias + i0);
                    acc0[4 * g + 0] = b4.x; acc0[4 * g + 1] = b4.y; acc0[4 * g + 2] = b4.z; acc0[4 * g + 3] = b4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc0[4 * g + e] = (i0 + e < a.n_items) ? a.item_bias[i0 + e] : 0.0f;
                }
            }
        }
        f32x16 acc1 = acc0;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bf[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bf[1][s], acc1, 0, 0, 0);
        }
        bool appended = false;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const f32x16& acc = f ? acc1 : acc0;
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) m |= (acc[r] > thr[f]) ? (1u << r) : 0u;
            if (!full) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile_base + (r & 3) + 8 * (r >> 2) + 4 * h >= a.n_items) m &= ~(1u << r);
            }
            if (m) {  // rare once the thresholds have warmed up
                appended = true;
                const int ul = 32 * f + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (m & (1u << r)) {
                        const int item = tile_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int p = atomicAdd(&cnt[ul], 1);  // < FE_CAP: see the compaction rule below
                        my_cand[static_cast<int64_t>(ul) * FE_CAP + p] = skr::rank_key(acc[r], item);
                    }
                }
            }
        }
        // ---- a list may take at most 32 new entries per tile: compact every list above CAP-32 -------
        if (__any(appended)) {
            uint64_t need = __ballot(cnt[lane] > FE_CAP - FE_TI);
            while (need) {
                const int ul = __ffsll(static_cast<long long>(need)) - 1;
                need &= need - 1;
                const int f = ul >> 5;
                const int u_id = __shfl(f ? uid[1] : uid[0], ul & 31, 64);
                const float nt = compact_user(a, lane, my_cand + static_cast<int64_t>(ul) * FE_CAP, &cnt[ul], u_id, -1);
                if (c == (ul & 31)) {
                    if (f) thr[1] = nt; else thr[0] = nt;
                }
            }
        }
        FE_STASH(nxt)
    }
#undef FE_FETCH
#undef FE_FETCH1
#undef FE_STASH
    // ---- final ranking of every user of this wave ---------------------------------------------------
    for (int ul = 0; ul < FE_UW; ++ul) {
        const int64_t row = ubase + ul;
        if (row >= a.B) break;
        const int u_id = __shfl((ul >> 5) ? uid[1] : uid[0], ul & 31, 64);
        compact_user(a, lane, my_cand + static_cast<int64_t>(ul) * FE_CAP, &cnt[ul], u_id, row);
    }
}


// ================================================================================================
// v2: same mapping, re-scheduled for one/two waves per SIMD.
//   * item tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write,
//     and the data's latency is covered by a whole 64-MFMA chain (the wait sits at the END of the
//     iteration; in v1 hipcc hoisted the ds_writes -- and with them the vmcnt waits -- into the chain,
//     which parked the wave for ~1950 cycles per tile: profiles/r01_eval_pmc.txt);
//   * two accumulator sets: while tile t is multiplied, the any-score-above-threshold test of tile
//     t-1 (64 v_cmp + scalar ORs, straight-line) sits in the same basic block and is interleaved
//     with the MFMAs by the scheduler; the candidate path is entered only when some lane passed;
//   * the item bias rides along through LDS (one 4-byte LDS-DMA per tile) and seeds the accumulators.
// ================================================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct WaveCtx {
    int lane, c, h;
    int64_t ubase;
    int* cnt;
    uint64_t* my_cand;
};

// candidate path for one finished tile: append every score above the user's threshold, then compact
// the lists that came within one tile of their capacity
__device__ __forceinline__ void tile_candidates(const FusedArgs& a, const WaveCtx& w, const f32x16& acc0,
                                                const f32x16& acc1, int tile_base, float (&thr)[2], const int (&uid)[2]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const f32x16& acc = f ? acc1 : acc0;
        const int ul = 32 * f + w.c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int item = tile_base + (r & 3) + 8 * (r >> 2) + 4 * w.h;
            if (acc[r] > thr[f] && item < a.n_items) {
                const int p = atomicAdd(&w.cnt[ul], 1);  // < FE_CAP by the compaction rule
                w.my_cand[static_cast<int64_t>(ul) * FE_CAP + p] = skr::rank_key(acc[r], item);
            }
        }
    }
    uint64_t need = __ballot(w.cnt[w.lane] > FE_CAP - FE_TI);
    while (need) {
        const int ul = __ffsll(static_cast<long long>(need)) - 1;
        need &= need - 1;
        const int f = ul >> 5;
        const int u_id = __shfl(f ? uid[1] : uid[0], ul & 31, 64);
        const float nt = compact_user(a, w.lane, w.my_cand + static_cast<int64_t>(ul) * FE_CAP, &w.cnt[ul], u_id, -1);
        if (w.c == (ul & 31)) {
            if (f) thr[1] = nt; else thr[0] = nt;
        }
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(FE_WAVES * 64, 2) void fused_topk_kernel_v2(FusedArgs a) {
    __shared__ float4 s_tile[FE_WAVES][2][FE_TI * FE_D / 4];  // wave-private double buffer: 2 x 8 KB
    __shared__ float4 s_bias[FE_WAVES][2][16];                // 64 floats per buffer (32 used twice)
    __shared__ int s_cnt[FE_WAVES][FE_UW];
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
    w.my_cand = a.cand + w.ubase * FE_CAP;

    float bf[2][32];
    float thr[2];
    int uid[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t row = w.ubase + 32 * f + c;
        const bool ok = row < a.B;
        uid[f] = a.users[ok ? row : (a.B - 1)];
        thr[f] = ok ? -INFINITY : INFINITY;
        const float4* up = reinterpret_cast<const float4*>(a.user_table + static_cast<int64_t>(uid[f]) * FE_D + 32 * h);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = up[q];
            bf[f][4 * q + 0] = v.x; bf[f][4 * q + 1] = v.y; bf[f][4 * q + 2] = v.z; bf[f][4 * q + 3] = v.w;
        }
    }
    const int n_tiles = (a.n_items + FE_TI - 1) / FE_TI;

    // LDS-DMA of tile T into buffer BUF: instruction j writes LDS bytes [j*1024, (j+1)*1024) = rows
    // 4j..4j+3; lane l supplies the 16-byte piece ((l&15) ^ (row&15)) of row 4j + (l>>4).
#define FE2_ISSUE(T, BUF)                                                                                    \
    {                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                      \
            const int rl_ = 4 * j + (lane >> 4);                                                             \
            int item_ = (T) * FE_TI + rl_;                                                                   \
            item_ = item_ < a.n_items ? item_ : a.n_items - 1;                                               \
            const int piece_ = (lane & 15) ^ (rl_ & 15);                                                     \
            __builtin_amdgcn_global_load_lds(                                                                \
                (gbl_ptr_t)(a.item_table + static_cast<int64_t>(item_) * FE_D + 4 * piece_),                 \
                (lds_ptr_t)(&s_tile[wv][BUF][j * 64]), 16, 0, 0);                                            \
        }                                                                                                    \
        if (HAS_BIAS) {                                                                                      \
            int bi_ = (T) * FE_TI + (lane & 31);                                                             \
            bi_ = bi_ < a.n_items ? bi_ : a.n_items - 1;                                                     \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.item_bias + bi_), (lds_ptr_t)(&s_bias[wv][BUF][0]), 4, 0, 0); \
        }                                                                                                    \
    }
#define FE2_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    // one tile: CUR accumulators <- tile T; fast test of the PREVious tile's accumulators meanwhile
#define FE2_STEP(C0, C1, P0, P1, T)                                                                          \
    {                                                                                                        \
        const int t_ = (T);                                                                                  \
        const int bc_ = t_ & 1;                                                                              \
        if (t_ + 1 < n_tiles) FE2_ISSUE(t_ + 1, bc_ ^ 1)                                                     \
        float av_[32];                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                      \
            const float4 v_ = s_tile[wv][bc_][c * 16 + ((8 * h + q) ^ (c & 15))];                            \
            av_[4 * q + 0] = v_.x; av_[4 * q + 1] = v_.y; av_[4 * q + 2] = v_.z; av_[4 * q + 3] = v_.w;      \
        }                                                                                                    \
        if (HAS_BIAS) {                                                                                      \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
                const float4 b4_ = s_bias[wv][bc_][2 * g + h];                                               \
                C0[4 * g + 0] = b4_.x; C0[4 * g + 1] = b4_.y; C0[4 * g + 2] = b4_.z; C0[4 * g + 3] = b4_.w;  \
            }                                                                                                \
        } else {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) C0[r] = 0.0f;                                     \
        }                                                                                                    \
        C1 = C0;                                                                                             \
        bool any_ = false;                                                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) any_ |= (P0[r] > thr[0]) | (P1[r] > thr[1]);          \
        _Pragma("unroll") for (int s = 0; s < 32; ++s) {                                                     \
            C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[s], bf[0][s], C0, 0, 0, 0);                        \
            C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[s], bf[1][s], C1, 0, 0, 0);                        \
        }                                                                                                    \
        if (__any(any_)) tile_candidates(a, w, P0, P1, (t_ - 1) * FE_TI, thr, uid);                          \
        FE2_WAIT();                                                                                          \
    }

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = -INFINITY;  // "tile -1": nothing passes
    f32x16 accA1 = accB, accB1 = accB;
    FE2_ISSUE(0, 0)
    FE2_WAIT();
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
        FE2_STEP(accA, accA1, accB, accB1, t)
        FE2_STEP(accB, accB1, accA, accA1, t + 1)
    }
    if (t < n_tiles) {  // odd tile count: the last tile lands in A
        FE2_STEP(accA, accA1, accB, accB1, t)
        tile_candidates(a, w, accA, accA1, t * FE_TI, thr, uid);
    } else {
        tile_candidates(a, w, accB, accB1, (n_tiles - 1) * FE_TI, thr, uid);
    }
#undef FE2_STEP
#undef FE2_WAIT
#undef FE2_ISSUE
    for (int ul = 0; ul < FE_UW; ++ul) {
        const int64_t row = w.ubase + ul;
        if (row >= a.B) break;
        const int u_id = __shfl((ul >> 5) ? uid[1] : uid[0], ul & 31, 64);
        compact_user(a, lane, w.my_cand + static_cast<int64_t>(ul) * FE_CAP, &w.cnt[ul], u_id, row);
    }
}

}  // namespace

extern "C" {

size_t skr_eval_fused_workspace(int B, int top_k) {
    (void)top_k;
    if (B <= 0) return 0;
    const size_t padded = (static_cast<size_t>(B) + FE_UW - 1) / FE_UW * FE_UW;
    return padded * FE_CAP * sizeof(uint64_t);
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
    if (B == 0) return SKR_OK;
    if (work_bytes < skr_eval_fused_workspace(B, top_k) || !d_work)
        return skr::fail(SKR_ENOMEM, "workspace too small: need %zu bytes", skr_eval_fused_workspace(B, top_k));
    FusedArgs a{d_user_table, d_users, B, d_item_table, d_item_bias, n_items, d_train_rowptr, d_train_items,
                top_k, static_cast<uint64_t*>(d_work), d_topk_ids, d_topk_scores};
    const int64_t waves = (static_cast<int64_t>(B) + FE_UW - 1) / FE_UW;
    const unsigned blocks = static_cast<unsigned>((waves + FE_WAVES - 1) / FE_WAVES);
    hipStream_t st = skr::as_stream(stream);
    static const int version = [] { const char* e = getenv("SKR_FUSED_V"); return e ? atoi(e) : 2; }();
    if (version == 1) {
        if (d_item_bias)
            hipLaunchKernelGGL(fused_topk_kernel<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
        else
            hipLaunchKernelGGL(fused_topk_kernel<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
    } else {
        if (d_item_bias)
            hipLaunchKernelGGL(fused_topk_kernel_v2<true>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
        else
            hipLaunchKernelGGL(fused_topk_kernel_v2<false>, dim3(blocks), dim3(FE_WAVES * 64), 0, st, a);
    }
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

}  // extern "C"
