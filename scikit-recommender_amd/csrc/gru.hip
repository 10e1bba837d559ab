// gru.hip -- SURVEY 8 row G / f-4: GRU4RecPlus, the session-parallel recurrent recommender.
//
// Replaces the TensorFlow-1.14 graph the reference builds (recommender/GRU4RecPlus.py:124-200):
//   :168-178  embedding lookup + tf.nn.rnn_cell.GRUCell stack           -> gru_fwd_kernel / gru_bwd_*
//   :180-186  logits against the batch's own positives + shared samples -> session_logits_kernel
//   :137-166  _softmax_neg, _bpr_max_loss, _top1_max_loss (+ autograd)  -> session_rowloss_kernel, session_dout_kernel
//   :190-192  l2_loss + AdamOptimizer.minimize (dense update)           -> session_out_grads_kernel,
//                                                                          scatter_add_rows_kernel, skr_adam_step
// The recurrent state is fed through placeholders, so a training step is a one-step truncated BPTT:
// gradients stop at the incoming state.  The work per step is tiny (b = 128 sessions, 2 176 targets),
// so these kernels are written for few launches and L2-resident operands, not for MFMA; the inference
// sweep (`_get_user_embeddings`, :256-302) runs ALL users in parallel, one GRU step per history
// position, through the same forward kernel.  PARITY UNPINNED (TensorFlow absent): the tests check
// against a torch-CPU restatement of the same graph.
#include "skr_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int G_ROWS = 16;     // sessions per workgroup (4 when the batch is small)
constexpr int G_SMALL_B = 2048;
constexpr int G_T = 256;
constexpr int G_KMAX = 256;    // in_dim + hid
constexpr int G_HMAX = 128;
constexpr int G_CH = 32;      // weights in flight per thread in the forward kernel (SKR tuning: 16 / 32 / 64)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float hidden_act(float x, int act) { return act == 0 ? tanhf(x) : fmaxf(x, 0.0f); }
// derivative of the hidden activation expressed through its OUTPUT c
__device__ __forceinline__ float hidden_act_grad(float c, int act) { return act == 0 ? 1.0f - c * c : (c > 0.0f ? 1.0f : 0.0f); }

struct GruIn {
    const float* x;          // [B, in] rows, or an embedding table when x_index != nullptr
    const int32_t* x_index;  // nullable: x row of session b = x[x_index[b]]
    const float* h;          // [B, hid]
    int B, in_dim, hid;
};

__device__ __forceinline__ const float* x_row(const GruIn& g, int row) {
    return g.x + static_cast<int64_t>(g.x_index ? g.x_index[row] : row) * g.in_dim;
}

// ------------------------------------------------------------------------------------------------
// forward: tf.nn.rnn_cell.GRUCell.call
//   [r | u] = sigmoid([x, h] Wg + bg) ;  c = act([x, r*h] Wc + bc) ;  h' = u*h + (1-u)*c
// 16 sessions per workgroup, their [x, h] rows staged in LDS; a thread owns one output column and
// 16 / (256 / columns) of the rows, so each weight is read once per workgroup from L2.
// ------------------------------------------------------------------------------------------------
// H (32 / 64 / 128) and ALIGNED (in_dim a multiple of G_CH) are template parameters so that every loop bound and the
// thread -> (column, rows) mapping are compile-time: with runtime bounds and per-element guards the kernel compiled to
// 33 k instructions (260 KB of code, four times the instruction cache) and took 91 us per call at B = 128.
template <int ROWS, int H, bool ALIGNED>
__global__ __launch_bounds__(G_T) void gru_fwd_kernel(GruIn g, const uint8_t* __restrict__ active,
                                                      const float* __restrict__ Wg, const float* __restrict__ bg,
                                                      const float* __restrict__ Wc, const float* __restrict__ bc, int act,
                                                      float* __restrict__ r_out, float* __restrict__ u_out,
                                                      float* __restrict__ c_out, float* __restrict__ h_new) {
    __shared__ float a[ROWS][G_KMAX];
    __shared__ float rh[ROWS][G_HMAX];
    __shared__ float us[ROWS][G_HMAX];
    const int tid = threadIdx.x;
    const int IN = g.in_dim, K = IN + H;
    const int row0 = blockIdx.x * ROWS;
    for (int idx = tid; idx < ROWS * K; idx += G_T) {
        const int r = idx / K, k = idx - r * K;
        const int row = row0 + r;
        float v = 0.0f;
        if (row < g.B) v = (k < IN) ? x_row(g, row)[k] : g.h[static_cast<int64_t>(row) * H + (k - IN)];
        a[r][k] = v;
    }
    __syncthreads();
    // acc[i] += sum_k src[rows of this thread][k] * W[(w_row0 + k) * ld + j] for k in [0, n): G_CH weights in flight
    auto accumulate = [&](auto& acc, auto per_c, const float (*src)[G_KMAX], int src_off, const float* __restrict__ W, int w_row0, int ld,
                          int j, int g0, int ng, int n, bool aligned) {
        constexpr int PER = decltype(per_c)::value;
        int k0 = 0;
        for (; k0 + G_CH <= n; k0 += G_CH) {
            float w[G_CH];
#pragma unroll
            for (int q = 0; q < G_CH; ++q) w[q] = W[static_cast<int64_t>(w_row0 + k0 + q) * ld + j];
#pragma unroll
            for (int q = 0; q < G_CH; ++q)
#pragma unroll
                for (int i = 0; i < PER; ++i) acc[i] = fmaf(src[g0 + i * ng][src_off + k0 + q], w[q], acc[i]);
        }
        if (!aligned)
            for (; k0 < n; ++k0) {
                const float w = W[static_cast<int64_t>(w_row0 + k0) * ld + j];
#pragma unroll
                for (int i = 0; i < PER; ++i) acc[i] = fmaf(src[g0 + i * ng][src_off + k0], w, acc[i]);
            }
    };
    {   // gates: C = 2H output columns, NG thread groups share the ROWS rows
        constexpr int C = 2 * H, NG = G_T / C, PER = ROWS / NG;      // NG in {1, 2, 4} <= ROWS
        const int j = tid % C, g0 = tid / C;
        float acc[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] = 0.0f;
        accumulate(acc, std::integral_constant<int, PER>{}, a, 0, Wg, 0, C, j, g0, NG, ALIGNED ? K : IN, ALIGNED);
        if (!ALIGNED) accumulate(acc, std::integral_constant<int, PER>{}, a, IN, Wg, IN, C, j, g0, NG, H, true);   // H is a multiple of G_CH
        const float b = bg[j];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = g0 + i * NG, row = row0 + r;
            const float s = sigmoidf_(acc[i] + b);
            if (j < H) {
                rh[r][j] = s * a[r][IN + j];
                if (r_out && row < g.B) r_out[static_cast<int64_t>(row) * H + j] = s;
            } else {
                us[r][j - H] = s;
                if (u_out && row < g.B) u_out[static_cast<int64_t>(row) * H + (j - H)] = s;
            }
        }
    }
    __syncthreads();
    {   // candidate and new state: H output columns
        constexpr int NG = G_T / H, PER = (ROWS >= NG) ? ROWS / NG : 1;
        const int j = tid % H;
        const bool live = (ROWS >= NG) || (tid / H) < ROWS;             // more thread groups than rows: the rest idle
        const int g0 = live ? tid / H : 0;
        float acc[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] = 0.0f;
        accumulate(acc, std::integral_constant<int, PER>{}, a, 0, Wc, 0, H, j, g0, NG, IN, ALIGNED);
        // the recurrent half reads r*h: same [ROWS][.] shape as `a` is needed by the helper, so it lives in `a`'s h columns
        // from here on (the old state is kept in registers first)
        float ho[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) ho[i] = a[g0 + i * NG][IN + j];
        __syncthreads();
        if (live)
#pragma unroll
            for (int i = 0; i < PER; ++i) a[g0 + i * NG][IN + j] = rh[g0 + i * NG][j];
        __syncthreads();
        accumulate(acc, std::integral_constant<int, PER>{}, a, IN, Wc, IN, H, j, g0, NG, H, true);
        const float b = bc[j];
        if (live)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int r = g0 + i * NG, row = row0 + r;
                if (row < g.B) {
                    const float c = hidden_act(acc[i] + b, act);
                    const float u = us[r][j];
                    float hn = u * ho[i] + (1.0f - u) * c;
                    if (active && !active[row]) hn = ho[i];        // finished history: the state is carried
                    if (c_out) c_out[static_cast<int64_t>(row) * H + j] = c;
                    h_new[static_cast<int64_t>(row) * H + j] = hn;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// forward for a TRAINING batch at the benchmarked shape (in_dim = hid = 128, a few hundred sessions): gru_fwd_kernel's thread
// walks all 256 k of its column -- sixteen batches of 32 weight loads one behind the other, ~1.2 us of L2 latency each: 22 us
// for 25 MFLOP.  Here 1 024 threads share the same 4 rows: the k range is cut in four (gates: 256 columns x 4 segments of
// 64 k) and eight (candidate: 128 columns x 8 segments of 32 k), every thread has two / one batch in flight, and the partial
// sums meet in LDS, added in segment order (a fixed order: results are reproducible; they differ from gru_fwd_kernel's
// k-ascending chain in the last bits).
// ------------------------------------------------------------------------------------------------
constexpr int GS_T = 1024, GS_ROWS = 4, GS_H = 128, GS_K = 256;
__global__ __launch_bounds__(GS_T) void gru_fwd_split_kernel(GruIn g, const uint8_t* __restrict__ active,
                                                             const float* __restrict__ Wg, const float* __restrict__ bg,
                                                             const float* __restrict__ Wc, const float* __restrict__ bc, int act,
                                                             float* __restrict__ r_out, float* __restrict__ u_out,
                                                             float* __restrict__ c_out, float* __restrict__ h_new) {
    __shared__ float a[GS_ROWS][GS_K];            // [x | h], later [x | r*h]
    __shared__ float ho[GS_ROWS][GS_H];           // the old state
    __shared__ float us[GS_ROWS][GS_H];
    __shared__ float part[8][GS_ROWS][GS_K];      // partial sums: 4 segments x 256 columns, then 8 x 128 (32 KB)
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * GS_ROWS;
    {
        const int r = tid >> 8, k = tid & 255, row = row0 + r;
        float v = 0.0f;
        if (row < g.B) v = (k < GS_H) ? x_row(g, row)[k] : g.h[static_cast<int64_t>(row) * GS_H + (k - GS_H)];
        a[r][k] = v;
        if (k >= GS_H) ho[r][k - GS_H] = v;
    }
    __syncthreads();
    {   // gates: column j of 256, k segment ks of 4
        const int j = tid & 255, ks = tid >> 8, k0 = ks * 64;
        float acc[GS_ROWS] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float w[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) w[q] = Wg[static_cast<int64_t>(k0 + 32 * c + q) * (2 * GS_H) + j];
#pragma unroll
            for (int q = 0; q < 32; ++q)
#pragma unroll
                for (int r = 0; r < GS_ROWS; ++r) acc[r] = fmaf(a[r][k0 + 32 * c + q], w[q], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < GS_ROWS; ++r) part[ks][r][j] = acc[r];
    }
    __syncthreads();
    {   // (row, column) of the gates: the four segments in order, the activation, r*h in place of h
        const int r = tid >> 8, j = tid & 255, row = row0 + r;
        const float s = sigmoidf_(((part[0][r][j] + part[1][r][j]) + part[2][r][j]) + part[3][r][j] + bg[j]);
        if (j < GS_H) {
            if (r_out && row < g.B) r_out[static_cast<int64_t>(row) * GS_H + j] = s;
        } else {
            us[r][j - GS_H] = s;
            if (u_out && row < g.B) u_out[static_cast<int64_t>(row) * GS_H + (j - GS_H)] = s;
        }
        __syncthreads();                       // every thread has read its partial sums and the old h before they are replaced
        if (j < GS_H) a[r][GS_H + j] = s * ho[r][j];
    }
    __syncthreads();
    {   // candidate: column j of 128, k segment ks of 8 (segments 0..3: x, 4..7: r*h)
        const int j = tid & 127, ks = tid >> 7, k0 = ks * 32;
        float acc[GS_ROWS] = {0.0f, 0.0f, 0.0f, 0.0f};
        float w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = Wc[static_cast<int64_t>(k0 + q) * GS_H + j];
#pragma unroll
        for (int q = 0; q < 32; ++q)
#pragma unroll
            for (int r = 0; r < GS_ROWS; ++r) acc[r] = fmaf(a[r][k0 + q], w[q], acc[r]);
#pragma unroll
        for (int r = 0; r < GS_ROWS; ++r) part[ks][r][j] = acc[r];
    }
    __syncthreads();
    if (tid < GS_ROWS * GS_H) {
        const int r = tid >> 7, j = tid & 127, row = row0 + r;
        if (row < g.B) {
            float sum = part[0][r][j];
#pragma unroll
            for (int q = 1; q < 8; ++q) sum += part[q][r][j];
            const float c = hidden_act(sum + bc[j], act);
            const float u = us[r][j];
            float hn = u * ho[r][j] + (1.0f - u) * c;
            if (active && !active[row]) hn = ho[r][j];        // finished history: the state is carried
            if (c_out) c_out[static_cast<int64_t>(row) * GS_H + j] = c;
            h_new[static_cast<int64_t>(row) * GS_H + j] = hn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward for MANY sessions (the inference sweep `_get_user_embeddings`, GRU4RecPlus.py:256-302, advances every user at
// once): the two products [x, h] Wg and [x, r*h] Wc on the matrix cores, v_mfma_f32_32x32x2_f32 (fp32 operands, fp32
// accumulation).  A workgroup takes 32 sessions: their [x, h] rows sit in LDS (row stride K + 1: the 32 rows of a column then
// fall into 32 banks) as the A operand; a wavefront owns 32-column tiles of the output and streams the weight rows of its
// columns from L2 (coalesced 128-byte reads, two 16-k chunks ahead); the gates' r*h passes to the candidate phase through LDS,
// u in the registers of the wavefront that owns those columns in both phases (50 KB of LDS: three workgroups per CU).
// 1 M sessions, d = 128 (tools/gru_sweep_lab.py): 3.19 ms per step = 62 TFLOP/s, the vector-FMA kernel (16 sessions per
// workgroup) 6.58 ms = 30 TFLOP/s.  (With u through LDS and two workgroups per CU: 3.99 ms.  Hardware exp2 / reciprocal in the
// activations gave 3 % and were not taken: the sweep would round differently from the training step.)  Sums are formed
// pairwise in k inside an MFMA and then in k order: equal to the FMA chain to rounding.
// ------------------------------------------------------------------------------------------------
typedef float gru_f32x16 __attribute__((ext_vector_type(16)));
constexpr int GM_ROWS = 32;

template <int H>
__global__ __launch_bounds__(G_T) void gru_fwd_mfma_kernel(GruIn g, const uint8_t* __restrict__ active,
                                                           const float* __restrict__ Wg, const float* __restrict__ bg,
                                                           const float* __restrict__ Wc, const float* __restrict__ bc, int act,
                                                           float* __restrict__ r_out, float* __restrict__ u_out,
                                                           float* __restrict__ c_out, float* __restrict__ h_new) {
    __shared__ float a[GM_ROWS][G_KMAX + 1];
    __shared__ float rh[GM_ROWS][H + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int IN = g.in_dim, K = IN + H;
    const int row0 = blockIdx.x * GM_ROWS;
    // staging: a wavefront takes 8 sessions, lanes along the row (coalesced, no index arithmetic per element, all of a
    // session's loads issued together); column K is a zero that the odd-K tail multiplies
#pragma unroll
    for (int rr = 0; rr < GM_ROWS / (G_T / 64); ++rr) {
        const int r = wv * (GM_ROWS / (G_T / 64)) + rr, row = row0 + r;
        const bool ok = row < g.B;
        const float* __restrict__ xr = ok ? x_row(g, row) : g.x;
        const float* __restrict__ hr = g.h + static_cast<int64_t>(ok ? row : 0) * H;
#pragma unroll
        for (int e = 0; e < (G_KMAX + 64) / 64; ++e) {
            const int k = lane + 64 * e;
            if (k <= K) a[r][k] = (ok && k < K) ? (k < IN ? xr[k] : hr[k - IN]) : 0.0f;
        }
    }
    __syncthreads();
    const int m = lane & 31, kh = lane >> 5;            // A[row m][k + kh], B[k + kh][col m]
    // NT 32 x 32 output tiles (columns col0 + tstride t): acc[t] += A * W(:, col0 + tstride t ...).  Rows [0, n1) of W meet
    // src's columns, rows [n1, n1 + n2) meet src2's (r*h in the candidate phase).  Two tiles share every A value and give the
    // matrix pipe two independent accumulation chains.
    auto tiles = [&](auto nt_c, gru_f32x16* acc, const float (*src)[G_KMAX + 1], const float (*src2)[H + 1], int n1, int n2,
                     const float* __restrict__ W, int ld, int col0, int tstride) {
        constexpr int NT = decltype(nt_c)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        const float* __restrict__ wcol = W + col0 + m;
        const int n = n1 + n2;
        // the operands of the next 16 k are fetched before the MFMAs of the current 16 are issued (two register sets): without
        // that every chunk waits out an L2 round trip with the matrix pipe idle
        float b[2][NT][8], av[2][8];
        auto fetch = [&](int k0, float (&bb)[NT][8], float (&aa)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + 2 * q + kh;
#pragma unroll
                for (int t = 0; t < NT; ++t) bb[t][q] = k < n ? wcol[static_cast<int64_t>(k) * ld + tstride * t] : 0.0f;
                aa[q] = k < n1 ? src[m][k] : (k < n ? (src2 ? src2[m][k - n1] : src[m][k]) : 0.0f);
            }
        };
        fetch(0, b[0], av[0]);
        for (int k0 = 0; k0 < n; k0 += 32) {
            if (k0 + 16 < n) fetch(k0 + 16, b[1], av[1]);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][q], b[0][t][q], acc[t], 0, 0, 0);
            if (k0 + 16 >= n) break;
            if (k0 + 32 < n) fetch(k0 + 32, b[0], av[0]);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][q], b[1][t][q], acc[t], 0, 0, 0);
        }
    };
    // accumulator element i of a lane: row 8 * (i / 4) + 4 * kh + i % 4, column m
    // A wavefront takes the r tile and the u tile of the SAME 32 hidden columns (two accumulation chains that share every A
    // value), then -- behind the barrier that publishes r*h -- the candidate tile of those columns: u stays in its registers.
    // (Hidden sizes below 128 leave wavefronts without a tile: 64 -> two work, 32 -> one.)
    const int ct = wv;
    const bool mine = ct < H / 32;
    float ureg[16];
    if (mine) {
        gru_f32x16 acc[2];
        tiles(std::integral_constant<int, 2>{}, acc, a, nullptr, K, 0, Wg, 2 * H, ct * 32, H);
        const int j = ct * 32 + m;
        const float b_r = bg[j], b_u = bg[H + j];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 8 * (i >> 2) + 4 * kh + (i & 3), row = row0 + r;
            const float rg = sigmoidf_(acc[0][i] + b_r);
            ureg[i] = sigmoidf_(acc[1][i] + b_u);
            rh[r][j] = rg * a[r][IN + j];
            if (r_out && row < g.B) r_out[static_cast<int64_t>(row) * H + j] = rg;
            if (u_out && row < g.B) u_out[static_cast<int64_t>(row) * H + j] = ureg[i];
        }
    }
    __syncthreads();
    if (mine) {                                            // candidate and new state
        gru_f32x16 acc1[1];
        tiles(std::integral_constant<int, 1>{}, acc1, a, rh, IN, H, Wc, H, ct * 32, 0);
        const gru_f32x16& acc = acc1[0];
        const int j = ct * 32 + m;
        const float b = bc[j];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 8 * (i >> 2) + 4 * kh + (i & 3), row = row0 + r;
            if (row < g.B) {
                const float c = hidden_act(acc[i] + b, act);
                const float ho = a[r][IN + j], u = ureg[i];
                float hn = u * ho + (1.0f - u) * c;
                if (active && !active[row]) hn = ho;        // finished history: the state is carried
                if (c_out) c_out[static_cast<int64_t>(row) * H + j] = c;
                h_new[static_cast<int64_t>(row) * H + j] = hn;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1 (per session): pre-activation gradients and dL/dx
//   dc~ = dh' (1-u) act'(c) ;  du~ = dh' (h - c) u (1-u) ;  d(rh) = dc~ Wc[in:,:]^T ;  dr~ = d(rh) h r (1-r)
//   dx  = dc~ Wc[:in,:]^T + [dr~ | du~] Wg[:in,:]^T
// ------------------------------------------------------------------------------------------------
// sixteen per-lane partial sums -> their sixteen totals in 17 shuffles instead of 16 x 6: each butterfly step hands the
// half of the values a lane does not keep to the partner that does.  Lane L ends with the total of value (L >> 2) & 15.
__device__ __forceinline__ float reduce16(float (&v)[16], int lane) {
    float w8[8], w4[4], w2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = (b5 ? v[8 + i] : v[i]) + __shfl_xor(b5 ? v[i] : v[8 + i], 32, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (b4 ? w8[4 + i] : w8[i]) + __shfl_xor(b4 ? w8[i] : w8[4 + i], 16, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = (b3 ? w4[2 + i] : w4[i]) + __shfl_xor(b3 ? w4[i] : w4[2 + i], 8, 64);
    float t = (b2 ? w2[1] : w2[0]) + __shfl_xor(b2 ? w2[0] : w2[1], 4, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 1, 64);
    return t;
}

// The two products with TRANSPOSED weights are taken a weight row per wavefront, lanes along the row (coalesced reads, 16 / ROWS
// rows in flight), partial products of all ROWS sessions reduced across the lanes by reduce16.  (A thread per (session,
// output) walking along its weight row reads 64 different cache lines per wave instruction: 31 us per call at B = 128.)
// T threads: 256, or (a training batch: few workgroups, each wavefront walking its weight rows one group behind the other) 1 024 --
// sixteen wavefronts share the same 4 rows and take a quarter of the weight rows each
template <int ROWS, int H, int T = G_T>
__global__ __launch_bounds__(T) void gru_bwd_rows_kernel(GruIn g, const float* __restrict__ Wg,
                                                           const float* __restrict__ Wc, int act,
                                                           const float* __restrict__ r_in, const float* __restrict__ u_in,
                                                           const float* __restrict__ c_in, const float* __restrict__ dh_new,
                                                           float* __restrict__ dcp_out, float* __restrict__ dg_out,
                                                           float* __restrict__ dx_out) {
    __shared__ float dcp[ROWS][G_HMAX];
    __shared__ float dg[ROWS][2 * G_HMAX];
    __shared__ float hh[ROWS][G_HMAX];
    __shared__ float rr[ROWS][G_HMAX];
    const int tid = threadIdx.x;
    const int IN = g.in_dim;
    const int row0 = blockIdx.x * ROWS;
    for (int idx = tid; idx < ROWS * H; idx += T) {
        const int r = idx / H, j = idx - r * H, row = row0 + r;
        float vdc = 0.0f, vdu = 0.0f, vh = 0.0f, vr = 0.0f;
        if (row < g.B) {
            const int64_t o = static_cast<int64_t>(row) * H + j;
            const float d = dh_new[o], u = u_in[o], c = c_in[o];
            vh = g.h[o];
            vr = r_in[o];
            vdc = d * (1.0f - u) * hidden_act_grad(c, act);
            vdu = d * (vh - c) * u * (1.0f - u);
        }
        dcp[r][j] = vdc;
        dg[r][H + j] = vdu;
        hh[r][j] = vh;
        rr[r][j] = vr;
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int QN = 16 / ROWS;                      // weight rows per reduction group (ROWS is 4 or 16)
    constexpr int EH = (H + 63) / 64, EG = (2 * H + 63) / 64;
    const int vq = ((lane >> 2) & 15) / ROWS, vr_ = ((lane >> 2) & 15) % ROWS;     // the value this lane ends up holding
    for (int k0 = wv * QN; k0 < H; k0 += (T / 64) * QN) {      // d(rh) = dc~ Wc[in:, :]^T ; dr~ = d(rh) h r (1 - r)
        float v[16];
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            float w[EH];
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                const int j = lane + 64 * e;
                const float x = Wc[static_cast<int64_t>(IN + k0 + q) * H + (j < H ? j : 0)];
                w[e] = j < H ? x : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                float part = 0.0f;
#pragma unroll
                for (int e = 0; e < EH; ++e) part = fmaf(dcp[r][(lane + 64 * e) & (G_HMAX - 1)], w[e], part);
                v[q * ROWS + r] = part;
            }
        }
        const float total = reduce16(v, lane);
        if ((lane & 3) == 0) {
            const int k = k0 + vq;
            const float rv = rr[vr_][k];
            dg[vr_][k] = total * hh[vr_][k] * rv * (1.0f - rv);
        }
    }
    __syncthreads();
    for (int i0 = wv * QN; i0 < IN; i0 += (T / 64) * QN) {      // dx = dc~ Wc[:in, :]^T + [dr~ | du~] Wg[:in, :]^T
        float v[16];
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int i = (i0 + q < IN) ? i0 + q : IN - 1;       // a row past the end is computed and not stored
            float wc[EH], wg[EG];
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                const int j = lane + 64 * e;
                const float x = Wc[static_cast<int64_t>(i) * H + (j < H ? j : 0)];
                wc[e] = j < H ? x : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < EG; ++e) {
                const int j = lane + 64 * e;
                const float x = Wg[static_cast<int64_t>(i) * 2 * H + (j < 2 * H ? j : 0)];
                wg[e] = j < 2 * H ? x : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                float part = 0.0f;
#pragma unroll
                for (int e = 0; e < EH; ++e) part = fmaf(dcp[r][(lane + 64 * e) & (G_HMAX - 1)], wc[e], part);
#pragma unroll
                for (int e = 0; e < EG; ++e) part = fmaf(dg[r][(lane + 64 * e) & (2 * G_HMAX - 1)], wg[e], part);
                v[q * ROWS + r] = part;
            }
        }
        const float total = reduce16(v, lane);
        const int i = i0 + vq, row = row0 + vr_;
        if ((lane & 3) == 0 && i < IN && row < g.B) dx_out[static_cast<int64_t>(row) * IN + i] = total;
    }
    for (int idx = tid; idx < ROWS * H; idx += T) {
        const int r = idx / H, j = idx - r * H, row = row0 + r;
        if (row >= g.B) continue;
        dcp_out[static_cast<int64_t>(row) * H + j] = dcp[r][j];
        dg_out[static_cast<int64_t>(row) * 2 * H + j] = dg[r][j];
        dg_out[static_cast<int64_t>(row) * 2 * H + H + j] = dg[r][H + j];
    }
}

// backward, part 2 (per weight): dWg += [x,h]^T [dr~|du~] ; dWc += [x, r*h]^T dc~ ; bias sums.
// One thread per weight, a loop over the (few) sessions of the batch.
// the input-embedding gradient of the FIRST layer (scatter_add_rows_kernel's body, below) may ride in the weights launch: it
// only needs dx, which the rows kernel in front of both has written, and its 32 workgroups (4.6 us as a launch of their own)
// disappear among the weights' ~390
struct ScatterArgs {
    const float* src;          // [n, dim]  (NULL: no scatter in this launch)
    const int32_t* index;
    int n, dim;
    const float* table;
    float reg;
    float* g_table;
    uint8_t* touch;
    const float* touch_base;
    int first_block;           // workgroups from this one on scatter
};
__device__ void scatter_add_rows_body(int block_x, const ScatterArgs& a);

__global__ __launch_bounds__(G_T) void gru_bwd_weights_kernel(GruIn g, const float* __restrict__ r_in,
                                                              const float* __restrict__ dcp, const float* __restrict__ dgp,
                                                              float* __restrict__ gWg, float* __restrict__ gbg,
                                                              float* __restrict__ gWc, float* __restrict__ gbc, ScatterArgs sc) {
    if (sc.src && static_cast<int>(blockIdx.x) >= sc.first_block) {
        scatter_add_rows_body(static_cast<int>(blockIdx.x) - sc.first_block, sc);
        return;
    }
    const int IN = g.in_dim, H = g.hid, K = IN + H;
    const int64_t n_g = static_cast<int64_t>(K) * 2 * H, n_c = static_cast<int64_t>(K) * H;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * G_T + threadIdx.x;
    // sum over the sessions of A[b][k] * D[b * ld + j]; where A comes from is decided ONCE, outside the loop: a choice inside it
    // (input or state column, gathered or dense input rows) kept the loads of consecutive sessions from being issued together
    // (36 us per call at B = 128; the sums and their order are unchanged)
    auto dot = [&](int k, const float* __restrict__ D, int ld, int j, const float* __restrict__ hsrc, const float* __restrict__ rsrc) {
        float s = 0.0f;
        if (k < IN) {
            if (g.x_index) {
#pragma unroll 32
                for (int b = 0; b < g.B; ++b)
                    s = fmaf(g.x[static_cast<int64_t>(g.x_index[b]) * IN + k], D[static_cast<int64_t>(b) * ld + j], s);
            } else {
#pragma unroll 32
                for (int b = 0; b < g.B; ++b) s = fmaf(g.x[static_cast<int64_t>(b) * IN + k], D[static_cast<int64_t>(b) * ld + j], s);
            }
        } else if (rsrc) {
#pragma unroll 32
            for (int b = 0; b < g.B; ++b) {
                const int64_t o = static_cast<int64_t>(b) * H + (k - IN);
                s = fmaf(rsrc[o] * hsrc[o], D[static_cast<int64_t>(b) * ld + j], s);
            }
        } else {
#pragma unroll 32
            for (int b = 0; b < g.B; ++b) s = fmaf(hsrc[static_cast<int64_t>(b) * H + (k - IN)], D[static_cast<int64_t>(b) * ld + j], s);
        }
        return s;
    };
    if (t < n_g) {
        const int k = static_cast<int>(t / (2 * H)), j = static_cast<int>(t - static_cast<int64_t>(k) * 2 * H);
        gWg[t] += dot(k, dgp, 2 * H, j, g.h, nullptr);
    } else if (t < n_g + n_c) {
        const int64_t q = t - n_g;
        const int k = static_cast<int>(q / H), j = static_cast<int>(q - static_cast<int64_t>(k) * H);
        gWc[q] += dot(k, dcp, H, j, g.h, r_in);
    } else if (t < n_g + n_c + 2 * H) {
        const int j = static_cast<int>(t - n_g - n_c);
        float s = 0.0f;
        for (int b = 0; b < g.B; ++b) s += dgp[static_cast<int64_t>(b) * 2 * H + j];
        gbg[j] += s;
    } else if (t < n_g + n_c + 3 * H) {
        const int j = static_cast<int>(t - n_g - n_c - 2 * H);
        float s = 0.0f;
        for (int b = 0; b < g.B; ++b) s += dcp[static_cast<int64_t>(b) * H + j];
        gbc[j] += s;
    }
}

// ------------------------------------------------------------------------------------------------
// loss, three launches: (a) logits[b, y] = final_act(out[b] . E[Y[y]] + bias[Y[y]]) as a small GEMM whose
// target rows are read once; (b) per session b the row-wise loss and its gradient; (c) dL/dout, again
// with every target row read once.  The session's own positive is column b.  _softmax_neg masks that column, bpr_max / top1_max weigh the
// pairwise terms with it; the hand-derived gradient (checked against autograd in the tests):
//   s_y = softmax over y != b of l_y (the masked column enters the max as 0, GRU4RecPlus.py:139-141)
//   bpr_max : P = sum s_y sig(l_b - l_y), R = sum s_y l_y^2, L = -log(P + 1e-24) + lam R
//             dL/dl_y = -[ s_y (sig_y - P) - sig_y (1 - sig_y) s_y ] / (P + eps) + lam [ 2 l_y s_y + s_y (l_y^2 - R) ]
//             dL/dl_b = -[ sum_y sig_y (1 - sig_y) s_y ] / (P + eps)
//   top1_max: q_y = sig(l_y - l_b) + sig(l_y^2), L = sum s_y q_y
//             dL/dl_y = s_y [ sig'(l_y - l_b) + 2 l_y sig'(l_y^2) ] + s_y (q_y - L) ;  dL/dl_b = -sum_y s_y sig'(l_y - l_b)
// ------------------------------------------------------------------------------------------------
constexpr int L_NY_MAX = 8192;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < G_T / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

__device__ __forceinline__ float final_act_fwd(float x, int kind) {
    return kind == 0 ? x : (kind == 1 ? fmaxf(x, 0.0f) : fmaxf(x, 0.2f * x));
}
__device__ __forceinline__ float final_act_grad(float l, int kind) {  // through the OUTPUT l
    return kind == 0 ? 1.0f : (kind == 1 ? (l > 0.0f ? 1.0f : 0.0f) : (l > 0.0f ? 1.0f : 0.2f));
}

// (a) logits[b, y] = final_act(out[b] . E[Y[y]] + bias[Y[y]]): one workgroup per 16 targets, whose rows are
//     read ONCE (coalesced) into LDS and used by every session -- b * n_y dot products, n_y row reads.
constexpr int L_TY = 16;

template <int H>      // compile-time hidden size: index arithmetic by shifts, staging loads issued together
__global__ __launch_bounds__(G_T) void session_logits_kernel(const float* __restrict__ out, int B, int /*hid*/,
                                                             const float* __restrict__ E, const float* __restrict__ bias,
                                                             const int32_t* __restrict__ Y, int n_y, int fact,
                                                             float* __restrict__ logits, float* __restrict__ dout_clear,
                                                             int n_clear, float* __restrict__ loss_clear) {
    __shared__ float et[L_TY][G_HMAX + 4];
    __shared__ float bs[L_TY];
    const int tid = threadIdx.x, y0 = blockIdx.x * L_TY;
    // the two accumulators of the launches behind this one start from zero: dL/dout (session_dout_kernel adds into it) and the
    // loss word (session_rowloss_kernel adds the rows' shares) -- cleared here instead of by a memset and a fill of their own
    for (int i = blockIdx.x * G_T + tid; i < n_clear; i += gridDim.x * G_T) dout_clear[i] = 0.0f;
    if (blockIdx.x == 0 && tid == 0) *loss_clear = 0.0f;
    for (int idx = tid; idx < L_TY * H; idx += G_T) {
        const int y = idx / H, d = idx - y * H;
        et[y][d] = (y0 + y < n_y) ? E[static_cast<int64_t>(Y[y0 + y]) * H + d] : 0.0f;
    }
    if (tid < L_TY) bs[tid] = (y0 + tid < n_y) ? bias[Y[y0 + tid]] : 0.0f;
    __syncthreads();
    for (int idx = tid; idx < 2 * B; idx += G_T) {     // (session, half of the 16 targets)
        const int b = idx >> 1, q0 = (idx & 1) * (L_TY / 2);
        const float4* o = reinterpret_cast<const float4*>(out + static_cast<int64_t>(b) * H);
        float acc[L_TY / 2];
#pragma unroll
        for (int q = 0; q < L_TY / 2; ++q) acc[q] = 0.0f;
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 v = o[k4];
#pragma unroll
            for (int q = 0; q < L_TY / 2; ++q) {
                const float* e = &et[q0 + q][4 * k4];
                acc[q] = fmaf(v.x, e[0], acc[q]); acc[q] = fmaf(v.y, e[1], acc[q]);
                acc[q] = fmaf(v.z, e[2], acc[q]); acc[q] = fmaf(v.w, e[3], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < L_TY / 2; ++q) {
            const int y = y0 + q0 + q;
            if (y < n_y) logits[static_cast<int64_t>(b) * n_y + y] = final_act_fwd(acc[q] + bs[q0 + q], fact);
        }
    }
}

// (b) per session: softmax over the other targets, loss, gradient w.r.t. the pre-activation logits (in place)
// pos_off / B_mean: session row r of this launch is slot pos_off + r of a batch of B_mean sessions (its positive is
// column pos_off + r, the loss is the mean over B_mean) -- one rank's share of a session-sharded step; 0 / B otherwise
__global__ __launch_bounds__(G_T) void session_rowloss_kernel(int B, int n_y, int fact, int loss_kind, float bpr_reg,
                                                              float* __restrict__ dlogits, float* __restrict__ loss,
                                                              int pos_off, int B_mean) {
    __shared__ float lg[L_NY_MAX];
    __shared__ float red[G_T / 64];
    const int tid = threadIdx.x;
    float* row = dlogits + static_cast<int64_t>(blockIdx.x) * n_y;
    const int b = pos_off + static_cast<int>(blockIdx.x);       // column of this session's positive
    float mx = 0.0f;   // the masked column contributes a 0 to the max
    for (int y = tid; y < n_y; y += G_T) {
        const float l = row[y];
        lg[y] = l;
        if (y != b) mx = fmaxf(mx, l);
    }
    mx = block_reduce(mx, red, true);
    const float pos = lg[b];
    float z = 0.0f;
    for (int y = tid; y < n_y; y += G_T)
        if (y != b) z += expf(lg[y] - mx);
    z = block_reduce(z, red, false);
    // first moments: P and R (bpr_max) or L (top1_max), plus the gradient of the positive
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int y = tid; y < n_y; y += G_T) {
        if (y == b) continue;
        const float l = lg[y], s = expf(l - mx) / z;
        if (loss_kind == 0) {
            const float sg = sigmoidf_(pos - l);
            a0 += sg * s;
            a1 += l * l * s;
            a2 += sg * (1.0f - sg) * s;
        } else {
            const float s1 = sigmoidf_(l - pos), s2 = sigmoidf_(l * l);
            a0 += (s1 + s2) * s;
            a2 += s1 * (1.0f - s1) * s;
        }
    }
    a0 = block_reduce(a0, red, false);
    a1 = block_reduce(a1, red, false);
    a2 = block_reduce(a2, red, false);
    const float invB = 1.0f / static_cast<float>(B_mean);
    float lb, gpos;
    if (loss_kind == 0) {
        lb = -logf(a0 + 1e-24f) + bpr_reg * a1;
        gpos = -a2 / (a0 + 1e-24f);
    } else {
        lb = a0;
        gpos = -a2;
    }
    if (tid == 0) atomicAdd(loss, lb * invB);
    for (int y = tid; y < n_y; y += G_T) {
        const float l = lg[y];
        float gy;
        if (y == b) {
            gy = gpos;
        } else {
            const float s = expf(l - mx) / z;
            if (loss_kind == 0) {
                const float sg = sigmoidf_(pos - l);
                const float dP = s * (sg - a0) - sg * (1.0f - sg) * s;
                const float dR = 2.0f * l * s + s * (l * l - a1);
                gy = -dP / (a0 + 1e-24f) + bpr_reg * dR;
            } else {
                const float s1 = sigmoidf_(l - pos), s2 = sigmoidf_(l * l);
                gy = s * (s1 * (1.0f - s1) + 2.0f * l * s2 * (1.0f - s2)) + s * ((s1 + s2) - a0);
            }
        }
        row[y] = gy * invB * final_act_grad(l, fact);
    }
}

// (c) dL/dout[b] += sum over this workgroup's 64 targets of dlogits[b, y] E[Y[y]] for 16 sessions; target rows
//     and gradient tile are staged in LDS, partial sums meet in dout through atomics (cleared by the caller)
constexpr int L_CY = 64;
constexpr int L_CB = 16;

template <int H>
__device__ __forceinline__ void session_dout_body(int block_x, int block_y, const float* __restrict__ dlogits, int B,
                                                  const float* __restrict__ E, const int32_t* __restrict__ Y, int n_y,
                                                  float* __restrict__ dout) {
    __shared__ float et[L_CY][G_HMAX];
    __shared__ float gl[L_CB][L_CY + 1];
    const int tid = threadIdx.x, y0 = block_x * L_CY, b0 = block_y * L_CB;
    const int ny = (n_y - y0) < L_CY ? (n_y - y0) : L_CY;
    for (int idx = tid; idx < ny * H; idx += G_T) {
        const int y = idx / H, d = idx - y * H;
        et[y][d] = E[static_cast<int64_t>(Y[y0 + y]) * H + d];
    }
    for (int idx = tid; idx < L_CB * L_CY; idx += G_T) {
        const int r = idx / L_CY, y = idx - r * L_CY;
        gl[r][y] = (b0 + r < B && y < ny) ? dlogits[static_cast<int64_t>(b0 + r) * n_y + y0 + y] : 0.0f;
    }
    __syncthreads();
    constexpr int ng = G_T / H, per = L_CB / ng;             // ng in {2, 4, 8}: L_CB / ng sessions per thread
    const int d = tid % H, grp = tid / H;
    float acc[per];
#pragma unroll
    for (int i = 0; i < per; ++i) acc[i] = 0.0f;
    for (int y = 0; y < ny; ++y) {
        const float e = et[y][d];
#pragma unroll
        for (int i = 0; i < per; ++i) acc[i] = fmaf(gl[grp + i * ng][y], e, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < per; ++i) {
        const int b = b0 + grp + i * ng;
        if (b < B) atomicAdd(&dout[static_cast<int64_t>(b) * H + d], acc[i]);
    }
}

template <int H>
__global__ __launch_bounds__(G_T) void session_dout_kernel(const float* __restrict__ dlogits, int B, int /*hid*/,
                                                           const float* __restrict__ E, const int32_t* __restrict__ Y,
                                                           int n_y, float* __restrict__ dout) {
    session_dout_body<H>(blockIdx.x, blockIdx.y, dlogits, B, E, Y, n_y, dout);
}

// mark the 64-float blocks of the flat gradient buffer that [a, a+n) overlaps (DenseAdam's touch bytes)
__device__ __forceinline__ void mark_range(uint8_t* touch, const float* base, const float* a, int n) {
    const int64_t b0 = (a - base) >> 6, b1 = (a + n - 1 - base) >> 6;
    for (int64_t k = b0; k <= b1; ++k)
        if (touch[k] == 0) touch[k] = 1;
}

// output-side gradients: one wavefront per target y.
//   gE[Y[y]] += sum_b dlogits[b, y] out[b] + reg E[Y[y]] ;  gb[Y[y]] += sum_b dlogits[b, y] + reg bias[Y[y]]
// (an item that occurs several times in Y is counted each time, like tf.gather's gradient and l2_loss)
template <int H>
__device__ __forceinline__ void session_out_grads_body(int block_x, const float* __restrict__ dlogits,
                                                       const float* __restrict__ out, int B,
                                                       const int32_t* __restrict__ Y, int n_y,
                                                       const float* __restrict__ E, const float* __restrict__ bias,
                                                       float reg, float* __restrict__ gE, float* __restrict__ gb,
                                                       uint8_t* __restrict__ touch, const float* touch_base) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = block_x * (G_T / 64) + wv;
    if (y >= n_y) return;
    const int64_t item = Y[y];
    float acc0 = 0.0f, acc1 = 0.0f, colsum = 0.0f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int bl = b0 + lane;
        const float mine = bl < B ? dlogits[static_cast<int64_t>(bl) * n_y + y] : 0.0f;
        colsum += mine;
        const int lim = (B - b0) < 64 ? (B - b0) : 64;
#pragma unroll 16
        for (int k = 0; k < lim; ++k) {
            const float gk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), k));   // k is wave-uniform
            const float* o = out + static_cast<int64_t>(b0 + k) * H;
            acc0 = fmaf(gk, o[lane & (H - 1)], acc0);        // H < 64: lanes >= H repeat a row element and are not stored
            if (H > 64) acc1 = fmaf(gk, o[(64 + lane) & (H - 1)], acc1);
        }
    }
    colsum = skr::wave_sum(colsum);
    float* ge = gE + item * H;
    if (lane < H) atomicAdd(&ge[lane], acc0 + reg * E[item * H + lane]);
    if (H > 64) atomicAdd(&ge[64 + lane], acc1 + reg * E[item * H + 64 + lane]);
    if (lane == 0) {
        atomicAdd(&gb[item], colsum + reg * bias[item]);
        if (touch) {
            mark_range(touch, touch_base, ge, H);
            mark_range(touch, touch_base, &gb[item], 1);
        }
    }
}

template <int H>
__global__ __launch_bounds__(G_T) void session_out_grads_kernel(const float* __restrict__ dlogits,
                                                                const float* __restrict__ out, int B, int /*hid*/,
                                                                const int32_t* __restrict__ Y, int n_y,
                                                                const float* __restrict__ E, const float* __restrict__ bias,
                                                                float reg, float* __restrict__ gE, float* __restrict__ gb,
                                                                uint8_t* __restrict__ touch, const float* touch_base) {
    session_out_grads_body<H>(blockIdx.x, dlogits, out, B, Y, n_y, E, bias, reg, gE, gb, touch, touch_base);
}

// (c) and the output-side gradients in ONE launch: both only read dlogits, and (c)'s few dozen workgroups would otherwise
// hold the chip for a launch of their own (10.8 of a step's ~115 us).  Workgroups [0, n_dx * n_dy) take tiles of (c), the
// rest one target each; a workgroup takes one branch as a whole, so the barrier inside (c) is met by all its threads.
template <int H>
__global__ __launch_bounds__(G_T) void session_grads_kernel(const float* __restrict__ dlogits, const float* __restrict__ out, int B,
                                                            const int32_t* __restrict__ Y, int n_y,
                                                            const float* __restrict__ E, const float* __restrict__ bias,
                                                            float reg, float* __restrict__ dout, float* __restrict__ gE,
                                                            float* __restrict__ gb, uint8_t* __restrict__ touch,
                                                            const float* touch_base, int n_dx, int n_dy) {
    const int n_d = n_dx * n_dy, bx = static_cast<int>(blockIdx.x);
    if (bx < n_d) session_dout_body<H>(bx % n_dx, bx / n_dx, dlogits, B, E, Y, n_y, dout);
    else session_out_grads_body<H>(bx - n_d, dlogits, out, B, Y, n_y, E, bias, reg, gE, gb, touch, touch_base);
}

// g_table[index[n]] += src[n] + reg * table[index[n]]   (the input-embedding gradient)
__device__ void scatter_add_rows_body(int block_x, const ScatterArgs& a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = block_x * (G_T / 64) + wv;
    if (r >= a.n) return;
    const int64_t row = a.index[r];
    for (int d = lane; d < a.dim; d += 64) {
        float v = a.src[static_cast<int64_t>(r) * a.dim + d];
        if (a.reg != 0.0f) v += a.reg * a.table[row * a.dim + d];
        atomicAdd(&a.g_table[row * a.dim + d], v);
    }
    if (a.touch && lane == 0) mark_range(a.touch, a.touch_base, a.g_table + row * a.dim, a.dim);
}

__global__ __launch_bounds__(G_T) void scatter_add_rows_kernel(const float* __restrict__ src,
                                                               const int32_t* __restrict__ index, int n, int dim,
                                                               const float* __restrict__ table, float reg,
                                                               float* __restrict__ g_table, uint8_t* __restrict__ touch,
                                                               const float* touch_base) {
    scatter_add_rows_body(static_cast<int>(blockIdx.x), ScatterArgs{src, index, n, dim, table, reg, g_table, touch, touch_base, 0});
}

bool dims_ok(int in_dim, int hid) {
    return (hid == 32 || hid == 64 || hid == 128) && in_dim >= 1 && in_dim <= 128 && in_dim + hid <= G_KMAX;
}

}  // namespace

extern "C" {

int skr_gru_cell_fwd(const float* d_x, const int32_t* d_x_index, const float* d_h, const uint8_t* d_active, int B,
                     int in_dim, int hid, const float* d_Wg, const float* d_bg, const float* d_Wc, const float* d_bc,
                     int hidden_act_kind, float* d_r, float* d_u, float* d_c, float* d_h_new, void* stream) {
    SKR_REQUIRE(d_x && d_h && d_Wg && d_bg && d_Wc && d_bc && d_h_new, "skr_gru_cell_fwd: NULL argument");
    SKR_REQUIRE(dims_ok(in_dim, hid), "skr_gru_cell_fwd: hid must be 32, 64 or 128 and in_dim <= 128 (got %d, %d)", in_dim, hid);
    SKR_REQUIRE(hidden_act_kind == 0 || hidden_act_kind == 1, "There is not hidden_act named '%d'.", hidden_act_kind);
    SKR_REQUIRE(B >= 0 && d_h != d_h_new, "skr_gru_cell_fwd: bad batch or in-place state");
    if (B == 0) return SKR_OK;
    GruIn g{d_x, d_x_index, d_h, B, in_dim, hid};
    // few sessions (a training batch): 4 rows per workgroup so that more than a handful of CUs work;
    // many (the inference sweep): 16 rows, each weight read once per 16 sessions
    const bool aligned = in_dim % G_CH == 0;
    auto launch = [&](auto rows_c, auto hid_c, auto al_c) {
        constexpr int R = decltype(rows_c)::value, HH = decltype(hid_c)::value;
        constexpr bool AL = decltype(al_c)::value;
        hipLaunchKernelGGL((gru_fwd_kernel<R, HH, AL>), dim3((B + R - 1) / R), dim3(G_T), 0, skr::as_stream(stream), g, d_active, d_Wg,
                           d_bg, d_Wc, d_bc, hidden_act_kind, d_r, d_u, d_c, d_h_new);
    };
    auto by_align = [&](auto rows_c, auto hid_c) {
        if (aligned) launch(rows_c, hid_c, std::true_type{}); else launch(rows_c, hid_c, std::false_type{});
    };
    auto by_hid = [&](auto rows_c) {
        if (hid == 32) by_align(rows_c, std::integral_constant<int, 32>{});
        else if (hid == 64) by_align(rows_c, std::integral_constant<int, 64>{});
        else by_align(rows_c, std::integral_constant<int, 128>{});
    };
    // many sessions (the inference sweep): the matrix cores.  SKR_GRU_MFMA=0 keeps the vector kernel (16 sessions per workgroup).
    static const bool use_mfma = [] { const char* e = getenv("SKR_GRU_MFMA"); return !(e && atoi(e) == 0); }();
    if (B > G_SMALL_B && use_mfma) {
        const dim3 grid((B + GM_ROWS - 1) / GM_ROWS), wg(G_T);
        if (hid == 32)
            hipLaunchKernelGGL(gru_fwd_mfma_kernel<32>, grid, wg, 0, skr::as_stream(stream), g, d_active, d_Wg, d_bg, d_Wc, d_bc,
                               hidden_act_kind, d_r, d_u, d_c, d_h_new);
        else if (hid == 64)
            hipLaunchKernelGGL(gru_fwd_mfma_kernel<64>, grid, wg, 0, skr::as_stream(stream), g, d_active, d_Wg, d_bg, d_Wc, d_bc,
                               hidden_act_kind, d_r, d_u, d_c, d_h_new);
        else
            hipLaunchKernelGGL(gru_fwd_mfma_kernel<128>, grid, wg, 0, skr::as_stream(stream), g, d_active, d_Wg, d_bg, d_Wc, d_bc,
                               hidden_act_kind, d_r, d_u, d_c, d_h_new);
    } else if (B <= G_SMALL_B) {
        // SKR_GRU_SPLIT=0 keeps gru_fwd_kernel at the benchmarked shape too
        static const bool split = [] { const char* e = getenv("SKR_GRU_SPLIT"); return !(e && atoi(e) == 0); }();
        if (split && in_dim == GS_H && hid == GS_H)
            hipLaunchKernelGGL(gru_fwd_split_kernel, dim3((B + GS_ROWS - 1) / GS_ROWS), dim3(GS_T), 0, skr::as_stream(stream), g, d_active,
                               d_Wg, d_bg, d_Wc, d_bc, hidden_act_kind, d_r, d_u, d_c, d_h_new);
        else
            by_hid(std::integral_constant<int, 4>{});
    } else by_hid(std::integral_constant<int, G_ROWS>{});
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

static int gru_cell_bwd_impl(const float* d_x, const int32_t* d_x_index, const float* d_h, int B, int in_dim, int hid,
                     const float* d_Wg, const float* d_Wc, int hidden_act_kind, const float* d_r, const float* d_u,
                     const float* d_c, const float* d_dh_new, float* d_gWg, float* d_gbg, float* d_gWc, float* d_gbc,
                     float* d_dx, float* d_work, ScatterArgs sc, void* stream) {
    SKR_REQUIRE(d_x && d_h && d_Wg && d_Wc && d_r && d_u && d_c && d_dh_new && d_gWg && d_gbg && d_gWc && d_gbc && d_dx &&
                d_work, "skr_gru_cell_bwd: NULL argument");
    SKR_REQUIRE(dims_ok(in_dim, hid), "skr_gru_cell_bwd: hid must be 32, 64 or 128 and in_dim <= 128 (got %d, %d)", in_dim, hid);
    SKR_REQUIRE(hidden_act_kind == 0 || hidden_act_kind == 1, "There is not hidden_act named '%d'.", hidden_act_kind);
    SKR_REQUIRE(B >= 0, "skr_gru_cell_bwd: negative batch");
    if (B == 0) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    GruIn g{d_x, d_x_index, d_h, B, in_dim, hid};
    float* dcp = d_work;                                   // [B, hid]
    float* dgp = d_work + static_cast<int64_t>(B) * hid;   // [B, 2 hid]
    static const bool wide = [] { const char* e = getenv("SKR_GRU_SPLIT"); return !(e && atoi(e) == 0); }();
    auto rows = [&](auto rows_c, auto hid_c) {
        constexpr int R = decltype(rows_c)::value, HH = decltype(hid_c)::value;
        if (R == 4 && wide)
            hipLaunchKernelGGL((gru_bwd_rows_kernel<R, HH, 1024>), dim3((B + R - 1) / R), dim3(1024), 0, st, g, d_Wg, d_Wc, hidden_act_kind,
                               d_r, d_u, d_c, d_dh_new, dcp, dgp, d_dx);
        else
            hipLaunchKernelGGL((gru_bwd_rows_kernel<R, HH>), dim3((B + R - 1) / R), dim3(G_T), 0, st, g, d_Wg, d_Wc, hidden_act_kind, d_r,
                               d_u, d_c, d_dh_new, dcp, dgp, d_dx);
    };
    auto by_hid = [&](auto rows_c) {
        if (hid == 32) rows(rows_c, std::integral_constant<int, 32>{});
        else if (hid == 64) rows(rows_c, std::integral_constant<int, 64>{});
        else rows(rows_c, std::integral_constant<int, 128>{});
    };
    if (B <= G_SMALL_B) by_hid(std::integral_constant<int, 4>{}); else by_hid(std::integral_constant<int, G_ROWS>{});
    SKR_LAUNCH_CHECK();
    const int64_t n_out = static_cast<int64_t>(in_dim + hid) * 3 * hid + 3 * hid;
    const unsigned n_w = static_cast<unsigned>((n_out + G_T - 1) / G_T);
    sc.first_block = static_cast<int>(n_w);
    const unsigned n_s = sc.src ? static_cast<unsigned>((sc.n + G_T / 64 - 1) / (G_T / 64)) : 0u;
    hipLaunchKernelGGL(gru_bwd_weights_kernel, dim3(n_w + n_s), dim3(G_T), 0, st, g, d_r, dcp, dgp, d_gWg, d_gbg, d_gWc, d_gbc, sc);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_gru_cell_bwd(const float* d_x, const int32_t* d_x_index, const float* d_h, int B, int in_dim, int hid,
                     const float* d_Wg, const float* d_Wc, int hidden_act_kind, const float* d_r, const float* d_u,
                     const float* d_c, const float* d_dh_new, float* d_gWg, float* d_gbg, float* d_gWc, float* d_gbc,
                     float* d_dx, float* d_work, void* stream) {
    return gru_cell_bwd_impl(d_x, d_x_index, d_h, B, in_dim, hid, d_Wg, d_Wc, hidden_act_kind, d_r, d_u, d_c, d_dh_new, d_gWg, d_gbg,
                             d_gWc, d_gbc, d_dx, d_work, ScatterArgs{}, stream);
}

int skr_gru_cell_bwd_scatter(const float* d_x, const int32_t* d_x_index, const float* d_h, int B, int in_dim, int hid,
                             const float* d_Wg, const float* d_Wc, int hidden_act_kind, const float* d_r, const float* d_u,
                             const float* d_c, const float* d_dh_new, float* d_gWg, float* d_gbg, float* d_gWc, float* d_gbc,
                             float* d_dx, float* d_work, float reg, float* d_g_table, uint8_t* d_touch, const float* d_touch_base,
                             void* stream) {
    SKR_REQUIRE(d_x && d_x_index && d_g_table, "skr_gru_cell_bwd_scatter: the first layer's gathered input (table + index) and its gradient table");
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
    return gru_cell_bwd_impl(d_x, d_x_index, d_h, B, in_dim, hid, d_Wg, d_Wc, hidden_act_kind, d_r, d_u, d_c, d_dh_new, d_gWg, d_gbg,
                             d_gWc, d_gbc, d_dx, d_work, ScatterArgs{d_dx, d_x_index, B, in_dim, d_x, reg, d_g_table, d_touch, d_touch_base, 0},
                             stream);
}

static int session_loss_launch(const float* d_out, int B, int hid, const float* d_item_table, const float* d_item_bias,
                               const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                               float* d_dout, float* d_loss, int pos_off, int B_mean, void* stream,
                               bool with_grads = false, float reg = 0.0f, float* d_g_table = nullptr, float* d_g_bias = nullptr,
                               uint8_t* d_touch = nullptr, const float* d_touch_base = nullptr) {
    SKR_REQUIRE(d_out && d_item_table && d_item_bias && d_y && d_dlogits && d_dout && d_loss, "skr_session_loss: NULL argument");
    SKR_REQUIRE(!with_grads || (d_g_table && d_g_bias), "skr_session_loss_grads: NULL argument");
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
    SKR_REQUIRE(hid == 32 || hid == 64 || hid == 128, "skr_session_loss: hid must be 32, 64 or 128 (got %d)", hid);
    SKR_REQUIRE(B >= 1 && pos_off >= 0 && pos_off + B <= B_mean && n_y >= B_mean && n_y <= L_NY_MAX,
                "skr_session_loss: need slots [%d, %d) inside a batch of %d <= n_y = %d <= %d", pos_off, pos_off + B, B_mean, n_y, L_NY_MAX);
    SKR_REQUIRE(final_act_kind >= 0 && final_act_kind <= 2, "There is not final_act named '%d'.", final_act_kind);
    SKR_REQUIRE(loss_kind == 0 || loss_kind == 1, "There is not loss named '%d'.", loss_kind);
    hipStream_t st = skr::as_stream(stream);
if (hid == 32) hipLaunchKernelGGL(session_logits_kernel<32>, dim3((n_y + L_TY - 1) / L_TY), dim3(G_T), 0, st, d_out, B, hid, d_item_table,
                       d_item_bias, d_y, n_y, final_act_kind, d_dlogits, d_dout, B * hid, d_loss);
    else if (hid == 64) hipLaunchKernelGGL(session_logits_kernel<64>, dim3((n_y + L_TY - 1) / L_TY), dim3(G_T), 0, st, d_out, B, hid, d_item_table,
                       d_item_bias, d_y, n_y, final_act_kind, d_dlogits, d_dout, B * hid, d_loss);
    else hipLaunchKernelGGL(session_logits_kernel<128>, dim3((n_y + L_TY - 1) / L_TY), dim3(G_T), 0, st, d_out, B, hid, d_item_table,
                       d_item_bias, d_y, n_y, final_act_kind, d_dlogits, d_dout, B * hid, d_loss);
    SKR_LAUNCH_CHECK();
    hipLaunchKernelGGL(session_rowloss_kernel, dim3(B), dim3(G_T), 0, st, B, n_y, final_act_kind, loss_kind, bpr_reg,
                       d_dlogits, d_loss, pos_off, B_mean);
    SKR_LAUNCH_CHECK();
    if (with_grads) {
        const int n_dx = (n_y + L_CY - 1) / L_CY, n_dy = (B + L_CB - 1) / L_CB, n_g = (n_y + G_T / 64 - 1) / (G_T / 64);
        const dim3 grid(static_cast<unsigned>(n_dx * n_dy + n_g));
        if (hid == 32) hipLaunchKernelGGL(session_grads_kernel<32>, grid, dim3(G_T), 0, st, d_dlogits, d_out, B, d_y, n_y, d_item_table,
                                          d_item_bias, reg, d_dout, d_g_table, d_g_bias, d_touch, d_touch_base, n_dx, n_dy);
        else if (hid == 64) hipLaunchKernelGGL(session_grads_kernel<64>, grid, dim3(G_T), 0, st, d_dlogits, d_out, B, d_y, n_y, d_item_table,
                                               d_item_bias, reg, d_dout, d_g_table, d_g_bias, d_touch, d_touch_base, n_dx, n_dy);
        else hipLaunchKernelGGL(session_grads_kernel<128>, grid, dim3(G_T), 0, st, d_dlogits, d_out, B, d_y, n_y, d_item_table,
                                d_item_bias, reg, d_dout, d_g_table, d_g_bias, d_touch, d_touch_base, n_dx, n_dy);
        SKR_LAUNCH_CHECK();
        return SKR_OK;
    }
if (hid == 32) hipLaunchKernelGGL(session_dout_kernel<32>, dim3((n_y + L_CY - 1) / L_CY, (B + L_CB - 1) / L_CB), dim3(G_T), 0, st, d_dlogits,
                       B, hid, d_item_table, d_y, n_y, d_dout);
    else if (hid == 64) hipLaunchKernelGGL(session_dout_kernel<64>, dim3((n_y + L_CY - 1) / L_CY, (B + L_CB - 1) / L_CB), dim3(G_T), 0, st, d_dlogits,
                       B, hid, d_item_table, d_y, n_y, d_dout);
    else hipLaunchKernelGGL(session_dout_kernel<128>, dim3((n_y + L_CY - 1) / L_CY, (B + L_CB - 1) / L_CB), dim3(G_T), 0, st, d_dlogits,
                       B, hid, d_item_table, d_y, n_y, d_dout);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_session_loss(const float* d_out, int B, int hid, const float* d_item_table, const float* d_item_bias,
                     const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                     float* d_dout, float* d_loss, void* stream) {
    return session_loss_launch(d_out, B, hid, d_item_table, d_item_bias, d_y, n_y, final_act_kind, loss_kind, bpr_reg, d_dlogits,
                               d_dout, d_loss, 0, B, stream);
}

int skr_session_loss_sharded(const float* d_out, int B_local, int hid, const float* d_item_table, const float* d_item_bias,
                             const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                             float* d_dout, float* d_loss, int slot_offset, int B_global, void* stream) {
    return session_loss_launch(d_out, B_local, hid, d_item_table, d_item_bias, d_y, n_y, final_act_kind, loss_kind, bpr_reg,
                               d_dlogits, d_dout, d_loss, slot_offset, B_global, stream);
}

int skr_session_loss_grads(const float* d_out, int B_local, int hid, const float* d_item_table, const float* d_item_bias,
                           const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                           float* d_dout, float* d_loss, int slot_offset, int B_global, float reg, float* d_g_table,
                           float* d_g_bias, uint8_t* d_touch, const float* d_touch_base, void* stream) {
    return session_loss_launch(d_out, B_local, hid, d_item_table, d_item_bias, d_y, n_y, final_act_kind, loss_kind, bpr_reg,
                               d_dlogits, d_dout, d_loss, slot_offset, B_global, stream, true, reg, d_g_table, d_g_bias, d_touch,
                               d_touch_base);
}

int skr_session_out_grads(const float* d_dlogits, const float* d_out, int B, int hid, const int32_t* d_y, int n_y,
                          const float* d_item_table, const float* d_item_bias, float reg, float* d_g_table,
                          float* d_g_bias, uint8_t* d_touch, const float* d_touch_base, void* stream) {
    SKR_REQUIRE(d_dlogits && d_out && d_y && d_item_table && d_item_bias && d_g_table && d_g_bias,
                "skr_session_out_grads: NULL argument");
    SKR_REQUIRE(hid == 32 || hid == 64 || hid == 128, "skr_session_out_grads: hid must be 32, 64 or 128 (got %d)", hid);
    SKR_REQUIRE(B >= 1 && n_y >= 1, "skr_session_out_grads: empty batch");
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
if (hid == 32) hipLaunchKernelGGL(session_out_grads_kernel<32>, dim3((n_y + G_T / 64 - 1) / (G_T / 64)), dim3(G_T), 0,
                       skr::as_stream(stream), d_dlogits, d_out, B, hid, d_y, n_y, d_item_table, d_item_bias, reg, d_g_table,
                       d_g_bias, d_touch, d_touch_base);
    else if (hid == 64) hipLaunchKernelGGL(session_out_grads_kernel<64>, dim3((n_y + G_T / 64 - 1) / (G_T / 64)), dim3(G_T), 0,
                       skr::as_stream(stream), d_dlogits, d_out, B, hid, d_y, n_y, d_item_table, d_item_bias, reg, d_g_table,
                       d_g_bias, d_touch, d_touch_base);
    else hipLaunchKernelGGL(session_out_grads_kernel<128>, dim3((n_y + G_T / 64 - 1) / (G_T / 64)), dim3(G_T), 0,
                       skr::as_stream(stream), d_dlogits, d_out, B, hid, d_y, n_y, d_item_table, d_item_bias, reg, d_g_table,
                       d_g_bias, d_touch, d_touch_base);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

// popularity^alpha negatives (GRU4RecPlus.py:198-200): np.searchsorted(pop_cumsum, u) -- the first index whose cumulative
// weight is >= u -- for n uniforms.  u comes from the host (numpy's global stream, as the reference draws it) or, with
// d_uniform == NULL, from a counter-keyed generator on the device (splitmix64 of (seed, k): equal to the reference in law only)
__global__ void pop_sample_kernel(const double* __restrict__ cumsum, int n_items, const double* __restrict__ uniform,
                                  unsigned long long seed, int64_t n, int32_t* __restrict__ out) {
    const int64_t k = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    double u;
    if (uniform) {
        u = uniform[k];
    } else {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * static_cast<unsigned long long>(k + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        u = static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);     // 53 random bits -> [0, 1)
    }
    // lower_bound over [0, n_items - 1]: a cumulative sum in float64 may end a few ulps below 1.0, and a uniform above it
    // would otherwise yield n_items -- an id the reference answers with an IndexError at its gather and that would be an
    // out-of-range row here; the last item takes those draws
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cumsum[mid] < u) lo = mid + 1; else hi = mid;
    }
    out[k] = lo;
}

int skr_pop_sample(const double* d_cumsum, int n_items, const double* d_uniform, uint64_t seed, int64_t n, int32_t* d_out,
                   void* stream) {
    SKR_REQUIRE(d_cumsum && d_out && n_items >= 1 && n >= 0, "skr_pop_sample: bad argument");
    if (n == 0) return SKR_OK;
    hipLaunchKernelGGL(pop_sample_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, skr::as_stream(stream), d_cumsum,
                       n_items, d_uniform, static_cast<unsigned long long>(seed), n, d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_scatter_add_rows(const float* d_src, const int32_t* d_index, int n, int dim, const float* d_table, float reg,
                         float* d_g_table, uint8_t* d_touch, const float* d_touch_base, void* stream) {
    SKR_REQUIRE(d_src && d_index && d_g_table, "skr_scatter_add_rows: NULL argument");
    SKR_REQUIRE(reg == 0.0f || d_table, "skr_scatter_add_rows: reg needs the table");
    SKR_REQUIRE(n >= 0 && dim >= 1, "skr_scatter_add_rows: bad shape");
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
    if (n == 0) return SKR_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((n + G_T / 64 - 1) / (G_T / 64)), dim3(G_T), 0, skr::as_stream(stream),
                       d_src, d_index, n, dim, d_table, reg, d_g_table, d_touch, d_touch_base);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

}  // extern "C"
