// train.hip -- T rows: BPR lookup-and-score forward/backward, dense Adam, CSR propagation.
//
// Replaces the stock torch ops the reference issues per step (no native code there):
//   recommender/BPRMF.py:77-82,114-127   gathers, inner_product, bpr_loss.sum(), l2_loss, backward, Adam
//   recommender/LightGCN.py:89-100       torch.sparse.mm per layer + stack/mean
//   recommender/LayerGCN.py:207-220      sparse.mm + cosine_similarity re-weighting + layer sum
//   utils/torch.py:20-21,62-74           inner_product, bpr_loss, l2_loss
//
// Every kernel maps the d = 64 embedding row onto the 64 lanes of one wavefront: a row is one
// coalesced 256-byte access, dot products are wave reductions, gradient scatter-adds are one
// 256-byte global_atomic_add_f32 per row (the shape the memory-side atomic units like).
#include "skr_common.h"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int D = 64;

// ------------------------------------------------------------------------------------------------
// K1: fused BPR batch (forward + backward)
// ------------------------------------------------------------------------------------------------
constexpr int BPR_WAVES = 4;

// C: the row is 64 C floats wide (embedding widths beyond 64, zero-padded to a multiple of 64 by the caller); lane l owns
// floats l, l + 64, ...: every access is still a coalesced 256-byte one
template <int C>
__global__ __launch_bounds__(BPR_WAVES * 64) void bpr_step_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, const float* __restrict__ bias,
    const float* __restrict__ RP, const float* __restrict__ RQ, const int32_t* __restrict__ u_ids,
    const int32_t* __restrict__ i_ids, const int32_t* __restrict__ j_ids, int n, float loss_scale, float reg,
    float reg_scale, float* __restrict__ gP, float* __restrict__ gQ, float* __restrict__ gb, float* __restrict__ gRP,
    float* __restrict__ gRQ, float* __restrict__ loss, uint8_t* __restrict__ touch, const float* touch_base,
    int loss_slots, int shard_world, int shard_rank, float grad_scale) {
    __shared__ float s_loss[BPR_WAVES], s_l2[BPR_WAVES];
    // mark the 64-float gradient block that starts at `a` as touched (one lane per row is enough)
    // (a byte that is already non-zero -- 1, or the sticky 2 -- is left alone)
    auto mark = [&](const float* a) {
        uint8_t* t = &touch[(a - touch_base) >> 6];
        if (*t == 0) *t = 1;
    };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc_loss = 0.0f, acc_l2 = 0.0f;
    const float rs = reg * reg_scale;
    const bool same_tables = RP == P && RQ == Q && gRP == gP && gRQ == gQ;
    for (int b = blockIdx.x * BPR_WAVES + wv; b < n; b += gridDim.x * BPR_WAVES) {
        int64_t u = u_ids[b];
        const int64_t i = i_ids[b], j = j_ids[b];
        if (shard_world > 1) {      // a GLOBAL batch on a user-sharded rank: only the triples of the users this rank owns
            if (u % shard_world != shard_rank) continue;
            u /= shard_world;       // row of the local user table
        }
        constexpr int DW = D * C;
        float pu[C], qi[C], qj[C];
        float di = 0.0f, dj = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            pu[c] = P[u * DW + c * D + lane]; qi[c] = Q[i * DW + c * D + lane]; qj[c] = Q[j * DW + c * D + lane];
            di += pu[c] * qi[c];
            dj += pu[c] * qj[c];
        }
        // x_ui - x_uj; the two inner products are reduced separately like inner_product() does
        float xi = skr::wave_sum(di), xj = skr::wave_sum(dj);
        float bi = 0.0f, bj = 0.0f;
        if (bias) {
            bi = bias[i];
            bj = bias[j];
            xi += bi;
            xj += bj;
        }
        const float x = xi - xj;
        // -logsigmoid(x) = -(min(0,x) - log1p(exp(-|x|)))   (torch's log_sigmoid forward)
        const float z = expf(-fabsf(x));
        const float l = -(fminf(0.0f, x) - log1pf(z));
        // d/dx = -sigmoid(-x)
        const float sig_neg = (x >= 0.0f) ? z / (1.0f + z) : 1.0f / (1.0f + z);
        // grad_scale: an extra factor on the SCORE part of the gradient only (LightGCN: dL/dE-bar enters the backward
        // propagation as H = dL/dE-bar / (K + 1); scaling here saves a pass over the [N, 64] buffer)
        const float c = -sig_neg * loss_scale * grad_scale;
        float sq;
        if (same_tables) {
            // BPRMF: the regulariser rows ARE the score rows -- no second read, one atomic per row for both gradient parts
            float sl = 0.0f;
#pragma unroll
            for (int c_ = 0; c_ < C; ++c_) {
                sl += pu[c_] * pu[c_] + qi[c_] * qi[c_] + qj[c_] * qj[c_];
                atomicAdd(&gP[u * DW + c_ * D + lane], c * (qi[c_] - qj[c_]) + rs * pu[c_]);
                atomicAdd(&gQ[i * DW + c_ * D + lane], c * pu[c_] + rs * qi[c_]);
                atomicAdd(&gQ[j * DW + c_ * D + lane], -c * pu[c_] + rs * qj[c_]);
            }
            sq = skr::wave_sum(sl);
        } else {
            // score-part gradients
            float sl = 0.0f;
#pragma unroll
            for (int c_ = 0; c_ < C; ++c_) {
                atomicAdd(&gP[u * DW + c_ * D + lane], c * (qi[c_] - qj[c_]));
                atomicAdd(&gQ[i * DW + c_ * D + lane], c * pu[c_]);
                atomicAdd(&gQ[j * DW + c_ * D + lane], -c * pu[c_]);
                // regulariser rows (other tables than the score tables: LightGCN's ego embeddings)
                const float ru = RP[u * DW + c_ * D + lane], ri = RQ[i * DW + c_ * D + lane], rj = RQ[j * DW + c_ * D + lane];
                sl += ru * ru + ri * ri + rj * rj;
                if (rs != 0.0f) {
                    atomicAdd(&gRP[u * DW + c_ * D + lane], rs * ru);
                    atomicAdd(&gRQ[i * DW + c_ * D + lane], rs * ri);
                    atomicAdd(&gRQ[j * DW + c_ * D + lane], rs * rj);
                }
            }
            sq = skr::wave_sum(sl);
        }
        if (bias) {
            sq += bi * bi + bj * bj;
            if (lane == 0 && gb) {
                atomicAdd(&gb[i], c + rs * bi);
                atomicAdd(&gb[j], -c + rs * bj);
            }
        }
        if (touch && lane < C) {      // lane c marks the c-th 64-float block of each row
            mark(&gP[u * DW + lane * D]); mark(&gQ[i * DW + lane * D]); mark(&gQ[j * DW + lane * D]);
            if (rs != 0.0f) { mark(&gRP[u * DW + lane * D]); mark(&gRQ[i * DW + lane * D]); mark(&gRQ[j * DW + lane * D]); }
        }
        if (touch && lane == 0) {
            if (bias && gb) { mark(&gb[i]); mark(&gb[j]); }
        }
        acc_loss += l;
        acc_l2 += 0.5f * sq;
    }
    if (lane == 0) {
        s_loss[wv] = acc_loss;
        s_l2[wv] = acc_l2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b2 = 0.0f;
        for (int w = 0; w < BPR_WAVES; ++w) {
            a += s_loss[w];
            b2 += s_l2[w];
        }
        // same-address float atomics serialise (~20 ns each): 256 workgroups on ONE pair of words cost 4.5 us of a
        // 10.9 us launch (by ablation).  loss_slots > 1 spreads them over that many pairs; the caller adds the pairs up.
        const int sl = 2 * (static_cast<int>(blockIdx.x) % loss_slots);
        atomicAdd(&loss[sl], a * loss_scale);
        atomicAdd(&loss[sl + 1], b2);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: dense Adam (torch.optim.Adam single-tensor path), 16-byte vectors, grid-stride
// ------------------------------------------------------------------------------------------------
struct AdamArgs {
    float one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamArgs& a) {
    m = m + a.one_minus_b1 * (g - m);           // exp_avg.lerp_(grad, 1-beta1)
    v = v * a.b2 + (a.one_minus_b2 * g) * g;    // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + (a.neg_step_size * m) / denom;      // addcdiv_(exp_avg, denom, value=-step_size)
}

// the same update when sqrt(1 - beta2^t) is exactly 1.0f (beta2 = 0.999: from step ~16 600 on): x / 1.0f == x, so the
// correctly rounded division by the bias correction (a dozen instructions) is left out -- results are identical
__device__ __forceinline__ void adam_elem_unit_bc2(float& p, float g, float& m, float& v, const AdamArgs& a) {
    m = m + a.one_minus_b1 * (g - m);
    v = v * a.b2 + (a.one_minus_b2 * g) * g;
    const float denom = sqrtf(v) + a.eps;
    p = p + (a.neg_step_size * m) / denom;
}

template <bool TOUCH, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, AdamArgs a, int zero_grad,
                                                   uint8_t* __restrict__ touch) {
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [](const float4* q) -> float4 {
        if (NT) {
            float4 r;
            r.x = __builtin_nontemporal_load(&q->x); r.y = __builtin_nontemporal_load(&q->y);
            r.z = __builtin_nontemporal_load(&q->z); r.w = __builtin_nontemporal_load(&q->w);
            return r;
        }
        return *q;
    };
    auto stv = [](float4* q, const float4& r) {
        if (NT) {
            __builtin_nontemporal_store(r.x, &q->x); __builtin_nontemporal_store(r.y, &q->y);
            __builtin_nontemporal_store(r.z, &q->z); __builtin_nontemporal_store(r.w, &q->w);
        } else {
            *q = r;
        }
    };
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i0 < n4; i0 += stride * UNROLL) {
        float4 pp[UNROLL], mm[UNROLL], vv[UNROLL], gg[UNROLL];
        uint8_t flag[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {   // issue every load of this trip before the first use
            const int64_t i = i0 + k * stride;
            gg[k] = zero4;
            flag[k] = 0;
            if (i < n4) {
                pp[k] = ld(&p4[i]);
                mm[k] = ld(&m4[i]);
                vv[k] = ld(&v4[i]);
                // 16 consecutive lanes share one 64-float block and its byte; they all read it in this
                // instruction, before the lane with (i & 15) == 0 clears it further down
                flag[k] = TOUCH ? touch[i >> 4] : 2;
                if (flag[k]) gg[k] = g4[i];
            }
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const int64_t i = i0 + k * stride;
            if (i < n4) {
                adam_elem(pp[k].x, gg[k].x, mm[k].x, vv[k].x, a);
                adam_elem(pp[k].y, gg[k].y, mm[k].y, vv[k].y, a);
                adam_elem(pp[k].z, gg[k].z, mm[k].z, vv[k].z, a);
                adam_elem(pp[k].w, gg[k].w, mm[k].w, vv[k].w, a);
                stv(&p4[i], pp[k]);
                stv(&m4[i], mm[k]);
                stv(&v4[i], vv[k]);
                if (flag[k]) {
                    if (zero_grad) g4[i] = zero4;
                    if (TOUCH && flag[k] == 1 && (i & 15) == 0) touch[i >> 4] = 0;
                }
            }
        }
    }
    // tail (n not a multiple of 4): always read
    for (int64_t i = (n4 << 2) + blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_elem(pp, g[i], mm, vv, a);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (zero_grad) g[i] = 0.f;
        if (TOUCH && touch[i >> 6] == 1) touch[i >> 6] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K3: CSR SpMM, d = 64, nnz-balanced: one wavefront per CH consecutive non-zeros
// ------------------------------------------------------------------------------------------------
constexpr int SP_CH = 512;     // non-zeros per wavefront
constexpr int SP_WAVES = 4;

__device__ __forceinline__ int64_t row_of(const int64_t* __restrict__ rowptr, int n_rows, int64_t e) {
    int64_t lo = 0, hi = n_rows - 1;  // largest r with rowptr[r] <= e
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ bool row_is_split(int64_t rb, int64_t re) { return (rb / SP_CH) != ((re - 1) / SP_CH); }

// rows no chunk completes: empty rows get their final value here, split rows are zeroed for atomics
__global__ void spmm_prep_kernel(int n_rows, const int64_t* __restrict__ rowptr, const float* __restrict__ addend,
                                 float* __restrict__ Y, float* __restrict__ accum, float accum_scale, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int64_t rb = rowptr[r], re = rowptr[r + 1];
    if (re == rb) {
        const float y = addend ? addend[r * ld + lane] : 0.0f;
        Y[r * ld + lane] = y;
        if (accum) accum[r * ld + lane] += accum_scale * y;
    } else if (row_is_split(rb, re)) {
        Y[r * ld + lane] = 0.0f;
    }
}

__global__ __launch_bounds__(SP_WAVES * 64) void spmm_main_kernel(int n_rows, const int64_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col,
                                                                  const float* __restrict__ val,
                                                                  const float* __restrict__ X,
                                                                  const float* __restrict__ addend, float* __restrict__ Y,
                                                                  float* __restrict__ accum, float accum_scale,
                                                                  int64_t nnz, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t w = blockIdx.x * static_cast<int64_t>(SP_WAVES) + (threadIdx.x >> 6);
    const int64_t e0 = w * SP_CH;
    if (e0 >= nnz) return;
    const int64_t e1 = (e0 + SP_CH < nnz) ? e0 + SP_CH : nnz;
    int64_t r = row_of(rowptr, n_rows, e0);
    int64_t e = e0;
    while (e < e1) {
        while (rowptr[r + 1] <= e) ++r;  // skip rows that ended (empty rows included)
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        const int64_t se = re < e1 ? re : e1;
        float acc = 0.0f;
        for (int64_t c0 = e; c0 < se; c0 += 64) {
            const int m = static_cast<int>(se - c0 < 64 ? se - c0 : 64);
            int cl = 0;
            float vl = 0.0f;
            if (lane < m) {
                cl = col[c0 + lane];
                vl = val[c0 + lane];
            }
            int k = 0;
            for (; k + 4 <= m; k += 4) {  // four independent 256-byte row gathers in flight
                const int c_0 = __shfl(cl, k), c_1 = __shfl(cl, k + 1), c_2 = __shfl(cl, k + 2), c_3 = __shfl(cl, k + 3);
                const float x0 = X[static_cast<int64_t>(c_0) * ld + lane], x1 = X[static_cast<int64_t>(c_1) * ld + lane];
                const float x2 = X[static_cast<int64_t>(c_2) * ld + lane], x3 = X[static_cast<int64_t>(c_3) * ld + lane];
                acc = fmaf(__shfl(vl, k), x0, acc);
                acc = fmaf(__shfl(vl, k + 1), x1, acc);
                acc = fmaf(__shfl(vl, k + 2), x2, acc);
                acc = fmaf(__shfl(vl, k + 3), x3, acc);
            }
            for (; k < m; ++k) {
                const int c_0 = __shfl(cl, k);
                acc = fmaf(__shfl(vl, k), X[static_cast<int64_t>(c_0) * ld + lane], acc);
            }
        }
        if (!row_is_split(rb, re)) {  // this wave saw the whole row: finish it here
            float y = acc;
            if (addend) y += addend[r * ld + lane];
            Y[r * ld + lane] = y;
            if (accum) accum[r * ld + lane] += accum_scale * y;
        } else {
            atomicAdd(&Y[r * ld + lane], acc);
        }
        e = se;
    }
}

__global__ void spmm_fix_kernel(int n_rows, const int64_t* __restrict__ rowptr, const float* __restrict__ addend,
                                float* __restrict__ Y, float* __restrict__ accum, float accum_scale, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int64_t rb = rowptr[r], re = rowptr[r + 1];
    if (re > rb && row_is_split(rb, re)) {
        float y = Y[r * ld + lane];
        if (addend) {
            y += addend[r * ld + lane];
            Y[r * ld + lane] = y;
        }
        if (accum) accum[r * ld + lane] += accum_scale * y;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerGCN refinement (LayerGCN.py:214-216) forward / backward, one wavefront per row
// ------------------------------------------------------------------------------------------------
constexpr float COS_EPS = 1e-8f;  // F.cosine_similarity default

template <int C>
__global__ void refine_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ E, int64_t n_rows,
                                  float* __restrict__ Z, float* __restrict__ w_out, float* __restrict__ accum) {
    constexpr int DW = D * C;
    const int lane = threadIdx.x & 63;
    const int64_t r = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    float y[C], e[C];
    float sy = 0.0f, se = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        y[c] = Y[r * DW + c * D + lane]; e[c] = E[r * DW + c * D + lane];
        sy += y[c] * y[c];
        se += e[c] * e[c];
    }
    const float ny = fmaxf(sqrtf(skr::wave_sum(sy)), COS_EPS);
    const float ne = fmaxf(sqrtf(skr::wave_sum(se)), COS_EPS);
    float sw = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) sw += (y[c] / ny) * (e[c] / ne);
    const float w = skr::wave_sum(sw);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float z = w * y[c];
        Z[r * DW + c * D + lane] = z;
        if (accum) accum[r * DW + c * D + lane] += z;
    }
    if (lane == 0) w_out[r] = w;
}

template <int C>
__global__ void refine_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ E, const float* __restrict__ w_in,
                                  const float* __restrict__ dZ, int64_t n_rows, float* __restrict__ dY,
                                  float* __restrict__ dE, const uint8_t* __restrict__ row_mask, int zero_skipped) {
    constexpr int DW = D * C;
    const int lane = threadIdx.x & 63;
    const int64_t r = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    if (row_mask && !row_mask[r]) {            // dZ_r is zero: dY_r is zero and nothing is added to dE_r
        if (zero_skipped) {
#pragma unroll
            for (int c = 0; c < C; ++c) dY[r * DW + c * D + lane] = 0.0f;
        }
        return;
    }
    float y[C], e[C], dz[C];
    float sy = 0.0f, se = 0.0f, sd = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        y[c] = Y[r * DW + c * D + lane]; e[c] = E[r * DW + c * D + lane]; dz[c] = dZ[r * DW + c * D + lane];
        sy += y[c] * y[c];
        se += e[c] * e[c];
        sd += dz[c] * y[c];
    }
    const float w = w_in[r];
    const float nyr = sqrtf(skr::wave_sum(sy)), ner = sqrtf(skr::wave_sum(se));
    const float ny = fmaxf(nyr, COS_EPS), ne = fmaxf(ner, COS_EPS);
    const float dw = skr::wave_sum(sd);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float yh = y[c] / ny, eh = e[c] / ne;
        // d(yh)/dy = (I - yh yh^T)/ny when the norm is not clamped, I/eps when it is (clamp_min has zero slope)
        const float gy = (nyr > COS_EPS) ? (eh - w * yh) / ny : eh / ny;
        const float ge = (ner > COS_EPS) ? (yh - w * eh) / ne : yh / ne;
        dY[r * DW + c * D + lane] = w * dz[c] + dw * gy;
        dE[r * DW + c * D + lane] += dw * ge;
    }
}

// rows whose mask byte is set are zeroed (and the byte cleared): restores the "all zero" state of a buffer of which only a
// batch's rows were written, without a fill of the whole buffer
__global__ void clear_marked_rows_kernel(uint8_t* __restrict__ mask, int64_t n_rows, int clear_mask, float* __restrict__ table, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6) * 64;     // 64 rows per wavefront
    if (r0 >= n_rows) return;
    const bool set = r0 + lane < n_rows && mask[r0 + lane] != 0;
    unsigned long long b = __ballot(set);
    if (set && clear_mask) mask[r0 + lane] = 0;
    while (b) {
        const int j = __ffsll(static_cast<long long>(b)) - 1;
        b &= b - 1;
        for (int c = lane; c < dim; c += 64) table[(r0 + j) * dim + c] = 0.0f;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx, int64_t n,
                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (k >= n) return;
    out[k * D + lane] = table[static_cast<int64_t>(idx[k]) * D + lane];
}
// any row width (the GRU4RecPlus tables: 32 / 64 / 128 floats, biases: 1): one thread per element
__global__ void gather_rows_any_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx, int64_t n, int dim,
                                       float* __restrict__ out) {
    const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (e >= n * dim) return;
    const int64_t k = e / dim;
    out[e] = table[static_cast<int64_t>(idx[k]) * dim + (e - k * dim)];
}
__global__ void scatter_rows_any_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int dim,
                                        float* __restrict__ table) {
    const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (e >= n * dim) return;
    const int64_t k = e / dim;
    if (idx[k] < 0) return;
    table[static_cast<int64_t>(idx[k]) * dim + (e - k * dim)] = src[e];
}

// table[idx[k]] = src[k]; negative ids are skipped; duplicate ids must carry identical rows
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                                    float* __restrict__ table) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (k >= n || idx[k] < 0) return;
    table[static_cast<int64_t>(idx[k]) * D + lane] = src[k * D + lane];
}

// out[i] = ((in[0][i] + in[1][i]) + in[2][i]) + ...   -- the ranks' blocks added in rank order, the same bits on every rank
__global__ void sum_blocks_kernel(const float* __restrict__ in, int n_blocks, int64_t n, float* __restrict__ out) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
        float acc = in[i];
        for (int b = 1; b < n_blocks; ++b) acc += in[b * n + i];
        out[i] = acc;
    }
}

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
        y[i] = fmaf(a, x[i], y[i]);
}

__global__ void scale_copy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) y[i] = a * x[i];
}

__global__ void scale_kernel(float a, float* __restrict__ x, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) x[i] *= a;
}

inline unsigned rows_to_blocks(int64_t n_rows) { return static_cast<unsigned>((n_rows * 64 + 255) / 256); }

// ------------------------------------------------------------------------------------------------
// K2b: the same dense Adam, temporally blocked.  The reference's optimiser moves EVERY parameter at EVERY
// step, but a BPR step puts a non-zero gradient into at most 3*batch of the ~1.1 M rows, and the batches of an
// epoch are known in advance.  For a block of k consecutive steps the 64-float blocks of the flat buffer
// are split into HOT (touched by at least one of the k steps) and COLD.  A cold block sees k zero-gradient
// updates: they are applied in ONE pass (p, m, v read and written once instead of k times), each of the k
// updates evaluated exactly as adam_elem does with g = 0 and that step's bias corrections.  Hot blocks get
// the ordinary update at every step, through the id lists of the block (each block claimed once per step).
// Every parameter still receives every update, in the same arithmetic: results are bit-identical to calling
// skr_adam_step after every step (tests/test_gpu_train.py::test_blocked_adam_is_bit_identical).
// ------------------------------------------------------------------------------------------------
constexpr int AB_KMAX = 64;
struct AdamBlockArgs {
    float one_minus_b1, b2, one_minus_b2, eps;
    float neg_step_size[AB_KMAX], bc2_sqrt[AB_KMAX];
    float nss_bound[AB_KMAX];   // max |neg_step_size[s']| over s' >= s: the bound the at-rest test of a run starting at s needs
                                // (torch's -lr / bc1 only shrinks with the step: then this IS |neg_step_size[s]|; TF's
                                //  -lr * sqrt(bc2) / bc1 falls, then rises again towards lr)
    int k;
    // thresholds of the "parameter at rest" test of adam_cold_rows_kernel (0 switches the test off)
    float rest_eps;   // 2^-28 * eps
    float rest_b2k;   // a lower bound of beta2^k
    // ranges of the "ordinary magnitudes" test (fast_mlo = +inf switches it off)
    float fast_vlo, fast_mlo, fast_mhi;
    unsigned long long* stats;   // optional census (SKR_COLD_STATS=1): cold blocks at rest / ordinary / general
};

__global__ __launch_bounds__(256) void adam_mark_kernel(const int32_t* __restrict__ ids, int64_t n, int64_t offset,
                                                        int stride, int32_t* __restrict__ tag, int32_t value,
                                                        int32_t* __restrict__ claim, int32_t claim_value) {
    const int64_t g = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (g < n && ids[g] >= 0) {   // negative id: an empty slot of a de-duplicated list
        const int64_t blk = (offset + static_cast<int64_t>(ids[g]) * stride) >> 6;
        tag[blk] = value;
        if (claim) claim[blk] = claim_value;
    }
}

// cold pass: every float4 whose 64-float block is not tagged gets k zero-gradient updates
template <int UNROLL>
__global__ __launch_bounds__(256) void adam_cold_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        int64_t n, AdamBlockArgs a, const int32_t* __restrict__ tag,
                                                        int32_t hot_value) {
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    auto ld = [](const float4* q) -> float4 {
        float4 r;
        r.x = __builtin_nontemporal_load(&q->x); r.y = __builtin_nontemporal_load(&q->y);
        r.z = __builtin_nontemporal_load(&q->z); r.w = __builtin_nontemporal_load(&q->w);
        return r;
    };
    auto stv = [](float4* q, const float4& r) {
        __builtin_nontemporal_store(r.x, &q->x); __builtin_nontemporal_store(r.y, &q->y);
        __builtin_nontemporal_store(r.z, &q->z); __builtin_nontemporal_store(r.w, &q->w);
    };
    auto steps = [&](float& pp, float& mm, float& vv) {
        for (int s = 0; s < a.k; ++s) {
            AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
            adam_elem(pp, 0.0f, mm, vv, one);
        }
    };
    // the same k updates for the four lanes of a float4, step-major: the four independent chains of one step sit
    // next to each other, which lets the compiler pair them into packed fp32 instructions
    auto steps4 = [&](float4& pp, float4& mm, float4& vv) {
        for (int s = 0; s < a.k; ++s) {
            AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
            if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000) {   // scalar branch, not a select
                adam_elem_unit_bc2(pp.x, 0.0f, mm.x, vv.x, one);
                adam_elem_unit_bc2(pp.y, 0.0f, mm.y, vv.y, one);
                adam_elem_unit_bc2(pp.z, 0.0f, mm.z, vv.z, one);
                adam_elem_unit_bc2(pp.w, 0.0f, mm.w, vv.w, one);
            } else {
                adam_elem(pp.x, 0.0f, mm.x, vv.x, one);
                adam_elem(pp.y, 0.0f, mm.y, vv.y, one);
                adam_elem(pp.z, 0.0f, mm.z, vv.z, one);
                adam_elem(pp.w, 0.0f, mm.w, vv.w, one);
            }
        }
    };
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i0 < n4; i0 += stride * UNROLL) {
        float4 pp[UNROLL], mm[UNROLL], vv[UNROLL];
        bool cold[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + u * stride;
            cold[u] = i < n4 && tag[i >> 4] != hot_value;
            if (cold[u]) {
                pp[u] = ld(&p4[i]);
                mm[u] = ld(&m4[i]);
                vv[u] = ld(&v4[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (cold[u]) {
                const int64_t i = i0 + u * stride;
                steps4(pp[u], mm[u], vv[u]);
                stv(&p4[i], pp[u]);
                stv(&m4[i], mm[u]);
                stv(&v4[i], vv[u]);
            }
        }
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
        if (tag[i >> 6] != hot_value) steps(p[i], m[i], v[i]);
}

// Square root and division for ORDINARY MAGNITUDES, bit-identical to sqrtf(x) and n / d as compiled under
// -fhip-fp32-correctly-rounded-divide-sqrt but cheaper:
//   div_ordinary   the compiler's own expansion (v_rcp_f32, one Newton step, two quotient corrections, final fma)
//                  minus v_div_scale_f32 and v_div_fixup_f32, which are the identity (VCC = 0) / a pass-through when
//                  d is normal with |d| < 2^126, |n| >= 2^-103 and -125 <= exponent(n) - exponent(d) < 96
//                  (CDNA3/4 ISA, V_DIV_SCALE_F32 / V_DIV_FIXUP_F32): same instructions on the same values;
//   sqrt_ordinary  v_rsq_f32 and one fused correction s + (x - s*s) * r/2 instead of v_sqrt_f32 and two residual tests:
//                  a different route to the correctly rounded root, so it is PROVEN BY ENUMERATION -- the self-test runs
//                  it against sqrtf on every float of [2^-96, FLT_MAX] (the range it is used on is [2^-90, 2^20]).
// skr_selftest_cold_math does that enumeration and tries the division on 2^32 hashed operand pairs of its range; as
// a control it also counts how often the raw v_sqrt_f32 differs from sqrtf (it must: that is why a correction exists).
__device__ __forceinline__ float sqrt_ordinary(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r, h = 0.5f * r;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}

__device__ __forceinline__ float div_ordinary(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    float t = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(t, r, q);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one zero-gradient update of two rows of ordinary magnitudes (v > 0, so v*b2 + (c2*0)*0 == v*b2)
template <bool UNIT_BC2>
__device__ __forceinline__ void adam_pair_ordinary(f32x2& p, f32x2& m, f32x2& v, const AdamArgs& a) {
    m = m + a.one_minus_b1 * (0.0f - m);
    v = v * a.b2;
    f32x2 sq;
    sq.x = sqrt_ordinary(v.x);
    sq.y = sqrt_ordinary(v.y);
    if (!UNIT_BC2) {
        sq.x = div_ordinary(sq.x, a.bc2_sqrt);
        sq.y = div_ordinary(sq.y, a.bc2_sqrt);
    }
    const f32x2 d = sq + a.eps, n = a.neg_step_size * m;
    f32x2 q;
    q.x = div_ordinary(n.x, d.x);
    q.y = div_ordinary(n.y, d.y);
    p = p + q;
}

template <bool UNIT_BC2>
__device__ __forceinline__ void adam_one_ordinary(float& p, float& m, float& v, const AdamArgs& a) {
    m = m + a.one_minus_b1 * (0.0f - m);
    v = v * a.b2;
    float sq = sqrt_ordinary(v);
    if (!UNIT_BC2) sq = div_ordinary(sq, a.bc2_sqrt);
    p = p + div_ordinary(a.neg_step_size * m, sq + a.eps);
}

__global__ __launch_bounds__(256) void selftest_cold_math_kernel(uint32_t lo, uint32_t hi, uint64_t n_pairs,
                                                                 unsigned long long* __restrict__ bad) {
    const uint64_t tid = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t nth = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long bs = 0, bd = 0, b1 = 0, b2 = 0;
    for (uint64_t b = lo + tid; b <= hi; b += nth) {
        const float x = __uint_as_float(static_cast<uint32_t>(b));
        const uint32_t want = __float_as_uint(sqrtf(x));
        bs += __float_as_uint(sqrt_ordinary(x)) != want;
        b1 += __float_as_uint(__builtin_amdgcn_sqrtf(x)) != want;   // control
        b2 += 1;
    }
    for (uint64_t i = tid; i < n_pairs; i += nth) {
        uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull;   // splitmix64
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
        // d: exponent in [-48, 21], n: exponent in [-100, 40], random mantissas and signs
        const uint32_t hd = static_cast<uint32_t>(h), hn = static_cast<uint32_t>(h >> 32);
        const uint32_t ed = 127 - 48 + (hd >> 23) % 70, en = 127 - 100 + ((hn >> 23) & 0xff) % 141;
        const float d = __uint_as_float((hd & 0x807fffffu) | (ed << 23)), n = __uint_as_float((hn & 0x807fffffu) | (en << 23));
        bd += __float_as_uint(div_ordinary(n, d)) != __float_as_uint(n / d);
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (bd) atomicAdd(&bad[1], bd);
    if (b1) atomicAdd(&bad[2], b1);
    if (b2) atomicAdd(&bad[3], b2);
}

// the two per-lane tests of the cold pass (and of the hot step's catch-up): see the comment below.  nss0 = the largest
// |neg_step_size| among the zero-gradient updates in question (AdamBlockArgs::nss_bound of the first of them)
__device__ __forceinline__ bool lane_at_rest(float pp, float mm, float vv, float nss0, const AdamBlockArgs& a) {
    const float ap = fabsf(pp), n0 = nss0 * fabsf(mm);
    const float r = ap * 0x1p-29f;
    const float bound = (r * r) * (vv * a.rest_b2k);
    const bool small = n0 < ap * a.rest_eps || (n0 * n0 < bound && bound >= 0x1p-120f);
    return __float_as_uint(vv) <= 0x7f800000u && ap >= 0x1p-60f && small;
}

__device__ __forceinline__ bool lane_ordinary(float mm, float vv, const AdamBlockArgs& a) {
    const float am = fabsf(mm);
    return vv >= a.fast_vlo && vv <= 0x1p20f && am >= a.fast_mlo && am <= a.fast_mhi;
}

// cold pass, one wavefront per 64-float block (= one embedding row), with a cheap exact path for rows AT REST.
//
// A zero-gradient update is p += (nss*m') / (sqrt(v')/bc2 + eps) with m' = m + c1*(0 - m), v' = v*b2.  A row that
// no batch has touched for a few hundred steps has |m| decayed so far that the quotient q is below a quarter of
// the spacing of the floats around p: then fl(p + q) == p and the correctly rounded sqrt and divisions (about 36 of
// the ~41 issue slots of an update) decide nothing.  A block is AT REST for all k updates of the pass when every lane
// passes, on the values the pass starts from,
//     sign(v) = +, v not NaN;  |p| >= 2^-60;
//     |nss[0]*m| < 2^-28 * |p| * eps            or    |nss[0]*m|^2 < 2^-58 * p^2 * v * lb(b2^k)   (and that bound is normal)
// Proof sketch: |m| and |nss[s]| never grow over the pass and v never drops below v*b2^k, so for every
// update |n| = |fl(nss[s]*m')| <= |fl(nss[0]*m)| and d = fl(fl(sqrt(v')/bc2) + eps) >= max(eps, sqrt(v*b2^k))*(1 - 2^-22);
// hence |fl(n/d)| < 2^-27 |p| < spacing(p)/4 and p is unchanged, bit for bit, by each of the k updates.  m and v still get
// their k decays in the arithmetic of adam_elem (v*b2 + (c2*0)*0 == v*b2 because v*b2 carries a + sign).  The thresholds
// are zero (tests off) unless 0 < beta1, beta2 < 1, lr > 0, eps >= 0.  Blocks not at rest take adam_elem as before.
template <int U>
__global__ __launch_bounds__(256) void adam_cold_rows_kernel(float* __restrict__ p, float* __restrict__ m,
                                                             float* __restrict__ v, int64_t n, AdamBlockArgs a,
                                                             const int32_t* __restrict__ tag, int32_t hot_value) {
    const int lane = threadIdx.x & 63;
    const int64_t nb = n >> 6;
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * 4;
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto general = [&](float& pp, float& mm, float& vv) {
        for (int s = 0; s < a.k; ++s) {
            AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
            if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)   // scalar branch, not a select
                adam_elem_unit_bc2(pp, 0.0f, mm, vv, one);
            else
                adam_elem(pp, 0.0f, mm, vv, one);
        }
    };
    // two rows of ordinary magnitudes advance together (packed fp32 for the element-wise parts, the two square
    // root / division chains interleaved): the first waits in `held` until the wavefront meets the second
    auto ordinary2 = [&](f32x2& p2, f32x2& m2, f32x2& v2) {
        for (int s = 0; s < a.k; ++s) {
            AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
            if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
                adam_pair_ordinary<true>(p2, m2, v2, one);
            else
                adam_pair_ordinary<false>(p2, m2, v2, one);
        }
    };
    bool have = false;
    float hp = 0.0f, hm = 0.0f, hv = 0.0f;
    int64_t hi = 0;
    for (int64_t b0 = wave0; b0 < nb; b0 += n_waves * U) {
        float pp[U], mm[U], vv[U];
        bool cold[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t b = b0 + u * n_waves;
            cold[u] = b < nb && tag[b] != hot_value;
            if (cold[u]) {
                const int64_t i = (b << 6) + lane;
                pp[u] = __builtin_nontemporal_load(&p[i]);
                mm[u] = __builtin_nontemporal_load(&m[i]);
                vv[u] = __builtin_nontemporal_load(&v[i]);
            }
        }
        // rows at rest (4 of 5 cold rows): only the moments decay, 3 vector instructions per element and step -- as much
        // vector-ALU time over the pass as the rows of ordinary magnitudes.  Neighbours (u, u + 1) that are both at rest
        // decay together in packed fp32 (the same multiply and add on each half); a row whose moments are all zero
        // (never touched) has nothing to decay.
        bool rest[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            rest[u] = cold[u] && __builtin_amdgcn_ballot_w64(!lane_at_rest(pp[u], mm[u], vv[u], a.nss_bound[0], a)) == 0;
        auto store_rest = [&](int64_t i, float m0, float v0, float m1, float v1) {
            if (__builtin_amdgcn_ballot_w64(__float_as_uint(m1) != __float_as_uint(m0)) != 0) __builtin_nontemporal_store(m1, &m[i]);
            if (__builtin_amdgcn_ballot_w64(__float_as_uint(v1) != __float_as_uint(v0)) != 0) __builtin_nontemporal_store(v1, &v[i]);
        };
#pragma unroll
        for (int u = 0; u + 1 < U; u += 2) {
            if (!(rest[u] && rest[u + 1])) continue;
            if (a.stats && lane == 0) atomicAdd(&a.stats[0], 2ull);
            f32x2 m2{mm[u], mm[u + 1]}, v2{vv[u], vv[u + 1]};
            if (__builtin_amdgcn_ballot_w64((__float_as_uint(m2.x) | __float_as_uint(m2.y) | __float_as_uint(v2.x) |
                                             __float_as_uint(v2.y)) != 0) != 0) {
                for (int s = 0; s < a.k; ++s) {
                    // -m for (0 - m): a sign modifier on the multiply instead of an instruction.  They differ for m = +-0
                    // only (+0 vs -0 into the product), and m + (+-0) is m, resp. +0 for m = +-0, either way
                    m2 = m2 + a.one_minus_b1 * (-m2);
                    v2 = v2 * a.b2;
                }
                store_rest(((b0 + u * n_waves) << 6) + lane, mm[u], vv[u], m2.x, v2.x);
                store_rest(((b0 + (u + 1) * n_waves) << 6) + lane, mm[u + 1], vv[u + 1], m2.y, v2.y);
            }
            cold[u] = cold[u + 1] = false;      // done
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!cold[u]) continue;
            const int64_t i = ((b0 + u * n_waves) << 6) + lane;
            if (rest[u]) {
                if (a.stats && lane == 0) atomicAdd(&a.stats[0], 1ull);
                float m1 = mm[u], v1 = vv[u];
                for (int s = 0; s < a.k; ++s) {
                    m1 = m1 + a.one_minus_b1 * (0.0f - m1);
                    v1 = v1 * a.b2;
                }
                store_rest(i, mm[u], vv[u], m1, v1);
                continue;
            }
            const bool lane_ord = lane_ordinary(mm[u], vv[u], a);
            if (__builtin_amdgcn_ballot_w64(!lane_ord) == 0) {
                if (a.stats && lane == 0) atomicAdd(&a.stats[1], 1ull);
                if (!have) {
                    hp = pp[u], hm = mm[u], hv = vv[u], hi = i;
                    have = true;
                    continue;
                }
                f32x2 p2{hp, pp[u]}, m2{hm, mm[u]}, v2{hv, vv[u]};
                ordinary2(p2, m2, v2);
                __builtin_nontemporal_store(p2.x, &p[hi]);
                __builtin_nontemporal_store(m2.x, &m[hi]);
                __builtin_nontemporal_store(v2.x, &v[hi]);
                __builtin_nontemporal_store(p2.y, &p[i]);
                __builtin_nontemporal_store(m2.y, &m[i]);
                __builtin_nontemporal_store(v2.y, &v[i]);
                have = false;
                continue;
            }
            if (a.stats && lane == 0) atomicAdd(&a.stats[2], 1ull);
            general(pp[u], mm[u], vv[u]);
            __builtin_nontemporal_store(pp[u], &p[i]);
            __builtin_nontemporal_store(mm[u], &m[i]);
            __builtin_nontemporal_store(vv[u], &v[i]);
        }
    }
    if (have) {   // an odd one out
        general(hp, hm, hv);
        __builtin_nontemporal_store(hp, &p[hi]);
        __builtin_nontemporal_store(hm, &m[hi]);
        __builtin_nontemporal_store(hv, &v[hi]);
    }
    // tail (n not a multiple of 64)
    const int64_t i = (nb << 6) + blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i < n && tag[nb] != hot_value) general(p[i], m[i], v[i]);
}

// hot step: one wavefront per id.  claim[block] holds the optimiser step the block has been advanced to (the mark
// kernel sets it to the step count the k-step block starts from).  The wavefront that raises it to step_t owns the
// block for this launch and ADVANCES it: zero-gradient updates for the steps it has not seen yet, then step_t's update
// with the accumulated gradient, which is consumed.  A caller that names every hot block at every step gets one
// update per launch; a caller that names only the rows of batch t and of batch t+1 (the next batch must READ current
// rows) visits a row when it matters and catches up there -- the same updates in the same order, fewer passes over
// HBM.  The last step of a k-step block must name every hot block, so that all of them end at the same step.
__global__ __launch_bounds__(256) void adam_hot_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, AdamBlockArgs a, int32_t t0,
                                                       int32_t t, const int32_t* __restrict__ ids, int64_t n_ids,
                                                       int64_t offset, int stride, int32_t* __restrict__ claim) {
    const int lane = threadIdx.x & 63;
    const int64_t e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= n_ids) return;
    const int32_t id = ids[e];
    if (id < 0) return;                            // an empty slot of a de-duplicated list
    const int64_t blk = (offset + static_cast<int64_t>(id) * stride) >> 6;
    // the row is loaded while the claim is in flight (one memory round trip less on a latency-bound kernel); a
    // wavefront that loses the claim drops what it loaded.  Nobody writes the row during this launch but its owner.
    const int64_t i = blk * 64 + lane;
    float pp = 1.0f, mm = 0.0f, vv = 0.0f, gg = 0.0f;   // lanes beyond n: values that pass every wavefront-wide test
    if (i < n) {
        pp = p[i];
        mm = m[i];
        vv = v[i];
        gg = g[i];
    }
    int old = 0;
    if (lane == 0) old = atomicExch(&claim[blk], t);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old >= t) return;
    if (old < t0) old = t0;
    // The steps the block is behind are zero-gradient updates, and they sit on the critical path of the training step
    // (the slowest wavefront of this launch is one with a row that is 20 steps behind).  The same three exact evaluations
    // as in the cold pass: AT REST (a row no batch has touched for long -- most user rows when their turn comes: only the
    // moments decay), ORDINARY MAGNITUDES (scaling-free square root / division: a dependent chain 2.5x shorter), general.
    const int s_grad = t - t0 - 1;                 // the step that takes the gradient
    int s = old - t0;
    if (s < s_grad) {
        if (__builtin_amdgcn_ballot_w64(!lane_at_rest(pp, mm, vv, a.nss_bound[s], a)) == 0) {
            for (; s < s_grad; ++s) {
                mm = mm + a.one_minus_b1 * (0.0f - mm);
                vv = vv * a.b2;
            }
        } else if (__builtin_amdgcn_ballot_w64(!lane_ordinary(mm, vv, a)) == 0) {
            for (; s < s_grad; ++s) {
                AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
                if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
                    adam_one_ordinary<true>(pp, mm, vv, one);
                else
                    adam_one_ordinary<false>(pp, mm, vv, one);
            }
        }
    }
    if (i < n) {
        for (; s < t - t0; ++s) {   // step t0 + s + 1: what is left of the zero-gradient steps, then the gradient step
            AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
            const float gs = (s == s_grad) ? gg : 0.0f;
            if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
                adam_elem_unit_bc2(pp, gs, mm, vv, one);
            else
                adam_elem(pp, gs, mm, vv, one);
        }
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
    }
    // a row named only because the NEXT batch reads it has no gradient yet: nothing to clear
    if (__builtin_amdgcn_ballot_w64(gg != 0.0f) != 0 && i < n) g[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// K2c: the BPR batch and the hot rows' Adam in ONE launch per step (single GPU).
//
// With bpr_step_kernel + adam_hot_kernel a training step is a chain of two dependent launches (6-8 us + 13 us): the
// gradient of batch t must be complete before any row moves, and batch t+1 reads rows that step t moved.  The chain is
// cut to one launch by evaluating the hot rows LAZILY: a row is advanced when a batch reads it, by the wavefronts that
// read it, and a gradient is applied at the row's NEXT naming (or by the block's end launch).  The batches of a k-step
// block are known in advance, so for every reference (step s, row r) the host precomputes (skrec/recommender/fused.py)
//     slot   the row's index in the block's compact workspace
//     n0     how many earlier steps of the block named the row (0-based naming index), modulo 6
//     prev   the step of the previous naming (or none)
//     owner  exactly one reference per (step, row) pair
// The row's state before step s:  n0 == 0: the dense tables (no step of the block has touched it);  n0 > 0: workspace copy
// n0 & 1, valid through optimiser index prev - 1, plus the gradient of step `prev` waiting in gradient buffer (n0 - 1) % 3.
// Every wavefront that reads the row applies, in registers, index `prev` with that gradient and the zero-gradient indices
// prev + 1 .. s - 1 -- the updates the dense optimiser makes, in its arithmetic (adam_elem and the at-rest / ordinary
// evaluations of the cold pass, which give the same bits) -- and uses the result for its scores.  The pair's OWNER also
// writes it to copy (n0 + 1) & 1 and clears gradient buffer (n0 + 1) % 3; all of them add this step's gradient into buffer
// n0 % 3.  Within one launch nobody writes what another wavefront reads: two state copies and three gradient buffers keep
// readers, the writer and the accumulators apart, so no wavefront waits for another and no hand-off crosses the L2s.
// bpr_fused_end_kernel brings every slot to the block's last index and writes it back to the dense tables.
// Same updates of every parameter in the same order and arithmetic as one dense Adam launch per step
// (tests/test_gpu_train.py::test_fused_step_is_bit_identical).
// ------------------------------------------------------------------------------------------------
constexpr int FUSED_SLOT_BITS = 20;
constexpr int FUSED_PRE = 7;     // value of a word's n0 field (n0 mod 6 otherwise): first naming, state waiting in the pre buffer

struct FusedRow {
    float p, m, v, g;
};

// A run of zero-gradient updates of one row of ordinary magnitudes, indices [s, s_to).  The quotient of update s depends on
// m_s and v_s only -- not on p -- so the square-root / division chains of consecutive updates are independent of each other:
// four of them are laid side by side (one wavefront alone on its SIMD otherwise waits out the latency of every one of the
// ~25 dependent instructions of a chain: ~200 cycles per update instead of ~70), and p takes the quotients in order -- the
// same operations on the same values as update after update.
template <bool UNIT_BC2>
__device__ __forceinline__ void ordinary_run(float& p, float& m, float& v, const AdamBlockArgs& a, int& s, int s_to) {
    for (; s + 4 <= s_to; s += 4) {
        float ms[4], vs[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m = m + a.one_minus_b1 * (0.0f - m);
            v = v * a.b2;
            ms[u] = m;
            vs[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float sq = sqrt_ordinary(vs[u]);
            if (!UNIT_BC2) sq = div_ordinary(sq, a.bc2_sqrt[s + u]);
            q[u] = div_ordinary(a.neg_step_size[s + u] * ms[u], sq + a.eps);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) p = p + q[u];
    }
    for (; s < s_to; ++s) {
        AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
        adam_one_ordinary<UNIT_BC2>(p, m, v, one);
    }
}

// zero-gradient indices [s_from, s_to) and, before them, index g_idx with gradient r.g (g_idx < 0: none)
__device__ __forceinline__ void fused_advance(FusedRow& r, int g_idx, int s_from, int s_to, const AdamBlockArgs& a) {
    if (g_idx >= 0) {
        AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[g_idx], a.bc2_sqrt[g_idx], a.eps};
        if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
            adam_elem_unit_bc2(r.p, r.g, r.m, r.v, one);
        else
            adam_elem(r.p, r.g, r.m, r.v, one);
    }
    int s = s_from;
    if (s < s_to) {
        if (__builtin_amdgcn_ballot_w64(!lane_at_rest(r.p, r.m, r.v, a.nss_bound[s], a)) == 0) {
            for (; s < s_to; ++s) {
                r.m = r.m + a.one_minus_b1 * (0.0f - r.m);
                r.v = r.v * a.b2;
            }
        } else if (__builtin_amdgcn_ballot_w64(!lane_ordinary(r.m, r.v, a)) == 0) {
            // sqrt(1 - beta2^t) rises with t and stays at 1.0f once it gets there: the run is all-unit, all-non-unit, or
            // (around step 16 600, once) mixed -- then update by update
            if (__builtin_amdgcn_readfirstlane(__float_as_int(a.bc2_sqrt[s])) == 0x3f800000)
                ordinary_run<true>(r.p, r.m, r.v, a, s, s_to);
            else if (__builtin_amdgcn_readfirstlane(__float_as_int(a.bc2_sqrt[s_to - 1])) != 0x3f800000)
                ordinary_run<false>(r.p, r.m, r.v, a, s, s_to);
            else
                for (; s < s_to; ++s) {
                    AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
                    if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
                        adam_one_ordinary<true>(r.p, r.m, r.v, one);
                    else
                        adam_one_ordinary<false>(r.p, r.m, r.v, one);
                }
        }
    }
    for (; s < s_to; ++s) {
        AdamArgs one{a.one_minus_b1, a.b2, a.one_minus_b2, a.neg_step_size[s], a.bc2_sqrt[s], a.eps};
        if (__builtin_amdgcn_readfirstlane(__float_as_int(one.bc2_sqrt)) == 0x3f800000)
            adam_elem_unit_bc2(r.p, 0.0f, r.m, r.v, one);
        else
            adam_elem(r.p, 0.0f, r.m, r.v, one);
    }
}

struct FusedWork {
    float *wp, *wm, *wv, *g;   // wp / wm / wv: [2][cap][64];  g: [3][cap][64]
    int64_t cap;
};

// One wavefront per interaction, its five rows one after the other (dbg: timing switches of tools/fused_lab.py).  A
// workgroup of five wavefronts per interaction (one per row, the rows meeting in LDS) was tried and is slower (18.4 vs 12.7 us
// per launch alone on the chip): a CU holds four interactions either way, so the catch-up arithmetic per SIMD is the same,
// and the barrier and the second id load come on top.
__global__ __launch_bounds__(BPR_WAVES * 64) void bpr_fused_step_kernel(
    const float* __restrict__ P, const float* __restrict__ M, const float* __restrict__ V, int64_t n_par, FusedWork w,
    const int32_t* __restrict__ u_ids, const int32_t* __restrict__ i_ids, const int32_t* __restrict__ j_ids,
    const int32_t* __restrict__ meta, int n, int64_t ublk0, int64_t iblk0, int64_t bblk0, int s_now, AdamBlockArgs a,
    float reg, float* __restrict__ loss, int loss_slots, int dbg, const float* __restrict__ pre) {
    __shared__ float s_loss[BPR_WAVES], s_l2[BPR_WAVES];
    // Issue priority over the cold pass that shares the SIMDs (side stream); dbg & 16 switches it off.  Round 2 measured it
    // alone (the step +6 %, but the cold pass 0.47 -> 0.57 ms, which then bounded the block) and left it off; round 3 pairs it
    // with one more cold-pass workgroup per CU (SKR_COLD_BPC 4 -> 5), which gives the pass back what the priority takes:
    // step launch 18.7 -> 15.3 us, cold pass 0.56 -> 0.54 ms, epoch 1.083 -> 1.010 s, the 20-step slice 36.2 -> 40.2 M
    // interactions/s on the same box (tools/cold_bpc_sweep.sh; 6 per CU: epoch 0.965 s but the short slice scatters 32-40 M,
    // 7 and more: the step kernel finds no room, 21.5 us)
    if (!(dbg & 16)) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc_loss = 0.0f, acc_l2 = 0.0f;
    for (int b = blockIdx.x * BPR_WAVES + wv; b < n; b += gridDim.x * BPR_WAVES) {
        const int64_t u = u_ids[b], i = i_ids[b], j = j_ids[b];
        int32_t mt[5];
        int64_t blk[5] = {ublk0 + u, iblk0 + i, iblk0 + j, bblk0 + (i >> 6), bblk0 + (j >> 6)};
#pragma unroll
        for (int r = 0; r < 5; ++r) mt[r] = __builtin_amdgcn_readfirstlane(meta[static_cast<int64_t>(r) * n + b]);
        FusedRow row[5];
        int from[5];      // first zero-gradient index still to apply
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int64_t slot = mt[r] & ((1 << FUSED_SLOT_BITS) - 1);
            int n0 = (mt[r] >> FUSED_SLOT_BITS) & 7;
            row[r] = FusedRow{1.0f, 0.0f, 0.0f, 0.0f};
            from[r] = (mt[r] >> 24) & 0x7f;
            if (n0 == FUSED_PRE) {
                // the row's first naming in the block, and the row was cold in the block before: its zero-gradient updates up
                // to this step were applied ahead of time, beside the previous block (bpr_fused_pre_kernel) -- same bits
                const int64_t e = slot * 64 + lane;
                row[r].p = pre[e];
                row[r].m = pre[w.cap * 64 + e];
                row[r].v = pre[2 * w.cap * 64 + e];
                from[r] = s_now;
                mt[r] &= ~(7 << FUSED_SLOT_BITS);        // n0 = 0 from here on
                n0 = 0;
            } else if (((mt[r] >> 24) & 0x7f) == 0) {
                const int64_t e = blk[r] * 64 + lane;
                if (e < n_par) {
                    row[r].p = P[e];
                    row[r].m = M[e];
                    row[r].v = V[e];
                }
            } else {
                const int64_t e = ((n0 & 1) * w.cap + slot) * 64 + lane;
                row[r].p = w.wp[e];
                row[r].m = w.wm[e];
                row[r].v = w.wv[e];
                row[r].g = w.g[(((n0 + 2) % 3) * w.cap + slot) * 64 + lane];
            }
        }
        if (!(dbg & 1)) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int prev1 = (mt[r] >> 24) & 0x7f;
                fused_advance(row[r], prev1 - 1, from[r], s_now, a);
            }
        }
        if (!(dbg & 4)) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if ((mt[r] >> 23) & 1) {
                    const int64_t slot = mt[r] & ((1 << FUSED_SLOT_BITS) - 1);
                    const int n0 = (mt[r] >> FUSED_SLOT_BITS) & 7;
                    const int64_t e = (((n0 + 1) & 1) * w.cap + slot) * 64 + lane;
                    w.wp[e] = row[r].p;
                    w.wm[e] = row[r].m;
                    w.wv[e] = row[r].v;
                    w.g[(((n0 + 1) % 3) * w.cap + slot) * 64 + lane] = 0.0f;
                }
            }
        }
        const float pu = row[0].p, qi = row[1].p, qj = row[2].p;
        const float bi = __shfl(row[3].p, static_cast<int>(i & 63)), bj = __shfl(row[4].p, static_cast<int>(j & 63));
        const float xi = skr::wave_sum(pu * qi) + bi, xj = skr::wave_sum(pu * qj) + bj;
        const float x = xi - xj;
        const float z = expf(-fabsf(x));
        const float l = -(fminf(0.0f, x) - log1pf(z));
        const float sig_neg = (x >= 0.0f) ? z / (1.0f + z) : 1.0f / (1.0f + z);
        const float c = -sig_neg;
        float sq = skr::wave_sum(pu * pu + qi * qi + qj * qj);
        float* gcur[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int64_t slot = mt[r] & ((1 << FUSED_SLOT_BITS) - 1);
            const int n0 = (mt[r] >> FUSED_SLOT_BITS) & 7;
            gcur[r] = w.g + ((n0 % 3) * w.cap + slot) * 64;
        }
        if (!(dbg & 2)) {
            // a row only this interaction names at this step takes a plain store (its buffer holds nothing that counts: it was
            // cleared two namings ago, or never written); shared rows are summed by the memory-side atomic units
            const float gu = c * (qi - qj) + reg * pu, gi = c * pu + reg * qi, gj = -c * pu + reg * qj;
            if (mt[0] < 0) gcur[0][lane] = gu; else atomicAdd(&gcur[0][lane], gu);
            if (mt[1] < 0) gcur[1][lane] = gi; else atomicAdd(&gcur[1][lane], gi);
            if (mt[2] < 0) gcur[2][lane] = gj; else atomicAdd(&gcur[2][lane], gj);
        }
        sq += bi * bi + bj * bj;
        if (lane == 0 && !(dbg & 2)) {
            atomicAdd(&gcur[3][i & 63], c + reg * bi);
            atomicAdd(&gcur[4][j & 63], -c + reg * bj);
        }
        acc_loss += l;
        acc_l2 += 0.5f * sq;
    }
    if (lane == 0) {
        s_loss[wv] = acc_loss;
        s_l2[wv] = acc_l2;
    }
    __syncthreads();
    if (threadIdx.x == 0 && !(dbg & 8)) {
        float x = 0.0f, y = 0.0f;
        for (int q = 0; q < BPR_WAVES; ++q) {
            x += s_loss[q];
            y += s_l2[q];
        }
        const int sl = 2 * (static_cast<int>(blockIdx.x) % loss_slots);
        atomicAdd(&loss[sl], x);
        atomicAdd(&loss[sl + 1], y);
    }
}

// Ahead of a block, beside the block before it (side stream, behind that block's cold pass): the rows the block names that
// were COLD in the previous block -- nearly every user row -- are advanced from the block's first index to the step of their
// first naming, into the pre buffer ([3][cap][64]: p, m, v).  These zero-gradient updates depend on nothing the block
// itself does; the step launch that first names such a row then finds it current and spends nothing on catching up
// (16 updates on average at k = 32, on the critical path of the step before).  The same fused_advance call the step
// would have made: the same bits.
__global__ __launch_bounds__(256) void bpr_fused_pre_kernel(const float* __restrict__ P, const float* __restrict__ M,
                                                            const float* __restrict__ V, int64_t n_par, float* __restrict__ pre,
                                                            int64_t cap, const int32_t* __restrict__ slot_blk,
                                                            const int32_t* __restrict__ slot_fin, const int32_t* __restrict__ n_slots,
                                                            AdamBlockArgs a, const int32_t* __restrict__ tag_prev,
                                                            int32_t tag_prev_value) {
    const int lane = threadIdx.x & 63;
    const int64_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= *n_slots) return;
    const int64_t blk = slot_blk[slot];
    const int first = (slot_fin[slot] >> 16) & 0xff;
    if (first == 0 || tag_prev[blk] == tag_prev_value) return;       // nothing to catch up / the previous block still owns the row
    const int64_t e = blk * 64 + lane;
    FusedRow r{1.0f, 0.0f, 0.0f, 0.0f};
    if (e < n_par) { r.p = P[e]; r.m = M[e]; r.v = V[e]; }
    fused_advance(r, -1, 0, first, a);
    const int64_t o = slot * 64 + lane;
    pre[o] = r.p;
    pre[cap * 64 + o] = r.m;
    pre[2 * cap * 64 + o] = r.v;
}

// end of a k-step block: every slot is brought to the block's last index and written back; its gradient buffers are
// left zero for the next block.  fin = (number of namings mod 6) | (step of the last naming << 8) | (step of the first << 16)
__global__ __launch_bounds__(256) void bpr_fused_end_kernel(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V,
                                                            int64_t n_par, FusedWork w, const int32_t* __restrict__ slot_blk,
                                                            const int32_t* __restrict__ slot_fin,
                                                            const int32_t* __restrict__ n_slots, AdamBlockArgs a,
                                                            const int32_t* __restrict__ tag_next, int32_t tag_next_value, int which) {
    const int lane = threadIdx.x & 63;
    const int64_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= *n_slots) return;
    const int64_t blk = slot_blk[slot];
    // which = 1: only the rows the NEXT block touches too (it must find them in the dense tables); 2: only the others (they
    // can be written back beside the next block's steps); 0: all
    if (which != 0 && ((tag_next[blk] == tag_next_value) != (which == 1))) return;
    const int fin = slot_fin[slot], nn = fin & 7, last = (fin >> 8) & 0xff;
    FusedRow r;
    {
        const int64_t e = ((nn & 1) * w.cap + slot) * 64 + lane;
        r.p = w.wp[e];
        r.m = w.wm[e];
        r.v = w.wv[e];
        r.g = w.g[(((nn + 2) % 3) * w.cap + slot) * 64 + lane];
    }
    const int64_t e = blk * 64 + lane;
    if (e >= n_par) r = FusedRow{1.0f, 0.0f, 0.0f, 0.0f};
    fused_advance(r, last, last + 1, a.k, a);
    if (e < n_par) {
        P[e] = r.p;
        M[e] = r.m;
        V[e] = r.v;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) w.g[(q * w.cap + slot) * 64 + lane] = 0.0f;
}

// ---- the references' words of a k-step block (skr_bpr_fused_plan): four small launches, no sort -------------------------
// named[blk] collects, as a 64-bit mask, the steps of the block that name flat block `blk`; everything a reference needs
// follows from the mask: n0 = popcount below its own step, prev = the highest set bit below it.  claimed[blk] hands out
// one owner per (step, row); the owner of a row's FIRST naming draws the row's slot.  named / claimed are all-zero between
// calls (the last launch clears what the first two set).
struct FusedPlanArgs {
    const int32_t *u, *i, *j;
    int k, b;
    int64_t ublk0, iblk0, bblk0;
    unsigned long long *named, *claimed, *shared;
    int32_t *slot_of, *meta, *slot_block, *slot_fin, *n_slots;
    const int32_t* tag_prev;      // hot-block tags of the block BEFORE this one (NULL: no row is pre-advanced)
    int32_t tag_prev_value;
};

__device__ __forceinline__ void fused_plan_refs(const FusedPlanArgs& a, int64_t t, int64_t blk[5]) {
    const int64_t u = a.u[t], i = a.i[t], j = a.j[t];
    blk[0] = a.ublk0 + u;
    blk[1] = a.iblk0 + i;
    blk[2] = a.iblk0 + j;
    blk[3] = a.bblk0 + (i >> 6);
    blk[4] = a.bblk0 + (j >> 6);
}

template <int PASS>
__global__ __launch_bounds__(256) void fused_plan_kernel(FusedPlanArgs a) {
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;      // position in the block's step-major columns
    if (PASS == 0 && t == 0) *a.n_slots = 0;
    if (t >= static_cast<int64_t>(a.k) * a.b) return;
    const int s = static_cast<int>(t / a.b);
    const int64_t col = t - static_cast<int64_t>(s) * a.b;
    int64_t blk[5];
    fused_plan_refs(a, t, blk);
    const unsigned long long bit = 1ull << s;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int64_t e = (static_cast<int64_t>(s) * 5 + r) * a.b + col;      // this reference's word
        if (PASS == 0) {
            atomicOr(&a.named[blk[r]], bit);
            a.slot_block[e] = -1;
        } else if (PASS == 1) {
            const bool owner = (atomicOr(&a.claimed[blk[r]], bit) & bit) == 0;
            if (!owner) atomicOr(&a.shared[blk[r]], bit);                     // a second reference to the pair
            a.meta[e] = owner ? (1 << 23) : 0;
            const unsigned long long mask = a.named[blk[r]];
            // the row's first naming draws its slot: one atomic per wavefront on the shared counter (its lanes' draws are
            // numbered by their rank among the drawing lanes) -- 60 k same-address atomics per block otherwise
            const bool draws = owner && (mask & (bit - 1)) == 0;
            const unsigned long long db = __ballot(draws);
            int base = 0;
            if (db) {
                const int leader = __ffsll(static_cast<long long>(db)) - 1;
                if (static_cast<int>(threadIdx.x & 63) == leader) base = atomicAdd(a.n_slots, __popcll(db));
                base = __shfl(base, leader);
            }
            if (draws) {
                const int slot = base + __popcll(db & ((1ull << (threadIdx.x & 63)) - 1ull));
                a.slot_of[blk[r]] = slot;
                a.slot_block[slot] = static_cast<int32_t>(blk[r]);
                a.slot_fin[slot] = (__popcll(mask) % 6) | ((63 - __clzll(static_cast<long long>(mask))) << 8) | (s << 16);
            }
        } else if (PASS == 2) {
            const unsigned long long below = a.named[blk[r]] & (bit - 1);
            const int n0 = __popcll(below), prev1 = below ? 64 - __clzll(static_cast<long long>(below)) : 0;
            const int sole = (a.shared[blk[r]] & bit) ? 0 : 1;                // the pair's only reference (bit 31)
            // a first naming at step s > 0 of a row the previous block did not touch: bpr_fused_pre_kernel catches it up
            const bool pre = a.tag_prev && below == 0 && s > 0 && a.tag_prev[blk[r]] != a.tag_prev_value;
            a.meta[e] = a.meta[e] | a.slot_of[blk[r]] | ((pre ? FUSED_PRE : n0 % 6) << FUSED_SLOT_BITS) | (prev1 << 24) |
                        static_cast<int32_t>(static_cast<uint32_t>(sole) << 31);
        } else {
            a.named[blk[r]] = 0;
            a.claimed[blk[r]] = 0;
            a.shared[blk[r]] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sparse exchange of the replicated item table's gradient (multi-GPU BPRMF).  A step touches at
// most 2 * batch of the I item rows, so the ranks exchange packed rows [id | dV (64) | db] instead of
// all-reducing the dense [I, 65] block (26 MB at I = 100 k): pack -> all-gather -> unpack.
// ------------------------------------------------------------------------------------------------
constexpr int PK_W = D + 2;   // floats per packed row

// one wavefront per slot; ids are unique within a call or negative (empty slot).  The dense rows are
// cleared after packing so that every rank re-accumulates all contributions in the same (rank) order.
__global__ __launch_bounds__(256) void pack_grad_rows_kernel(const int32_t* __restrict__ ids, int n, float* __restrict__ gV,
                                                             float* __restrict__ gb, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int id = ids[k];
    float* o = out + static_cast<int64_t>(k) * PK_W;
    if (id < 0) {
        if (lane == 0) o[0] = __int_as_float(-1);
        return;
    }
    float* src = gV + static_cast<int64_t>(id) * D;
    o[1 + lane] = src[lane];
    src[lane] = 0.0f;
    if (lane == 0) {
        o[0] = __int_as_float(id);
        o[1 + D] = gb ? gb[id] : 0.0f;
        if (gb) gb[id] = 0.0f;
    }
}

// dense += packed rows of ONE rank (unique ids: plain read-modify-write, no atomics, deterministic)
__global__ __launch_bounds__(256) void unpack_grad_rows_kernel(const float* __restrict__ in, int n, float* __restrict__ gV,
                                                               float* __restrict__ gb, uint8_t* __restrict__ touch,
                                                               const float* touch_base) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const float* r = in + static_cast<int64_t>(k) * PK_W;
    const int id = __float_as_int(r[0]);
    if (id < 0) return;
    float* dst = gV + static_cast<int64_t>(id) * D;
    dst[lane] += r[1 + lane];
    if (lane == 0) {
        if (gb) gb[id] += r[1 + D];
        if (touch) {
            uint8_t* t = &touch[(dst - touch_base) >> 6];
            if (*t == 0) *t = 1;
            if (gb) {
                uint8_t* tb = &touch[(gb + id - touch_base) >> 6];
                if (*tb == 0) *tb = 1;
            }
        }
    }
}

// The same sum in ONE launch, for packs whose valid ids ascend with the empty slots last.  One wavefront per (rank, slot):
// it looks the id up in every other rank's id column (64 lanes probe a stride, then the 32 entries of the stride that
// can hold it); the lowest rank that holds the id owns it and adds the ranks' rows to the dense row in rank order --
// the order of the per-rank launches above, so the results are the same bits -- with a plain read-modify-write.
constexpr int UP_MAX_RANKS = 16;

__device__ __forceinline__ int pack_find(const float* __restrict__ col0, int n, int id, int lane) {
    // col0: &in[rank][0][0]; ids sit PK_W floats apart.  -> slot of `id`, or -1
    auto key = [&](int j) -> int {
        const int v = j < n ? __float_as_int(col0[static_cast<int64_t>(j) * PK_W]) : -1;
        return v < 0 ? 0x7fffffff : v;
    };
    const int stride = (n + 63) >> 6;                       // 64 probes cover the column
    const int k0 = key(lane * stride);
    // first probe whose key exceeds id -> the stride before it may hold id
    const uint64_t gt = __builtin_amdgcn_ballot_w64(k0 > id);
    const int first_gt = gt ? __builtin_ctzll(gt) : 64;
    if (first_gt == 0) return -1;
    const int base = (first_gt - 1) * stride;
    int found = -1;
    for (int off = 0; off < stride; off += 64) {
        const int j = base + off + lane;
        const bool hit = off + lane < stride && key(j) == id;
        const uint64_t hb = __builtin_amdgcn_ballot_w64(hit);
        if (hb) found = base + off + __builtin_ctzll(hb);
    }
    return found;
}

__global__ __launch_bounds__(256) void unpack_grad_rows_sorted_kernel(const float* __restrict__ in, int n, int n_ranks,
                                                                      float* __restrict__ gV, float* __restrict__ gb,
                                                                      uint8_t* __restrict__ touch, const float* touch_base) {
    const int lane = threadIdx.x & 63;
    const int64_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= static_cast<int64_t>(n) * n_ranks) return;
    const int r = static_cast<int>(w / n), k = static_cast<int>(w % n);
    const float* mine = in + (static_cast<int64_t>(r) * n + k) * PK_W;
    const int id = __float_as_int(mine[0]);
    if (id < 0) return;
    int pos[UP_MAX_RANKS];
#pragma unroll
    for (int q = 0; q < UP_MAX_RANKS; ++q)
        pos[q] = (q < n_ranks && q != r) ? pack_find(in + static_cast<int64_t>(q) * n * PK_W, n, id, lane) : -1;
#pragma unroll
    for (int q = 0; q < UP_MAX_RANKS; ++q)
        if (q < r && pos[q] >= 0) return;                    // a lower rank holds the id: that wavefront owns it
    float* dst = gV + static_cast<int64_t>(id) * D;
    float acc = dst[lane] + mine[1 + lane];
    float accb = 0.0f;
    if (gb && lane == 0) accb = gb[id] + mine[1 + D];
#pragma unroll
    for (int q = 0; q < UP_MAX_RANKS; ++q) {
        if (q > r && pos[q] >= 0) {
            const float* row = in + (static_cast<int64_t>(q) * n + pos[q]) * PK_W;
            acc += row[1 + lane];
            if (gb && lane == 0) accb += row[1 + D];
        }
    }
    dst[lane] = acc;
    if (lane == 0) {
        if (gb) gb[id] = accb;
        if (touch) {
            uint8_t* t = &touch[(dst - touch_base) >> 6];
            if (*t == 0) *t = 1;
            if (gb) {
                uint8_t* tb = &touch[(gb + id - touch_base) >> 6];
                if (*tb == 0) *tb = 1;
            }
        }
    }
}

}  // namespace

extern "C" {

static int bpr_step_launch(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                 const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, float loss_scale, float reg,
                 float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss,
                 uint8_t* d_touch, const float* d_touch_base, int loss_slots, void* stream, int shard_world = 1,
                 int shard_rank = 0, float grad_scale = 1.0f, int dim = D) {
    SKR_REQUIRE(d_P && d_Q && d_RP && d_RQ && d_u && d_i && d_j && d_gP && d_gQ && d_gRP && d_gRQ && d_loss,
                "skr_bpr_step: NULL argument");
    SKR_REQUIRE(shard_world >= 1 && shard_rank >= 0 && shard_rank < shard_world, "skr_bpr_step_sharded: rank %d of %d", shard_rank,
                shard_world);
    SKR_REQUIRE(n >= 0, "skr_bpr_step: negative batch size");
    SKR_REQUIRE(!d_touch || d_touch_base, "skr_bpr_step: d_touch needs d_touch_base");
    if (n == 0) return SKR_OK;
    int blocks = (n + BPR_WAVES - 1) / BPR_WAVES;
    if (blocks > 4096) blocks = 4096;
    SKR_REQUIRE(dim == 64 || dim == 128 || dim == 192 || dim == 256, "skr_bpr_step: dim must be 64, 128, 192 or 256 (got %d); pad narrower "
                "rows with zeros", dim);
#define SKR_BPR_LAUNCH(C_)                                                                                                       \
    hipLaunchKernelGGL(bpr_step_kernel<C_>, dim3(blocks), dim3(BPR_WAVES * 64), 0, skr::as_stream(stream), d_P, d_Q, d_bias,     \
                       d_RP, d_RQ, d_u, d_i, d_j, n, loss_scale, reg, reg_scale, d_gP, d_gQ, d_gb, d_gRP, d_gRQ, d_loss,        \
                       d_touch, d_touch_base, loss_slots, shard_world, shard_rank, grad_scale)
    switch (dim / 64) {
        case 1: SKR_BPR_LAUNCH(1); break;
        case 2: SKR_BPR_LAUNCH(2); break;
        case 3: SKR_BPR_LAUNCH(3); break;
        default: SKR_BPR_LAUNCH(4); break;
    }
#undef SKR_BPR_LAUNCH
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_bpr_step_dim(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                     const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, int dim, float loss_scale, float reg,
                     float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss,
                     int loss_slots, uint8_t* d_touch, const float* d_touch_base, float grad_scale, void* stream) {
    SKR_REQUIRE(loss_slots == 1 || loss_slots == SKR_LOSS_SLOTS, "skr_bpr_step_dim: loss_slots must be 1 or %d", SKR_LOSS_SLOTS);
    return bpr_step_launch(d_P, d_Q, d_bias, d_RP, d_RQ, d_u, d_i, d_j, n, loss_scale, reg, reg_scale, d_gP, d_gQ, d_gb, d_gRP,
                           d_gRQ, d_loss, d_touch, d_touch_base, loss_slots, stream, 1, 0, grad_scale, dim);
}

int skr_bpr_step(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                 const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, float loss_scale, float reg,
                 float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss,
                 uint8_t* d_touch, const float* d_touch_base, void* stream) {
    return bpr_step_launch(d_P, d_Q, d_bias, d_RP, d_RQ, d_u, d_i, d_j, n, loss_scale, reg, reg_scale, d_gP, d_gQ, d_gb, d_gRP,
                           d_gRQ, d_loss, d_touch, d_touch_base, 1, stream);
}

int skr_bpr_step_sharded(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                         const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, float loss_scale, float reg,
                         float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss,
                         uint8_t* d_touch, const float* d_touch_base, int shard_world, int shard_rank, float grad_scale,
                         void* stream) {
    return bpr_step_launch(d_P, d_Q, d_bias, d_RP, d_RQ, d_u, d_i, d_j, n, loss_scale, reg, reg_scale, d_gP, d_gQ, d_gb, d_gRP,
                           d_gRQ, d_loss, d_touch, d_touch_base, 1, stream, shard_world, shard_rank, grad_scale);
}

int skr_bpr_step_spread(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                        const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, float loss_scale, float reg,
                        float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss64,
                        uint8_t* d_touch, const float* d_touch_base, void* stream) {
    return bpr_step_launch(d_P, d_Q, d_bias, d_RP, d_RQ, d_u, d_i, d_j, n, loss_scale, reg, reg_scale, d_gP, d_gQ, d_gb, d_gRP,
                           d_gRQ, d_loss64, d_touch, d_touch_base, SKR_LOSS_SLOTS, stream);
}

static int adam_step_impl(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int64_t step_t, int zero_grad, uint8_t* d_touch, bool tf, void* stream);
int skr_adam_step(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t step_t, int zero_grad, uint8_t* d_touch, void* stream) {
    return adam_step_impl(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, step_t, zero_grad, d_touch, false, stream);
}
int skr_adam_step_tf(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                     float eps, int64_t step_t, int zero_grad, uint8_t* d_touch, void* stream) {
    return adam_step_impl(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, step_t, zero_grad, d_touch, true, stream);
}
static int adam_step_impl(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int64_t step_t, int zero_grad, uint8_t* d_touch, bool tf, void* stream) {
    SKR_REQUIRE(d_p && d_g && d_m && d_v, "skr_adam_step: NULL argument");
    SKR_REQUIRE(n >= 0 && step_t >= 1, "skr_adam_step: n must be >= 0 and step_t >= 1");
    SKR_REQUIRE(((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_g) | reinterpret_cast<uintptr_t>(d_m) |
                  reinterpret_cast<uintptr_t>(d_v)) & 15) == 0, "skr_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return SKR_OK;
    // torch/optim/adam.py _single_tensor_adam: python-double scalars, cast to fp32 at the tensor ops
    const double b1 = static_cast<double>(beta1), b2 = static_cast<double>(beta2);
    const double bc1 = 1.0 - std::pow(b1, static_cast<double>(step_t));
    const double bc2 = 1.0 - std::pow(b2, static_cast<double>(step_t));
    AdamArgs a;
    a.one_minus_b1 = static_cast<float>(1.0 - b1);
    a.b2 = beta2;
    a.one_minus_b2 = static_cast<float>(1.0 - b2);
    a.neg_step_size = static_cast<float>(-(static_cast<double>(lr) / bc1));
    a.bc2_sqrt = static_cast<float>(std::sqrt(bc2));
    if (tf) {      // tf.train.AdamOptimizer's placement of the second bias correction (adam_scalars)
        a.neg_step_size = static_cast<float>(-(static_cast<double>(lr) * std::sqrt(bc2) / bc1));
        a.bc2_sqrt = 1.0f;
    }
    a.eps = eps;
    // Launch shape measured on MI355X (tools/tune_adam.sh, profiles/r01_adam_tuning.txt): 2 workgroups
    // per CU, 4 float4 per lane in flight, non-temporal accesses.  SKR_ADAM_CFG="<blocks_per_cu>,
    // <unroll>,<nt>" overrides it for tuning runs.
    static int cfg_bpc = 2, cfg_unroll = 4, cfg_nt = 1;
    static bool cfg_read = false;
    if (!cfg_read) {
        cfg_read = true;
        if (const char* e = getenv("SKR_ADAM_CFG")) sscanf(e, "%d,%d,%d", &cfg_bpc, &cfg_unroll, &cfg_nt);
    }
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 256 * cfg_bpc) blocks = 256 * cfg_bpc;
    if (blocks < 1) blocks = 1;
    const dim3 grid(static_cast<unsigned>(blocks)), blk(256);
    hipStream_t st = skr::as_stream(stream);
#define SKR_ADAM_LAUNCH(T, U_, N_) \
    hipLaunchKernelGGL((adam_kernel<T, U_, N_>), grid, blk, 0, st, d_p, d_g, d_m, d_v, n, a, zero_grad, d_touch)
#define SKR_ADAM_PICK(T)                                                        \
    if (cfg_nt) {                                                               \
        if (cfg_unroll == 4) SKR_ADAM_LAUNCH(T, 4, true);                       \
        else if (cfg_unroll == 2) SKR_ADAM_LAUNCH(T, 2, true);                  \
        else SKR_ADAM_LAUNCH(T, 1, true);                                       \
    } else {                                                                    \
        if (cfg_unroll == 4) SKR_ADAM_LAUNCH(T, 4, false);                      \
        else if (cfg_unroll == 2) SKR_ADAM_LAUNCH(T, 2, false);                 \
        else SKR_ADAM_LAUNCH(T, 1, false);                                      \
    }
    if (d_touch) { SKR_ADAM_PICK(true) } else { SKR_ADAM_PICK(false) }
#undef SKR_ADAM_PICK
#undef SKR_ADAM_LAUNCH
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_csr_spmm(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val, const float* d_X,
                 int dim, int64_t nnz, const float* d_addend, float* d_Y, float* d_accum, float accum_scale,
                 void* stream) {
    return skr_csr_spmm_strided(n_rows, d_rowptr, d_col, d_val, d_X, dim, D, nnz, d_addend, d_Y, d_accum, accum_scale, stream);
}

int skr_csr_spmm_strided(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val, const float* d_X,
                         int dim, int ld, int64_t nnz, const float* d_addend, float* d_Y, float* d_accum, float accum_scale,
                         void* stream) {
    SKR_REQUIRE(d_rowptr && d_col && d_val && d_X && d_Y, "skr_csr_spmm: NULL argument");
    SKR_REQUIRE(dim == D, "skr_csr_spmm: dim must be 64 (got %d): wider tables are multiplied in 64-column slices (ld)", dim);
    SKR_REQUIRE(ld >= D, "skr_csr_spmm: the row stride must be at least 64 floats (got %d)", ld);
    SKR_REQUIRE(n_rows >= 0 && nnz >= 0, "skr_csr_spmm: negative size");
    SKR_REQUIRE(d_Y != d_X, "skr_csr_spmm: in-place propagation is not supported");
    if (n_rows == 0) return SKR_OK;
    hipStream_t st = skr::as_stream(stream);
    hipLaunchKernelGGL(spmm_prep_kernel, dim3(rows_to_blocks(n_rows)), dim3(256), 0, st, n_rows, d_rowptr, d_addend, d_Y,
                       d_accum, accum_scale, ld);
    SKR_LAUNCH_CHECK();
    if (nnz > 0) {
        const int64_t waves = (nnz + SP_CH - 1) / SP_CH;
        hipLaunchKernelGGL(spmm_main_kernel, dim3(static_cast<unsigned>((waves + SP_WAVES - 1) / SP_WAVES)),
                           dim3(SP_WAVES * 64), 0, st, n_rows, d_rowptr, d_col, d_val, d_X, d_addend, d_Y, d_accum,
                           accum_scale, nnz, ld);
        SKR_LAUNCH_CHECK();
        hipLaunchKernelGGL(spmm_fix_kernel, dim3(rows_to_blocks(n_rows)), dim3(256), 0, st, n_rows, d_rowptr, d_addend,
                           d_Y, d_accum, accum_scale, ld);
        SKR_LAUNCH_CHECK();
    }
    return SKR_OK;
}

int skr_layer_refine_fwd(const float* d_Y, const float* d_E, int64_t n_rows, int dim, float* d_Z, float* d_w,
                         float* d_accum, void* stream) {
    SKR_REQUIRE(d_Y && d_E && d_Z && d_w, "skr_layer_refine_fwd: NULL argument");
    SKR_REQUIRE(dim == 64 || dim == 128 || dim == 192 || dim == 256, "skr_layer_refine_fwd: dim must be 64, 128, 192 or 256 (got %d)", dim);
    if (n_rows <= 0) return SKR_OK;
#define SKR_RF(C_) hipLaunchKernelGGL(refine_fwd_kernel<C_>, dim3(rows_to_blocks(n_rows)), dim3(256), 0, skr::as_stream(stream), d_Y, d_E, \
                                      n_rows, d_Z, d_w, d_accum)
    switch (dim / 64) { case 1: SKR_RF(1); break; case 2: SKR_RF(2); break; case 3: SKR_RF(3); break; default: SKR_RF(4); break; }
#undef SKR_RF
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_layer_refine_bwd_masked(const float* d_Y, const float* d_E, const float* d_w, const float* d_dZ, int64_t n_rows, int dim,
                                float* d_dY, float* d_dE, const uint8_t* d_row_mask, int zero_skipped, void* stream) {
    SKR_REQUIRE(d_Y && d_E && d_w && d_dZ && d_dY && d_dE, "skr_layer_refine_bwd: NULL argument");
    SKR_REQUIRE(dim == 64 || dim == 128 || dim == 192 || dim == 256, "skr_layer_refine_bwd: dim must be 64, 128, 192 or 256 (got %d)", dim);
    if (n_rows <= 0) return SKR_OK;
#define SKR_RB(C_) hipLaunchKernelGGL(refine_bwd_kernel<C_>, dim3(rows_to_blocks(n_rows)), dim3(256), 0, skr::as_stream(stream), d_Y, d_E, \
                                      d_w, d_dZ, n_rows, d_dY, d_dE, d_row_mask, zero_skipped)
    switch (dim / 64) { case 1: SKR_RB(1); break; case 2: SKR_RB(2); break; case 3: SKR_RB(3); break; default: SKR_RB(4); break; }
#undef SKR_RB
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_layer_refine_bwd(const float* d_Y, const float* d_E, const float* d_w, const float* d_dZ, int64_t n_rows, int dim,
                         float* d_dY, float* d_dE, void* stream) {
    return skr_layer_refine_bwd_masked(d_Y, d_E, d_w, d_dZ, n_rows, dim, d_dY, d_dE, nullptr, 0, stream);
}

int skr_clear_marked_rows(uint8_t* d_mask, int64_t n_rows, int64_t clear_mask, float* d_table, int dim, void* stream) {
    SKR_REQUIRE(n_rows >= 0 && dim >= 1, "skr_clear_marked_rows: bad shape");
    if (n_rows == 0) return SKR_OK;
    SKR_REQUIRE(d_mask && d_table, "skr_clear_marked_rows: NULL argument");
    const int64_t waves = (n_rows + 63) / 64;
    hipLaunchKernelGGL(clear_marked_rows_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, skr::as_stream(stream), d_mask,
                       n_rows, clear_mask ? 1 : 0, d_table, dim);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_gather_rows(const float* d_table, const int32_t* d_idx, int64_t n, int dim, float* d_out, void* stream) {
    SKR_REQUIRE(d_table && d_idx && d_out, "skr_gather_rows: NULL argument");
    SKR_REQUIRE(dim >= 1, "skr_gather_rows: dim must be positive (got %d)", dim);
    if (n <= 0) return SKR_OK;
    if (dim == D)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_to_blocks(n)), dim3(256), 0, skr::as_stream(stream), d_table, d_idx,
                           n, d_out);
    else
        hipLaunchKernelGGL(gather_rows_any_kernel, dim3(static_cast<unsigned>((n * dim + 255) / 256)), dim3(256), 0,
                           skr::as_stream(stream), d_table, d_idx, n, dim, d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_scatter_rows(const float* d_src, const int32_t* d_idx, int64_t n, int dim, float* d_table, void* stream) {
    SKR_REQUIRE(d_table && d_idx && d_src, "skr_scatter_rows: NULL argument");
    SKR_REQUIRE(dim >= 1, "skr_scatter_rows: dim must be positive (got %d)", dim);
    if (n <= 0) return SKR_OK;
    if (dim == D)
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows_to_blocks(n)), dim3(256), 0, skr::as_stream(stream), d_src, d_idx, n,
                           d_table);
    else
        hipLaunchKernelGGL(scatter_rows_any_kernel, dim3(static_cast<unsigned>((n * dim + 255) / 256)), dim3(256), 0,
                           skr::as_stream(stream), d_src, d_idx, n, dim, d_table);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_scale_copy(float a, const float* d_x, float* d_y, int64_t n, void* stream) {
    SKR_REQUIRE(d_x && d_y, "skr_scale_copy: NULL argument");
    if (n <= 0) return SKR_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_copy_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream), a, d_x, d_y, n);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_sum_blocks(const float* d_in, int n_blocks, int64_t n, float* d_out, void* stream) {
    SKR_REQUIRE(d_in && d_out && n_blocks >= 1 && n >= 0, "skr_sum_blocks: bad argument");
    if (n == 0) return SKR_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream), d_in, n_blocks, n,
                       d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_axpy(float a, const float* d_x, float* d_y, int64_t n, void* stream) {
    SKR_REQUIRE(d_x && d_y, "skr_axpy: NULL argument");
    if (n <= 0) return SKR_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream), a, d_x, d_y,
                       n);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_scale(float a, float* d_x, int64_t n, void* stream) {
    SKR_REQUIRE(d_x, "skr_scale: NULL argument");
    if (n <= 0) return SKR_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream), a, d_x, n);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_pack_grad_rows(const int32_t* d_ids, int n, float* d_g_table, float* d_g_bias, int dim, float* d_out, void* stream) {
    SKR_REQUIRE(d_ids && d_g_table && d_out, "skr_pack_grad_rows: NULL argument");
    SKR_REQUIRE(dim == D, "skr_pack_grad_rows: dim must be 64 (got %d)", dim);
    if (n <= 0) return SKR_OK;
    hipLaunchKernelGGL(pack_grad_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, skr::as_stream(stream), d_ids, n, d_g_table,
                       d_g_bias, d_out);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_unpack_grad_rows(const float* d_in, int n_per_rank, int n_ranks, float* d_g_table, float* d_g_bias, int dim,
                         uint8_t* d_touch, const float* d_touch_base, void* stream) {
    SKR_REQUIRE(d_in && d_g_table, "skr_unpack_grad_rows: NULL argument");
    SKR_REQUIRE(dim == D, "skr_unpack_grad_rows: dim must be 64 (got %d)", dim);
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
    if (n_per_rank <= 0 || n_ranks <= 0) return SKR_OK;
    // one launch per rank, in rank order: ids are unique inside a rank's block, and every replica adds the
    // blocks in the same order, so the replicated tables stay bit-identical
    for (int r = 0; r < n_ranks; ++r) {
        hipLaunchKernelGGL(unpack_grad_rows_kernel, dim3((n_per_rank + 3) / 4), dim3(256), 0, skr::as_stream(stream),
                           d_in + static_cast<int64_t>(r) * n_per_rank * PK_W, n_per_rank, d_g_table, d_g_bias, d_touch,
                           d_touch_base);
        SKR_LAUNCH_CHECK();
    }
    return SKR_OK;
}

int skr_unpack_grad_rows_sorted(const float* d_in, int n_per_rank, int n_ranks, float* d_g_table, float* d_g_bias, int dim,
                                uint8_t* d_touch, const float* d_touch_base, void* stream) {
    SKR_REQUIRE(d_in && d_g_table, "skr_unpack_grad_rows_sorted: NULL argument");
    SKR_REQUIRE(dim == D, "skr_unpack_grad_rows_sorted: dim must be 64 (got %d)", dim);
    SKR_REQUIRE((d_touch == nullptr) == (d_touch_base == nullptr), "touch: both pointers or neither");
    if (n_per_rank <= 0 || n_ranks <= 0) return SKR_OK;
    if (n_ranks > UP_MAX_RANKS)   // beyond the kernel's register table: the per-rank launches give the same bits
        return skr_unpack_grad_rows(d_in, n_per_rank, n_ranks, d_g_table, d_g_bias, dim, d_touch, d_touch_base, stream);
    const int64_t waves = static_cast<int64_t>(n_per_rank) * n_ranks;
    hipLaunchKernelGGL(unpack_grad_rows_sorted_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0,
                       skr::as_stream(stream), d_in, n_per_rank, n_ranks, d_g_table, d_g_bias, d_touch, d_touch_base);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

static void adam_scalars(float lr, float beta1, float beta2, int64_t step_t, float* neg_step_size, float* bc2_sqrt, bool tf = false) {
    // torch/optim/adam.py _single_tensor_adam: python-double scalars, cast to fp32 at the tensor ops (as skr_adam_step)
    const double b1 = static_cast<double>(beta1), b2 = static_cast<double>(beta2);
    const double bc1 = 1.0 - std::pow(b1, static_cast<double>(step_t));
    const double bc2 = 1.0 - std::pow(b2, static_cast<double>(step_t));
    if (tf) {
        // tf.train.AdamOptimizer (GRU4RecPlus.py:192): p -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t):
        // the second bias correction sits in the step size, the denominator has none
        *neg_step_size = static_cast<float>(-(static_cast<double>(lr) * std::sqrt(bc2) / bc1));
        *bc2_sqrt = 1.0f;
        return;
    }
    *neg_step_size = static_cast<float>(-(static_cast<double>(lr) / bc1));
    *bc2_sqrt = static_cast<float>(std::sqrt(bc2));
}

// the k steps' scalars of a block that starts after step_t0, and the running maximum the at-rest tests use
static void adam_block_scalars(AdamBlockArgs& a, float lr, float beta1, float beta2, int64_t step_t0, int k, bool tf) {
    for (int s = 0; s < k; ++s) adam_scalars(lr, beta1, beta2, step_t0 + 1 + s, &a.neg_step_size[s], &a.bc2_sqrt[s], tf);
    float mx = 0.0f;
    for (int s = k - 1; s >= 0; --s) {
        mx = std::fmax(mx, std::fabs(a.neg_step_size[s]));
        a.nss_bound[s] = mx;
    }
}

int skr_adam_block_mark(const int32_t* d_ids, int64_t n_ids, int64_t offset_floats, int stride_floats, int32_t* d_tag,
                        int32_t tag_value, int32_t* d_claim, int64_t step_t0, void* stream) {
    SKR_REQUIRE(d_ids && d_tag, "skr_adam_block_mark: NULL argument");
    SKR_REQUIRE(n_ids >= 0 && offset_floats >= 0 && stride_floats >= 1, "skr_adam_block_mark: bad shape");
    SKR_REQUIRE(step_t0 >= 0 && step_t0 < INT32_MAX - AB_KMAX, "skr_adam_block_mark: step_t0 out of range");
    if (n_ids == 0) return SKR_OK;
    hipLaunchKernelGGL(adam_mark_kernel, dim3(static_cast<unsigned>((n_ids + 255) / 256)), dim3(256), 0, skr::as_stream(stream),
                       d_ids, n_ids, offset_floats, stride_floats, d_tag, tag_value, d_claim, static_cast<int32_t>(step_t0));
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

// SKR_COLD_STATS=1: a device census of how the cold passes sorted their blocks (read with skr_cold_pass_census)
static unsigned long long* cold_stats_buffer() {
    static unsigned long long* buf = [] {
        unsigned long long* p = nullptr;
        const char* e = getenv("SKR_COLD_STATS");
        if (e && atoi(e) == 1 && hipMalloc(&p, 4 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(p, 0, 4 * sizeof(unsigned long long));
            return p;
        }
        return static_cast<unsigned long long*>(nullptr);
    }();
    return buf;
}

// thresholds of the at-rest / ordinary-magnitude tests for a run of up to k zero-gradient updates whose scalars sit in
// a.neg_step_size[0 .. k-1] / a.bc2_sqrt[0 .. k-1] (|neg_step_size| falls, bc2_sqrt rises with the step)
static void adam_block_thresholds(AdamBlockArgs& a, float lr, float beta1, float beta2, float eps, int k) {
    const bool sane = beta1 > 0.0f && beta1 < 1.0f && beta2 > 0.0f && beta2 < 1.0f && lr > 0.0f && eps >= 0.0f &&
                      std::isfinite(lr) && std::isfinite(eps);
    a.rest_eps = sane ? eps * 0x1p-28f : 0.0f;
    a.rest_b2k = sane ? static_cast<float>(std::pow(static_cast<double>(beta2), k) * (1.0 - 1e-4)) : 0.0f;
    // ordinary magnitudes for all k updates (ranges of sqrt_ordinary / div_ordinary with room to spare): v in
    // [2^-90, 2^20] throughout, |nss*m| in [2^-100, 2^40] throughout, eps <= 2^20, sqrt(1 - beta2^t) >= 2^-10
    double nss_max = 0.0, nss_min = INFINITY;      // torch's scalars: the first and the last step's; TF's are not monotone
    for (int s_ = 0; s_ < k; ++s_) {
        nss_max = std::fmax(nss_max, std::fabs(static_cast<double>(a.neg_step_size[s_])));
        nss_min = std::fmin(nss_min, std::fabs(static_cast<double>(a.neg_step_size[s_])));
    }
    const double m_lo = 0x1p-100 / (nss_min * std::pow(static_cast<double>(beta1), k) * 0.99), m_hi = 0x1p40 / nss_max;
    const bool ord = sane && eps <= 0x1p20f && a.bc2_sqrt[0] >= 0x1p-10f && a.rest_b2k > 0.0f && m_lo < 1e30 && m_hi > 1e-30 &&
                     std::isfinite(m_lo) && std::isfinite(m_hi);
    a.fast_vlo = ord ? static_cast<float>(0x1p-90 / static_cast<double>(a.rest_b2k)) : 0.0f;
    a.fast_mlo = ord ? static_cast<float>(m_lo) : INFINITY;
    a.fast_mhi = ord ? static_cast<float>(std::fmin(m_hi, 1e38)) : 0.0f;
}

int skr_cold_pass_census(uint64_t* h_counts3, int reset) {
    SKR_REQUIRE(h_counts3, "skr_cold_pass_census: NULL argument");
    unsigned long long* buf = cold_stats_buffer();
    h_counts3[0] = h_counts3[1] = h_counts3[2] = 0;
    if (!buf) return SKR_OK;
    unsigned long long h[4];
    SKR_HIP(hipDeviceSynchronize());
    SKR_HIP(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    if (reset) SKR_HIP(hipMemset(buf, 0, sizeof(h)));
    for (int i = 0; i < 3; ++i) h_counts3[i] = h[i];
    return SKR_OK;
}

static int adam_block_cold_impl(float* d_p, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                               int64_t step_t0, int k, const int32_t* d_tag, int32_t hot_value, bool tf, void* stream) {
    SKR_REQUIRE(d_p && d_m && d_v && d_tag, "skr_adam_block_cold: NULL argument");
    SKR_REQUIRE(n >= 0 && step_t0 >= 0 && k >= 1 && k <= AB_KMAX, "skr_adam_block_cold: need 1 <= k <= %d", AB_KMAX);
    SKR_REQUIRE(((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_m) | reinterpret_cast<uintptr_t>(d_v)) & 15) == 0,
                "skr_adam_block_cold: buffers must be 16-byte aligned");
    if (n == 0) return SKR_OK;
    AdamBlockArgs a;
    a.one_minus_b1 = static_cast<float>(1.0 - static_cast<double>(beta1));
    a.b2 = beta2;
    a.one_minus_b2 = static_cast<float>(1.0 - static_cast<double>(beta2));
    a.eps = eps;
    a.k = k;
    adam_block_scalars(a, lr, beta1, beta2, step_t0, k, tf);
    static const int bpc = [] { const char* e = getenv("SKR_COLD_BPC"); const int v = e ? atoi(e) : 5; return v < 1 ? 1 : (v > 8 ? 8 : v); }();   // workgroups per CU.  The pass runs beside the k-step block's small launches: round 2 settled on 4 (960 timed steps: 24.9 / 31.3 / 30.3 / 28.1 M interactions/s at 2 / 3 / 4 / 8); round 3: 5, together with the step kernel's issue priority (see bpr_fused_step_kernel)
    // SKR_COLD_REST=0 keeps every cold block on the full update (the float4 kernel): the A/B switch of tools/microbench.py
    static const bool rest = [] { const char* e = getenv("SKR_COLD_REST"); return !(e && atoi(e) == 0); }();
    // A pass over a WHOLE block of the default length (32 steps and more) of the BPR tables takes six workgroups per CU
    // (SKR_COLD_BPC_FULL; 0: SKR_COLD_BPC for every pass): in the steady state of an epoch the pass and the step stream are
    // balanced (0.53 ms against 32 x 15 us + the write-back), and the sixth workgroup takes 0.04 ms off the pass for 0.4 us per
    // step launch -- an epoch 0.997 -> 0.956 s on the same box (tools/r3_bpc_full.sh).  Shorter blocks (the 20-step slice of
    // the bench line, an epoch's ragged last block) leave the step stream less work to hide the pass behind and keep five: with
    // six for every pass the short slice scatters (34.5-39.5 M interactions/s against 39.4-41.0).  GRU4RecPlus's pass (TF
    // arithmetic) keeps SKR_COLD_BPC: its step is a chain of eight small launches that was measured with five.
    static const int bpc_full = [] { const char* e = getenv("SKR_COLD_BPC_FULL"); const int v = e ? atoi(e) : 6; return v < 1 ? 0 : (v > 8 ? 8 : v); }();
    const int bpc_k = (bpc_full && k >= 32 && !tf) ? bpc_full : bpc;
    adam_block_thresholds(a, lr, beta1, beta2, eps, k);
    a.stats = cold_stats_buffer();
    if (rest) {
        int64_t blocks = ((n >> 6) + 4 * 4 - 1) / (4 * 4);
        if (blocks > 256 * bpc_k) blocks = 256 * bpc_k;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(adam_cold_rows_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream),
                           d_p, d_m, d_v, n, a, d_tag, hot_value);
    } else {
        int64_t blocks = ((n >> 2) + 255) / 256;
        if (blocks > 256 * bpc) blocks = 256 * bpc;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(adam_cold_kernel<2>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, skr::as_stream(stream), d_p,
                           d_m, d_v, n, a, d_tag, hot_value);
    }
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_adam_block_cold(float* d_p, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                        int64_t step_t0, int k, const int32_t* d_tag, int32_t hot_value, void* stream) {
    return adam_block_cold_impl(d_p, d_m, d_v, n, lr, beta1, beta2, eps, step_t0, k, d_tag, hot_value, false, stream);
}
int skr_adam_block_cold_tf(float* d_p, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                           int64_t step_t0, int k, const int32_t* d_tag, int32_t hot_value, void* stream) {
    return adam_block_cold_impl(d_p, d_m, d_v, n, lr, beta1, beta2, eps, step_t0, k, d_tag, hot_value, true, stream);
}

static int adam_block_hot_impl(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                               float eps, int64_t step_t0, int64_t step_t, const int32_t* d_ids, int64_t n_ids, int64_t offset_floats,
                               int stride_floats, int32_t* d_claim, bool tf, void* stream) {
    SKR_REQUIRE(d_p && d_g && d_m && d_v && d_ids && d_claim, "skr_adam_block_hot: NULL argument");
    SKR_REQUIRE(n >= 0 && step_t0 >= 0 && step_t > step_t0 && step_t - step_t0 <= AB_KMAX && step_t < INT32_MAX,
                "skr_adam_block_hot: need step_t0 < step_t <= step_t0 + %d", AB_KMAX);
    SKR_REQUIRE(n_ids >= 0 && offset_floats >= 0 && stride_floats >= 1, "skr_adam_block_hot: bad shape");
    if (n_ids == 0) return SKR_OK;
    AdamBlockArgs a{};
    a.one_minus_b1 = static_cast<float>(1.0 - static_cast<double>(beta1));
    a.b2 = beta2;
    a.one_minus_b2 = static_cast<float>(1.0 - static_cast<double>(beta2));
    a.eps = eps;
    a.k = static_cast<int>(step_t - step_t0);
    adam_block_scalars(a, lr, beta1, beta2, step_t0, a.k, tf);
    adam_block_thresholds(a, lr, beta1, beta2, eps, a.k);   // for the zero-gradient steps a block may be behind
    hipLaunchKernelGGL(adam_hot_kernel, dim3(static_cast<unsigned>((n_ids + 3) / 4)), dim3(256), 0, skr::as_stream(stream), d_p,
                       d_g, d_m, d_v, n, a, static_cast<int32_t>(step_t0), static_cast<int32_t>(step_t), d_ids, n_ids,
                       offset_floats, stride_floats, d_claim);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}
int skr_adam_block_hot(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                       float eps, int64_t step_t0, int64_t step_t, const int32_t* d_ids, int64_t n_ids, int64_t offset_floats,
                       int stride_floats, int32_t* d_claim, void* stream) {
    return adam_block_hot_impl(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, step_t0, step_t, d_ids, n_ids, offset_floats, stride_floats,
                               d_claim, false, stream);
}
int skr_adam_block_hot_tf(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int64_t step_t0, int64_t step_t, const int32_t* d_ids, int64_t n_ids, int64_t offset_floats,
                          int stride_floats, int32_t* d_claim, void* stream) {
    return adam_block_hot_impl(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, step_t0, step_t, d_ids, n_ids, offset_floats, stride_floats,
                               d_claim, true, stream);
}

// scalars and thresholds of a k-step block, kept between the k + 1 launches of the block (2k pow() calls otherwise)
static const AdamBlockArgs& fused_block_args(float lr, float beta1, float beta2, float eps, int64_t step_t0, int k) {
    struct Key {
        float lr, b1, b2, eps;
        int64_t t0;
        int k;
    };
    thread_local Key key{0, 0, 0, 0, -1, 0};
    thread_local AdamBlockArgs a{};
    if (key.lr != lr || key.b1 != beta1 || key.b2 != beta2 || key.eps != eps || key.t0 != step_t0 || key.k != k) {
        a = AdamBlockArgs{};
        a.one_minus_b1 = static_cast<float>(1.0 - static_cast<double>(beta1));
        a.b2 = beta2;
        a.one_minus_b2 = static_cast<float>(1.0 - static_cast<double>(beta2));
        a.eps = eps;
        a.k = k;
        adam_block_scalars(a, lr, beta1, beta2, step_t0, k, false);
        adam_block_thresholds(a, lr, beta1, beta2, eps, k);
        key = Key{lr, beta1, beta2, eps, step_t0, k};
    }
    return a;
}

static FusedWork fused_work(float* d_work, int64_t cap) {
    const int64_t plane = cap * 64;
    return FusedWork{d_work, d_work + 2 * plane, d_work + 4 * plane, d_work + 6 * plane, cap};
}

int skr_bpr_fused_step(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_work, int64_t cap,
                       const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch,
                       int64_t user_block0, int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2,
                       float eps, int64_t step_t0, int k, int s, float reg, float* d_loss64, void* stream) {
    return skr_bpr_fused_step2(d_p, d_m, d_v, n, d_work, cap, d_u, d_i, d_j, d_meta, n_batch, user_block0, item_block0, bias_block0, lr,
                               beta1, beta2, eps, step_t0, k, s, reg, d_loss64, nullptr, stream);
}

int skr_bpr_fused_step2(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_work, int64_t cap,
                        const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch,
                        int64_t user_block0, int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2,
                        float eps, int64_t step_t0, int k, int s, float reg, float* d_loss64, const float* d_pre, void* stream) {
    SKR_REQUIRE(d_p && d_m && d_v && d_work && d_u && d_i && d_j && d_meta && d_loss64, "skr_bpr_fused_step: NULL argument");
    SKR_REQUIRE(n >= 0 && n_batch >= 0 && cap >= 1 && cap <= (1 << FUSED_SLOT_BITS), "skr_bpr_fused_step: need 1 <= cap <= 2^%d",
                FUSED_SLOT_BITS);
    SKR_REQUIRE(step_t0 >= 0 && k >= 1 && k <= AB_KMAX && s >= 0 && s < k, "skr_bpr_fused_step: need 0 <= s < k <= %d", AB_KMAX);
    SKR_REQUIRE(user_block0 >= 0 && item_block0 >= 0 && bias_block0 >= 0, "skr_bpr_fused_step: bad table offsets");
    if (n_batch == 0) return SKR_OK;
    const AdamBlockArgs& a = fused_block_args(lr, beta1, beta2, eps, step_t0, k);
    static const int dbg = [] { const char* e = getenv("SKR_FUSED_DBG"); return e ? atoi(e) : 0; }();
    int blocks = (n_batch + BPR_WAVES - 1) / BPR_WAVES;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(bpr_fused_step_kernel, dim3(blocks), dim3(BPR_WAVES * 64), 0, skr::as_stream(stream), d_p, d_m, d_v, n,
                       fused_work(d_work, cap), d_u, d_i, d_j, d_meta, n_batch, user_block0, item_block0, bias_block0, s, a, reg,
                       d_loss64, SKR_LOSS_SLOTS, dbg, d_pre);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_bpr_fused_plan(const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n_batch, int k, int64_t user_block0,
                       int64_t item_block0, int64_t bias_block0, int64_t n_flat_blocks, void* d_scratch, int32_t* d_meta,
                       int32_t* d_slot_block, int32_t* d_slot_fin, int32_t* d_n_slots, void* stream) {
    return skr_bpr_fused_plan2(d_u, d_i, d_j, n_batch, k, user_block0, item_block0, bias_block0, n_flat_blocks, d_scratch, d_meta,
                               d_slot_block, d_slot_fin, d_n_slots, nullptr, 0, stream);
}

int skr_bpr_fused_plan2(const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n_batch, int k, int64_t user_block0,
                        int64_t item_block0, int64_t bias_block0, int64_t n_flat_blocks, void* d_scratch, int32_t* d_meta,
                        int32_t* d_slot_block, int32_t* d_slot_fin, int32_t* d_n_slots, const int32_t* d_tag_prev,
                        int32_t tag_prev_value, void* stream) {
    SKR_REQUIRE(d_u && d_i && d_j && d_scratch && d_meta && d_slot_block && d_slot_fin && d_n_slots, "skr_bpr_fused_plan: NULL argument");
    SKR_REQUIRE(n_batch >= 1 && k >= 1 && k <= AB_KMAX && static_cast<int64_t>(k) * 5 * n_batch <= (1 << FUSED_SLOT_BITS),
                "skr_bpr_fused_plan: need 1 <= k <= %d and k * 5 * n_batch <= 2^%d", AB_KMAX, FUSED_SLOT_BITS);
    SKR_REQUIRE(user_block0 >= 0 && item_block0 >= 0 && bias_block0 >= 0 && n_flat_blocks >= 1 && n_flat_blocks < INT32_MAX,
                "skr_bpr_fused_plan: bad table offsets");
    SKR_REQUIRE((reinterpret_cast<uintptr_t>(d_scratch) & 7) == 0, "skr_bpr_fused_plan: scratch must be 8-byte aligned");
    FusedPlanArgs a;
    a.u = d_u, a.i = d_i, a.j = d_j;
    a.k = k, a.b = n_batch;
    a.ublk0 = user_block0, a.iblk0 = item_block0, a.bblk0 = bias_block0;
    a.named = static_cast<unsigned long long*>(d_scratch);
    a.claimed = a.named + n_flat_blocks;
    a.shared = a.claimed + n_flat_blocks;
    a.slot_of = reinterpret_cast<int32_t*>(a.shared + n_flat_blocks);
    a.meta = d_meta, a.slot_block = d_slot_block, a.slot_fin = d_slot_fin, a.n_slots = d_n_slots;
    a.tag_prev = d_tag_prev, a.tag_prev_value = tag_prev_value;
    const dim3 grid(static_cast<unsigned>((static_cast<int64_t>(k) * n_batch + 255) / 256)), wg(256);
    hipLaunchKernelGGL(fused_plan_kernel<0>, grid, wg, 0, skr::as_stream(stream), a);
    hipLaunchKernelGGL(fused_plan_kernel<1>, grid, wg, 0, skr::as_stream(stream), a);
    hipLaunchKernelGGL(fused_plan_kernel<2>, grid, wg, 0, skr::as_stream(stream), a);
    hipLaunchKernelGGL(fused_plan_kernel<3>, grid, wg, 0, skr::as_stream(stream), a);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_bpr_fused_pre(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_pre, int64_t cap,
                      const int32_t* d_slot_block, const int32_t* d_slot_fin, const int32_t* d_n_slots, float lr, float beta1,
                      float beta2, float eps, int64_t step_t0, int k, const int32_t* d_tag_prev, int32_t tag_prev_value,
                      void* stream) {
    SKR_REQUIRE(d_p && d_m && d_v && d_pre && d_slot_block && d_slot_fin && d_n_slots && d_tag_prev, "skr_bpr_fused_pre: NULL argument");
    SKR_REQUIRE(n >= 0 && cap >= 1 && cap <= (1 << FUSED_SLOT_BITS), "skr_bpr_fused_pre: need 1 <= cap <= 2^%d", FUSED_SLOT_BITS);
    SKR_REQUIRE(step_t0 >= 0 && k >= 1 && k <= AB_KMAX, "skr_bpr_fused_pre: need 1 <= k <= %d", AB_KMAX);
    // (scalars of its own: this runs on another stream than the block's steps, one block ahead of them)
    AdamBlockArgs a{};
    a.one_minus_b1 = static_cast<float>(1.0 - static_cast<double>(beta1));
    a.b2 = beta2;
    a.one_minus_b2 = static_cast<float>(1.0 - static_cast<double>(beta2));
    a.eps = eps;
    a.k = k;
    adam_block_scalars(a, lr, beta1, beta2, step_t0, k, false);
    adam_block_thresholds(a, lr, beta1, beta2, eps, k);
    hipLaunchKernelGGL(bpr_fused_pre_kernel, dim3(static_cast<unsigned>((cap + 3) / 4)), dim3(256), 0, skr::as_stream(stream), d_p, d_m,
                       d_v, n, d_pre, cap, d_slot_block, d_slot_fin, d_n_slots, a, d_tag_prev, tag_prev_value);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_bpr_fused_end(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_slot_block,
                      const int32_t* d_slot_fin, const int32_t* d_n_slots, float lr, float beta1, float beta2, float eps,
                      int64_t step_t0, int k, const int32_t* d_tag_next, int32_t tag_next_value, int which, void* stream) {
    SKR_REQUIRE(d_p && d_m && d_v && d_work && d_slot_block && d_slot_fin && d_n_slots, "skr_bpr_fused_end: NULL argument");
    SKR_REQUIRE(n >= 0 && cap >= 1 && cap <= (1 << FUSED_SLOT_BITS), "skr_bpr_fused_end: need 1 <= cap <= 2^%d", FUSED_SLOT_BITS);
    SKR_REQUIRE(step_t0 >= 0 && k >= 1 && k <= AB_KMAX, "skr_bpr_fused_end: need 1 <= k <= %d", AB_KMAX);
    SKR_REQUIRE(which >= 0 && which <= 2 && (which == 0 || d_tag_next), "skr_bpr_fused_end: which must be 0, or 1 / 2 with the next block's tags");
    const AdamBlockArgs& a = fused_block_args(lr, beta1, beta2, eps, step_t0, k);
    hipLaunchKernelGGL(bpr_fused_end_kernel, dim3(static_cast<unsigned>((cap + 3) / 4)), dim3(256), 0, skr::as_stream(stream), d_p,
                       d_m, d_v, n, fused_work(d_work, cap), d_slot_block, d_slot_fin, d_n_slots, a, d_tag_next, tag_next_value, which);
    SKR_LAUNCH_CHECK();
    return SKR_OK;
}

int skr_bpr_fused_block(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_u,
                        const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch, int64_t user_block0,
                        int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2, float eps, int64_t step_t0,
                        int k, float reg, float* d_loss64, int64_t loss_stride_floats, const int32_t* d_slot_block,
                        const int32_t* d_slot_fin, const int32_t* d_n_slots, const int32_t* d_tag_next, int32_t tag_next_value,
                        void* stream) {
    return skr_bpr_fused_block2(d_p, d_m, d_v, n, d_work, cap, d_u, d_i, d_j, d_meta, n_batch, user_block0, item_block0, bias_block0, lr,
                                beta1, beta2, eps, step_t0, k, reg, d_loss64, loss_stride_floats, d_slot_block, d_slot_fin, d_n_slots,
                                d_tag_next, tag_next_value, nullptr, stream);
}

int skr_bpr_fused_block2(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_u,
                         const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch, int64_t user_block0,
                         int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2, float eps, int64_t step_t0,
                         int k, float reg, float* d_loss64, int64_t loss_stride_floats, const int32_t* d_slot_block,
                         const int32_t* d_slot_fin, const int32_t* d_n_slots, const int32_t* d_tag_next, int32_t tag_next_value,
                         const float* d_pre, void* stream) {
    SKR_REQUIRE(k >= 1 && k <= AB_KMAX && n_batch >= 0 && loss_stride_floats >= 0, "skr_bpr_fused_block: bad shape");
    for (int s = 0; s < k; ++s) {
        const int64_t o = static_cast<int64_t>(s) * n_batch;
        const int rc = skr_bpr_fused_step2(d_p, d_m, d_v, n, d_work, cap, d_u + o, d_i + o, d_j + o, d_meta + 5 * o, n_batch, user_block0,
                                           item_block0, bias_block0, lr, beta1, beta2, eps, step_t0, k, s, reg,
                                           d_loss64 + s * loss_stride_floats, d_pre, stream);
        if (rc != SKR_OK) return rc;
    }
    return skr_bpr_fused_end(d_p, d_m, d_v, n, d_work, cap, d_slot_block, d_slot_fin, d_n_slots, lr, beta1, beta2, eps, step_t0, k,
                             d_tag_next, tag_next_value, d_tag_next ? 1 : 0, stream);
}

int skr_selftest_cold_math(uint64_t n_pairs, uint64_t* h_mismatches, void* stream) {
    SKR_REQUIRE(h_mismatches, "skr_selftest_cold_math: NULL argument");
    unsigned long long* d_bad = nullptr;
    SKR_HIP(hipMalloc(&d_bad, 4 * sizeof(unsigned long long)));
    SKR_HIP(hipMemsetAsync(d_bad, 0, 4 * sizeof(unsigned long long), skr::as_stream(stream)));
    // square root: every float in [2^-96, largest finite]
    hipLaunchKernelGGL(selftest_cold_math_kernel, dim3(256 * 8), dim3(256), 0, skr::as_stream(stream), 0x0f800000u, 0x7f7fffffu,
                       n_pairs, d_bad);
    SKR_LAUNCH_CHECK();
    unsigned long long h[4] = {0, 0, 0, 0};
    SKR_HIP(hipMemcpyAsync(h, d_bad, sizeof(h), hipMemcpyDeviceToHost, skr::as_stream(stream)));
    SKR_HIP(hipStreamSynchronize(skr::as_stream(stream)));
    (void)hipFree(d_bad);
    h_mismatches[0] = h[0];
    h_mismatches[1] = h[1];
    h_mismatches[2] = h[2];
    h_mismatches[3] = h[3];
    return SKR_OK;
}

}  // extern "C"
