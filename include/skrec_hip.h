/* skrec_hip.h -- C ABI of libskrec_hip.so, the MI355X (gfx950) implementation of
 * scikit-recommender's data-parallel hot path.
 *
 * Rules of the boundary
 *   - plain C: no C++ types, no exceptions, no torch types; every pointer named d_* is a DEVICE
 *     pointer (HBM), every other pointer is a host pointer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it
 *     and nothing synchronises the host unless the function says so;
 *   - caller-owned buffers, like the reference's native functions (randint.h:75, evaluate.h:57);
 *   - every function returns 0 on success or a negative skr_status; skr_last_error() gives the text.
 *     Where the reference would spin forever or crash on bad input (randint.h:38-48 with an
 *     exclusion set covering the range; metric_dict[] with an unknown id, evaluate.h:50) this
 *     library returns SKR_EINVAL instead.
 *   - integer widths follow the reference: item/user ids and sizes are 32-bit `int`
 *     (pyx_init.pyx:6-16 asserts it), CSR row pointers are int64.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * root, skrec/...).  INTEGRATION.md shows the reference-side binding for each.
 */
#ifndef SKREC_HIP_H
#define SKREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum skr_status {
    SKR_OK = 0,
    SKR_EINVAL = -1,   /* bad argument (the text says which) */
    SKR_EHIP = -2,     /* a HIP runtime call failed */
    SKR_ENOMEM = -3,   /* workspace too small / allocation failed */
    SKR_ENODEV = -4,   /* no gfx950 device visible */
    SKR_EOVERFLOW = -5 /* an internal bounded buffer overflowed (result invalid) */
} skr_status;

/* metric ids = skrec/utils/py/evaluator.py:57 and utils/py/cython/include/metric.h:112-118 */
enum { SKR_PRECISION = 1, SKR_RECALL = 2, SKR_MAP = 3, SKR_NDCG = 4, SKR_MRR = 5 };

int skr_abi_version(void);            /* bumped on every signature change */
const char* skr_last_error(void);     /* thread-local, valid until the next failing call */
int skr_device_count(void);           /* number of visible HIP devices (0 on a CPU-only host) */
int skr_device_summary(char* buf, size_t n); /* "gfx950 256CU ..." of the current device */
/* Longest row of a device CSR (host result; synchronises).  The host mirror uses it for the
 * reference's argument checks (pyx_random.pyx:49) and for the fused evaluator's precondition. */
int skr_csr_max_row_len(const int64_t* d_rowptr, int n_rows, int* out_max, void* stream);

/* ============================================================================================
 * S -- negative sampling
 * replaces: c_randint_choice / _random_int / global `std::mt19937 _gen(2020)`
 *           (utils/py/cython/include/randint.h:20-88), its Cython glue pyx_randint_choice
 *           (utils/py/cython/pyx_random.pyx:20-76) and the per-user Python loop
 *           _sampling_negative_items (io/data_iterator.py:81-94).
 * ========================================================================================== */
typedef struct skr_sampler skr_sampler; /* owns one MT19937 stream, resident in HBM */

/* seed 2020 reproduces the reference's process-global stream (randint.h:20). */
int skr_sampler_create(uint32_t seed, skr_sampler** out);
int skr_sampler_destroy(skr_sampler* s);
/* Raw state exchange (624 untempered words + position 0..624), host buffers; synchronises. */
int skr_sampler_get_state(skr_sampler* s, uint32_t* words624, int* pos);
int skr_sampler_set_state(skr_sampler* s, const uint32_t* words624, int pos);
/* Total raw 32-bit words consumed since creation / set_state (synchronises). */
int skr_sampler_draws(skr_sampler* s, uint64_t* n);
/* How the last exact epoch ran (test / measurement hook; synchronises the device): h_info4 = {status, handed_over,
 * slots filled, words consumed}.  status 0 = the one-workgroup path (dense data, small calls, SKR_EXACT_PATH=serial),
 * 1 = the slab path (sampler.hip 2d) and every slot was filled, 2 = slab path, the generated words ran out (the next
 * exact-epoch call reports it as an error).  handed_over 1 = the slab path met a case it leaves to the serial kernel and
 * that kernel finished the stream. */
int skr_sampler_last_epoch(skr_sampler* s, int64_t* h_info4);

/* One call of c_randint_choice (randint.h:75): `size` draws from [0, high), written to d_result.
 *   replace      as the reference's bool;
 *   d_prob       NULL (uniform, randint.h:84) or float[high] weights (discrete, randint.h:79);
 *   d_exclusion  NULL or int[n_exclusion] SORTED ascending, unique (replaces unordered_set<int>).
 * Runs the reference's serial algorithm on the device (one lane), bit-exact with the reference
 * stream; meant for the API-surface calls (random.py:9), not for epochs.  Validates like
 * pyx_random.pyx:34-54 and returns SKR_EINVAL where the reference raises ValueError. */
int skr_randint_choice(skr_sampler* s, int high, int size, int replace, const float* d_prob,
                       const int32_t* d_exclusion, int n_exclusion, int32_t* d_result, void* stream);

/* A whole epoch of _sampling_negative_items (data_iterator.py:81-94) in one call, EXACT STREAM:
 * for users 0..n_users-1 ascending, rows with no positives skipped, n_pos(u)*num_neg uniform draws
 * from [0, num_items) rejecting u's train positives -- consuming the sampler's MT19937 stream in
 * exactly the reference's order, so d_out is bit-identical to the reference's concatenated array
 * and the stream continues into the next epoch.
 *   d_rowptr      int64[n_users+1] CSR offsets of the train positives
 *   d_pos_sorted  int32[nnz] positives, ascending within each row (membership test only; the
 *                 pairing with positives in file order is the caller's, data_iterator.py:30-42)
 *   nnz           == rowptr[n_users] (the caller knows it; avoids a device read-back)
 *   d_out         int32[nnz*num_neg]  (== reshape [nnz, num_neg] when num_neg > 1, :91)
 * Synchronises the stream once at the end (it must learn how many words were consumed). */
int skr_sample_epoch_exact(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                           const int32_t* d_pos_sorted, int64_t nnz, int num_neg, int32_t* d_out, void* stream);
/* The same epoch without the read-back at the start of the call: the row statistics of the CSR (longest row, sum of squared
 * row lengths: h_stats2[0], [1]) are taken ONCE with skr_csr_row_stats (which waits for its own kernel) -- a data set's CSR does
 * not change between epochs (data_iterator.py:30-42 builds it in the constructor) -- and the call only queues work.  A failure of
 * the PREVIOUS epoch (stream exhausted) is reported from a status word that was copied to the host behind that epoch. */
int skr_csr_row_stats(const int64_t* d_rowptr, int n_rows, int64_t* h_stats2, void* stream);
int skr_sample_epoch_exact_stats(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr, const int32_t* d_pos_sorted,
                                 int64_t nnz, int num_neg, int32_t* d_out, const int64_t* h_stats2, void* stream);

/* The same exact-stream epoch when the number of draws of a user is NOT the size of its exclusion set:
 * the sequential iterators (data_iterator.py:237-331: one draw group per training sequence, exclusion =
 * the user's whole history, `user_n_pos` from _generative_time_order_positive_items :44-78) and the
 * knowledge-graph iterator (_sampling_negative_tails :406-420: one group per triple of a head,
 * exclusion = its distinct tails).  Users ascending, users with no draws skipped, exactly as the
 * reference's loop over `user_n_pos.items()`.
 *   d_rowptr / d_excl_sorted  CSR of the exclusion sets (ascending within a row), nnz entries
 *   d_drawptr                 int64[n_users+1] cumulative number of draws; n_draws == d_drawptr[n_users]
 *   d_out                     int32[n_draws] */
int skr_sample_epoch_exact_counts(skr_sampler* s, int num_items, int n_users, const int64_t* d_rowptr,
                                  const int32_t* d_excl_sorted, int64_t nnz, const int64_t* d_drawptr, int64_t n_draws,
                                  int32_t* d_out, void* stream);

/* The same distribution, embarrassingly parallel: slot a of user u in epoch e draws from a
 * xoshiro128++ stream keyed by (seed, epoch, global slot index), Lemire-mapped to [0, num_items),
 * retried until not in u's positives.  Bit-exact with its CPU twin (tests/fast_sampler_twin.py),
 * equal to the reference in law only; independent of how users are sharded over GPUs when
 * slot_offset is the global index of this shard's first slot.  Never synchronises.  A row that
 * covers the whole catalogue (rejected with ValueError by pyx_random.pyx:49; the host mirror checks
 * it) yields -1 instead of spinning. */
int skr_sample_epoch_fast(uint64_t seed, uint64_t epoch, int64_t slot_offset, int num_items, int n_users,
                          const int64_t* d_rowptr, const int32_t* d_pos_sorted, int64_t nnz, int num_neg,
                          int32_t* d_out, void* stream);

/* ============================================================================================
 * E -- full-catalogue top-K evaluation
 * replaces: cpp_evaluate_matrix / eval_one_user (utils/py/cython/include/evaluate.h:24-76),
 *           the five metric functions (include/metric.h:19-118), eval_score_matrix
 *           (utils/py/cython/pyx_eval_matrix.pyx:22-37), and -- in the fused form -- also
 *           _MF.predict / _LightGCN.predict's U[b] @ V.T (+bias) (recommender/BPRMF.py:84-88,
 *           LightGCN.py:102-107) and the -inf train masking loop (utils/py/evaluator.py:197-200).
 * Equal scores: skr_eval_scores reproduces the reference's order, which is whatever libstdc++'s
 * partial_sort_copy heap leaves (evaluate.h:42-43) -- rows with ties among their best K+1 scores are
 * re-ranked by that very algorithm.  The fused form ranks equal scores by ascending item id (its fp32
 * summation order differs from any host GEMM's anyway, so score ties are not reproducible there).
 * ========================================================================================== */

/* Drop-in for cpp_evaluate_matrix (evaluate.h:57): d_scores [n_users, ld] row-major fp32 (first
 * n_items columns used), already train-masked by the caller.
 *   d_test_rowptr/d_test_items  CSR of each row's ground truth, items SORTED ascending, unique
 *   metric[n_metric]            host array of metric ids 1..5, output is metric-major
 *   d_rows      float[n_users, n_metric*top_k]   per-user metric rows (may be NULL)
 *   d_topk_ids  int32[n_users, top_k]            the arg-top-K lists (may be NULL)
 *   d_sums      double[n_metric*top_k]           += column sums over users (may be NULL)
 * Requires 1 <= top_k <= min(n_items, SKR_MAX_TOPK). */
#define SKR_MAX_TOPK 128
/* skr_eval_scores and skr_rank_metrics (the score-matrix path) rank deeper: the reference's top_k is a free integer
 * (run_config.py:16; its default (10, ..., 100) fits the fused kernel too) */
#define SKR_MAX_TOPK_SCORES 512
int skr_eval_scores(const float* d_scores, int n_users, int n_items, int64_t ld,
                    const int64_t* d_test_rowptr, const int32_t* d_test_items,
                    const int* metric, int n_metric, int top_k,
                    float* d_rows, int32_t* d_topk_ids, double* d_sums, void* stream);

/* Fused scoring + masking + top-K, no [B, I] matrix in HBM:
 *   score(b, i) = dot(d_user_table[d_users[b]], d_item_table[i]) (+ d_item_bias[i] if non-NULL),
 *   train positives of user d_users[b] excluded, K best (score desc, id asc) written per row.
 *   d_user_table [*, dim], d_item_table [n_items, dim] fp32 row-major, dim == 64
 *   d_users      int32[B] user ids (also index the train CSR rows)
 *   d_train_rowptr/d_train_items  CSR over ALL users, items sorted ascending (may be NULL: no mask)
 *   d_topk_ids int32[B, top_k], d_topk_scores float[B, top_k] (may be NULL)
 *   d_work / work_bytes   scratch from skr_eval_fused_workspace(B, top_k)
 * Arithmetic (environment SKR_FUSED_MODE, read per call).  "f16x2" (default): every operand, scaled by a power of two
 * per table, is split into TWO fp16 pieces and a product formed from three fp16 MFMA products with fp32 accumulation;
 * fp32-level accuracy (measured error vs float64 below the plain chain's) holds while the scores that decide a list
 * lie well above an absolute floor set by the largest elements of the two tables -- checked per user on the device,
 * and every user that fails is recomputed by the bf16x3 kernel inside the same call (skr_eval_fused_rejected counts
 * them).  "bf16x3": three bf16 pieces, six MFMA products, no condition on the operands (fused_topk_kernel_v6, 16-item
 * steps on v_mfma_f32_16x16x32_bf16).
 * "fp32": exact fp32 FMA chains on the FP32 MFMA.  The split modes keep library-owned device buffers for the split
 * item table (n_items*64*6 bytes, f16x2: + n_items*64*4), one set per process: calls in a split mode must be issued
 * on ONE stream at a time (calls on the same stream queue behind each other, which is what the evaluator does).
 * Requires n_items - max train row length >= top_k (else SKR_EINVAL: use skr_eval_scores). */
size_t skr_eval_fused_workspace(int B, int top_k);
/* SKR_FUSED_MODE=f16x2 only: how many rows of the LAST skr_eval_fused_topk call on `stream` its guard did not accept and
 * handed to the bf16x3 kernel (written to *h_count on the host; blocks until that call is done; 0 in the other modes). */
int skr_eval_fused_rejected(int32_t* h_count, void* stream);
int skr_eval_fused_topk(const float* d_user_table, const int32_t* d_users, int B,
                        const float* d_item_table, const float* d_item_bias, int n_items, int dim,
                        const int64_t* d_train_rowptr, const int32_t* d_train_items,
                        int top_k, int32_t* d_topk_ids, float* d_topk_scores,
                        void* d_work, size_t work_bytes, void* stream);

/* Dense score matrix of the predict() API surface (_MF.predict, BPRMF.py:84-88; _LightGCN.predict,
 * LightGCN.py:102-107): d_scores[b, i] = <user_table[d_users[b]], item_table[i]> (+ d_item_bias[i]),
 * fp32 fma chain in ascending k, bias added last.  The evaluator does not use it (it never needs
 * the matrix); it exists so that predict() has no torch kernel behind it.  dim == 64. */
int skr_score_matrix(const float* d_user_table, const int32_t* d_users, int B, const float* d_item_table,
                     const float* d_item_bias, int n_items, int dim, float* d_scores, int64_t ld, void* stream);

/* evaluator.py:197-200 on the device: d_scores[b, i] = -inf for every train item i of user
 * d_users[b] (generic path, when a foreign model hands over a dense [B, n_items] score matrix). */
int skr_mask_train(float* d_scores, int B, int n_items, int64_t ld, const int32_t* d_users,
                   const int64_t* d_train_rowptr, const int32_t* d_train_items, void* stream);

/* Metric rows from arg-top-K lists (metric.h:19-109); truth row of list b is d_truth_rows[b]
 * (NULL: row b).  Same outputs as skr_eval_scores. */
int skr_rank_metrics(const int32_t* d_topk_ids, int B, int top_k, const int32_t* d_truth_rows,
                     const int64_t* d_test_rowptr, const int32_t* d_test_items,
                     const int* metric, int n_metric, float* d_rows, double* d_sums, void* stream);

/* ============================================================================================
 * T -- BPR lookup-and-score forward/backward, optimiser, graph propagation
 * replaces stock torch ops of the reference (no native counterpart there):
 *   _MF.forward + bpr_loss + l2_loss + autograd backward   recommender/BPRMF.py:77-82,114-126,
 *                                                           utils/torch.py:20-21,62-74
 *   torch.optim.Adam (dense, every row every step)          BPRMF.py:99,127
 *   torch.sparse.mm(norm_adj, E) per layer, layer mean      LightGCN.py:89-100
 *   cosine re-weighting, layer sum                          LayerGCN.py:207-220
 * All tables fp32 row-major with dim == 64.
 * ========================================================================================== */

/* One BPR batch, forward + backward, gradients ACCUMULATED (atomically) into dense buffers that
 * the caller zeroed:  x = <P_u,Q_i> + b_i - <P_u,Q_j> - b_j ;  loss_b = softplus(-x)
 *   d_loss[0] += sum_b loss_b * loss_scale      (BPRMF: 1, sum, BPRMF.py:117; LightGCN: 1/n, mean)
 *   d_loss[1] += 0.5*sum of squares of the gathered REG rows (l2_loss, torch.py:66-74)
 *   grads of the score part go to d_gP/d_gQ/d_gb (tables the scores were computed from);
 *   reg * reg_scale * row goes to d_gRP/d_gRQ (+ d_gb for the bias), computed from d_RP/d_RQ --
 *   for BPRMF these are the same tables (BPRMF.py:118-124); for LightGCN/LayerGCN the scores use
 *   the propagated tables and the regulariser the ego tables (LightGCN.py:192-196).
 *   d_bias / d_gb may be NULL (no item bias).
 *   d_touch (may be NULL): one byte per 64-float block of the gradient allocation that starts at
 *   d_touch_base; every block this call adds into gets its byte set to 1, so that skr_adam_step
 *   can skip reading (and re-zeroing) the gradient of blocks nobody touched. */
int skr_bpr_step(const float* d_P, const float* d_Q, const float* d_bias,
                 const float* d_RP, const float* d_RQ,
                 const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n,
                 float loss_scale, float reg, float reg_scale,
                 float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ,
                 float* d_loss, uint8_t* d_touch, const float* d_touch_base, void* stream);
/* skr_bpr_step on one rank of a user-sharded job: d_u / d_i / d_j hold a GLOBAL batch of n triples with
 * GLOBAL user ids; only the triples whose user this rank owns (u % shard_world == shard_rank) are processed, against row
 * u / shard_world of the rank's local user tables.  No selection on the host, no read-back of a count.
 * grad_scale multiplies the score part of the gradient only (d_gP / d_gQ / d_gb), not the loss sums nor the regulariser's
 * gradient: LightGCN passes 1 / (n_layers + 1), the factor of the layer mean (LightGCN.py:97-99), instead of scaling the
 * [N, 64] gradient buffer in a pass of its own. */
int skr_bpr_step_sharded(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                         const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, float loss_scale, float reg,
                         float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ, float* d_loss,
                         uint8_t* d_touch, const float* d_touch_base, int shard_world, int shard_rank, float grad_scale,
                         void* stream);
/* skr_bpr_step with the two loss sums spread over SKR_LOSS_SLOTS pairs: d_loss64 is float[2 * SKR_LOSS_SLOTS], pair q
 * receives the sums of the workgroups q, q + SKR_LOSS_SLOTS, ...; the batch's sums are the sums over the pairs.  (One
 * pair of words for all workgroups makes their atomics serialise: 4.5 us of a 10.9 us launch at batch 1024.) */
#define SKR_LOSS_SLOTS 32
int skr_bpr_step_spread(const float* d_P, const float* d_Q, const float* d_bias,
                 const float* d_RP, const float* d_RQ,
                 const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n,
                 float loss_scale, float reg, float reg_scale,
                 float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ,
                 float* d_loss64, uint8_t* d_touch, const float* d_touch_base, void* stream);
/* The general form: rows of `dim` floats (64, 128, 192 or 256; an embedding width that is not a multiple of 64 -- n_dim /
 * embed_size are free integers in the reference, BPRMF.py:27,51, LightGCN.py:34 -- is zero-padded by the caller: padded
 * columns have zero gradients and stay zero under Adam), loss_slots 1 (d_loss[2]) or SKR_LOSS_SLOTS (d_loss[2 * slots]),
 * grad_scale as in skr_bpr_step_sharded. */
int skr_bpr_step_dim(const float* d_P, const float* d_Q, const float* d_bias, const float* d_RP, const float* d_RQ,
                     const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n, int dim, float loss_scale,
                     float reg, float reg_scale, float* d_gP, float* d_gQ, float* d_gb, float* d_gRP, float* d_gRQ,
                     float* d_loss, int loss_slots, uint8_t* d_touch, const float* d_touch_base, float grad_scale,
                     void* stream);

/* torch.optim.Adam.step for one dense parameter (single-tensor path): for every element
 *   m = m + (g-m)*(1-b1);  v = v*b2 + (1-b2)*g*g;
 *   p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * step_t is the 1-based step count.  If zero_grad != 0 the gradient buffer is zeroed in the same
 * pass (the next step's optimizer.zero_grad(), BPRMF.py:125).
 * d_touch (may be NULL): one byte per 64-float block of d_g.  0 = the block's gradient is known to
 * be all zero (it is neither read nor written: 24 instead of 32 bytes of traffic per parameter,
 * same result bit for bit); 1 = read it, then clear the byte; 2 = always read, never cleared. */
int skr_adam_step(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n,
                  float lr, float beta1, float beta2, float eps, int64_t step_t, int zero_grad,
                  uint8_t* d_touch, void* stream);

/* Y = A * X for a CSR matrix with fp32 values and dim == 64 (torch.sparse.mm, LightGCN.py:94);
 *   optional fused epilogues:  Y += d_addend (same shape, may be NULL);
 *                              d_accum += accum_scale * Y (layer mean/sum, may be NULL). */
int skr_csr_spmm(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val,
                 const float* d_X, int dim, int64_t nnz, const float* d_addend, float* d_Y,
                 float* d_accum, float accum_scale, void* stream);

/* The same with a row stride: X, addend, Y and accum are 64-column slices of [*, ld] tables (ld >= 64 floats) -- a wider
 * embedding is multiplied slice by slice (pointers advanced by 64 c), the product being separable in the columns. */
int skr_csr_spmm_strided(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val,
                         const float* d_X, int dim, int ld, int64_t nnz, const float* d_addend, float* d_Y,
                         float* d_accum, float accum_scale, void* stream);

/* The same product through a per-matrix PLAN (csrc/spmm.hip): rows shorter than `long_rows_from` entries are gathered
 * one wavefront per row with 16 bytes per lane; longer rows are cut into tasks of <= 256 entries inside one block of
 * 16 384 columns, run block by block on the workgroups that share an XCD's L2, and their partial rows are added in a
 * fixed order (no float atomics: results do not depend on timing).  Same epilogues as skr_csr_spmm.
 *   skr_spmm_plan_create  analyses the CSR once (device work, synchronises the stream twice); the plan keeps the three
 *                         CSR pointers -- the arrays must outlive it and keep their contents -- plus a task list and a
 *                         scratch buffer of its own.  Columns must ascend within a row and be < n_cols.
 *                         long_rows_from: 0 = default (512), otherwise >= 2.
 *   skr_spmm_plan_info    h_info4 = {long rows + (hot rows << 32), tasks, column blocks, long_rows_from + (column windows << 32)};
 *                         hot rows: the densest long rows (>= SKR_SPMM_HOT_DENSITY, default 4, entries per 128 columns on
 *                         average; at most 96) are not gathered at all in calls without d_col_mask -- X is streamed through
 *                         LDS in blocks of 128 rows and their entries, re-packed block-major at plan time, read it there
 *                         (SKR_SPMM_HOT=0: off)
 *                         SKR_SPMM_WINDOWS=n (default 1) makes the short-row kernel gather from X in n column windows, one
 *                         launch each (an experiment switch: no gain measured at X = 256 MB)
 * A plan may be run any number of times, by one stream at a time (the scratch buffer is shared between runs). */
typedef struct skr_spmm_plan skr_spmm_plan;
int skr_spmm_plan_create(int n_rows, int n_cols, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val,
                         int64_t nnz, int long_rows_from, skr_spmm_plan** out, void* stream);
int skr_spmm_plan_run(const skr_spmm_plan* plan, const float* d_X, int dim, const float* d_addend, float* d_Y,
                      float* d_accum, float accum_scale, void* stream);
/* The same product where parts of it are known not to be needed (LightGCN's step: only the batch's rows of the LAST
 * forward layer are read, and the gradient that enters the FIRST backward hop is zero outside the batch's rows):
 *   d_row_mask  uint8[n_rows] or NULL: rows with a 0 byte are skipped entirely (their Y / accum rows are left as they are)
 *   d_col_mask  uint8[n_cols] or NULL: entries whose column has a 0 byte are skipped -- the caller guarantees that those
 *               rows of X are zero, so the result is the full product's (up to the sign of a zero)
 * skr_mark_ids sets d_mask[offset + ids[k]] = 1 (negative ids skipped); clear the mask first. */
int skr_spmm_plan_run_masked(const skr_spmm_plan* plan, const float* d_X, int dim, const float* d_addend, float* d_Y,
                             float* d_accum, float accum_scale, const uint8_t* d_row_mask, const uint8_t* d_col_mask,
                             void* stream);
int skr_mark_ids(const int32_t* d_ids, int64_t n, int64_t offset, uint8_t* d_mask, void* stream);
/* The product with the TRANSPOSE of a CSR matrix A [n_rows, *] where X [n_rows, 64] is zero outside the rows marked in
 * d_row_mask:  Y[c] += A[r, c] * X[r]  for every entry of every marked row (Y initialised by the caller: zeros, or the
 * addend).  The item side of LightGCN's FIRST backward hop is this: dL/dE-bar is zero outside the batch's <= batch users, so
 * instead of every item row scanning its users for marked ones (48 M entries looked at) the batch users' own rows -- the
 * user-side block, whose entries are the same non-zeros -- are walked (~50 k entries) with one 256-byte row atomic each.
 * Float atomics: the order of the additions into a row varies from run to run, as it does in skr_bpr_step's scatter. */
int skr_csr_scatter_marked_rows(int n_rows, const int64_t* d_rowptr, const int32_t* d_col, const float* d_val,
                                const uint8_t* d_row_mask, const float* d_X, int dim, float* d_Y, void* stream);
/* The general form: what happens to a finished row y = (A X)_r (+ addend_r) is described by a skr_spmm_epilogue, so that the
 * row-local passes around a propagation ride in the product's row epilogue instead of being launches of their own (a
 * wavefront holds the whole 64-float row there):
 *   SKR_EPI_PLAIN       Y_r = y (Y may be NULL when only accum is wanted); accum_r = accum_r + accum_scale * y, or, with
 *                       accum_base, accum_r = accum_scale * accum_base_r + accum_scale * y (LightGCN's layer mean including
 *                       its E0 term: LightGCN.py:89-100), or, with accum_init, accum_r = accum_scale * y (accum is not read);
 *   SKR_EPI_REFINE_FWD  LayerGCN.py:214-216 on the finished row: w_r = cos(y, E_r) (torch eps 1e-8) -> w[r]; Z_r = w_r * y;
 *                       Y_r = y (the raw row the backward needs; Y may be NULL for inference); accum_r += Z_r (accum_init:
 *                       accum_r = Z_r);
 *   SKR_EPI_REFINE_BWD  the finished row is dZ of the layer below (hop k+1 of the backward feeds layer k's refinement):
 *                       Y_r = dY_r = the backward of that refinement at (rawY_r, E_r, w[r]); dE_r += its E0 part
 *                       (skr_layer_refine_bwd's arithmetic).
 * Rows skipped by d_row_mask get no epilogue. */
enum { SKR_EPI_PLAIN = 0, SKR_EPI_REFINE_FWD = 1, SKR_EPI_REFINE_BWD = 2 };
typedef struct skr_spmm_epilogue {
    int32_t mode;
    int32_t accum_init;
    const float* addend;
    float* Y;
    float* accum;
    const float* accum_base;
    float accum_scale;
    int32_t ld;                   /* row stride, in floats, of X and of every dense operand named here; 0 = 64.  With ld = 64 C
                                     and the pointers advanced by 64 c the call multiplies the c-th 64-column slice of [n, 64 C]
                                     tables (embedding widths beyond 64: the product is separable in the columns);
                                     SKR_EPI_PLAIN only */
    const float* E;
    float* w;
    float* Z;
    const float* rawY;
    float* dE;
    const uint8_t* accum_mask;    /* uint8[n_rows] or NULL: accum_r is only updated where the byte is set (a training step reads
                                     the layer mean / sum at its batch's rows only) */
    const uint8_t* addend_mask;   /* uint8[n_rows] or NULL: addend_r IS zero where the byte is 0 and is not read there (dL/d output
                                     is zero outside the batch's rows) */
} skr_spmm_epilogue;
int skr_spmm_plan_run_ex(const skr_spmm_plan* plan, const float* d_X, int dim, const skr_spmm_epilogue* epi,
                         const uint8_t* d_row_mask, const uint8_t* d_col_mask, void* stream);
int skr_spmm_plan_info(const skr_spmm_plan* plan, int64_t* h_info4);
int skr_spmm_plan_destroy(skr_spmm_plan* plan);

/* LayerGCN layer refinement (LayerGCN.py:214-216): w_r = cos(Y_r, E_r) (torch eps 1e-8),
 * Z_r = w_r * Y_r; d_accum += Z (may be NULL); d_w[n_rows] keeps w for the backward. */
int skr_layer_refine_fwd(const float* d_Y, const float* d_E, int64_t n_rows, int dim,
                         float* d_Z, float* d_w, float* d_accum, void* stream);
/* Backward of the above: given dZ -> dY (written) and dE (accumulated). */
int skr_layer_refine_bwd(const float* d_Y, const float* d_E, const float* d_w, const float* d_dZ,
                         int64_t n_rows, int dim, float* d_dY, float* d_dE, void* stream);

/* The same restricted to the rows whose byte in d_row_mask (uint8[n_rows], NULL = all) is set: where dZ_r is known to be
 * zero (LayerGCN's first backward refinement: dL/d out is zero outside the batch's rows) dY_r = 0 and dE_r gets nothing, so
 * the row is skipped.  zero_skipped != 0: dY_r = 0 is written for the skipped rows; 0: dY_r is left as it is (the product
 * that follows must then skip those columns: d_col_mask). */
int skr_layer_refine_bwd_masked(const float* d_Y, const float* d_E, const float* d_w, const float* d_dZ, int64_t n_rows,
                                int dim, float* d_dY, float* d_dE, const uint8_t* d_row_mask, int zero_skipped, void* stream);
/* Rows of d_table [n_rows, dim] whose byte in d_mask is set are zeroed; with clear_mask != 0 the bytes are cleared too.
 * Restores the all-zero state of a gradient buffer of which only one batch's rows were written (instead of a fill of
 * the whole buffer; no reference counterpart: autograd allocates a fresh zero tensor per step). */
int skr_clear_marked_rows(uint8_t* d_mask, int64_t n_rows, int64_t clear_mask, float* d_table, int dim, void* stream);

/* Row gather out[k] = table[idx[k]] (F.embedding; any row width `dim`, 64 has its own kernel) and  y = a*x + y  helpers
 * used by the host mirror so that no torch kernel sits on the hot path. */
int skr_gather_rows(const float* d_table, const int32_t* d_idx, int64_t n, int dim, float* d_out, void* stream);
/* table[idx[k]] = src[k] (negative ids skipped; rows of duplicate ids must be identical): with skr_gather_rows, the
 * compact form in which the user-sharded engines exchange the FEW rows of a replicated block a step touches */
int skr_scatter_rows(const float* d_src, const int32_t* d_idx, int64_t n, int dim, float* d_table, void* stream);
/* d_out[i] = ((d_in[0][i] + d_in[1][i]) + d_in[2][i]) + ... over n_blocks blocks of n floats: the ranks' all-gathered
 * blocks added in rank order, so that every replica ends with the same bits whatever the collective library does */
int skr_sum_blocks(const float* d_in, int n_blocks, int64_t n, float* d_out, void* stream);
int skr_axpy(float a, const float* d_x, float* d_y, int64_t n, void* stream);
int skr_scale_copy(float a, const float* d_x, float* d_y, int64_t n, void* stream);    /* y = a * x */
int skr_scale(float a, float* d_x, int64_t n, void* stream);                 /* x *= a */

/* ---------------------------------------------------------------------------------------------
 * Epoch shuffle + batch assembly.  replaces: BatchIterator's per-element Python batching
 * (utils/py/batch_iterator.py:48-66,132-155: idx = np.random.permutation(n); [data[i] for i in idx] per column) and
 * the zip of the users / pos / neg columns (io/data_iterator.py:226-234).
 * ONE launch per epoch: for r < n_out, row r of every output column = row src(r) of its input column, with
 *   src(r) = d_perm[r]     when d_perm != NULL (int32[>= n_out], values in [0, n_src): the numpy-compatible contract --
 *                          the values of the epoch's np.random.permutation(n), e.g. from skr_host_permutation), or
 *   src(r) = pi_seed(r)    when d_perm == NULL: a keyed bijection of [0, n_src) evaluated in registers (no permutation
 *                          array, no sort; equal to the reference's shuffle in law only).
 * d_cols / d_outs: HOST arrays of n_cols (1..4) DEVICE pointers; widths[k] = 32-bit words per row of column k (1 for an
 * id or label column, num_neg for a [n, num_neg] block).  Input and output must not alias.  n_out <= n_src < 2^31. */
int skr_shuffle_gather(const int32_t* d_perm, uint64_t seed, int64_t n_src, int64_t n_out, int n_cols,
                       const void* const* d_cols, const int* widths, void* const* d_outs, void* stream);
/* pi_seed(0 .. n_out-1) itself (int32), on the device / on the host (the host form needs no GPU) */
int skr_shuffle_permutation(uint64_t seed, int64_t n, int64_t n_out, int32_t* d_out, void* stream);
int skr_shuffle_permutation_host(uint64_t seed, int64_t n, int64_t n_out, int32_t* h_out);

/* HOST function (no GPU): np.random.permutation(n) of numpy's legacy global generator, restated natively.  The
 * reference shuffles each epoch with it (skrec/io/batch_iterator.py:61-63).  key624 / pos: the generator's MT19937 state
 * as np.random.get_state() returns it (uint32[624], position 0..624); both are advanced exactly as numpy advances
 * them, so np.random.set_state() with the returned values leaves the generator where the reference's would be.
 * h_out int32[n], n < 2^31.  Runs outside the GIL when called through ctypes. */
int skr_host_permutation(uint32_t* h_key624, int* h_pos, int64_t n, int32_t* h_out);

/* The same dense Adam, temporally blocked over k consecutive steps whose batches are known in advance (an epoch's
 * batches are: data_iterator.py:230-234).  64-float blocks of the flat buffer that none of the k steps touches get
 * their k zero-gradient updates in ONE pass; touched blocks get the ordinary update at every step.  Every
 * parameter receives every update in the arithmetic of skr_adam_step: the results are bit-identical.
 *   skr_adam_block_mark   d_tag[(offset + id*stride) >> 6] = tag_value and d_claim[same] = step_t0 for every id (call
 *                         once per id list: user rows offset 0 stride 64, item rows offset U*64, bias offset
 *                         (U+I)*64 stride 1).  step_t0 = optimiser steps taken before the k-step block.
 *   skr_adam_block_cold   steps step_t0+1 .. step_t0+k with zero gradient on every block whose tag != hot_value
 *   skr_adam_block_hot    ADVANCES the blocks the ids name to step_t (step_t0 < step_t <= step_t0 + 64): a block that
 *                         d_claim says is at step c gets zero-gradient updates for steps c+1 .. step_t-1 and then
 *                         step_t's update with its accumulated gradient, which is read and cleared; d_claim becomes
 *                         step_t (duplicate ids: one wavefront wins).  Either name every hot block at every step, or
 *                         at step t only the rows of batch t and of batch t+1 (which must read current rows) and
 *                         every hot block at the LAST step of the k-step block -- all hot blocks must end at
 *                         step_t0 + k.  Both orders of visiting apply the same updates in the same order.
 *   d_tag, d_claim        int32[ceil(n / 64)], zero-initialised; use a fresh non-zero tag_value for every k-step block
 *   ids                   negative entries are skipped by _mark and _hot (empty slots of a de-duplicated list) */
int skr_adam_block_mark(const int32_t* d_ids, int64_t n_ids, int64_t offset_floats, int stride_floats, int32_t* d_tag,
                        int32_t tag_value, int32_t* d_claim, int64_t step_t0, void* stream);
int skr_adam_block_cold(float* d_p, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                        int64_t step_t0, int k, const int32_t* d_tag, int32_t hot_value, void* stream);
int skr_adam_block_hot(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                       float eps, int64_t step_t0, int64_t step_t, const int32_t* d_ids, int64_t n_ids, int64_t offset_floats,
                       int stride_floats, int32_t* d_claim, void* stream);
/* The same three entry points with tf.train.AdamOptimizer's arithmetic (GRU4RecPlus.py:192): p -= lr_t * m / (sqrt(v) + eps)
 * with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) -- the second bias correction sits in the step size, eps is added to
 * the raw sqrt(v).  |lr_t| falls and then rises again with t, so the at-rest test of a run of zero-gradient updates uses
 * the largest |lr_t| of the run.  Blocked == one skr_adam_step_tf per step, bit for bit. */
int skr_adam_step_tf(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                     int64_t step_t, int zero_grad, uint8_t* d_touch, void* stream);
int skr_adam_block_cold_tf(float* d_p, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                           int64_t step_t0, int k, const int32_t* d_tag, int32_t hot_value, void* stream);
int skr_adam_block_hot_tf(float* d_p, float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int64_t step_t0, int64_t step_t, const int32_t* d_ids, int64_t n_ids,
                          int64_t offset_floats, int stride_floats, int32_t* d_claim, void* stream);

/* One training step of BPRMF in ONE launch: skr_bpr_step (score rows == regulariser rows, loss_scale 1) and the hot rows'
 * part of the blocked dense Adam together, for a k-step block whose batches are known (replaces, per step, the pair
 * skr_bpr_step_spread + skr_adam_block_hot; BPRMF.py:108-127).  Hot rows are evaluated lazily: the wavefronts that read a
 * row first apply, in registers, the updates the row is behind (its pending gradient of the previous naming, then
 * zero-gradient updates -- the arithmetic of skr_adam_step, bit for bit), and a step's gradient waits in the block's
 * workspace until the row is named again or the block ends.  Per reference (step, row) the caller supplies one word
 *   d_meta[r * n_batch + b], r = 0..4: user row, positive item row, negative item row, the 64-word bias block of the
 *   positive item, of the negative item;   bits 0-19 slot of the row in the block's workspace (dense numbering 0 .. n_slots-1),
 *   bits 20-22 (number of EARLIER steps of the block that name the row) mod 6, bit 23 set on exactly one reference per
 *   (step, row), bits 24-30 1 + the step of the previous naming (0: none), bit 31 set when the reference is the only one
 *   of its (step, row) pair (its gradient is then stored, not added atomically)
 * -- skr_bpr_fused_plan writes them, and the slot tables of skr_bpr_fused_end, for the k batches of a block (d_u / d_i / d_j:
 * k * n_batch entries, step-major) in four small launches:  d_scratch = 28 * n_flat_blocks bytes, 8-byte aligned, ZERO before
 * the first call and left usable by every call (n_flat_blocks = ceil(n / 64));  d_meta int32[k * 5 * n_batch];
 * d_slot_block / d_slot_fin int32[k * 5 * n_batch] (d_slot_block: -1 beyond n_slots -- a valid id list for
 * skr_adam_block_mark);  d_n_slots int32[1].  Which slot a row gets may differ between calls; nothing depends on it.
 *   d_work   float[9 * cap * 64], ZERO before the first block and left as skr_bpr_fused_end leaves it; cap >= n_slots
 *   *_block0 index (in 64-float blocks of the flat buffer d_p) of user row 0, item row 0, bias word 0
 *   s        the step inside the block (0 .. k-1); the k launches of a block are followed by ONE skr_bpr_fused_end, which
 *            brings every slot (d_slot_block[slot] = its block of the flat buffer, d_slot_fin[slot] = (number of namings
 *            mod 6) | (step of the last naming << 8), *d_n_slots on the device) to step step_t0 + k and writes it back.
 * The cold blocks of the k-step block are skr_adam_block_cold's as before (tags from skr_adam_block_mark).
 * d_loss64: as skr_bpr_step_spread.  1 <= k <= 64. */
int skr_bpr_fused_plan(const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n_batch, int k, int64_t user_block0,
                       int64_t item_block0, int64_t bias_block0, int64_t n_flat_blocks, void* d_scratch, int32_t* d_meta,
                       int32_t* d_slot_block, int32_t* d_slot_fin, int32_t* d_n_slots, void* stream);
int skr_bpr_fused_step(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_work, int64_t cap,
                       const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch,
                       int64_t user_block0, int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2,
                       float eps, int64_t step_t0, int k, int s, float reg, float* d_loss64, void* stream);
/* skr_bpr_fused_end, `which`: 0 every slot; with the NEXT block's tags (d_tag_next[flat block] == tag_next_value: the next block
 * touches the row too -- skr_adam_block_mark of its slot table) 1 = only those rows, which the next block must find in the
 * dense tables, 2 = only the others, which may be written back beside the next block's steps (another stream; they must be
 * back before the next block's cold pass, and the two blocks then need workspaces of their own). */
int skr_bpr_fused_end(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_slot_block,
                      const int32_t* d_slot_fin, const int32_t* d_n_slots, float lr, float beta1, float beta2, float eps,
                      int64_t step_t0, int k, const int32_t* d_tag_next, int32_t tag_next_value, int which, void* stream);
/* The k skr_bpr_fused_step launches of a block (batch s at d_u / d_i / d_j + s * n_batch, its words at d_meta + 5 * s * n_batch, its
 * loss sums at d_loss64 + s * loss_stride_floats) followed by skr_bpr_fused_end (which = 1 when d_tag_next is given, else 0), in one call: the host side of a step is then a
 * loop in C, not 24-argument calls from the caller's language. */
int skr_bpr_fused_block(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_u,
                        const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch, int64_t user_block0,
                        int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2, float eps, int64_t step_t0,
                        int k, float reg, float* d_loss64, int64_t loss_stride_floats, const int32_t* d_slot_block,
                        const int32_t* d_slot_fin, const int32_t* d_n_slots, const int32_t* d_tag_next, int32_t tag_next_value,
                        void* stream);
/* The first naming of a row in a block costs the step launch the zero-gradient updates from the block's start to that step
 * (16 on average at k = 32) -- on the critical path.  For the rows that were COLD in the block before (nearly every user
 * row) these updates depend on nothing the block itself does, so they are made ahead of time, beside the previous block:
 *   skr_bpr_fused_plan2   as skr_bpr_fused_plan; with d_tag_prev / tag_prev_value = the hot-block tags of the PREVIOUS block
 *                         (skr_adam_block_mark) a first naming at a step > 0 of a row that block did not touch gets the
 *                         value 7 in its word's n0 field: "the state waits in the pre buffer"
 *   skr_bpr_fused_pre     fills d_pre (float [3][cap][64]: p, m, v by slot) for those rows from the dense tables, which
 *                         must hold the state the block starts from for them (i.e. behind the previous block's cold
 *                         pass; it never reads a row the previous block named).  step_t0 / k: the block's own.
 *   skr_bpr_fused_step2 / _block2   as _step / _block, reading d_pre where a word says so (NULL if no word does).
 * The same updates in the same arithmetic (the call the step launch would have made): bit-identical results. */
int skr_bpr_fused_plan2(const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, int n_batch, int k, int64_t user_block0,
                        int64_t item_block0, int64_t bias_block0, int64_t n_flat_blocks, void* d_scratch, int32_t* d_meta,
                        int32_t* d_slot_block, int32_t* d_slot_fin, int32_t* d_n_slots, const int32_t* d_tag_prev,
                        int32_t tag_prev_value, void* stream);
int skr_bpr_fused_pre(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_pre, int64_t cap,
                      const int32_t* d_slot_block, const int32_t* d_slot_fin, const int32_t* d_n_slots, float lr, float beta1,
                      float beta2, float eps, int64_t step_t0, int k, const int32_t* d_tag_prev, int32_t tag_prev_value,
                      void* stream);
int skr_bpr_fused_step2(const float* d_p, const float* d_m, const float* d_v, int64_t n, float* d_work, int64_t cap,
                        const int32_t* d_u, const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch,
                        int64_t user_block0, int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2,
                        float eps, int64_t step_t0, int k, int s, float reg, float* d_loss64, const float* d_pre, void* stream);
int skr_bpr_fused_block2(float* d_p, float* d_m, float* d_v, int64_t n, float* d_work, int64_t cap, const int32_t* d_u,
                         const int32_t* d_i, const int32_t* d_j, const int32_t* d_meta, int n_batch, int64_t user_block0,
                         int64_t item_block0, int64_t bias_block0, float lr, float beta1, float beta2, float eps,
                         int64_t step_t0, int k, float reg, float* d_loss64, int64_t loss_stride_floats,
                         const int32_t* d_slot_block, const int32_t* d_slot_fin, const int32_t* d_n_slots,
                         const int32_t* d_tag_next, int32_t tag_next_value, const float* d_pre, void* stream);

/* The cold pass sorts each 64-float block, by the values it starts from, into one of three exact evaluations of
 * the same k updates: AT REST (the update provably rounds to p + q == p for all k steps: only the moments decay),
 * ORDINARY MAGNITUDES (square root and divisions without the scaling / fix-up steps that cannot act there), or the
 * general form.  skr_selftest_cold_math checks the second one's building blocks against the compiler's sqrtf and
 * division on the device.  h_mismatches (uint64[4], host): [0] differing results of the square root over EVERY float
 * in [2^-96, FLT_MAX], [1] of the division over n_pairs hashed (n, d) pairs of its range -- both must be 0;
 * [2] control: how often the raw v_sqrt_f32 differs from sqrtf over the same floats (> 0), [3] floats enumerated.
 * (Test hook; synchronises the stream.) */
int skr_selftest_cold_math(uint64_t n_pairs, uint64_t* h_mismatches, void* stream);
/* With SKR_COLD_STATS=1 in the environment the cold passes count their blocks by evaluation: h_counts3 = {at rest,
 * ordinary magnitudes, general} since the start / the last reset (all zero when the census is off).  Synchronises the
 * device.  (Measurement hook: tools/e2e_scale.py prints the shares of a real epoch.) */
int skr_cold_pass_census(uint64_t* h_counts3, int reset);

/* Sparse exchange of a replicated table's gradient between ranks (no reference counterpart -- the
 * reference is single-process).  A BPR step touches at most 2*batch item rows, so instead of all-reducing the
 * dense [I, 65] block each rank packs its touched rows, the ranks all-gather the packs, and every rank adds
 * all packs to its dense gradient in rank order (replicas stay bit-identical).
 *   skr_pack_grad_rows:   d_ids int32[n], unique or negative (empty slot); d_out float32[n, dim+2] rows
 *                         [id bits | dV[id] | db[id]]; the packed dense rows are CLEARED.  d_g_bias may be NULL.
 *   skr_unpack_grad_rows: d_in float32[n_ranks, n_per_rank, dim+2]; dense += every valid row, rank 0 first. */
int skr_pack_grad_rows(const int32_t* d_ids, int n, float* d_g_table, float* d_g_bias, int dim, float* d_out, void* stream);
/* skr_unpack_grad_rows_sorted: the same sum, same order of additions, same bits, in ONE launch instead of n_ranks -- for
 * packs made from id lists whose valid ids ASCEND with the empty slots (-1) last (every rank's). */
int skr_unpack_grad_rows_sorted(const float* d_in, int n_per_rank, int n_ranks, float* d_g_table, float* d_g_bias, int dim,
                                uint8_t* d_touch, const float* d_touch_base, void* stream);
int skr_unpack_grad_rows(const float* d_in, int n_per_rank, int n_ranks, float* d_g_table, float* d_g_bias, int dim,
                         uint8_t* d_touch, const float* d_touch_base, void* stream);

/* ---------------------------------------------------------------------------------------------
 * G rows: GRU4RecPlus, recommender/GRU4RecPlus.py.  PARITY UNPINNED -- the reference runs this
 * model on TensorFlow 1.14, absent here; these entry points follow the graph of GRU4RecPlus.py:124-200 and
 * the published semantics of tf.nn.rnn_cell.GRUCell / tf.train.AdamOptimizer (oracle/gru4rec.py).
 * hid in {32, 64, 128}, in_dim <= 128.  hidden_act_kind: 0 tanh, 1 relu (:76-81).  final_act_kind:
 * 0 linear, 1 relu, 2 leaky_relu(0.2) (:84-91).  loss_kind: 0 bpr_max, 1 top1_max (:93-98).
 * ------------------------------------------------------------------------------------------- */

/* One GRUCell.call for B sessions (GRU4RecPlus.py:168-178):
 *   [r|u] = sigmoid([x,h] Wg + bg), c = act([x, r*h] Wc + bc), h' = u*h + (1-u)*c.
 *   d_x        float32 [B, in_dim], or -- when d_x_index != NULL -- an embedding table whose row
 *              d_x_index[b] is the input of session b (tf.nn.embedding_lookup fused in)
 *   d_active   uint8 [B] or NULL: 0 keeps h' = h (histories that already ended, _get_user_embeddings)
 *   d_Wg [in_dim+hid, 2*hid], d_bg [2*hid], d_Wc [in_dim+hid, hid], d_bc [hid]   (TF kernel layout)
 *   d_r, d_u, d_c  float32 [B, hid] saved for the backward pass, each may be NULL;  d_h_new [B, hid] */
int skr_gru_cell_fwd(const float* d_x, const int32_t* d_x_index, const float* d_h, const uint8_t* d_active, int B,
                     int in_dim, int hid, const float* d_Wg, const float* d_bg, const float* d_Wc, const float* d_bc,
                     int hidden_act_kind, float* d_r, float* d_u, float* d_c, float* d_h_new, void* stream);

/* Its backward for one step (the state enters through a placeholder, :127: no gradient into h).
 * Accumulates (+=) into d_gWg, d_gbg, d_gWc, d_gbc; writes dL/dx to d_dx [B, in_dim].
 * d_work: float32 scratch of 3*B*hid elements. */
int skr_gru_cell_bwd(const float* d_x, const int32_t* d_x_index, const float* d_h, int B, int in_dim, int hid,
                     const float* d_Wg, const float* d_Wc, int hidden_act_kind, const float* d_r, const float* d_u,
                     const float* d_c, const float* d_dh_new, float* d_gWg, float* d_gbg, float* d_gWc, float* d_gbc,
                     float* d_dx, float* d_work, void* stream);

/* skr_gru_cell_bwd of the FIRST layer (its input rows are gathered: d_x = the input item table, d_x_index = the batch's
 * items) followed by skr_scatter_add_rows(d_dx, d_x_index, B, in_dim, d_x, reg, d_g_table, ...): the input-embedding
 * gradient rides in the weight-gradient launch (both only need what the rows kernel in front of them has written). */
int skr_gru_cell_bwd_scatter(const float* d_x, const int32_t* d_x_index, const float* d_h, int B, int in_dim, int hid,
                             const float* d_Wg, const float* d_Wc, int hidden_act_kind, const float* d_r, const float* d_u,
                             const float* d_c, const float* d_dh_new, float* d_gWg, float* d_gbg, float* d_gWc, float* d_gbc,
                             float* d_dx, float* d_work, float reg, float* d_g_table, uint8_t* d_touch,
                             const float* d_touch_base, void* stream);

/* logits = final_act(out . E[Y]^T + bias[Y]) for B sessions against n_y targets (the batch's own next
 * items first: column b is session b's positive, :180-186), bpr_max / top1_max loss (:137-166) and its
 * gradient.  d_loss[0] = mean loss (the call clears the word first); d_dlogits [B, n_y] = dL/d(pre-activation logits);
 * d_dout [B, hid] = dL/d out.  B <= n_y <= 8192. */
int skr_session_loss(const float* d_out, int B, int hid, const float* d_item_table, const float* d_item_bias,
                     const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                     float* d_dout, float* d_loss, void* stream);

/* One rank's share of a SESSION-SHARDED step (BASELINE configs[4]): the B_local sessions of this call are the
 * slots slot_offset .. slot_offset + B_local - 1 of a batch of B_global sessions whose next items are the first B_global
 * entries of d_y on every rank (row r's positive is column slot_offset + r); the loss is the mean over B_global. */
int skr_session_loss_sharded(const float* d_out, int B_local, int hid, const float* d_item_table, const float* d_item_bias,
                             const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                             float* d_dout, float* d_loss, int slot_offset, int B_global, void* stream);

/* skr_session_loss_sharded followed by skr_session_out_grads (arguments as there; slot_offset = 0 and B_global = B_local for
 * a whole batch) with dL/dout and the output-side gradients in ONE launch: three launches per call instead of four. */
int skr_session_loss_grads(const float* d_out, int B_local, int hid, const float* d_item_table, const float* d_item_bias,
                           const int32_t* d_y, int n_y, int final_act_kind, int loss_kind, float bpr_reg, float* d_dlogits,
                           float* d_dout, float* d_loss, int slot_offset, int B_global, float reg, float* d_g_table,
                           float* d_g_bias, uint8_t* d_touch, const float* d_touch_base, void* stream);

/* Output-side gradients of the same step, accumulated into dense gradient tables:
 *   d_g_table[Y[y]] += sum_b dlogits[b,y] out[b] + reg * E[Y[y]],  d_g_bias[Y[y]] += sum_b dlogits[b,y] + reg * bias[Y[y]]
 * (reg: the l2_loss term of :189-191; repeated targets count each time).  d_touch / d_touch_base as in skr_bpr_step. */
int skr_session_out_grads(const float* d_dlogits, const float* d_out, int B, int hid, const int32_t* d_y, int n_y,
                          const float* d_item_table, const float* d_item_bias, float reg, float* d_g_table,
                          float* d_g_bias, uint8_t* d_touch, const float* d_touch_base, void* stream);

/* The popularity^alpha negatives of the session-parallel loop (`_sample_neg_items`, GRU4RecPlus.py:198-200:
 * np.searchsorted(pop_cumsum, np.random.rand(size))) on the device: d_out[k] = first index with d_cumsum[idx] >= u_k.
 * d_uniform float64[n]: the uniforms drawn on the host from numpy's global generator (the reference's stream), or NULL:
 * a counter-keyed device generator (seed, k) -- equal to the reference in law only.  d_cumsum float64[n_items], ascending. */
int skr_pop_sample(const double* d_cumsum, int n_items, const double* d_uniform, uint64_t seed, int64_t n, int32_t* d_out,
                   void* stream);

/* d_g_table[index[n]] += src[n] + reg * table[index[n]]: gradient of an embedding lookup (+ l2_loss of the looked-up rows) */
int skr_scatter_add_rows(const float* d_src, const int32_t* d_index, int n, int dim, const float* d_table, float reg,
                         float* d_g_table, uint8_t* d_touch, const float* d_touch_base, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SKREC_HIP_H */
