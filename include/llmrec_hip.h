/* llmrec_hip.h - C ABI of the MI355X (gfx950) hot path of LLMRec Stage 2.
 *
 * The reference (HKUDS/LLMRec) has no plugin / FFI layer: every FLOP on its hot path is an ATen
 * call made from Python. Each entry point below replaces one group of those call
 * sites; the reference file:line it stands in for is cited on the declaration. The Python host
 * (llmrec_amd/ops.py) binds these with ctypes and passes raw device pointers
 * (tensor.data_ptr()) plus the hipStream_t of torch's current stream; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = LLMREC_OK, negative = error (llmrec_status_string());
 *     llmrec_last_error() gives the thread-local detail message. Nothing throws.
 *   - all pointers are DEVICE pointers unless the name ends in _host; matrices are row-major fp32
 *     with an explicit leading dimension in ELEMENTS (ld >= number of columns).
 *   - no entry point allocates, frees or synchronises: scratch comes from the caller
 *     (sizes from the matching *_workspace_bytes query), so every call is hipGraph-capturable.
 *     The two exceptions are the one-time set-up routines llmrec_csr_build and
 *     llmrec_spmm_plan_count, which synchronise the stream to return a count to the host.
 *   - indices: COO input int64 (torch's sparse layout, reference main.py:131); CSR row pointers
 *     and column indices int32 (nnz < 2^31 in every BASELINE.json config); byte/element offsets
 *     into X/Y are computed in 64 bit inside the kernels (50 M rows x 128 floats > 2^31).
 */
#ifndef LLMREC_HIP_H
#define LLMREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLMREC_ABI_VERSION 6

enum {
    LLMREC_OK = 0,
    LLMREC_EINVAL = -1,        /* bad argument (null pointer, negative size, misaligned ld ...) */
    LLMREC_EHIP = -2,          /* a HIP runtime call or kernel launch failed */
    LLMREC_EWORKSPACE = -3,    /* caller workspace smaller than the *_workspace_bytes answer */
    LLMREC_EUNSUPPORTED = -4   /* shape outside the compiled kernel family */
};

typedef void* llmrec_stream_t; /* hipStream_t */

int llmrec_abi_version(void);
const char* llmrec_status_string(int status);
const char* llmrec_last_error(void);

/* ------------------------------------------------------------------------------------------
 * R1  graph ingest: COO -> CSR.          replaces scipy csr_norm + matrix_to_tensor and the
 *                                        COO->CSR conversion torch.sparse.mm does per call
 *                                        (reference main.py:84-93,114-134)
 * Sorts (row, col) pairs (radix sort, 64-bit keys) so each CSR row has ascending columns;
 * duplicate (row, col) entries are kept (torch.sparse.mm sums them too). val may be NULL
 * (pattern-only). Passing (col, row) builds the CSR of the transpose.
 * ------------------------------------------------------------------------------------------ */
int64_t llmrec_csr_build_workspace_bytes(int64_t n_rows, int64_t nnz);
int llmrec_csr_build(int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int64_t* coo_row, const int64_t* coo_col, const float* coo_val,
                     int32_t* rowptr /* n_rows+1 */, int32_t* colidx /* nnz */, float* val /* nnz or NULL */,
                     void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);

/* scale[r] = fp32((deg_r + 1e-8)^-1/2), 0 for empty rows     (reference main.py:115-117) */
int llmrec_degree_scale(int64_t n_rows, const int32_t* rowptr, float* scale, llmrec_stream_t stream);

/* flag_out[0] = 1 if every row's values are one constant (then row_const[r] = that constant,
 * 0 for empty rows) else 0. Lets the SpMM drop the 4 B/nnz value stream: the reference's
 * normalised adjacency is diag(s) * R with binary R (reference main.py:123-126). */
int llmrec_csr_row_constant(int64_t n_rows, const int32_t* rowptr, const float* val,
                            float* row_const, int32_t* flag_out, llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R2  SpMM  Y = diag(row_scale) * (P (.) val) * diag(col_scale) * X
 *                                        replaces torch.sparse.mm / torch.mm(sparse, dense)
 *                                        (reference Models.py:57-61 and its 20 call sites
 *                                        :153-157,162-163,166-167,176-180) and, with col_scale,
 *                                        the transposed SpMM autograd runs for dX = A^T dY.
 * val, row_scale, col_scale may each be NULL (= all ones). d = columns of X and Y.
 * Rounding of the col_scale form: the kernel accumulates fma(col_scale[c], X[c], acc). The Python host (llmrec_amd/ops.py::spmm_raw) runs
 * pattern-only operands of HBM-bound graphs (nnz >= 4 M) as A (c . X) instead - X pre-scaled by one dense pass into a cached scratch, then
 * this call with col_scale = NULL: acc += round(c x), one rounding more per term (<= 1 ulp of each product; the sums agree to fp32 round-off,
 * not bit for bit, with the fma form used below that size).
 * Rows are bucketed by length (the plan calls below; thresholds chosen by the host) and all buckets run in ONE launch:
 *   nnz <= LLMREC_SPMM_LONG_ROW   one lane group (d/4 lanes) per row,
 *   nnz <= t_wave                 one wavefront per row                  (list wave_rows),
 *   nnz <= t_block                one 512-thread block per row           (list block_rows),
 *   longer                        pieces of `segment` nnz, one block each, partial sums in `partials` (caller scratch,
 *                                 plan.n_segments * d floats) added in a fixed order by a second launch
 *                                 (lists split_rows / split_seg_begin / seg_split).
 * Every summation tree is fixed by (nnz, thresholds, d): results are run-to-run deterministic, no float atomics.
 * llmrec_amd/ops.py picks 32 nnz per lane group for launch-bound (L2-resident) graphs, 128 for HBM-bound ones.
 * slice_width > 0 (must divide d; epilogue op NONE only): the operand is processed as d / slice_width independent
 * column slices - (row, slice) tasks - which gives a launch-bound graph the parallelism and latency of a narrow product.
 * Epilogue on the finished row r (t = alpha * Z[r] + result[r], Z may be NULL or alias Y):
 *   LLMREC_SPMM_EPI_NONE          Y[r] = t                       (Z = Y, alpha = 1: "Y += A X")
 *   LLMREC_SPMM_EPI_SOFTMAX       Y[r] = softmax(t) over the d columns           (reference Models.py:176-177)
 *   LLMREC_SPMM_EPI_SOFTMAX_BWD   Y[r] = S[r] * (t - sum(t * S[r]))              (its backward; S = the forward output)
 * and finally an optional per-row scale of the OUTPUT (epilogue.post_scale).
 * ------------------------------------------------------------------------------------------ */
#define LLMREC_SPMM_LONG_ROW 32    /* rows with more nnz leave the lane-group bucket */

typedef struct {
    int32_t t_wave, t_block, segment;                      /* LONG_ROW <= t_wave <= t_block, segment >= LONG_ROW */
    int32_t n_wave_rows;  const int32_t* wave_rows;        /* rows with LONG_ROW < nnz <= t_wave */
    int32_t n_block_rows; const int32_t* block_rows;       /* rows with t_wave < nnz <= t_block */
    int32_t n_split_rows; const int32_t* split_rows;       /* rows with nnz > t_block */
    const int32_t* split_seg_begin;                        /* [n_split_rows] first piece of each split row */
    int32_t n_segments;   const int32_t* seg_split;        /* [n_segments] index into split_rows */
    /* Round 6 - PERMUTED CSR (pattern-only operands). slot_row != NULL: the rowptr / colidx arguments of llmrec_spmm_f32 are a copy of the operand's
     * CSR whose rows are stored in the order they are visited (llmrec_csr_permute_rows): slots [0, n_short_rows) = the lane-group bucket (nnz <=
     * LONG_ROW, empty rows included), then the wavefront, block and split rows; CSR row `slot` is output row slot_row[slot] (Y, Z, S, row_scale,
     * post_scale and the row flags are addressed by OUTPUT row) and the four lists hold slots. llmrec_amd/ops.py orders every bucket by descending
     * length class (power-of-two bucket of nnz), ascending row id within a class: a wavefront's 64 / LPR rows then have (almost) equal lengths
     * and finish together, and the index stream stays sequential - profiles/experiments/r06_spmm_order.md. Results do not depend on the order. */
    int32_t n_short_rows; const int32_t* slot_row;
} llmrec_spmm_plan_t;

#define LLMREC_SPMM_EPI_NONE 0
#define LLMREC_SPMM_EPI_SOFTMAX 1
#define LLMREC_SPMM_EPI_SOFTMAX_BWD 2
typedef struct {
    int32_t op; float alpha;
    const float* Z; int64_t ldz;
    const float* S; int64_t lds;
    const float* post_scale;     /* [n_rows] or NULL: Y[r] = post_scale[r] * op(t) - lets the CONSUMER of Y run without a per-edge
                                    col_scale gather (dX = R diag(s) g is computed as R (s . g)) */
    /* Operand sparsity (the backward of the last propagation layer: the gradient that enters it is non-zero only in the rows of the
     * batch's items, its product only in the rows of their neighbours - reference Models.py:176-186 differentiated):
     *   x_row_mask [n_cols] or NULL: rows of X whose byte differs from x_mask_active (1..255) are PROMISED all-zero and are not read
     *                 (they may hold anything); the result equals the unmasked product on the promised zeros up to the sign of a zero;
     *   y_row_flag [n_rows] or NULL: written for every row: x_mask_active if the row's result can be non-zero - an active X row was
     *                 gathered, or z_row_flag[row] == x_mask_active (a non-zero row of Z); rows of the long-row buckets are always
     *                 flagged - else 0: the mask for the next product in the chain (row-local epilogues map zero rows to zero rows);
     *   z_row_flag [n_rows] or NULL (may alias y_row_flag);
     *   y_row_gate [n_rows] or NULL: rows whose byte differs from x_mask_active are PROMISED to have no active neighbour and a zero Z row:
     *                 they are written as zeros before the index list of the row is even read (llmrec_mark_neighbours_u8 computes such a gate
     *                 from the list of active columns - a sweep over THEIR adjacency instead of the one of every row);
     *   y_row_needed [n_rows] or NULL: output rows whose byte differs from x_mask_active are NOT COMPUTED AND NOT WRITTEN (their previous
     *                 contents stay) - for a product whose consumers read only marked rows (the last propagation layer of a training step:
     *                 the loss reads the rows of the batch, the next product the rows those reach). Needs x_mask_active, not x_row_mask;
     *                 rows of the long-row buckets are always computed.
     *   rows_listed_only != 0: ONLY the rows in the lists of the plan (wavefront / block / split rows) are computed - with a plan built over an
     *                 explicit row list this is "these rows of A X"; all other rows of Y keep their contents. */
    const uint8_t* x_row_mask; int32_t x_mask_active; uint8_t* y_row_flag; const uint8_t* z_row_flag; const uint8_t* y_row_gate;
    const uint8_t* y_row_needed; int32_t rows_listed_only;
    /* Cache policy (round 5 experiment, profiles/experiments/r05_spmm_nt.md): x_nt_from_row > 0 gathers the X rows with index >= x_nt_from_row
     * with non-temporal loads, the rows below it with the default policy - meant for operands whose columns are ordered by descending
     * degree (hot rows first), so that the cold 256-B rows stream through the L2 without displacing the hot set. A hint only: results are
     * bit-identical. Pattern-only, unmasked products; ignored otherwise. 0 = off. */
    int32_t x_nt_from_row;
    /* Block -> row map (round 6 experiment, profiles/experiments/r06_spmm_order.md): xcd_contiguous != 0 gives the workgroups of one XCD (ids
     * equal mod 8: each XCD has a private L2) a CONTIGUOUS piece of the short-row / wavefront-row / block-row tasks instead of every 8th
     * block of them - for graphs whose row ids expose communities. Results are bit-identical. 0 = the linear map. */
    int32_t xcd_contiguous;
    /* Round 6: the tasks of the lane-group bucket run as software pipelines - a lane group takes several tasks and keeps the next task's indices
     * and the one after's row pointers in flight while it gathers (one memory round trip per task instead of three; results bit-identical).
     * no_pipeline != 0 restores one task per lane group (A/B switch). */
    int32_t no_pipeline;
} llmrec_spmm_epilogue_t;
/* Round 5 - "these rows of A X" with the row list AND its length in device memory (the row-restricted forward of the row-sharded step,
 * without a host read-back: llmrec_amd/dist_fused.py). Pattern-only operands (A = diag(row_scale) P).
 *   llmrec_sort_unique_ids_i32   list[0 .. *n_out) = the distinct ids >= 0 among ids[0 .. n), ascending (n <= LLMREC_SORT_UNIQUE_MAX: one block,
 *                                bitonic network over 32-bit keys in LDS); list holds n entries.
 *   llmrec_spmm_rows_compact_f32 out[j] = row_scale[r_j] * sum_{c in row r_j} X[c] for j < *n_list_dev, r_j = row_list[j]; out[j] = 0 for
 *                                *n_list_dev <= j < capacity (a fixed-size message for the exchange that follows). Long rows are cut into up to
 *                                LLMREC_SPMM_COMPACT_PARTS pieces (one block each, partial sums added in piece order by a second launch:
 *                                deterministic). workspace: llmrec_spmm_rows_compact_workspace_bytes(capacity, d).
 *   llmrec_scatter_set_rows_f32  dst[row_list[j]] = src[j] for j < *n_list_dev. */
#define LLMREC_SORT_UNIQUE_MAX 32768
#define LLMREC_SPMM_COMPACT_PARTS 8
int llmrec_sort_unique_ids_i32(int64_t n, const int64_t* ids, int32_t* list, int32_t* n_out, llmrec_stream_t stream);
int64_t llmrec_spmm_rows_compact_workspace_bytes(int64_t capacity, int32_t d);
int llmrec_spmm_rows_compact_f32(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* colidx, const float* row_scale,
                                 const float* X, int64_t ldx, int32_t d, const int32_t* row_list, const int32_t* n_list_dev, int32_t capacity,
                                 float* out, int64_t ldo, void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
int llmrec_scatter_set_rows_f32(int32_t capacity, const int32_t* row_list, const int32_t* n_list_dev, int32_t d, const float* src, int64_t lds,
                                float* dst, int64_t ldd, llmrec_stream_t stream);
/* flags[ids[j]] = value for j < n (ids[j] < 0 skipped): marks the rows a batch touches (x_row_mask / z_row_flag above) */
int llmrec_mark_rows_u8(int64_t n, const int64_t* ids, int32_t value, uint8_t* flags, llmrec_stream_t stream);
/* flags[c] = value for every column c of the CSR rows ids[j], j < n (ids[j] < 0 skipped): the rows of the transposed operand that have
 * one of the listed rows as a neighbour - with the by-item CSR and the batch's items: the users whose gradient the items' gradient reaches */
int llmrec_mark_neighbours_u8(int64_t n, const int64_t* ids, const int32_t* rowptr, const int32_t* colidx, int32_t value, uint8_t* flags,
                              llmrec_stream_t stream);
/* The user rows a training step's batch reaches, as an ascending row list (two launches, no host read-back - capturable): the batch's
 * users and every user adjacent to one of the batch's positive / negative items. These are the only rows in which the gradient of the
 * attribute streams' projected features is non-zero (reference Models.py:160-163 + main.py:249-254: an attribute stream reaches the loss
 * through the fused embeddings and the BPR terms of the batch's rows; its gradient travels back through A_iu^T from the batch's items) -
 * the row list of llmrec_wgrad_problem_t. users / pos / neg: the batch (int64, B_cap entries, the first *n_valid valid; n_valid NULL =
 * all; negative ids skipped); item_rowptr / item_colidx: the by-item CSR (row = item, columns = its users); flags: n_users bytes of
 * scratch, ALL-ZERO on entry and left all-zero; row_list: n_users + 32 entries; n_rows: device scalar. */
int llmrec_batch_reach_rows(int64_t n_users, int64_t n_items, const int64_t* users, const int64_t* pos, const int64_t* neg, int32_t B_cap,
                            const int32_t* n_valid, const int32_t* item_rowptr, const int32_t* item_colidx, uint8_t* flags,
                            int32_t* row_list, int32_t* n_rows, llmrec_stream_t stream);

/* counts_host[0..3] = n_wave_rows, n_block_rows, n_split_rows, n_segments (synchronises the stream). */
int llmrec_spmm_plan_count(int64_t n_rows, const int32_t* rowptr, int32_t t_wave, int32_t t_block, int32_t segment,
                           int32_t* scratch4 /* device, 4 ints */, int32_t* counts_host, llmrec_stream_t stream);
int llmrec_spmm_plan_fill(int64_t n_rows, const int32_t* rowptr, int32_t t_wave, int32_t t_block, int32_t segment,
                          int32_t* scratch4, int32_t* wave_rows, int32_t* block_rows, int32_t* split_rows,
                          int32_t* split_seg_begin, int32_t* seg_split, llmrec_stream_t stream);

/* p_colidx[p_rowptr[s] + j] = colidx[rowptr[perm[s]] + j]: the CSR with its rows stored in the order perm (p_rowptr = the prefix sums of the
 * permuted row lengths, computed by the caller). One lane group per row. */
int llmrec_csr_permute_rows(int64_t n_rows, const int32_t* rowptr, const int32_t* colidx, const int32_t* perm, const int32_t* p_rowptr,
                            int32_t* p_colidx, llmrec_stream_t stream);

int llmrec_spmm_f32(int64_t n_rows, int64_t n_cols,
                    const int32_t* rowptr, const int32_t* colidx, const float* val,
                    const float* row_scale, const float* col_scale,
                    const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t d, int32_t slice_width,
                    const llmrec_spmm_plan_t* plan_host, float* partials,
                    const llmrec_spmm_epilogue_t* epilogue_host /* NULL = none */, llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R4  side-feature projection            replaces nn.Linear forward / weight-grad
 *                                        (reference Models.py:30-37,145-150; aten::addmm, aten::mm)
 * Y[M x N] = X[M x K] * W[N x K]^T + b     (fp32 MFMA v_mfma_f32_16x16x4_f32, exact fp32)
 * dW[N x K] (+)= dY[M x N]^T * X[M x K],  db[N] (+)= column sums of dY.  The inputs X are
 * constants in the reference, so no dX is ever needed.
 * N must be a multiple of 16 and <= 128; K a multiple of 4.
 * ------------------------------------------------------------------------------------------ */
int llmrec_linear_fwd_f32(int64_t M, int32_t N, int32_t K, const float* X, int64_t ldx,
                          const float* W, int64_t ldw, const float* bias,
                          float* Y, int64_t ldy, llmrec_stream_t stream);
/* Several projections in one launch (the reference runs 8 per forward, Models.py:145-150):
 * Y_p = X_p W_p^T + b_p with a common N <= 64. Y_p may be a column slice of a wider buffer (ldy). */
#define LLMREC_LINEAR_MAX_PROBLEMS 8
typedef struct {
    const float* X; int64_t ldx; int64_t M; int32_t K;
    const float* W; int64_t ldw; const float* bias;
    float* Y; int64_t ldy;
    const float* bias_scale;   /* [M] or NULL (= all ones): Y[r] = X[r] W^T + bias_scale[r] * bias. With X = A F (a constant feature
                                  matrix propagated once through a constant adjacency) and bias_scale = the row sums of A this is
                                  A (F W^T + 1 b^T), i.e. projection + propagation (reference Models.py:145-157) as ONE product */
} llmrec_linear_problem_t;
int llmrec_linear_fwd_grouped_f32(int32_t n_problems, const llmrec_linear_problem_t* problems_host, int32_t N,
                                  llmrec_stream_t stream);
/* Weight gradient of one Linear fed by several (dY_p, X_p) pairs - the reference's shared
 * item_trans receives 5 per step (Models.py:150): dW (+)= sum_p dY_p^T X_p, db (+)= sum_p colsum(dY_p).
 * Workspace: llmrec_linear_wgrad_workspace_bytes(sum_p M_p, N, K). */
typedef struct { const float* dY; int64_t lddy; const float* X; int64_t ldx; int64_t M;
                 const float* db_row_weight;   /* [M] or NULL (= ones): db (+)= sum_r db_row_weight[r] dY[r] - the bias gradient of a projection
                                                  with llmrec_linear_problem_t.bias_scale (bf16x3, N = 64, K % 128 == 0 only) */
                 /* ROW LIST (llmrec_linear_wgrad_multi_* with every K % 128 == 0 only; NULL elsewhere): the rows of dY that can be non-zero.
                  * A training step's dY is exactly zero outside the rows its batch reaches (llmrec_batch_reach_rows), and a zero row adds
                  * nothing to dY^T X: the kernel streams the LISTED rows of dY and X only. Contract: row_list = device int32 ids, ascending,
                  * distinct, < M, readable up to n_rows rounded up to 16 entries (+ 16); every row of dY NOT listed is all-zero (it is
                  * never read; a non-zero one would be silently dropped); n_rows = device scalar (<= M), read by the kernel, so the list
                  * can change between replays of a captured launch. rows_expected (host, 0 = M) sizes the launch geometry: the problem
                  * gets ceil(rows_expected / slab) slabs, which the kernel re-cuts into equal pieces of the actual n_rows - a poor
                  * guess costs balance, never correctness. Deterministic; equal to the dense launch up to fp32 summation order. */
                 const int32_t* row_list; const int32_t* n_rows; int64_t rows_expected;
} llmrec_wgrad_problem_t;
int llmrec_linear_wgrad_grouped_f32(int32_t n_problems, const llmrec_wgrad_problem_t* problems_host, int32_t N, int32_t K,
                                    float* dW, int64_t lddw, float* db, int32_t accumulate,
                                    void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
/* Same contract, split-precision arithmetic (three-term bf16 split of both operands, six bf16 MFMAs per tile pair,
 * fp32 accumulate: fp32-roundoff-class error, bound by the X stream from HBM instead of the fp32 matrix pipe).
 * Shapes outside the fast path (N % 64, K % 64, 16-byte aligned rows) run the exact fp32 kernel. */
int llmrec_linear_wgrad_grouped_bf16x3(int32_t n_problems, const llmrec_wgrad_problem_t* problems_host, int32_t N, int32_t K,
                                       float* dW, int64_t lddw, float* db, int32_t accumulate,
                                       void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
/* The weight gradients of several Linears (targets) in ONE launch + one reduction launch, split-precision arithmetic as above:
 * target t has its own K, dW, db and (dY_p, X_p) pairs; N is common. All targets are cut into slabs of one length, chosen so
 * that the launch is whole rounds of equal blocks (three back-to-back launches each pay their own ramp-up and ragged last
 * round). Fast path only - N % 64 == 0, K_t % 64 == 0, non-empty problems with 16-byte aligned rows and byte offsets below
 * 2^32 - otherwise LLMREC_EUNSUPPORTED (workspace query: -1) and the caller issues llmrec_linear_wgrad_grouped_bf16x3 per
 * target. Deterministic (fixed slab order); the slab length differs from the single-target launches', so the sums are
 * equal to those only up to fp32 rounding. */
#define LLMREC_WGRAD_MAX_TARGETS 4
typedef struct { int32_t n_problems; const llmrec_wgrad_problem_t* problems; int32_t K; float* dW; int64_t lddw; float* db; int32_t accumulate;
                 int32_t block_budget;   /* read from targets[0] only: lay the launch out as whole rounds of this many blocks (0 = 256, one block per
                                            CU). A budget below 256 leaves CUs to kernels on other streams: a block of this launch owns its CU's
                                            whole register file, so whatever runs beside a 256-block round waits for it to drain */
} llmrec_wgrad_target_t;
int64_t llmrec_linear_wgrad_multi_workspace_bytes(int32_t n_targets, const llmrec_wgrad_target_t* targets_host, int32_t N);
int llmrec_linear_wgrad_multi_bf16x3(int32_t n_targets, const llmrec_wgrad_target_t* targets_host, int32_t N,
                                     void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
/* llmrec_linear_wgrad_multi_bf16x3 with the AdamW update of every target's W and b inside the reduction launch (the arithmetic of
 * llmrec_adamw_multi_f32, element by element, on the gradient that launch has just summed and stored: parameters, moments and
 * gradients come out bit-identical to the two calls in sequence). A fused training step ends on this launch instead of
 * reduction -> update. updates_host[t]: W / b (the parameters dW / db are the gradients of; b, m_b, v_b NULL when the target has no
 * db), their moments, g_scale as in llmrec_adamw_tensor_t; W and dW contiguous (lddw == K); state3 = the device state of
 * llmrec_adamw_advance. */
typedef struct { float* W; float* m_W; float* v_W; float* b; float* m_b; float* v_b; float g_scale; } llmrec_wgrad_update_t;
int llmrec_linear_wgrad_multi_adamw_bf16x3(int32_t n_targets, const llmrec_wgrad_target_t* targets_host, int32_t N,
                                           void* workspace, int64_t workspace_bytes, const llmrec_wgrad_update_t* updates_host,
                                           const float* state3, float lr, float beta1, float beta2, float eps, float weight_decay,
                                           llmrec_stream_t stream);
/* Same contract, split precision: each fp32 operand = exact sum of three bf16 numbers, product by the
 * six bf16 MFMAs (v_mfma_f32_16x16x32_bf16, fp32 accumulate) whose terms are >= 2^-24 relative.
 * fp32-roundoff-class error (not the bit-identical fma chain); 3/8 of the fp32 matrix time, so the
 * kernel becomes HBM-bound on the X stream. */
int llmrec_linear_fwd_grouped_bf16x3(int32_t n_problems, const llmrec_linear_problem_t* problems_host, int32_t N,
                                     llmrec_stream_t stream);
int64_t llmrec_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K);
int llmrec_linear_wgrad_f32(int64_t M, int32_t N, int32_t K, const float* dY, int64_t lddy,
                            const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                            int32_t accumulate, void* workspace, int64_t workspace_bytes,
                            llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R3  last-layer row softmax + mean over the L+1 layer outputs
 *                                        replaces nn.Softmax(dim=-1) and stack+mean
 *                                        (reference Models.py:176-177,185-186)
 * ------------------------------------------------------------------------------------------ */
int llmrec_softmax_rows_fwd_f32(int64_t rows, int32_t d, const float* Z, int64_t ldz,
                                float* Y, int64_t ldy, llmrec_stream_t stream);
/* dZ = Y (.) (dY - <dY, Y>_row) */
int llmrec_softmax_rows_bwd_f32(int64_t rows, int32_t d, const float* Y, int64_t ldy,
                                const float* dY, int64_t lddy, float* dZ, int64_t lddz,
                                llmrec_stream_t stream);
/* dZ = softmax_bwd(Y, alpha * dY): the scaling of the incoming gradient (the "+ mean term" factor 1 / (L + 1) of the ID chain) in the
 * same pass; alpha = 1 is llmrec_softmax_rows_bwd_f32 bit for bit. */
int llmrec_softmax_rows_bwd_scaled_f32(int64_t rows, int32_t d, float alpha, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                                       float* dZ, int64_t lddz, llmrec_stream_t stream);
/* The same for the listed rows only, with a per-row scale of the result: dZ[r] = post_scale[r] * softmax_bwd(Y[r], alpha * dY[r]) for
 * r = ids[j], j < n (ids[j] < 0 skipped; duplicates write the same values). The other rows of dZ are not touched: the first gradient of
 * the row-sharded step's backward is needed (and, behind an x_row_mask, read) only in the rows of the batch's items. */
int llmrec_softmax_rows_bwd_listed_f32(int64_t n, const int64_t* ids, int32_t d, float alpha, const float* Y, int64_t ldy,
                                       const float* dY, int64_t lddy, const float* post_scale, float* dZ, int64_t lddz,
                                       llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R6  fusion  out = scale * sum_t mean_terms[t] + sum_t rates[t] * normalize(terms[t])
 *                                        replaces torch.mean(torch.stack(..)) + the 16
 *                                        F.normalize / scaled-add calls (reference Models.py:185-197)
 * normalize(x) = x / max(||x||_2, 1e-12) row-wise. Term pointer/ld/rate tables live on the HOST
 * (<= LLMREC_MAX_TERMS each) and are passed by value to the kernel.
 * bwd: d_mean (shared by all mean terms) = scale * dOut is NOT written (the caller scales);
 *      d_terms[t] (+)= rates[t] * (dOut - n <n, dOut>) / max(||x||, 1e-12)   [+ reg_two_coef * x for t < n_reg_terms]
 * ------------------------------------------------------------------------------------------ */
#define LLMREC_MAX_TERMS 12
int llmrec_fuse_fwd_f32(int64_t rows, int32_t d, float mean_scale,
                        int32_t n_mean, const float* const* mean_terms_host, const int64_t* mean_ld_host,
                        int32_t n_norm, const float* const* norm_terms_host, const int64_t* norm_ld_host,
                        const float* rates_host, float* out, int64_t ldo, llmrec_stream_t stream);
int llmrec_fuse_bwd_f32(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                        int32_t n_norm, const float* const* norm_terms_host, const int64_t* norm_ld_host,
                        const float* rates_host, float* const* d_terms_host, const int64_t* d_ld_host,
                        int32_t accumulate, int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream);
/* The same with a SOURCE per term: d_terms[t] = src_terms[t] (or 0 where src_terms[t] is NULL) + the term's gradient, written
 * without reading d_terms. With the loss backward scattering into separate, otherwise all-zero source buffers
 * (llmrec_bpr_multi_bwd_f32) and llmrec_bpr_multi_zero_rows_f32 clearing the touched rows afterwards, a training step needs no
 * dense memset of its gradient buffers. */
int llmrec_fuse_bwd_src_f32(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                            int32_t n_norm, const float* const* norm_terms_host, const int64_t* norm_ld_host,
                            const float* rates_host, float* const* d_terms_host, const int64_t* d_ld_host,
                            const float* const* src_terms_host, const int64_t* src_ld_host,
                            int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream);
/* The user-side and the item-side fusion (or fusion backward, llmrec_fuse_bwd_src_f32 semantics) of one step as ONE launch
 * (n_problems <= 2; the two row ranges are independent): on two streams each launch paid a cross-queue fork and join for
 * 15 - 26 us of work (reference Models.py:188-197 for users and items). */
typedef struct {
    int64_t rows; float mean_scale;
    int32_t n_mean; const float* const* mean_terms; const int64_t* mean_ld;
    int32_t n_norm; const float* const* norm_terms; const int64_t* norm_ld; const float* rates;
    float* out; int64_t ldo;
} llmrec_fuse_fwd_problem_t;
int llmrec_fuse_fwd_multi_f32(int32_t n_problems, const llmrec_fuse_fwd_problem_t* problems_host, int32_t d, llmrec_stream_t stream);
/* The same launch, which ALSO leaves the sum of squares of the first n_sumsq_terms norm terms of every problem (the image / text streams the
 * feature regulariser of reference main.py:151-156 sums: the kernel holds those rows and their squared norms in registers anyway) as one
 * partial sum per block: sumsq_partial[0 .. *n_partial_host) (fixed block partition, fixed in-block tree: deterministic). *n_partial_host is
 * written on the HOST at call time (a pure function of the problems' row counts, <= partial_capacity or the call fails); the consumer is
 * llmrec_bpr_multi_losses_assemble_f32. Replaces the four llmrec_sumsq_f32 launches of a step. */
int llmrec_fuse_fwd_multi_sumsq_f32(int32_t n_problems, const llmrec_fuse_fwd_problem_t* problems_host, int32_t d, int32_t n_sumsq_terms,
                                    float* sumsq_partial, int32_t partial_capacity, int32_t* n_partial_host, llmrec_stream_t stream);
typedef struct {
    int64_t rows; const float* dOut; int64_t lddo;
    int32_t n_norm; const float* const* norm_terms; const int64_t* norm_ld; const float* rates;
    float* const* d_terms; const int64_t* d_ld;
    const float* const* src_terms; const int64_t* src_ld;      /* src_terms[t] may be NULL (= 0) */
    int32_t n_reg_terms; float reg_two_coef;
    const uint8_t* row_flags;   /* [rows] or NULL. A row whose flag is not ACTIVE (see LLMREC_ROW_STAMP below) PROMISES that dOut[r] and every
                                   source row r are all-zero (a row no sample of the batch touched - llmrec_bpr_multi_select_bwd_f32 stamps the
                                   rows it touches): the kernel then writes the regulariser's term for the first n_reg_terms streams and zeros
                                   for the others without reading anything else - the same bits, ~half the traffic at 10 % touched rows */
    const int32_t* row_stamp;   /* device counter that defines ACTIVE, or NULL (active = non-zero flag) */
} llmrec_fuse_bwd_problem_t;
int llmrec_fuse_bwd_src_multi_f32(int32_t n_problems, const llmrec_fuse_bwd_problem_t* problems_host, int32_t d, llmrec_stream_t stream);

/* n_reg_terms / reg_two_coef: the first n_reg_terms terms additionally receive reg_two_coef * x - the gradient of a
 * sum-of-squares regulariser coef * sum x^2 on them (reference main.py:151-156 on the image / text streams), folded in
 * because the kernel has x in registers anyway; pass 0, 0 for the plain backward. */

/* ------------------------------------------------------------------------------------------
 * R7  fused BPR + prune loss             replaces the 3 gathers, mul/sum, logsigmoid, the
 *                                        D2H argsort of prune_loss and its backward
 *                                        (reference main.py:158-165,232-254,330-342)
 * For sample b < B:  s_b = <Eu[u_b], Ei[p_b]> - <Eu[u_b], Ei[q_b]>,  m_b = logsigmoid(s_b + 1e-8)
 * keep = the k = (int)(remember_rate * B) samples with the SMALLEST m_b (ties: lower b first)
 * out[0] = mf  = -(1/k) * sum_keep m_b
 * out[1] = emb = decay / batch_size_flag * (1/(2 Su + 1e-8) + 1/(2 Sp + 1e-8) + 1/(2 Sq + 1e-8)),
 *          S* = squared Frobenius norms of the three gathered B x d blocks
 * The forward stores what backward needs (per-sample ds_b, the three norms, and per-sample
 * scratch) in `saved` (LLMREC_BPR_SAVED_FLOATS(B_max) floats). B may come from device memory (n_valid_dev != NULL) so a captured graph can
 * replay with a varying number of augmented triples; B_max bounds it (<= LLMREC_BPR_MAX_B).
 * ------------------------------------------------------------------------------------------ */
#define LLMREC_BPR_MAX_B 4096
#define LLMREC_BPR_MAX_PROBLEMS 8
#define LLMREC_BPR_SAVED_FLOATS(B) (6 * (B) + 8)   /* floats of `saved` per problem */
int llmrec_bpr_prune_fwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev,
                             double remember_rate, float decay, float batch_size_flag,
                             float* out2, float* saved, llmrec_stream_t stream);
/* User-sharded batch: every rank holds B_local samples of a global batch of
 * global_B. Pass 1 (scores_only = 1) writes the local m_b to saved[0..B_local) and the local
 * squared norms per sample to the scratch part of saved (m_b at saved[B_local + 4 + b]); the
 * host all-gathers the m_b (RCCL) and calls
 * pass 2 with them: selection ranks against the GLOBAL batch (ties: lower global index), out2[0]
 * is this rank's share -(1/k) * sum_{kept local} m_b (all-reduce it), and saved is ready for
 * llmrec_bpr_prune_bwd_f32 once saved[B_local..B_local+2] hold the all-reduced norms. */
int llmrec_bpr_prune_fwd_sharded_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_local, double remember_rate, float decay, float batch_size_flag,
                                     const float* global_m, int32_t global_B, int32_t my_offset, int32_t scores_only,
                                     float* out2, float* saved, llmrec_stream_t stream);
/* Several (user table, item table) pairs over ONE batch in two launches - the reference computes
 * 8 such losses per step (main.py:232-254). out: [n_problems][2], saved: n_problems blocks of
 * LLMREC_BPR_SAVED_FLOATS(B_max). The backward takes the loss weights from the (host) problem
 * table (g_mf, g_emb) and scatter-adds (deterministically, see llmrec_bpr_scatter_plan) into dEu / dEi, which the caller zero-initialises. */
typedef struct {
    const float* Eu; int64_t ldu; const float* Ei; int64_t ldi;
    float* dEu; int64_t lddu; float* dEi; int64_t lddi;
    float g_mf, g_emb;
} llmrec_bpr_problem_t;
int llmrec_bpr_multi_fwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                             float batch_size_flag, float* out, float* saved, llmrec_stream_t stream);
/* Batch-sharded form of llmrec_bpr_multi_fwd_f32 (data-parallel replicas or user-sharded ranks;
 * SURVEY.md 8(e) "all-gather the B log-sigmoids, select the threshold redundantly"). The global
 * batch is the concatenation of the ranks' valid samples in rank order; prune keeps the
 * k = (int)(remember_rate * B_global) smallest m_b of the GLOBAL batch (ties: lower global index).
 *   phase 1: scores + this rank's gather block -> gather_block[LLMREC_BPR_GATHER_FLOATS(P, B_max)]:
 *            [P][B_max] m_b (+inf beyond n_valid) | [P][4] Su, Sp, Sq, - (local sums) | n_valid as float
 *   (host: all-gather the blocks, RCCL; every rank's B_max must be equal)
 *   phase 2: selection against `gathered` ([n_ranks] blocks, rank_stride floats apart): out[p][0] is
 *            this rank's SHARE of mf_p (sum over ranks = mf_p), out[p][1] = emb_p from the global
 *            norms (identical on every rank); `saved` is then ready for llmrec_bpr_multi_bwd_f32.
 * n_ranks * B_max <= 4 * LLMREC_BPR_MAX_B. */
#define LLMREC_BPR_GATHER_FLOATS(P, B) ((P) * (B) + 4 * (P) + 1)
int llmrec_bpr_multi_fwd_sharded_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                     float batch_size_flag, int32_t phase, float* gather_block,
                                     const float* gathered, int32_t n_ranks, int64_t rank_stride, int32_t my_rank,
                                     float* out, float* saved, llmrec_stream_t stream);
/* DETERMINISTIC gradient scatter (ABI 6). The reference's backward is index_put(accumulate) on CPU - the rows several samples share
 * are added in one fixed order and same-seed runs agree (main.py:232-254,330-342). Every backward entry point below therefore takes a
 * SCATTER PLAN of the batch instead of adding rows with float atomics:
 *   llmrec_bpr_scatter_plan   plan[0 .. B_max)        = the keys  users[b] << 32 | b                          sorted ascending
 *                             plan[B_max .. 3 B_max)  = the keys  pos[b] << 32 | b  and  neg[b] << 32 | (B_max + b), sorted ascending
 *                             (samples b >= n_valid: id 0xffffffff, at the end), then as int32 runlen[3 B_max]: at the first position of a
 *                             run of equal ids the run's length, 0 elsewhere. One launch (rank counting in LDS, ceil(3 B_max / 16) blocks);
 *                             it depends on the batch only, so a step builds it right behind its sampler, off the critical path.
 * In the backward launch the head of every run of equal ids owns the destination row and adds the run's contributions in ascending
 * problem index, then ascending slot. Problems whose dEu (or dEi) POINTERS are equal share one target (the five attribute problems of a
 * step all scatter into d prof_u): the first of them sums the others' shares; targets must otherwise be disjoint. */
#define LLMREC_BPR_PLAN_WORDS(B) (5 * (B))         /* 64-bit words of `plan`: 3 B keys + 3 B int32 run lengths */
int llmrec_bpr_scatter_plan(const int64_t* users, const int64_t* pos, const int64_t* neg, int32_t B_max, const int32_t* n_valid_dev,
                            uint64_t* plan, llmrec_stream_t stream);
int llmrec_bpr_multi_bwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, float decay, float batch_size_flag,
                             const float* saved, const uint64_t* plan, llmrec_stream_t stream);
/* The same step of one LOCAL batch (no gathered layout) with the loss VALUES off the critical path - the fused training step's
 * critical path is scores -> selection -> backward rows instead of scores -> rank -> reduce -> backward:
 *   llmrec_bpr_multi_scores_f32       launch 1: m_b, sigmoid(-x_b) and the per-sample squared norms into `saved`
 *   llmrec_bpr_multi_select_bwd_f32   launch 2 (selection): ranks every sample against the batch in LDS and writes the kept coefficients /
 *                                     values into `saved`; one more block per problem sums the batch's squared norms with the summation
 *                                     tree of the loss launch below (so the bits agree) into saved[B_max .. B_max + 2];
 *                                     launch 3 (backward rows): the deterministic scatter of llmrec_bpr_multi_bwd_f32 through `plan`
 *   llmrec_bpr_multi_losses_f32       any time after launch 2, on any stream: out[p] = {mf_p, emb_p} and the norm / k slots of `saved`
 *                                     (only the logged loss values depend on it)
 * Results: `out`, `saved` and the gradient rows bit-identical to llmrec_bpr_multi_fwd_f32 + llmrec_bpr_multi_bwd_f32. */
/* Row stamps: the rows a batch touches are marked in byte arrays with the value LLMREC_ROW_STAMP(counter) of a device counter that
 * the scores launch advances once per step; consumers (llmrec_fuse_bwd_problem_t.row_flags) treat a
 * row as touched iff its byte equals the current stamp. Nothing ever has to be cleared - a stale byte differs from the current stamp for
 * the next 254 steps, and a false "touched" only costs the general path (same result). */
#define LLMREC_ROW_STAMP(counter) ((uint8_t)((uint32_t)(counter) % 255u + 1u))
int llmrec_bpr_multi_scores_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                const int64_t* users, const int64_t* pos, const int64_t* neg,
                                int32_t B_max, const int32_t* n_valid_dev, float* saved,
                                int32_t* row_stamp /* optional: device counter, += 1 by this launch */, llmrec_stream_t stream);
int llmrec_bpr_multi_select_bwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                    const int64_t* users, const int64_t* pos, const int64_t* neg,
                                    int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                    float batch_size_flag, float* saved,
                                    uint8_t* user_row_flags, uint8_t* item_row_flags,   /* optional: [u_b] / [p_b], [q_b] := stamp for b < n_valid */
                                    const int32_t* row_stamp,                           /* optional device counter (NULL: stamp = 1) */
                                    const uint64_t* plan,                               /* llmrec_bpr_scatter_plan of this batch */
                                    llmrec_stream_t stream);
int llmrec_bpr_multi_losses_f32(int32_t n_problems, int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                float batch_size_flag, float* out, float* saved, llmrec_stream_t stream);
/* Round 5 - fewer launches per step:
 * llmrec_bpr_multi_scores_step_f32 = llmrec_bpr_multi_scores_f32 that also BEGINS THE STEP on the device: beside the row stamp it advances
 *   AdamW's step counter / bias corrections (llmrec_adamw_advance on adamw_state3; NULL = leave it) - every AdamW launch of the step is
 *   stream-ordered behind the loss launches anyway;
 * llmrec_bpr_multi_losses_assemble_f32 = llmrec_bpr_multi_losses_f32 (same trees, same bits in out / saved) followed, in the same single-block
 *   launch, by the feature regulariser's value feat_reg = feat_reg_coef * sum(sumsq_partial[0 .. n_partial)) (llmrec_fuse_fwd_multi_sumsq_f32;
 *   fixed-order tree) -> scal4[0] and by llmrec_loss_assemble_f32 mode 0 (scal4[1..3] = loss, mf, emb; running_sums3 += them in double). */
int llmrec_bpr_multi_scores_step_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_max, const int32_t* n_valid_dev, float* saved, int32_t* row_stamp,
                                     float* adamw_state3, float lr, float beta1, float beta2, llmrec_stream_t stream);
int llmrec_bpr_multi_losses_assemble_f32(int32_t n_problems, int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                         float batch_size_flag, float* out, float* saved, const float* w_mf_host,
                                         const float* sumsq_partial, int32_t n_partial, float feat_reg_coef,
                                         float* scal4, double* running_sums3, llmrec_stream_t stream);
/* Clears exactly the rows llmrec_bpr_multi_bwd_f32 added into (dEu[u_b], dEi[p_b], dEi[q_b] of every problem, b < n_valid),
 * so scatter targets that start all-zero are all-zero again. */
int llmrec_bpr_multi_zero_rows_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                   const int64_t* users, const int64_t* pos, const int64_t* neg,
                                   int32_t B_max, const int32_t* n_valid_dev, llmrec_stream_t stream);
/* dEu[u_b] += g_mf * ds_b * (Ei[p_b] - Ei[q_b]) + g_emb * c_u * Eu[u_b]   (deterministic scatter-add through `plan`, see above)
 * dEi[p_b] += g_mf * ds_b * Eu[u_b] + g_emb * c_p * Ei[p_b] ; dEi[q_b] likewise with -ds_b, c_q
 * g_mf / g_emb are upstream gradients read from device memory (grads2[0], grads2[1]). dEu and dEi are distinct buffers. */
int llmrec_bpr_prune_bwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev,
                             float decay, float batch_size_flag,
                             const float* saved, const float* grads2,
                             float* dEu, int64_t lddu, float* dEi, int64_t lddi, const uint64_t* plan, llmrec_stream_t stream);

/* The same gradient as COMPACT rows instead of a scatter (row-sharded step): rows3 = [3][B_max][d],
 * block 0 = d/dEu[u_b], block 1 = d/dEi[p_b], block 2 = d/dEi[q_b]; rows of samples b >= B are zero. The item rows are
 * what the ranks exchange (an all-gather of 2 B rows instead of an all-reduce of the dense I x d gradient). */
int llmrec_bpr_prune_bwd_rows_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg,
                                  int32_t B_max, const int32_t* n_valid_dev,
                                  float decay, float batch_size_flag,
                                  const float* saved, const float* grads2, float* rows3, llmrec_stream_t stream);

/* dst[ids[j]] += alpha * rows[j] for j < n, ids[j] < 0 skipped. Rows with the same id are added in ascending j
 * (radix sort of (id, j), one lane group per run): DETERMINISTIC, so ranks that scatter the same gathered rows keep
 * bit-identical replicas - a float-atomic scatter would not. */
int64_t llmrec_scatter_rows_workspace_bytes(int64_t n);
int llmrec_scatter_rows_f32(int64_t n, const int64_t* ids, const float* rows, int64_t ldr, int32_t d, float alpha,
                            float* dst, int64_t ldd, void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R8  feature regulariser, optimiser     replaces (x**2).sum() x4 (reference main.py:151-156)
 *                                        and torch.optim.AdamW.step (main.py:100-104,278)
 * ------------------------------------------------------------------------------------------ */
/* out[0] (+)= coef * sum of squares of the rows x d block (deterministic two-level reduction) */
int64_t llmrec_sumsq_workspace_bytes(int64_t rows, int32_t d);
int llmrec_sumsq_f32(int64_t rows, int32_t d, const float* X, int64_t ldx, float coef, int32_t accumulate,
                     float* out, void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
/* Y (+)= alpha_dev[0] * alpha * X   (gradient of the regulariser; alpha_dev may be NULL = 1) */
int llmrec_axpy_f32(int64_t rows, int32_t d, float alpha, const float* alpha_dev, const float* X, int64_t ldx,
                    float* Y, int64_t ldy, int32_t accumulate, llmrec_stream_t stream);

/* out_g[j] (+)= sum_r w[r] * X[r][group_width g + j] for the n_groups column groups of X (w NULL = ones); groups that name the same
 * destination are summed in group order; two-level, fixed order (deterministic). The bias gradient of projections whose operand was
 * propagated beforehand (llmrec_linear_problem_t.bias_scale): db = sum_r bias_scale[r] dY[r]; one call serves every Linear of the
 * [rows, 7 d] gradient buffer (reference: the autograd bias gradients of Models.py:145-150). group_width <= 64, n_groups <= MAX_GROUPS. */
#define LLMREC_COLSUM_MAX_GROUPS 8
int64_t llmrec_weighted_colsum_workspace_bytes(int32_t n_columns);
int llmrec_weighted_colsum_f32(int64_t rows, int32_t n_groups, int32_t group_width, const float* X, int64_t ldx, const float* w,
                               float* const* group_out_host, int32_t accumulate,
                               void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);

/* Clear up to LLMREC_ZERO_MAX_TENSORS fp32 buffers in ONE launch (the backward's scatter targets; replaces
 * one aten::zero_ per tensor inside the captured step). */
#define LLMREC_ZERO_MAX_TENSORS 8
typedef struct { float* p; int64_t n; } llmrec_zero_tensor_t;
int llmrec_zero_multi_f32(int32_t n_tensors, const llmrec_zero_tensor_t* tensors_host, llmrec_stream_t stream);
/* dst[ids[j]][0..d) = 0 for j < n (ids[j] < 0 skipped): row-wise clean-up of a scatter target whose other rows are already zero
 * (the row-sharded step scatters B gradient rows into 10^7-row tables; replaces a dense aten::zero_ per step) */
int llmrec_zero_rows_f32(int64_t n, const int64_t* ids, int32_t d, float* dst, int64_t ldd, llmrec_stream_t stream);

/* Loss assembly of reference main.py:273 / :280-283 on the device, one single-wave launch (replaces the
 * aten mul / sum / add / copy launches that assembled the logged scalars):
 *   bpr_out [n_problems][2] = (mf, emb) of each BPR problem; w_mf_host[n_problems] = its weight in the loss;
 *   scal[4] = [feat_reg (input), loss, mf, emb].
 *   mode 0 (single GPU): scal[1] = sum_p w_mf[p] * mf_p + emb_0 + scal[0]; scal[2] = mf_0; scal[3] = emb_0
 *   mode 1 (batch-sharded replica, before the gradient all-reduce): tail[p] = mf_p (this rank's share),
 *          tail[n_problems] = (emb_0 + scal[0]) * inv_world
 *   mode 2 (after it): scal[2] = tail[0]; scal[3] = emb_0; scal[1] = sum_p w_mf[p] * tail[p] + tail[n_problems]
 *   running_sums3 (optional, modes 0 and 2): running_sums3[0..2] += (loss, mf, emb) in double - the epoch sums of the reference's log line
 *   (main.py:280-283) accumulated inside the step graph, so a loop of graph replays needs no host-side arithmetic per step */
int llmrec_loss_assemble_f32(int32_t mode, int32_t n_problems, const float* bpr_out, const float* w_mf_host,
                             float* scal4, float* tail, float inv_world, double* running_sums3, llmrec_stream_t stream);

/* out[b] = scale * sum_t terms[t][idx[b]] for b < n: the layer mean of reference Models.py:185-186 for the rows a batch
 * needs only (row-sharded step: B rows instead of a pass over the 10^7-row user tables) */
int llmrec_gather_mean_f32(int64_t n, const int64_t* idx, int32_t d, float scale, int32_t n_terms,
                           const float* const* terms, const int64_t* term_ld, float* out, int64_t ldo, llmrec_stream_t stream);
/* Y[r] = s[r] * X[r] (Y may alias X) */
int llmrec_scale_rows_f32(int64_t rows, int32_t d, const float* s, const float* X, int64_t ldx, float* Y, int64_t ldy,
                          llmrec_stream_t stream);

/* state[0] = step count (as float bits of an int32), state[1] = lr / (1 - b1^t), state[2] = sqrt(1 - b2^t).
 * llmrec_adamw_advance increments t on the device and refreshes state[1..2]. */
int llmrec_adamw_advance(float* state3, float lr, float beta1, float beta2, llmrec_stream_t stream);
/* decoupled weight decay, then Adam (torch.optim.AdamW, amsgrad=False, maximize=False) */
int llmrec_adamw_f32(int64_t n, float* p, const float* g, float* m, float* v, const float* state3,
                     float lr, float beta1, float beta2, float eps, float weight_decay, llmrec_stream_t stream);

/* the same update for up to LLMREC_ADAMW_MAX_TENSORS parameters in one launch */
#define LLMREC_ADAMW_MAX_TENSORS 16
typedef struct { float* p; const float* g; float* m; float* v; int64_t n;
                 float g_scale;   /* the gradient is g_scale * g (non-zero; 1 = plain): saves a scaling pass over a large table */
                 float* g_out;    /* optional: receives the gradient the update used (g_scale * g) - the parameter's .grad when g is another
                                     buffer (the user table's gradient IS inv * dE_u: no separate axpy launch); NULL = not stored */
} llmrec_adamw_tensor_t;
int llmrec_adamw_multi_f32(int32_t n_tensors, const llmrec_adamw_tensor_t* tensors_host, const float* state3,
                           float lr, float beta1, float beta2, float eps, float weight_decay, llmrec_stream_t stream);
/* The same update and, in the same launch (extra blocks), the row-wise clean-up of scatter targets: for every job j < n_jobs and every
 * b < min(*n_valid_dev, B_cap): dst_j[ids_j[b], 0 .. d_j) = 0 (what llmrec_bpr_multi_zero_rows_f32 does with its own launch; a job may span
 * the adjacent column slices several problems scattered into). The caller orders the launch behind the last reader of those rows. */
#define LLMREC_ZERO_ROWS_MAX_JOBS 24
typedef struct { const int64_t* ids; float* dst; int64_t ldd; int32_t d; } llmrec_zero_rows_job_t;
int llmrec_adamw_multi_zero_rows_f32(int32_t n_tensors, const llmrec_adamw_tensor_t* tensors_host, const float* state3,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int32_t n_jobs, const llmrec_zero_rows_job_t* jobs_host, int32_t B_cap, const int32_t* n_valid_dev,
                                     llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R9/R10  full-rank scoring + masked top-K + hit vectors
 *                                        replaces torch.matmul(E_u[blk], E_i^T), the D2H copy of
 *                                        the 2048 x I score block and the per-user Python
 *                                        set-difference + heapq.nlargest
 *                                        (reference utility/batch_test.py:21-36,83-109,149-157)
 * For each listed user: scores over all items (fp32 MFMA), train items removed (CSR row of the
 * user, ascending columns), top K by (score desc, item id asc). The U x I matrix is never
 * written. Missing entries (fewer than K candidates) are -1 / -inf.
 * K <= LLMREC_TOPK_MAX. d a multiple of 16 and <= 128.
 * ------------------------------------------------------------------------------------------ */
#define LLMREC_TOPK_MAX 64
int llmrec_score_topk_f32(int32_t n_query, const int64_t* query_users,
                          const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                          int64_t n_items, int32_t d,
                          const int32_t* train_rowptr, const int32_t* train_colidx,
                          int32_t K, int32_t* out_idx /* n_query x K */, float* out_score /* n_query x K */,
                          llmrec_stream_t stream);
/* The same with a workspace of llmrec_score_topk_workspace_bytes(n_query, n_items, d) bytes (16-byte aligned):
 *   - the item table is first re-laid in MFMA fragment order (one pass over Ei, written into the workspace), so that every load of
 *     the sweep is one contiguous KB instead of 64 row pieces;
 *   - the user tiles left over after the last full round of one 16-user tile per compute unit are each swept by several
 *     blocks (parts of the item range) whose 64-slot lists are merged by a second launch, spreading the left-over blocks over the device;
 *   - with a train CSR and at most 131 072 items: a block turns the rows of up to two of its users with more than 48 train items into
 *     bitmaps (one word per 32-item tile, in its own slice of the workspace) and reads one word per round instead of walking the row.
 * The lists and scores are the same, bit for bit. workspace == NULL behaves like llmrec_score_topk_f32. */
/* Tuning knob of the bf16 sweep's ITEM PARTS (tables beyond the L2: every user tile is cut into parts of `items` items whose fragments fit an
 * XCD's L2, block ids part-major; csrc/topk.hip plan_parts): 0 = the library's policy (8 MB of fragments per part for tables beyond 131 072 items),
 * -1 = never, else a multiple of 32 >= 1024. Process-wide; set it BEFORE sizing a workspace (llmrec_score_topk_workspace_bytes depends on it). */
int llmrec_topk_set_part_items(int32_t items);
int64_t llmrec_score_topk_workspace_bytes(int32_t n_query, int64_t n_items, int32_t d);
int llmrec_score_topk_ws_f32(int32_t n_query, const int64_t* query_users,
                             const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                             int64_t n_items, int32_t d,
                             const int32_t* train_rowptr, const int32_t* train_colidx,
                             int32_t K, int32_t* out_idx /* n_query x K */, float* out_score /* n_query x K */,
                             void* workspace, int64_t workspace_bytes, llmrec_stream_t stream);
/* debug / test entry: the full score block with the same MFMA arithmetic as above */
/* The same call with the sweep's arithmetic chosen explicitly (round 5). Lists and scores are BIT-IDENTICAL in both modes:
 *   LLMREC_TOPK_MODE_EXACT_SWEEP  every score by the exact-fp32 MFMA chain (what llmrec_score_topk_ws_f32 runs);
 *   LLMREC_TOPK_MODE_PREFILTER    the sweep on bf16 MFMAs (both operands as two round-to-nearest bf16 terms, three products:
 *                                 |s - s'| <= 2^-14 ||u|| ||i||) keeps each user's 64 best items by the UPPER BOUND ub = s' + 2^-14 ||u|| ||i||;
 *                                 those 64 are then scored with the exact fp32 fma chain, re-ranked, and VERIFIED: (64th ub) < (K-th exact
 *                                 score) proves that no other item can be in the top K. User tiles that fail the proof are swept again by the
 *                                 exact kernel (second launch; the other tiles' blocks exit at once). Needs the workspace;
 *                                 K <= LLMREC_TOPK_PREFILTER_MAX_K (else the exact sweep runs). llmrec_score_topk_stats_offset: byte offset inside
 *                                 the workspace of three uint32 words the mode leaves behind - [0] = the drains of the sweep's candidate
 *                                 pools, summed over its blocks (round 6), [1] = the number of user tiles the exact sweep had to redo, [2] = the
 *                                 number of train rows the blocks swept as bitmaps (long rows; all 16 rows of a user tile with dense rows). */
#define LLMREC_TOPK_PREFILTER_MAX_K 56
#define LLMREC_TOPK_MODE_EXACT_SWEEP 0
#define LLMREC_TOPK_MODE_PREFILTER 1
int llmrec_score_topk_mode_f32(int32_t n_query, const int64_t* query_users,
                               const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                               int64_t n_items, int32_t d,
                               const int32_t* train_rowptr, const int32_t* train_colidx,
                               int32_t K, int32_t* out_idx, float* out_score,
                               void* workspace, int64_t workspace_bytes, int32_t mode, llmrec_stream_t stream);
int64_t llmrec_score_topk_stats_offset(int32_t n_query, int64_t n_items);
int llmrec_scores_f32(int32_t n_query, const int64_t* query_users,
                      const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                      int64_t n_items, int32_t d, float* S, int64_t lds, llmrec_stream_t stream);
/* hits[q, j] = 1 if topk_idx[q, j] is in the held-out CSR row of query_users[q]
 * (reference batch_test.py:29-34) */
int llmrec_topk_hits(int32_t n_query, const int64_t* query_users, int32_t K, const int32_t* topk_idx,
                     const int32_t* test_rowptr, const int32_t* test_colidx, uint8_t* hits,
                     llmrec_stream_t stream);
/* R10 (reference utility/metrics.py:8-18,43-87 through batch_test.py:70-80): per-user precision, recall, ndcg and
 * hit-ratio at each of the n_ks <= 8 cut-offs (host array ks), in double, from the hit matrix of llmrec_topk_hits and
 * the ranked lists: out[q][4][n_ks]. ndcg uses the reference's IDCG (the retrieved hit vector sorted descending);
 * precision averages over the ranked items that exist. The caller sums over users (12 doubles leave the device
 * instead of n x K hits). */
int llmrec_topk_metrics(int32_t n_query, const int64_t* query_users, int32_t K, const uint8_t* hits, const int32_t* topk_idx,
                        const int32_t* test_rowptr, int32_t n_ks, const int32_t* ks_host, double* out, llmrec_stream_t stream);
/* Round 6 - a whole evaluation's R10 in two launches (what batch_test.py:160-165 accumulates over the users): out[4][n_ks] = the SUMS over the
 * query users of precision, recall, ndcg, hit-ratio at every cut-off (the caller divides by the number of users), from the ranked lists and the
 * held-out CSR directly - the hit flags and the per-user values stay in registers, the users are added by fixed trees (deterministic). K <= 128.
 * `out` may be device memory or mapped (pinned) host memory: an evaluation graph then ends with the twelve doubles already on the host. */
int64_t llmrec_topk_eval_sums_workspace_bytes(int32_t n_query, int32_t n_ks);
int llmrec_topk_eval_sums(int32_t n_query, const int64_t* query_users, int32_t K, const int32_t* topk_idx, const int32_t* test_rowptr,
                          const int32_t* test_colidx, int32_t n_ks, const int32_t* ks_host, void* workspace, int64_t workspace_bytes,
                          double* out, llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R11  on-device BPR sampler             replaces Data.sample (reference
 *                                        utility/load_data.py:157-195): B distinct users
 *                                        (keyed permutation of the users that have train items),
 *                                        one uniform positive from the user's CSR row, one
 *                                        uniform negative rejected while it is in that row.
 * Counter-based (Philox4x32-10) on (seed, step): reproducible and order-free.
 * ------------------------------------------------------------------------------------------ */
int llmrec_sample_bpr(uint64_t seed, uint64_t step, int64_t n_exist_users, const int64_t* exist_users,
                      int64_t n_items, const int32_t* train_rowptr, const int32_t* train_colidx,
                      int32_t B, int64_t* users, int64_t* pos, int64_t* neg, llmrec_stream_t stream);
/* The whole mini-batch of one step in one single-block launch a HIP graph can replay (the step counter is
 * device memory, advanced by the launch): slots [0, B) = this rank's slice [slice_begin, slice_begin + B) of
 * a global batch of B_global triples drawn exactly as llmrec_sample_bpr(seed, *step_dev, ..., B_global)
 * would; slots [B, B + n_aug) = the LLM-augmented triples of reference main.py:216-224 (n_aug distinct
 * users of the slice, their (aug_pos[u], aug_neg[u]) kept only if both are valid item ids; kept pairs first,
 * in draw order); n_valid_dev[0] = B + kept. users/pos/neg hold B + n_aug entries. */
int llmrec_sample_batch(uint64_t seed, uint64_t* step_dev, int64_t n_exist_users, const int64_t* exist_users,
                        int64_t n_items, const int32_t* train_rowptr, const int32_t* train_colidx,
                        int32_t B_global, int32_t slice_begin, int32_t B, int32_t n_aug,
                        const int64_t* aug_pos, const int64_t* aug_neg,
                        int64_t* users, int64_t* pos, int64_t* neg, int32_t* n_valid_dev, llmrec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py `roofline`; nothing on the product path calls it). One single-lane launch that
 * stores the device's constant-rate real-time counter (s_memrealtime) into *slot when the stream reaches it.
 * A captured HIP graph cannot carry event-record nodes for external events; a pair of these around a launch
 * gives that launch's duration INSIDE the replayed graph. llmrec_timestamp_rate_hz: ticks per second of
 * that counter on the current device (hipDeviceAttributeWallClockRate), <= 0 on error.
 * ------------------------------------------------------------------------------------------ */
int llmrec_timestamp(uint64_t* slot, llmrec_stream_t stream);
int64_t llmrec_timestamp_rate_hz(void);

#ifdef __cplusplus
}
#endif
#endif /* LLMREC_HIP_H */
