// spmm.hip - R2: Y = epilogue(alpha * Z + diag(row_scale) * (P (.) val) * diag(col_scale) * X), fp32, CSR.
// Replaces torch.sparse.mm / torch.mm(sparse, dense) at reference Models.py:57-61 (call sites
// :153-157,162-163,166-167,176-180), the transposed SpMM autograd runs for the backward, and - as fused
// epilogues on the finished row - the row softmax of the last propagation layer (Models.py:176-177), its
// backward, and the "+ mean-term" additions of the hand-written backward (llmrec_amd/fused.py).
//
// HBM / fabric-bound gather kernel, organised for 64-lane wavefronts:
//   * a row of X / Y is d floats; LPR = d/4 lanes (16 for d = 64) each own one float4 of the row, so one
//     wavefront instruction moves 64/LPR complete rows as 16-byte-per-lane accesses (coalesced 256-B rows);
//   * CSR row buckets by length (llmrec_spmm_plan_*, thresholds chosen by the host), all in ONE launch of 512-thread
//     blocks, heaviest blocks first:
//       <= 32 nnz            one lane group per row (4 rows per wavefront at d = 64),
//       <= plan.t_wave       one wavefront per row (lane groups take contiguous parts, butterfly sum),
//       <= plan.t_block      one block (8 wavefronts) per row, waves summed through LDS in wave order,
//       longer               plan.segment-nnz segments, one block each, partial sums added in a fixed order by a second,
//                            tiny launch (only graphs with such hubs pay for it);
//     every summation tree is fixed by (nnz, thresholds, d): results are run-to-run deterministic, no float atomics.
//     (A ticket scheme - the last piece's wave sums the partials behind agent-scope fences, no second launch - was
//     measured and dropped: buffer_wbl2 writes back EVERY dirty line of the XCD's L2, 2.4x slower at 36 M edges and
//     3x slower in situ at Netflix scale, where concurrent kernels keep the L2 dirty.)
//   * wide operands ([rows, 7 x 64] side-feature blocks) can be cut into 64-column SLICES: (row, slice) tasks, so a
//     launch-bound graph gets 7x the parallelism and the latency of a d = 64 product instead of one wave per row;
//   * a lane group loads LPR column indices with one coalesced access and broadcasts them with ds_bpermute
//     (__shfl), then issues UNROLL independent row gathers before accumulating (UNROLL x 16 B in flight per lane);
//   * the adjacency values are not read at all in the reference's case (A = diag(s) R, R binary): a per-row
//     scale is applied once at the end (4 B/nnz of traffic instead of 8-20 B/nnz).
// What bounds it at scale (tools/gatherbench.py, profiles/r02_gatherbench_*.txt): the L2-miss path of the fabric,
// ~7.4 TB/s of 128-B lines from the Infinity Cache or HBM alike, plus ~25 TB/s for the L2 hits; a constant-degree
// graph without any imbalance reaches 35 G gathers/s on the synthetic graphs' popularity - this kernel's ceiling.
#include "common.h"

namespace llmrec {

constexpr int UNROLL = 8;

template <int VEC> struct Vec;
template <> struct Vec<4> {
    float4 v;
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    // the same 16 bytes with the non-temporal cache policy (global_load_dwordx4 ... nt): a hint that the line will not be reused
    __device__ __forceinline__ void load_nt(const float* p) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        v = make_float4(t.x, t.y, t.z, t.w);
    }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ void add(const Vec& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
    __device__ __forceinline__ void fma(float w, const Vec& o) {
        v.x = fmaf(w, o.v.x, v.x); v.y = fmaf(w, o.v.y, v.y); v.z = fmaf(w, o.v.z, v.z); v.w = fmaf(w, o.v.w, v.w);
    }
    __device__ __forceinline__ void scale(float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
    __device__ __forceinline__ void xor_add(int off) {
        v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
        v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    __device__ __forceinline__ float hmax() const { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }
    __device__ __forceinline__ float hsum() const { return (v.x + v.y) + (v.z + v.w); }
    __device__ __forceinline__ float dot(const Vec& o) const { return (v.x * o.v.x + v.y * o.v.y) + (v.z * o.v.z + v.w * o.v.w); }
    __device__ __forceinline__ void exp_sub(float m) { v.x = expf(v.x - m); v.y = expf(v.y - m); v.z = expf(v.z - m); v.w = expf(v.w - m); }
    // this = s * (this - c)    (softmax backward: y * (g - sum(g y)))
    __device__ __forceinline__ void sub_mul(float c, const Vec& s) { v.x = s.v.x * (v.x - c); v.y = s.v.y * (v.y - c); v.z = s.v.z * (v.z - c); v.w = s.v.w * (v.w - c); }
};
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void load_nt(const float* p) { v = __builtin_nontemporal_load(p); }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
    __device__ __forceinline__ void add(const Vec& o) { v += o.v; }
    __device__ __forceinline__ void fma(float w, const Vec& o) { v = fmaf(w, o.v, v); }
    __device__ __forceinline__ void scale(float s) { v *= s; }
    __device__ __forceinline__ void xor_add(int off) { v += __shfl_xor(v, off, 64); }
    __device__ __forceinline__ float hmax() const { return v; }
    __device__ __forceinline__ float hsum() const { return v; }
    __device__ __forceinline__ float dot(const Vec& o) const { return v * o.v; }
    __device__ __forceinline__ void exp_sub(float m) { v = expf(v - m); }
    __device__ __forceinline__ void sub_mul(float c, const Vec& s) { v = s.v * (v - c); }
};

struct SpmmArgs {
    int64_t n_rows;
    const int32_t* rowptr;
    const int32_t* colidx;
    const float* val;
    const float* row_scale;
    const float* col_scale;
    const float* X;
    int64_t ldx;
    float* Y;
    int64_t ldy;
    int32_t d;                   // columns handled per task (= the slice width when the operand is cut into column slices)
    int32_t n_slices;            // column slices of width d: task (row, slice) reads / writes columns [slice d, (slice + 1) d)
    // epilogue: t = alpha * Z[row] + result; Y[row] = op(t)
    int32_t epi_op;
    float alpha;
    const float* Z;
    int64_t ldz;
    const float* S;              // softmax backward: rows of the forward softmax output
    int64_t lds;
    const float* post_scale;     // per-row scale of the output, applied last
    // operand sparsity (llmrec_spmm_epilogue_t): X rows whose byte in x_mask differs from x_active are all-zero and are not read;
    // y_flag[row] := x_active if the row's result can be non-zero (an active X row was gathered, or z_flag[row] == x_active), else 0
    const uint8_t* x_mask;
    int32_t x_active;
    uint8_t* y_flag;
    const uint8_t* z_flag;
    const uint8_t* y_gate;       // rows whose byte != x_active: no active neighbour, zero Z row - written as zeros at once
    const uint8_t* y_needed;     // rows whose byte != x_active are neither computed nor written (short-row range)
    int32_t listed_only;         // no short-row range at all: only the plan's row lists are computed
    int32_t nt_from;             // (NT kernels) X rows with index >= nt_from are gathered with the non-temporal policy: cache hint only
    int32_t no_pipeline;         // llmrec_spmm_epilogue_t.no_pipeline
    int32_t pipe;                // > 1: tasks per lane group of the short-row range, run as a software pipeline (rows_body_pipe)
    int32_t xcd;                 // block -> row map of the short-row / wavefront-row / block-row ranges: XCD-contiguous (xcd_range_logical)
    // plan
    const int32_t* slot_row;     // non-NULL: rowptr / colidx are the PERMUTED CSR of the plan - CSR row `slot` is output row slot_row[slot], the
    int32_t n_short_rows;        // first n_short_rows slots are the lane-group bucket, the plan's lists hold slots (llmrec_spmm_plan_t)
    const int32_t* wave_rows;
    const int32_t* block_rows;
    const int32_t* split_rows;
    const int32_t* split_seg_begin;
    const int32_t* seg_split;
    float* partials;             // [slice][segment][d]
    int32_t n_wave_rows, n_block_rows, n_split_rows, n_segments, segment;
    int32_t blk_seg, blk_block, blk_wave;     // first block of the block-row / wave-row / short-row ranges
};

// Accumulate sum_{j in [s, e)} w_j * X[col_j, chunk columns] into acc, in ascending j order. MASKED: rows of X that are not active
// (a.x_mask) are known to be all-zero and are not fetched; `any` collects whether this lane saw an active column.
// unmasked ranges (round 6): the NEXT chunk's column indices (and adjacency values) are fetched while the current chunk's rows are gathered, the
// col_scale factors of the current chunk ride beside its gathers - one memory round trip per chunk of LPR non-zeros instead of two or three.
// Same gathers, same order of additions: bit-identical to the loop it replaces.
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool NT>
__device__ __forceinline__ void accumulate_range_plain(const SpmmArgs& a, int64_t col0, int32_t s, int32_t e, int gl, Vec<VEC> (&acc)[NCHUNK]) {
    int32_t c_next = 0;
    float v_next = 1.0f;
    if (s + gl < e) { c_next = a.colidx[s + gl]; if (WEIGHTED && a.val) v_next = a.val[s + gl]; }
    for (int32_t base = s; base < e; base += LPR) {
        const int n = min(LPR, e - base);
        const int32_t myc = c_next;
        float myw = v_next;
        const int32_t nb = base + LPR;
        if (nb + gl < e) { c_next = a.colidx[nb + gl]; if (WEIGHTED && a.val) v_next = a.val[nb + gl]; }
        float cs = 1.0f;
        if (WEIGHTED && a.col_scale && gl < n) cs = a.col_scale[myc];        // (in flight beside the gathers below)
        for (int t = 0; t < n; t += UNROLL) {
            Vec<VEC> v[UNROLL][NCHUNK];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int tt = t + u;
                const int32_t c = __shfl(myc, tt & (LPR - 1), LPR);
                const float* xr = a.X + (int64_t)c * a.ldx + col0;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k) {
                    const int col = (k * LPR + gl) * VEC;
                    if (tt < n && col < a.d) {
                        if (NT && c >= a.nt_from) v[u][k].load_nt(xr + col);     // (uniform per lane group) cold row: do not displace the hot set
                        else v[u][k].load(xr + col);
                    } else v[u][k].zero();
                }
            }
            if (WEIGHTED && t == 0) myw = gl < n ? myw * cs : 0.f;            // val[e] * col_scale[c], as the loop it replaces forms it
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float w = 0.f;
                if (WEIGHTED) w = __shfl(myw, (t + u) & (LPR - 1), LPR);
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k) {
                    if (WEIGHTED) acc[k].fma(w, v[u][k]);
                    else acc[k].add(v[u][k]);
                }
            }
        }
    }
}

template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__device__ __forceinline__ void accumulate_range(const SpmmArgs& a, int64_t col0, int32_t s, int32_t e, int gl, Vec<VEC> (&acc)[NCHUNK], int& any) {
    if constexpr (!MASKED) {
        if (!a.no_pipeline) { accumulate_range_plain<LPR, NCHUNK, VEC, WEIGHTED, NT>(a, col0, s, e, gl, acc); return; }
    }
    for (int32_t base = s; base < e; base += LPR) {        const int n = min(LPR, e - base);
        int32_t myc = 0;
        float myw = 0.f;
        int myact = 0;
        if (gl < n) {
            myc = a.colidx[base + gl];
            if (WEIGHTED) {
                myw = a.val ? a.val[base + gl] : 1.0f;
                if (a.col_scale) myw *= a.col_scale[myc];
            }
            if (MASKED) { myact = (int)a.x_mask[myc] == a.x_active; any |= myact; }
        }
        if (MASKED) {
            // only the ACTIVE columns of this chunk are visited (ascending, as in the dense loop: the skipped terms are exact zeros)
            const unsigned long long bal = __ballot(myact != 0);
            const int g0 = (int)(threadIdx.x & 63) / LPR * LPR;
            unsigned long long bits = LPR >= 64 ? bal : ((bal >> g0) & ((1ull << LPR) - 1ull));      // uniform per lane group
            while (bits) {
                Vec<VEC> v[UNROLL][NCHUNK];
                float w[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const bool valid = bits != 0ull;
                    const int j = valid ? __builtin_ctzll(bits) : 0;
                    bits &= bits - 1ull;                                   // (0 stays 0)
                    const int32_t c = __shfl(myc, j, LPR);
                    if (WEIGHTED) w[u] = valid ? __shfl(myw, j, LPR) : 0.f; else w[u] = 0.f;
                    const float* xr = a.X + (int64_t)c * a.ldx + col0;
#pragma unroll
                    for (int k = 0; k < NCHUNK; ++k) {
                        const int col = (k * LPR + gl) * VEC;
                        if (valid && col < a.d) v[u][k].load(xr + col);
                        else v[u][k].zero();
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
                    for (int k = 0; k < NCHUNK; ++k) {
                        if (WEIGHTED) acc[k].fma(w[u], v[u][k]);
                        else acc[k].add(v[u][k]);
                    }
                }
            }
            continue;
        }
        for (int t = 0; t < n; t += UNROLL) {
            Vec<VEC> v[UNROLL][NCHUNK];
            float w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int tt = t + u;
                const int32_t c = __shfl(myc, tt & (LPR - 1), LPR);
                if (WEIGHTED) w[u] = __shfl(myw, tt & (LPR - 1), LPR);
                const float* xr = a.X + (int64_t)c * a.ldx + col0;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k) {
                    const int col = (k * LPR + gl) * VEC;
                    if (tt < n && col < a.d) {
                        if (NT && c >= a.nt_from) v[u][k].load_nt(xr + col);     // (uniform per lane group) cold row: do not displace the hot set
                        else v[u][k].load(xr + col);
                    } else v[u][k].zero();
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k) {
                    if (WEIGHTED) acc[k].fma(w[u], v[u][k]);
                    else acc[k].add(v[u][k]);
                }
            }
        }
    }
}

// the output-row flag of a masked product (see SpmmArgs); returns whether the row is known to be all-zero AND may be written as
// zeros without its epilogue (Z absent or known zero through z_flag; the forward softmax of a zero row is not zero)
__device__ __forceinline__ bool write_row_flag(const SpmmArgs& a, int64_t row, bool any_active, bool write) {
    const bool z = a.z_flag && (int)a.z_flag[row] == a.x_active;
    if (write && a.y_flag) a.y_flag[row] = (any_active || z) ? (uint8_t)a.x_active : (uint8_t)0;
    return !any_active && !z && (a.Z == nullptr || a.z_flag != nullptr) && a.epi_op != LLMREC_SPMM_EPI_SOFTMAX;
}

// The finished row (unscaled sum in acc, held by the LPR lanes of one lane group): scale, init term, epilogue, store.
template <int LPR, int NCHUNK, int VEC>
__device__ __forceinline__ void finish_row(const SpmmArgs& a, int64_t col0, int64_t row, int gl, Vec<VEC> (&acc)[NCHUNK], bool zero_row = false) {
    const float rs = a.row_scale ? a.row_scale[row] : 1.0f;
    float* yr = a.Y + row * a.ldy + col0;
    if (zero_row) {                                              // (uniform per lane group) a masked product's row without an active neighbour
#pragma unroll                                                   // and with an unflagged (= zero) Z row: op(0) = 0 for op in {none, softmax
        for (int k = 0; k < NCHUNK; ++k) {                       // backward} - written without reading Z / S
            const int col = (k * LPR + gl) * VEC;
            acc[k].zero();
            if (col < a.d) acc[k].store(yr + col);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int col = (k * LPR + gl) * VEC;
        if (a.row_scale) acc[k].scale(rs);
        if (a.Z && col < a.d) { Vec<VEC> z; z.load(a.Z + row * a.ldz + col0 + col); acc[k].fma(a.alpha, z); }
        if (col >= a.d) acc[k].zero();
    }
    if (a.epi_op == LLMREC_SPMM_EPI_SOFTMAX) {                   // wave-uniform
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) if ((k * LPR + gl) * VEC < a.d) m = fmaxf(m, acc[k].hmax());
        m = group_max<LPR>(m);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            if ((k * LPR + gl) * VEC < a.d) { acc[k].exp_sub(m); s += acc[k].hsum(); }
        }
        s = group_sum<LPR>(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].scale(inv);
    } else if (a.epi_op == LLMREC_SPMM_EPI_SOFTMAX_BWD) {
        Vec<VEC> y[NCHUNK];
        float c = 0.f;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + gl) * VEC;
            if (col < a.d) { y[k].load(a.S + row * a.lds + col0 + col); c += acc[k].dot(y[k]); }
            else y[k].zero();
        }
        c = group_sum<LPR>(c);
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].sub_mul(c, y[k]);
    }
    if (a.post_scale) {
        const float ps = a.post_scale[row];
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].scale(ps);
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int col = (k * LPR + gl) * VEC;
        if (col < a.d) acc[k].store(yr + col);
    }
}

constexpr int TPB = 512;                                        // threads per block: 8 wavefronts

// one lane group per (row, slice) task; rows with more than LLMREC_SPMM_LONG_ROW nnz belong to the other ranges
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__device__ __forceinline__ void rows_body(const SpmmArgs& a, int64_t block) {
    constexpr int GPB = TPB / LPR;
    const int gl = threadIdx.x & (LPR - 1);
    const int64_t task = block * GPB + (threadIdx.x / LPR);
    const int64_t n_list = a.slot_row ? (int64_t)a.n_short_rows : a.n_rows;
    if (task >= n_list * a.n_slices) return;
    const int64_t slice = task / n_list, slot = task - slice * n_list;         // slice-major: neighbouring lane groups take neighbouring rows
    const int64_t row = a.slot_row ? (int64_t)a.slot_row[slot] : slot;         // (permuted CSR: by descending length class - equal lengths side by side)
    const int64_t col0 = slice * a.d;
    if (a.y_needed && (int)a.y_needed[row] != a.x_active) return;                // (uniform per lane group) nobody reads this row
    const int32_t s = a.rowptr[slot], e = a.rowptr[slot + 1];
    if (!a.slot_row && (e - s) > LLMREC_SPMM_LONG_ROW) return;
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    int any = 0;
    bool zero_row = false;
    if (MASKED && a.y_gate && (int)a.y_gate[row] != a.x_active) {         // (uniform per lane group) gated out: nothing of the row is read
        if (a.y_flag && slice == 0 && gl == 0) a.y_flag[row] = 0;
        finish_row<LPR, NCHUNK, VEC>(a, col0, row, gl, acc, true);
        return;
    }
    if (MASKED) {
        // most rows of a masked product have no active neighbour at all: look at the index list and the mask bytes first (two dependent
        // loads, no per-edge arithmetic) and leave at once when nothing is active - the accumulation loop runs for the few other rows only
        for (int32_t base = s; base < e; base += LPR)
            if (base + gl < e) any |= (int)a.x_mask[a.colidx[base + gl]] == a.x_active;
        const unsigned long long bal = __ballot(any != 0);                // (uniform per lane group below)
        const int g0 = (int)(threadIdx.x & 63) / LPR * LPR;
        const unsigned long long gm = LPR >= 64 ? ~0ull : (((1ull << LPR) - 1ull) << g0);
        const bool any_g = (bal & gm) != 0ull;
        zero_row = write_row_flag(a, row, any_g, slice == 0 && gl == 0);
        if (any_g) accumulate_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, s, e, gl, acc, any);
    } else {
        accumulate_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, s, e, gl, acc, any);
    }
    finish_row<LPR, NCHUNK, VEC>(a, col0, row, gl, acc, zero_row);
}

// PIPELINED short rows (round 6). A (row, slice) task of the lane-group bucket is three DEPENDENT memory round trips - {slot -> output row,
// row pointers} -> column indices -> X rows - for a handful of gathers, and a launch-bound product (the step's [rows, 7 x 64] side products:
// 121 k tasks, ~32 k lane groups resident) pays them once per round of resident blocks: 4 rounds x 3 latencies. Here a lane group takes
// `pipe` tasks (task, task + G, task + 2 G, ...: neighbouring groups keep neighbouring slots) and runs them as a software pipeline: while the
// rows of task i are gathered, the indices of task i + 1 and the row pointers of task i + 2 are already in flight - ONE round trip per task
// after a two-deep prologue, and pipe x fewer blocks to schedule. Same gathers in the same order: bit-identical sums. (The other shape tried -
// one task per row looping over the slices, indices loaded once - was no faster in the step and slower alone: 29.4 vs 28.3 us.)
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__device__ __forceinline__ void rows_body_pipe(const SpmmArgs& a, int64_t block, int64_t n_blocks) {
    constexpr int GPB = TPB / LPR;
    constexpr int NI = (LLMREC_SPMM_LONG_ROW + LPR - 1) / LPR;         // index registers per lane
    const int gl = threadIdx.x & (LPR - 1);
    // (32-bit task arithmetic: the host enables the pipeline only below 2^31 tasks - fewer registers, no 64-bit divisions)
    const int32_t n_list = (int32_t)(a.slot_row ? (int64_t)a.n_short_rows : a.n_rows);
    const int32_t n_tasks = n_list * a.n_slices, G = (int32_t)n_blocks * GPB;
    int32_t t0 = (int32_t)block * GPB + (int32_t)(threadIdx.x / LPR);
    if (t0 >= n_tasks) return;
    auto level1 = [&](int32_t t, int32_t& row, int32_t& s_, int32_t& n_, int32_t& col0) {    // slot -> output row, row pointers (independent loads)
        const int32_t slice = t / n_list, slot = t - slice * n_list;
        row = a.slot_row ? a.slot_row[slot] : slot;
        s_ = a.rowptr[slot]; n_ = a.rowptr[slot + 1] - s_;
        col0 = slice * a.d;
    };
    auto level2 = [&](int32_t s_, int32_t n_, int32_t (&c)[NI]) {                              // the row's column indices (<= LONG_ROW of them)
#pragma unroll
        for (int k = 0; k < NI; ++k) { const int j = k * LPR + gl; c[k] = (j < n_ && n_ <= LLMREC_SPMM_LONG_ROW) ? a.colidx[s_ + j] : 0; }
    };
    int32_t row0, col00, row1 = 0, col01 = 0;
    int32_t s0, n0, s1 = 0, n1 = 0, c0[NI], c1[NI];
    level1(t0, row0, s0, n0, col00);
    level2(s0, n0, c0);
    int32_t t1 = t0 + G;
    bool has1 = t1 < n_tasks;
    if (has1) level1(t1, row1, s1, n1, col01);
    for (;;) {
        int32_t row2 = 0, col02 = 0;
        int32_t s2 = 0, n2 = 0;
        const int32_t t2 = t1 + G;
        const bool has2 = has1 && t2 < n_tasks;
        if (has1) level2(s1, n1, c1);                                   // in flight while task 0's rows are gathered
        if (has2) level1(t2, row2, s2, n2, col02);
        if (n0 <= LLMREC_SPMM_LONG_ROW && !(a.y_needed && (int)a.y_needed[row0] != a.x_active)) {   // (longer: a row of another bucket, plain plans only)
            float w0[NI];
            if (WEIGHTED) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int j = k * LPR + gl;
                    w0[k] = 0.f;
                    if (j < n0) { w0[k] = a.val ? a.val[s0 + j] : 1.0f; if (a.col_scale) w0[k] *= a.col_scale[c0[k]]; }
                }
            }
            Vec<VEC> acc[NCHUNK];
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
#pragma unroll
            for (int ki = 0; ki < NI; ++ki) {
                const int nk = min(LPR, n0 - ki * LPR);                 // indices of this register that are in the row (uniform per lane group)
                for (int t = 0; t < nk; t += UNROLL) {
                    Vec<VEC> v[UNROLL][NCHUNK];
                    float w[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int tt = t + u;
                        const int32_t c = __shfl(c0[ki], tt & (LPR - 1), LPR);
                        const float* xr = a.X + (int64_t)c * a.ldx + col00;
#pragma unroll
                        for (int k = 0; k < NCHUNK; ++k) {
                            const int col = (k * LPR + gl) * VEC;
                            if (tt < nk && col < a.d) v[u][k].load(xr + col);
                            else v[u][k].zero();
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        if (WEIGHTED) w[u] = __shfl(w0[ki], (t + u) & (LPR - 1), LPR);
#pragma unroll
                        for (int k = 0; k < NCHUNK; ++k) {
                            if (WEIGHTED) acc[k].fma(w[u], v[u][k]);
                            else acc[k].add(v[u][k]);
                        }
                    }
                }
            }
            finish_row<LPR, NCHUNK, VEC>(a, col00, row0, gl, acc, false);
        }
        if (!has1) break;
        row0 = row1; col00 = col01; s0 = s1; n0 = n1;
#pragma unroll
        for (int k = 0; k < NI; ++k) c0[k] = c1[k];
        row1 = row2; col01 = col02; s1 = s2; n1 = n2;
        t1 = t2; has1 = has2;
    }
}

// one wavefront over [s, e): the 64/LPR lane groups take contiguous parts, butterfly sum -> every group holds the total
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__device__ __forceinline__ void wave_range(const SpmmArgs& a, int64_t col0, int32_t s, int32_t e, int lane, Vec<VEC> (&acc)[NCHUNK], int& any) {
    constexpr int G = 64 / LPR;
    const int gl = lane & (LPR - 1), g = lane / LPR;
    const int32_t per = (((e - s) + G - 1) / G + LPR - 1) / LPR * LPR;      // multiple of LPR: aligned index loads
    const int32_t gs = min(s + g * per, e), ge = min(gs + per, e);
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    accumulate_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, gs, ge, gl, acc, any);
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].xor_add(off);
    }
}

// (row, slice) tasks of a row list, slice-major
// crow: the row of the CSR arrays (a slot of the permuted CSR when there is one), row: the output row
__device__ __forceinline__ void list_task(const SpmmArgs& a, const int32_t* list, int32_t n_list, int64_t task, int32_t& crow, int32_t& row, int64_t& col0, int32_t& slot) {
    const int64_t slice = task / n_list;
    slot = (int32_t)(task - slice * n_list);
    crow = list[slot];
    row = a.slot_row ? a.slot_row[crow] : crow;
    col0 = slice * a.d;
}

// one wavefront per (row, slice) of the wave-row list
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__device__ __forceinline__ void wave_rows_body(const SpmmArgs& a, int32_t block) {
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)block * (TPB / 64) + (threadIdx.x >> 6);
    if (task >= (int64_t)a.n_wave_rows * a.n_slices) return;
    int32_t crow, row, slot; int64_t col0;
    list_task(a, a.wave_rows, a.n_wave_rows, task, crow, row, col0, slot);
    Vec<VEC> acc[NCHUNK];
    int any = 0;
    bool zero_row = false;
    if (MASKED) {                                                        // as in rows_body: scan the mask first, accumulate only if something is active
        const int32_t rs = a.rowptr[crow], re = a.rowptr[crow + 1];
        for (int32_t base = rs; base < re; base += 64)
            if (base + lane < re) any |= (int)a.x_mask[a.colidx[base + lane]] == a.x_active;
        const bool any_w = __ballot(any != 0) != 0ull;
        zero_row = write_row_flag(a, row, any_w, col0 == 0 && lane == 0);
        if (any_w) wave_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, rs, re, lane, acc, any);
        else {
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
        }
    } else {
        wave_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, a.rowptr[crow], a.rowptr[crow + 1], lane, acc, any);
    }
    if (lane < LPR) finish_row<LPR, NCHUNK, VEC>(a, col0, row, lane, acc, zero_row);
}

// one block over [s, e): the 8 waves take contiguous parts, summed through LDS in wave order; the total ends up in the
// first lane group of wave 0 (returns true there)
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__device__ __forceinline__ bool block_range(const SpmmArgs& a, int64_t col0, int32_t s, int32_t e, float* lds, Vec<VEC> (&acc)[NCHUNK]) {
    constexpr int ROWW = NCHUNK * LPR * VEC;
    constexpr int NW = TPB / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int32_t per = (((e - s) + NW - 1) / NW + 63) / 64 * 64;
    const int32_t ws = min(s + w * per, e), we = min(ws + per, e);
    int any = 0;                                                         // (long rows: their output flag is set unconditionally)
    wave_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, ws, we, lane, acc, any);
    if (w > 0 && lane < LPR) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].store(lds + (w - 1) * ROWW + (k * LPR + lane) * VEC);
    }
    __syncthreads();
    if (w != 0 || lane >= LPR) return false;
#pragma unroll
    for (int ww = 0; ww < NW - 1; ++ww) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) { Vec<VEC> o; o.load(lds + ww * ROWW + (k * LPR + lane) * VEC); acc[k].add(o); }
    }
    return true;
}

// Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with a private L2 (4 MB). With the linear block -> row map every XCD
// therefore walks the WHOLE row range (every 8th block), and on a graph whose ids expose communities each community's X rows are fetched
// into all eight L2s. xcd = 1 (llmrec_spmm_epilogue_t.xcd_contiguous): within a range [base, base + n) of block ids the blocks of XCD x
// (ids = x mod 8) take the x-th contiguous piece of the range's tasks, in id order - neighbouring rows run on ONE XCD and each L2 holds
// the columns of its own eighth of the rows. Only the block -> row map changes: results are bit-identical.
__device__ __forceinline__ int32_t xcd_range_logical(int32_t b, int32_t base, int32_t n) {
    const int x = b & 7, r0 = base & 7;
    int32_t off = 0;
    for (int xx = 0; xx < x; ++xx) { const int fl = (xx - r0) & 7; off += fl < n ? (n - fl + 7) >> 3 : 0; }
    return off + ((b - base - ((x - r0) & 7)) >> 3);
}

// ONE launch: [0, blk_seg) segments of the split rows, [blk_seg, blk_block) block rows, [blk_block, blk_wave) wave
// rows (8 per block), the rest short rows. The heavy blocks come first.
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED, bool MASKED, bool NT>
__global__ __launch_bounds__(TPB) void spmm_kernel(SpmmArgs a) {
    constexpr int ROWW = NCHUNK * LPR * VEC;
    __shared__ __attribute__((aligned(16))) float red_lds[(TPB / 64 - 1) * ROWW];
    const int32_t b = blockIdx.x;
    if (b >= a.blk_wave) {
        const int64_t lb = a.xcd ? (int64_t)xcd_range_logical(b, a.blk_wave, (int32_t)gridDim.x - a.blk_wave) : (int64_t)b - a.blk_wave;
        if (!MASKED && !NT && a.pipe > 1) rows_body_pipe<LPR, NCHUNK, VEC, WEIGHTED>(a, lb, (int64_t)gridDim.x - a.blk_wave);     // (block-uniform)
        else rows_body<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, lb);
        return;
    }
    if (b >= a.blk_block) {
        wave_rows_body<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, a.xcd ? xcd_range_logical(b, a.blk_block, a.blk_wave - a.blk_block) : b - a.blk_block);
        return;
    }
    Vec<VEC> acc[NCHUNK];
    int32_t crow, row, slot; int64_t col0;
    if (b >= a.blk_seg) {
        list_task(a, a.block_rows, a.n_block_rows, a.xcd ? xcd_range_logical(b, a.blk_seg, a.blk_block - a.blk_seg) : b - a.blk_seg, crow, row, col0, slot);
        if (block_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, a.rowptr[crow], a.rowptr[crow + 1], red_lds, acc)) {
            if (MASKED && col0 == 0 && threadIdx.x == 0 && a.y_flag) a.y_flag[row] = (uint8_t)a.x_active;   // conservative: "may be non-zero"
            finish_row<LPR, NCHUNK, VEC>(a, col0, row, threadIdx.x, acc);
        }
        return;
    }
    const int32_t slice = b / a.n_segments, seg = b - slice * a.n_segments;
    slot = a.seg_split[seg];
    crow = a.split_rows[slot];
    row = a.slot_row ? a.slot_row[crow] : crow;
    col0 = (int64_t)slice * a.d;
    const int32_t k_in_row = seg - a.split_seg_begin[slot];
    const int32_t re = a.rowptr[crow + 1];
    const int32_t s = a.rowptr[crow] + k_in_row * a.segment;
    const int32_t e = min(s + a.segment, re);
    if (block_range<LPR, NCHUNK, VEC, WEIGHTED, MASKED, NT>(a, col0, s, e, red_lds, acc)) {
        if (MASKED && col0 == 0 && k_in_row == 0 && threadIdx.x == 0 && a.y_flag) a.y_flag[row] = (uint8_t)a.x_active;   // conservative
        float* pr = a.partials + (int64_t)b * a.d;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + (int)threadIdx.x) * VEC;
            if (col < a.d) acc[k].store(pr + col);
        }
    }
}

// one block per (split row, slice): the 256/LPR lane groups each add every (256/LPR)-th segment partial (ascending),
// then the groups are combined through LDS in group order - a fixed summation tree; then the row's epilogue
template <int LPR, int NCHUNK, int VEC>
__global__ __launch_bounds__(256) void spmm_finalize_kernel(SpmmArgs a) {
    constexpr int G = 256 / LPR;
    constexpr int ROWW = NCHUNK * LPR * VEC;
    __shared__ __attribute__((aligned(16))) float fin_lds[G * ROWW];
    const int gl = threadIdx.x & (LPR - 1), g = threadIdx.x / LPR;
    const int32_t slice = blockIdx.x / a.n_split_rows, slot = blockIdx.x - slice * a.n_split_rows;
    const int32_t crow = a.split_rows[slot];
    const int32_t row = a.slot_row ? a.slot_row[crow] : crow;
    const int32_t deg = a.rowptr[crow + 1] - a.rowptr[crow];
    const int32_t nseg = (deg + a.segment - 1) / a.segment;
    const float* pr = a.partials + ((int64_t)slice * a.n_segments + a.split_seg_begin[slot]) * a.d;
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    for (int s0 = g; s0 < nseg; s0 += G) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + gl) * VEC;
            if (col < a.d) { Vec<VEC> v; v.load(pr + (int64_t)s0 * a.d + col); acc[k].add(v); }
        }
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].store(fin_lds + g * ROWW + (k * LPR + gl) * VEC);
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        acc[k].zero();
        for (int gg = 0; gg < G; ++gg) { Vec<VEC> o; o.load(fin_lds + gg * ROWW + (k * LPR + gl) * VEC); acc[k].add(o); }
    }
    finish_row<LPR, NCHUNK, VEC>(a, (int64_t)slice * a.d, row, gl, acc);
}

// "These rows of A X" with the list and its length on the device (llmrec_spmm_rows_compact_f32): block (j, p) sums piece p of listed row j
// (block_range: 8 waves, LDS reduction in wave order) into partial[j][p]; rows up to COMPACT_MIN_PIECE nnz are one piece.
constexpr int COMPACT_MIN_PIECE = 2048;
__device__ __forceinline__ void compact_pieces(int32_t deg, int32_t& per, int32_t& n_pieces) {
    per = (deg + LLMREC_SPMM_COMPACT_PARTS - 1) / LLMREC_SPMM_COMPACT_PARTS;
    per = (per + 511) / 512 * 512;                                    // (whole 64-index chunks per wave)
    if (per < COMPACT_MIN_PIECE) per = COMPACT_MIN_PIECE;
    n_pieces = deg > 0 ? (deg + per - 1) / per : 0;
}
template <int LPR, int NCHUNK, int VEC>
__global__ __launch_bounds__(TPB) void spmm_rows_compact_kernel(SpmmArgs a, const int32_t* __restrict__ row_list, const int32_t* __restrict__ n_list,
                                                                float* __restrict__ partial) {
    constexpr int ROWW = NCHUNK * LPR * VEC;
    __shared__ __attribute__((aligned(16))) float red_lds[(TPB / 64 - 1) * ROWW];
    const int j = blockIdx.x, p = blockIdx.y;
    if (j >= n_list[0]) return;                                         // block-uniform
    const int32_t row = row_list[j];
    const int32_t s = a.rowptr[row], e = a.rowptr[row + 1];
    int32_t per, n_pieces;
    compact_pieces(e - s, per, n_pieces);
    if (p >= n_pieces) return;
    Vec<VEC> acc[NCHUNK];
    if (block_range<LPR, NCHUNK, VEC, false, false, false>(a, 0, s + p * per, min(s + (p + 1) * per, e), red_lds, acc)) {
        float* pr = partial + ((int64_t)j * LLMREC_SPMM_COMPACT_PARTS + p) * a.d;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + (int)threadIdx.x) * VEC;
            if (col < a.d) acc[k].store(pr + col);
        }
    }
}
// one lane group per slot j < capacity: the pieces of row j in piece order, the row scale; zeros past the list's end
template <int LPR, int NCHUNK, int VEC>
__global__ __launch_bounds__(256) void spmm_rows_compact_finish_kernel(SpmmArgs a, const int32_t* __restrict__ row_list, const int32_t* __restrict__ n_list,
                                                                       int capacity, const float* __restrict__ partial, float* __restrict__ out, int64_t ldo) {
    const int gl = threadIdx.x & (LPR - 1);
    const int j = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (j >= capacity) return;
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    float rs = 0.f;
    if (j < n_list[0]) {
        const int32_t row = row_list[j];
        int32_t per, n_pieces;
        compact_pieces(a.rowptr[row + 1] - a.rowptr[row], per, n_pieces);
        for (int p = 0; p < n_pieces; ++p) {
            const float* pr = partial + ((int64_t)j * LLMREC_SPMM_COMPACT_PARTS + p) * a.d;
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k) {
                const int col = (k * LPR + gl) * VEC;
                if (col < a.d) { Vec<VEC> v; v.load(pr + col); acc[k].add(v); }
            }
        }
        rs = a.row_scale ? a.row_scale[row] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int col = (k * LPR + gl) * VEC;
        acc[k].scale(rs);
        if (col < a.d) acc[k].store(out + (int64_t)j * ldo + col);
    }
}

template <int LPR, int NCHUNK, int VEC>
static int launch_rows_compact(SpmmArgs& a, const int32_t* row_list, const int32_t* n_list, int capacity, float* out, int64_t ldo, float* partial,
                               hipStream_t stream) {
    spmm_rows_compact_kernel<LPR, NCHUNK, VEC><<<dim3((unsigned)capacity, LLMREC_SPMM_COMPACT_PARTS), TPB, 0, stream>>>(a, row_list, n_list, partial);
    LLMREC_LAUNCH_CHECK();
    spmm_rows_compact_finish_kernel<LPR, NCHUNK, VEC><<<(unsigned)ceil_div(capacity, 256 / LPR), 256, 0, stream>>>(a, row_list, n_list, capacity, partial, out, ldo);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

template <int LPR, int NCHUNK, int VEC>
static int launch_spmm(SpmmArgs& a, hipStream_t stream) {
    const bool weighted = a.val != nullptr || a.col_scale != nullptr;
    constexpr int GPB = TPB / LPR;
    const int64_t S = a.n_slices;
    // unmasked products: the short rows' tasks as software pipelines (rows_body_pipe) - as many tasks per lane group as keep ~4 blocks per CU
    const int64_t short_tasks = a.listed_only ? 0 : (a.slot_row ? (int64_t)a.n_short_rows : a.n_rows) * S;
    int64_t pipe = 1;
    if (!a.x_mask && !(a.nt_from > 0 && !weighted) && !a.no_pipeline) {
        pipe = short_tasks + (int64_t)GPB * 8192 < (1ll << 31) ? ceil_div(short_tasks, (int64_t)GPB * 1024) : 1;      // (32-bit task ids in the kernel)
        if (pipe > 8) pipe = 8;
        if (pipe < 1) pipe = 1;
    }
    a.pipe = (int32_t)pipe;
    const int64_t row_blocks = ceil_div(short_tasks, GPB * pipe);
    const int64_t wave_blocks = ceil_div(a.n_wave_rows * S, TPB / 64);
    const int64_t total = a.n_segments * S + a.n_block_rows * S + wave_blocks + row_blocks;
    if (total > 0x7fffffffll) { set_error("spmm: too many rows for one launch"); return LLMREC_EUNSUPPORTED; }
    a.blk_seg = (int32_t)(a.n_segments * S);
    a.blk_block = a.blk_seg + (int32_t)(a.n_block_rows * S);
    a.blk_wave = a.blk_block + (int32_t)wave_blocks;
    if (total > 0) {
        if (a.x_mask) {
            if (weighted) spmm_kernel<LPR, NCHUNK, VEC, true, true, false><<<(unsigned)total, TPB, 0, stream>>>(a);
            else spmm_kernel<LPR, NCHUNK, VEC, false, true, false><<<(unsigned)total, TPB, 0, stream>>>(a);
        } else if (a.nt_from > 0 && !weighted) {             // cache-policy split (pattern-only operands: the propagation products)
            spmm_kernel<LPR, NCHUNK, VEC, false, false, true><<<(unsigned)total, TPB, 0, stream>>>(a);
        } else {
            if (weighted) spmm_kernel<LPR, NCHUNK, VEC, true, false, false><<<(unsigned)total, TPB, 0, stream>>>(a);
            else spmm_kernel<LPR, NCHUNK, VEC, false, false, false><<<(unsigned)total, TPB, 0, stream>>>(a);
        }
        LLMREC_LAUNCH_CHECK();
    }
    if (a.n_split_rows > 0) {
        spmm_finalize_kernel<LPR, NCHUNK, VEC><<<(unsigned)(a.n_split_rows * S), 256, 0, stream>>>(a);
        LLMREC_LAUNCH_CHECK();
    }
    return LLMREC_OK;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" int llmrec_spmm_f32(int64_t n_rows, int64_t n_cols,
                               const int32_t* rowptr, const int32_t* colidx, const float* val,
                               const float* row_scale, const float* col_scale,
                               const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t d, int32_t slice_width,
                               const llmrec_spmm_plan_t* plan_host, float* partials,
                               const llmrec_spmm_epilogue_t* epilogue_host, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && d >= 0, "spmm: negative size");
    if (n_rows == 0 || d == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(rowptr && Y && ldy >= d && ldx >= d, "spmm: null pointer or ld < d");
    LLMREC_CHECK_ARG(plan_host, "spmm: a row plan is required (llmrec_spmm_plan_count / _fill)");
    const llmrec_spmm_plan_t& p = *plan_host;
    LLMREC_CHECK_ARG(p.n_wave_rows >= 0 && p.n_block_rows >= 0 && p.n_split_rows >= 0 && p.n_segments >= 0 && p.segment >= LLMREC_SPMM_LONG_ROW,
                     "spmm: bad plan sizes");
    LLMREC_CHECK_ARG((p.n_wave_rows == 0 || p.wave_rows) && (p.n_block_rows == 0 || p.block_rows) &&
                     (p.n_split_rows == 0 || (p.split_rows && p.split_seg_begin && p.seg_split && partials)), "spmm: row plan incomplete");
    LLMREC_CHECK_ARG(slice_width == 0 || (slice_width > 0 && d % slice_width == 0), "spmm: slice_width must divide d");
    SpmmArgs a = {};
    a.n_rows = n_rows; a.rowptr = rowptr; a.colidx = colidx; a.val = val; a.row_scale = row_scale;
    a.col_scale = col_scale; a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy;
    a.d = slice_width > 0 ? slice_width : d;
    a.n_slices = slice_width > 0 ? d / slice_width : 1;
    a.wave_rows = p.wave_rows; a.block_rows = p.block_rows; a.split_rows = p.split_rows; a.split_seg_begin = p.split_seg_begin;
    a.seg_split = p.seg_split; a.partials = partials;
    LLMREC_CHECK_ARG(p.n_short_rows >= 0 && p.n_short_rows <= n_rows, "spmm: bad short-row count");
    LLMREC_CHECK_ARG(!p.slot_row || (!val && !(epilogue_host && epilogue_host->rows_listed_only)), "spmm: a permuted CSR is pattern-only and complete");
    a.slot_row = p.slot_row; a.n_short_rows = p.slot_row ? p.n_short_rows : 0;
    a.n_wave_rows = p.n_wave_rows; a.n_block_rows = p.n_block_rows; a.n_split_rows = p.n_split_rows; a.n_segments = p.n_segments;
    a.segment = p.segment;
    a.epi_op = LLMREC_SPMM_EPI_NONE; a.alpha = 0.f;
    bool epi_aligned = true;
    if (epilogue_host) {
        const llmrec_spmm_epilogue_t& e = *epilogue_host;
        LLMREC_CHECK_ARG(e.op >= LLMREC_SPMM_EPI_NONE && e.op <= LLMREC_SPMM_EPI_SOFTMAX_BWD, "spmm: unknown epilogue op %d", e.op);
        LLMREC_CHECK_ARG(e.op == LLMREC_SPMM_EPI_NONE || a.n_slices == 1, "spmm: the softmax epilogues need the whole row (slice_width = 0)");
        LLMREC_CHECK_ARG(!e.Z || e.ldz >= d, "spmm: epilogue Z with ld < d");
        LLMREC_CHECK_ARG(e.op != LLMREC_SPMM_EPI_SOFTMAX_BWD || (e.S && e.lds >= d), "spmm: softmax backward needs S with ld >= d");
        a.epi_op = e.op; a.alpha = e.alpha; a.Z = e.Z; a.ldz = e.ldz; a.S = e.S; a.lds = e.lds; a.post_scale = e.post_scale;
        a.listed_only = e.rows_listed_only != 0;
        LLMREC_CHECK_ARG(e.x_nt_from_row >= 0, "spmm: negative x_nt_from_row");
        a.nt_from = e.x_nt_from_row;
        a.xcd = e.xcd_contiguous != 0;
        a.no_pipeline = e.no_pipeline != 0;
        if (e.y_row_needed) {
            LLMREC_CHECK_ARG(e.x_mask_active >= 1 && e.x_mask_active <= 255, "spmm: y_row_needed needs x_mask_active in 1..255");
            a.y_needed = e.y_row_needed; a.x_active = e.x_mask_active;
        }
        if (e.x_row_mask) {
            LLMREC_CHECK_ARG(e.x_mask_active >= 1 && e.x_mask_active <= 255, "spmm: x_mask_active must be a byte value 1..255");
            a.x_mask = e.x_row_mask; a.x_active = e.x_mask_active; a.y_flag = e.y_row_flag; a.z_flag = e.z_row_flag;
            LLMREC_CHECK_ARG(!e.y_row_gate || e.op != LLMREC_SPMM_EPI_SOFTMAX, "spmm: y_row_gate with the forward softmax (a zero row is not a zero output)");
            a.y_gate = e.y_row_gate;
        } else {
            LLMREC_CHECK_ARG(!e.y_row_flag, "spmm: y_row_flag needs x_row_mask");
        }
        epi_aligned = (!e.Z || (e.ldz % 4 == 0 && (uintptr_t)e.Z % 16 == 0)) && (!e.S || (e.lds % 4 == 0 && (uintptr_t)e.S % 16 == 0));
    }
    const int dd = a.d;
    const bool vec4 = (dd % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && epi_aligned &&
                      (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)partials) % 16 == 0);
    if (vec4) {
        if (dd <= 16) return launch_spmm<4, 1, 4>(a, stream);
        if (dd <= 32) return launch_spmm<8, 1, 4>(a, stream);
        if (dd <= 64) return launch_spmm<16, 1, 4>(a, stream);
        if (dd <= 128) return launch_spmm<32, 1, 4>(a, stream);
        if (dd <= 256) return launch_spmm<64, 1, 4>(a, stream);
        if (dd <= 512) return launch_spmm<64, 2, 4>(a, stream);
        if (dd <= 1024) return launch_spmm<64, 4, 4>(a, stream);
    } else {
        if (dd <= 16) return launch_spmm<16, 1, 1>(a, stream);
        if (dd <= 64) return launch_spmm<64, 1, 1>(a, stream);
        if (dd <= 256) return launch_spmm<64, 4, 1>(a, stream);
    }
    set_error("spmm: d = %d outside the compiled kernel family (vec4 = %d)", dd, (int)vec4);
    return LLMREC_EUNSUPPORTED;
}

extern "C" int64_t llmrec_spmm_rows_compact_workspace_bytes(int64_t capacity, int32_t d) {
    if (capacity < 0 || d <= 0) return -1;
    return align_up(capacity * LLMREC_SPMM_COMPACT_PARTS * (int64_t)d * 4, 256);
}

extern "C" int llmrec_spmm_rows_compact_f32(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* colidx, const float* row_scale,
                                            const float* X, int64_t ldx, int32_t d, const int32_t* row_list, const int32_t* n_list_dev,
                                            int32_t capacity, float* out, int64_t ldo, void* workspace, int64_t workspace_bytes,
                                            llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && d > 0 && capacity >= 0, "spmm_rows_compact: bad sizes");
    if (capacity == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(rowptr && colidx && X && row_list && n_list_dev && out && ldx >= d && ldo >= d, "spmm_rows_compact: null pointer or ld < d");
    LLMREC_CHECK_ARG(capacity <= 65535 * 16, "spmm_rows_compact: capacity too large for one launch");
    if (!workspace || workspace_bytes < llmrec_spmm_rows_compact_workspace_bytes(capacity, d)) {
        set_error("spmm_rows_compact: workspace %lld < %lld", (long long)workspace_bytes, (long long)llmrec_spmm_rows_compact_workspace_bytes(capacity, d));
        return LLMREC_EWORKSPACE;
    }
    SpmmArgs a = {};
    a.n_rows = n_rows; a.rowptr = rowptr; a.colidx = colidx; a.row_scale = row_scale; a.X = X; a.ldx = ldx; a.d = d; a.n_slices = 1;
    hipStream_t stream = (hipStream_t)stream_;
    float* partial = (float*)workspace;
    const bool vec4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)X | (uintptr_t)out | (uintptr_t)partial) % 16 == 0);
    if (vec4) {
        if (d <= 16) return launch_rows_compact<4, 1, 4>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
        if (d <= 32) return launch_rows_compact<8, 1, 4>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
        if (d <= 64) return launch_rows_compact<16, 1, 4>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
        if (d <= 128) return launch_rows_compact<32, 1, 4>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
        if (d <= 256) return launch_rows_compact<64, 1, 4>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
    } else {
        if (d <= 16) return launch_rows_compact<16, 1, 1>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
        if (d <= 64) return launch_rows_compact<64, 1, 1>(a, row_list, n_list_dev, capacity, out, ldo, partial, stream);
    }
    set_error("spmm_rows_compact: d = %d outside the compiled kernel family (vec4 = %d)", d, (int)vec4);
    return LLMREC_EUNSUPPORTED;
}
