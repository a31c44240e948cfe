// spmm.hip - R2: Y = diag(row_scale) * (P (.) val) * diag(col_scale) * X, fp32, CSR.
// Replaces torch.sparse.mm / torch.mm(sparse, dense) at reference Models.py:57-61 (call sites
// :153-157,162-163,166-167,176-180) and the transposed SpMM autograd runs for the backward.
//
// HBM-bound gather kernel, organised for 64-lane wavefronts:
//   * a row of X / Y is d floats; LPR = d/4 lanes (16 for d = 64) each own one float4 of the row,
//     so one wavefront instruction moves 64/LPR complete rows as 16-byte-per-lane accesses
//     (coalesced 256-B row segments);
//   * "row buckets": rows with <= LLMREC_SPMM_LONG_ROW nnz are processed one lane-group per row
//     (4 rows per wavefront at d = 64); longer rows are cut into LLMREC_SPMM_SEGMENT-nnz segments,
//     one wavefront per segment, whose partial sums a third kernel adds in a fixed order
//     (deterministic, no float atomics);
//   * a lane group loads LPR column indices with one coalesced access and broadcasts them with
//     ds_bpermute (__shfl), then issues UNROLL independent row gathers before accumulating, so
//     each lane keeps UNROLL x 16 B in flight;
//   * the adjacency values are not read at all in the reference's case (A = diag(s) R, R binary):
//     a per-row scale is applied once at the end (4 B/nnz of traffic instead of 8-20 B/nnz).
#include "common.h"

namespace llmrec {

constexpr int UNROLL = 8;

template <int VEC> struct Vec;
template <> struct Vec<4> {
    float4 v;
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
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
};
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
    __device__ __forceinline__ void add(const Vec& o) { v += o.v; }
    __device__ __forceinline__ void fma(float w, const Vec& o) { v = fmaf(w, o.v, v); }
    __device__ __forceinline__ void scale(float s) { v *= s; }
    __device__ __forceinline__ void xor_add(int off) { v += __shfl_xor(v, off, 64); }
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
    int32_t d;
    int32_t accumulate;          // 1: Y += result
    int32_t skip_long;           // 1: rows longer than LLMREC_SPMM_LONG_ROW are left to the segment pass
    const int32_t* long_rows;
    const int32_t* long_seg_begin;
    const int32_t* seg_long;
    float* partials;
};

// Accumulate sum_{j in [s, e)} w_j * X[col_j, chunk columns] into acc, in ascending j order.
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__device__ __forceinline__ void accumulate_range(const SpmmArgs& a, int32_t s, int32_t e, int gl, Vec<VEC> (&acc)[NCHUNK]) {
    for (int32_t base = s; base < e; base += LPR) {
        const int n = min(LPR, e - base);
        int32_t myc = 0;
        float myw = 0.f;
        if (gl < n) {
            myc = a.colidx[base + gl];
            if (WEIGHTED) {
                myw = a.val ? a.val[base + gl] : 1.0f;
                if (a.col_scale) myw *= a.col_scale[myc];
            }
        }
        for (int t = 0; t < n; t += UNROLL) {
            Vec<VEC> v[UNROLL][NCHUNK];
            float w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int tt = t + u;
                const int32_t c = __shfl(myc, tt & (LPR - 1), LPR);
                if (WEIGHTED) w[u] = __shfl(myw, tt & (LPR - 1), LPR);
                const float* xr = a.X + (int64_t)c * a.ldx;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k) {
                    const int col = (k * LPR + gl) * VEC;
                    if (tt < n && col < a.d) v[u][k].load(xr + col);
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
    }
}

// one lane group per row
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__device__ __forceinline__ void rows_body(const SpmmArgs& a, int64_t block) {
    constexpr int GPB = 256 / LPR;
    const int gl = threadIdx.x & (LPR - 1);
    const int64_t row = block * GPB + (threadIdx.x / LPR);
    if (row >= a.n_rows) return;
    const int32_t s = a.rowptr[row], e = a.rowptr[row + 1];
    if (a.skip_long && (e - s) > LLMREC_SPMM_LONG_ROW) return;
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    accumulate_range<LPR, NCHUNK, VEC, WEIGHTED>(a, s, e, gl, acc);
    const float rs = a.row_scale ? a.row_scale[row] : 1.0f;
    float* yr = a.Y + row * a.ldy;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int col = (k * LPR + gl) * VEC;
        if (col < a.d) {
            if (a.row_scale) acc[k].scale(rs);
            if (a.accumulate) { Vec<VEC> old; old.load(yr + col); acc[k].add(old); }
            acc[k].store(yr + col);
        }
    }
}

template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__global__ __launch_bounds__(256) void spmm_rows_kernel(SpmmArgs a) {
    rows_body<LPR, NCHUNK, VEC, WEIGHTED>(a, blockIdx.x);
}

// one wavefront per segment of a long row; the 64/LPR lane groups take contiguous quarters
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__device__ __forceinline__ void segments_body(const SpmmArgs& a, int32_t n_seg, int32_t block) {
    constexpr int G = 64 / LPR;                       // lane groups per wavefront
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const int g = lane / LPR;
    const int32_t seg = block * 4 + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int32_t slot = a.seg_long[seg];
    const int32_t row = a.long_rows[slot];
    const int32_t k_in_row = seg - a.long_seg_begin[slot];
    const int32_t rs_ = a.rowptr[row], re_ = a.rowptr[row + 1];
    const int32_t s = rs_ + k_in_row * LLMREC_SPMM_SEGMENT;
    const int32_t e = min(s + LLMREC_SPMM_SEGMENT, re_);
    constexpr int PER_G = LLMREC_SPMM_SEGMENT / G;
    const int32_t gs = min(s + g * PER_G, e), ge = min(gs + PER_G, e);
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    accumulate_range<LPR, NCHUNK, VEC, WEIGHTED>(a, gs, ge, gl, acc);
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) acc[k].xor_add(off);
    }
    if (g == 0) {
        float* pr = a.partials + (int64_t)seg * a.d;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + gl) * VEC;
            if (col < a.d) acc[k].store(pr + col);
        }
    }
}

// rows and segments are independent: one launch, blocks [0, seg_blocks) take four segments each (the heavy
// blocks first), the rest the short rows (the 20 SpMMs of a Netflix-scale step are latency-bound; a
// dependent launch costs ~7 us)
template <int LPR, int NCHUNK, int VEC, bool WEIGHTED>
__global__ __launch_bounds__(256) void spmm_rows_segments_kernel(SpmmArgs a, int32_t seg_blocks, int32_t n_seg) {
    if ((int32_t)blockIdx.x < seg_blocks) segments_body<LPR, NCHUNK, VEC, WEIGHTED>(a, n_seg, blockIdx.x);
    else rows_body<LPR, NCHUNK, VEC, WEIGHTED>(a, (int64_t)blockIdx.x - seg_blocks);
}

// one block per long row: the 256/LPR lane groups each add every (256/LPR)-th segment partial
// (ascending), then the groups are combined through LDS in group order - a fixed summation tree
template <int LPR, int NCHUNK, int VEC>
__global__ __launch_bounds__(256) void spmm_finalize_kernel(SpmmArgs a) {
    constexpr int G = 256 / LPR;
    extern __shared__ __attribute__((aligned(16))) float fin_lds[];   // [G][NCHUNK * LPR * VEC]
    constexpr int ROWW = NCHUNK * LPR * VEC;
    const int gl = threadIdx.x & (LPR - 1), g = threadIdx.x / LPR;
    const int32_t slot = blockIdx.x;
    const int32_t row = a.long_rows[slot];
    const int32_t deg = a.rowptr[row + 1] - a.rowptr[row];
    const int32_t nseg = (deg + LLMREC_SPMM_SEGMENT - 1) / LLMREC_SPMM_SEGMENT;
    const float* pr = a.partials + (int64_t)a.long_seg_begin[slot] * a.d;
    Vec<VEC> acc[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].zero();
    for (int s0 = g; s0 < nseg; s0 += 4 * G) {
        Vec<VEC> v[4][NCHUNK];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sidx = s0 + u * G;
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k) {
                const int col = (k * LPR + gl) * VEC;
                if (sidx < nseg && col < a.d) v[u][k].load(pr + (int64_t)sidx * a.d + col);
                else v[u][k].zero();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k) acc[k].add(v[u][k]);
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) acc[k].store(fin_lds + g * ROWW + (k * LPR + gl) * VEC);
    __syncthreads();
    if (g == 0) {
        const float rs = a.row_scale ? a.row_scale[row] : 1.0f;
        float* yr = a.Y + (int64_t)row * a.ldy;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int col = (k * LPR + gl) * VEC;
            if (col >= a.d) continue;
            Vec<VEC> t;
            t.zero();
            for (int gg = 0; gg < G; ++gg) { Vec<VEC> o; o.load(fin_lds + gg * ROWW + col); t.add(o); }
            if (a.row_scale) t.scale(rs);
            if (a.accumulate) { Vec<VEC> old; old.load(yr + col); t.add(old); }
            t.store(yr + col);
        }
    }
}

template <int LPR, int NCHUNK, int VEC>
static int launch_spmm(const SpmmArgs& a, int32_t n_long, int32_t n_seg, hipStream_t stream) {
    const bool weighted = a.val != nullptr || a.col_scale != nullptr;
    constexpr int GPB = 256 / LPR;
    const int64_t blocks = ceil_div(a.n_rows, GPB);
    if (blocks > 0x7fffffffll) { set_error("spmm: too many rows for one launch"); return LLMREC_EUNSUPPORTED; }
    const int sb = (int)ceil_div(n_seg, 4);
    if (n_long > 0 && blocks + sb <= 0x7fffffffll) {
        const int grid = (int)(blocks + sb);
        if (weighted) spmm_rows_segments_kernel<LPR, NCHUNK, VEC, true><<<grid, 256, 0, stream>>>(a, sb, n_seg);
        else spmm_rows_segments_kernel<LPR, NCHUNK, VEC, false><<<grid, 256, 0, stream>>>(a, sb, n_seg);
        LLMREC_LAUNCH_CHECK();
        spmm_finalize_kernel<LPR, NCHUNK, VEC><<<n_long, 256, sizeof(float) * (256 / LPR) * NCHUNK * LPR * VEC, stream>>>(a);
        LLMREC_LAUNCH_CHECK();
        return LLMREC_OK;
    }
    if (blocks > 0) {
        if (weighted) spmm_rows_kernel<LPR, NCHUNK, VEC, true><<<(int)blocks, 256, 0, stream>>>(a);
        else spmm_rows_kernel<LPR, NCHUNK, VEC, false><<<(int)blocks, 256, 0, stream>>>(a);
        LLMREC_LAUNCH_CHECK();
    }
    return LLMREC_OK;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" int llmrec_spmm_f32(int64_t n_rows, int64_t n_cols,
                               const int32_t* rowptr, const int32_t* colidx, const float* val,
                               const float* row_scale, const float* col_scale,
                               const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t d,
                               int32_t n_long, const int32_t* long_rows, const int32_t* long_seg_begin,
                               int32_t n_seg, const int32_t* seg_long, float* partials,
                               int32_t accumulate, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && d >= 0, "spmm: negative size");
    if (n_rows == 0 || d == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(rowptr && Y && ldy >= d && ldx >= d, "spmm: null pointer or ld < d");
    LLMREC_CHECK_ARG(n_long >= 0 && n_seg >= 0, "spmm: negative plan size");
    LLMREC_CHECK_ARG(n_long == 0 || (long_rows && long_seg_begin && seg_long && partials), "spmm: long-row plan incomplete");
    SpmmArgs a;
    a.n_rows = n_rows; a.rowptr = rowptr; a.colidx = colidx; a.val = val; a.row_scale = row_scale;
    a.col_scale = col_scale; a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.d = d;
    a.accumulate = accumulate; a.skip_long = n_long > 0; a.long_rows = long_rows; a.long_seg_begin = long_seg_begin;
    a.seg_long = seg_long; a.partials = partials;
    const bool vec4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                      (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)partials) % 16 == 0);
    if (vec4) {
        if (d <= 16) return launch_spmm<4, 1, 4>(a, n_long, n_seg, stream);
        if (d <= 32) return launch_spmm<8, 1, 4>(a, n_long, n_seg, stream);
        if (d <= 64) return launch_spmm<16, 1, 4>(a, n_long, n_seg, stream);
        if (d <= 128) return launch_spmm<32, 1, 4>(a, n_long, n_seg, stream);
        if (d <= 256) return launch_spmm<64, 1, 4>(a, n_long, n_seg, stream);
        if (d <= 512) return launch_spmm<64, 2, 4>(a, n_long, n_seg, stream);
        if (d <= 1024) return launch_spmm<64, 4, 4>(a, n_long, n_seg, stream);
    } else {
        if (d <= 16) return launch_spmm<16, 1, 1>(a, n_long, n_seg, stream);
        if (d <= 64) return launch_spmm<64, 1, 1>(a, n_long, n_seg, stream);
        if (d <= 256) return launch_spmm<64, 4, 1>(a, n_long, n_seg, stream);
    }
    set_error("spmm: d = %d outside the compiled kernel family (vec4 = %d)", d, (int)vec4);
    return LLMREC_EUNSUPPORTED;
}
