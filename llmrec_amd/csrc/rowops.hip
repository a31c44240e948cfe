// rowops.hip - R3/R6/R8: row-wise kernels (softmax, normalise-and-add fusion, squared-norm
// regulariser, axpy, AdamW). All HBM-bound elementwise/row-reduction work; one 16-lane group per
// row with 16-byte accesses, so a wavefront instruction touches 4 complete 256-B rows at d = 64.
// Replaces nn.Softmax / torch.mean(torch.stack) / F.normalize + scaled adds (reference
// Models.py:176-177,185-197), (x**2).sum() (main.py:151-156) and AdamW.step (main.py:100-104,278).
#include "common.h"
#include <type_traits>

namespace llmrec {

constexpr int RL = 16;                 // lanes per row
constexpr int ROWS_PER_BLOCK = 256 / RL;

// A row held in registers: lane gl owns elements {(k*RL + gl)*VEC + q}, k < NCHUNK, q < VEC.
template <int VEC, int NCHUNK>
struct RowReg {
    float x[NCHUNK][VEC];
    __device__ __forceinline__ void load(const float* row, int gl, int d) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int c = (k * RL + gl) * VEC;
            if (VEC == 4) {
                float4 v = (c < d) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                x[k][0] = v.x; x[k][1 % VEC] = v.y; x[k][2 % VEC] = v.z; x[k][3 % VEC] = v.w;
            } else {
                x[k][0] = (c < d) ? row[c] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* row, int gl, int d) const {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k) {
            const int c = (k * RL + gl) * VEC;
            if (c < d) {
                if (VEC == 4) *reinterpret_cast<float4*>(row + c) = make_float4(x[k][0], x[k][1 % VEC], x[k][2 % VEC], x[k][3 % VEC]);
                else row[c] = x[k][0];
            }
        }
    }
    __device__ __forceinline__ void fill(float v) {
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) x[k][q] = v;
    }
    __device__ __forceinline__ float dot(const RowReg& o) const {      // full-row dot (all lanes get it)
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) s = fmaf(x[k][q], o.x[k][q], s);
        return group_sum<RL>(s);
    }
};

#define ROW_LOOP_HEADER                                                                           \
    const int gl = threadIdx.x & (RL - 1);                                                        \
    const int64_t row_stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;                               \
    for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / RL; row < rows; row += row_stride)

// ---------------------------------------------------------------------------------------------
// softmax over d
// ---------------------------------------------------------------------------------------------
template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int64_t rows, int d, const float* __restrict__ Z, int64_t ldz,
                                                          float* __restrict__ Y, int64_t ldy) {
    ROW_LOOP_HEADER {
        RowReg<VEC, NCHUNK> z;
        z.load(Z + row * ldz, gl, d);
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                if ((k * RL + gl) * VEC + q < d) mx = fmaxf(mx, z.x[k][q]);
        mx = group_max<RL>(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const bool in = (k * RL + gl) * VEC + q < d;
                z.x[k][q] = in ? expf(z.x[k][q] - mx) : 0.f;
                s += z.x[k][q];
            }
        s = group_sum<RL>(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) z.x[k][q] *= inv;
        z.store(Y + row * ldy, gl, d);
    }
}

template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int64_t rows, int d, const float* __restrict__ Y, int64_t ldy,
                                                          const float* __restrict__ dY, int64_t lddy,
                                                          float* __restrict__ dZ, int64_t lddz, float alpha) {
    ROW_LOOP_HEADER {
        RowReg<VEC, NCHUNK> y, g;
        y.load(Y + row * ldy, gl, d);
        g.load(dY + row * lddy, gl, d);
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)                               // dY pre-scaled (alpha = 1: exact): the "+ mean term" axpy ahead of
#pragma unroll
            for (int q = 0; q < VEC; ++q) g.x[k][q] *= alpha;          // the ID chain's first softmax backward rides here
        const float dot = y.dot(g);
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) g.x[k][q] = y.x[k][q] * (g.x[k][q] - dot);
        g.store(dZ + row * lddz, gl, d);
    }
}

// ---------------------------------------------------------------------------------------------
// fusion
// ---------------------------------------------------------------------------------------------
struct FuseArgs {
    const float* mean_terms[LLMREC_MAX_TERMS];
    int64_t mean_ld[LLMREC_MAX_TERMS];
    const float* norm_terms[LLMREC_MAX_TERMS];
    int64_t norm_ld[LLMREC_MAX_TERMS];
    float* d_terms[LLMREC_MAX_TERMS];
    int64_t d_ld[LLMREC_MAX_TERMS];
    const float* s_terms[LLMREC_MAX_TERMS];   // backward, accumulate == 2: d_terms[t] = s_terms[t] (or 0 if null) + the term's gradient
    int64_t s_ld[LLMREC_MAX_TERMS];
    float rates[LLMREC_MAX_TERMS];
    int n_mean, n_norm;
    float mean_scale;
    int n_reg;          // backward: the first n_reg norm terms also receive reg2 * x (a sum-of-squares regulariser on them)
    float reg2;
};

template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(int64_t rows, int d, FuseArgs a, float* __restrict__ out, int64_t ldo) {
    ROW_LOOP_HEADER {
        RowReg<VEC, NCHUNK> acc, t;
        acc.fill(0.f);
        for (int i = 0; i < a.n_mean; ++i) {
            t.load(a.mean_terms[i] + row * a.mean_ld[i], gl, d);
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc.x[k][q] += t.x[k][q];
        }
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc.x[k][q] *= a.mean_scale;
        for (int i = 0; i < a.n_norm; ++i) {
            t.load(a.norm_terms[i] + row * a.norm_ld[i], gl, d);
            const float nrm = fmaxf(sqrtf(t.dot(t)), 1e-12f);           // F.normalize eps
            const float w = a.rates[i] / nrm;
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc.x[k][q] = fmaf(w, t.x[k][q], acc.x[k][q]);
        }
        acc.store(out + row * ldo, gl, d);
    }
}

template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void fuse_bwd_kernel(int64_t rows, int d, FuseArgs a, const float* __restrict__ dOut,
                                                       int64_t lddo, int accumulate) {
    ROW_LOOP_HEADER {
        RowReg<VEC, NCHUNK> g, t, o;
        g.load(dOut + row * lddo, gl, d);
        for (int i = 0; i < a.n_norm; ++i) {
            t.load(a.norm_terms[i] + row * a.norm_ld[i], gl, d);
            const float nn = sqrtf(t.dot(t));
            float* dst = a.d_terms[i] + row * a.d_ld[i];
            if (accumulate == 1) o.load(dst, gl, d);
            else if (accumulate == 2 && a.s_terms[i]) o.load(a.s_terms[i] + row * a.s_ld[i], gl, d);
            else o.fill(0.f);
            if (nn >= 1e-12f) {
                const float inv = 1.0f / nn;
                const float proj = t.dot(g) * inv * inv;                 // <n, g> / ||x||
                const float w = a.rates[i] * inv;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] += w * (g.x[k][q] - t.x[k][q] * proj);
            } else {                                                     // clamp active: x / eps
                const float w = a.rates[i] * 1e12f;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] += w * g.x[k][q];
            }
            if (i < a.n_reg) {                                           // d/dx of coef * sum x^2, folded in (x is already loaded)
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] = fmaf(a.reg2, t.x[k][q], o.x[k][q]);
            }
            o.store(dst, gl, d);
        }
    }
}

// The user-side and the item-side fusion (or its backward) as ONE launch: the two row ranges are independent, on two streams each
// launch paid a cross-queue fork and join (~10 us each way in a replayed graph) for 15 - 26 us of work. Blocks
// [0, block_begin) sweep problem 0, the rest problem 1.
constexpr int FUSE_MAX_PROBLEMS = 2;
struct FuseMulti {
    FuseArgs a[FUSE_MAX_PROBLEMS];
    int64_t rows[FUSE_MAX_PROBLEMS];
    float* out[FUSE_MAX_PROBLEMS]; int64_t ldo[FUSE_MAX_PROBLEMS];            // forward
    const float* dOut[FUSE_MAX_PROBLEMS]; int64_t lddo[FUSE_MAX_PROBLEMS];    // backward
    const uint8_t* row_flags[FUSE_MAX_PROBLEMS];                              // backward: not active = dOut and the sources of this row are all-zero
    const int32_t* row_stamp[FUSE_MAX_PROBLEMS];                              // device counter defining "active" (NULL: non-zero)
    int block_begin;                                                          // first block of problem 1
    int n_sumsq;                                                              // forward: the first n_sumsq norm terms' squared norms are summed
    float* sumsq_partial;                                                     // ... into one partial per block [gridDim.x]
};
#define ROW_LOOP_MULTI(m)                                                                          \
    const int prob = (int)blockIdx.x >= (m).block_begin ? 1 : 0;                                   \
    const FuseArgs& a = (m).a[prob];                                                               \
    const int64_t rows = (m).rows[prob];                                                           \
    const int gl = threadIdx.x & (RL - 1);                                                         \
    const int blk = (int)blockIdx.x - (prob ? (m).block_begin : 0);                                \
    const int nblk = prob ? (int)gridDim.x - (m).block_begin : (m).block_begin;                    \
    const int64_t row_stride = (int64_t)nblk * ROWS_PER_BLOCK;                                     \
    for (int64_t row = (int64_t)blk * ROWS_PER_BLOCK + threadIdx.x / RL; row < rows; row += row_stride)

template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void fuse_fwd_multi_kernel(int d, FuseMulti m) {
    __shared__ float ss_red[256];
    float ss = 0.f;                                                     // this thread's share of sum ||x||^2 over the first n_sumsq terms
    ROW_LOOP_MULTI(m) {
        RowReg<VEC, NCHUNK> acc, t;
        acc.fill(0.f);
        for (int i = 0; i < a.n_mean; ++i) {
            t.load(a.mean_terms[i] + row * a.mean_ld[i], gl, d);
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc.x[k][q] += t.x[k][q];
        }
#pragma unroll
        for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc.x[k][q] *= a.mean_scale;
        for (int i = 0; i < a.n_norm; ++i) {
            t.load(a.norm_terms[i] + row * a.norm_ld[i], gl, d);
            const float sq = t.dot(t);
            if (i < m.n_sumsq && gl == 0) ss += sq;                     // (rows of a thread in ascending order, terms in order: fixed)
            const float nrm = fmaxf(sqrtf(sq), 1e-12f);                 // F.normalize eps
            const float w = a.rates[i] / nrm;
#pragma unroll
            for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc.x[k][q] = fmaf(w, t.x[k][q], acc.x[k][q]);
        }
        acc.store(m.out[prob] + row * m.ldo[prob], gl, d);
    }
    if (m.n_sumsq > 0) {                                                // uniform: one partial per block, pairwise tree over the 256 threads
        ss_red[threadIdx.x] = ss;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) ss_red[threadIdx.x] += ss_red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) m.sumsq_partial[blockIdx.x] = ss_red[0];
    }
}

// (accumulate == 2 semantics of fuse_bwd_kernel: d_terms[t] = s_terms[t] (or 0) + the term's gradient)
template <int VEC, int NCHUNK>
__global__ __launch_bounds__(256) void fuse_bwd_src_multi_kernel(int d, FuseMulti m) {
    ROW_LOOP_MULTI(m) {
        RowReg<VEC, NCHUNK> g, t, o;
        if (m.row_flags[prob] && !(m.row_stamp[prob] ? m.row_flags[prob][row] == LLMREC_ROW_STAMP(m.row_stamp[prob][0]) : m.row_flags[prob][row] != 0)) {
            // a row the batch did not touch: dOut = 0 and the sources are 0, so every stream's gradient is w (0 - x 0) = +0 and
            // what is left is the regulariser's reg2 * x on the first n_reg streams - the same bits as the general path below
            for (int i = 0; i < a.n_norm; ++i) {
                o.fill(0.f);
                if (i < a.n_reg) {
                    t.load(a.norm_terms[i] + row * a.norm_ld[i], gl, d);
#pragma unroll
                    for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                        for (int q = 0; q < VEC; ++q) o.x[k][q] = fmaf(a.reg2, t.x[k][q], o.x[k][q]);
                }
                o.store(a.d_terms[i] + row * a.d_ld[i], gl, d);
            }
            continue;
        }
        g.load(m.dOut[prob] + row * m.lddo[prob], gl, d);
        for (int i = 0; i < a.n_norm; ++i) {
            t.load(a.norm_terms[i] + row * a.norm_ld[i], gl, d);
            const float nn = sqrtf(t.dot(t));
            float* dst = a.d_terms[i] + row * a.d_ld[i];
            if (a.s_terms[i]) o.load(a.s_terms[i] + row * a.s_ld[i], gl, d);
            else o.fill(0.f);
            if (nn >= 1e-12f) {
                const float inv = 1.0f / nn;
                const float proj = t.dot(g) * inv * inv;                 // <n, g> / ||x||
                const float w = a.rates[i] * inv;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] += w * (g.x[k][q] - t.x[k][q] * proj);
            } else {                                                     // clamp active: x / eps
                const float w = a.rates[i] * 1e12f;
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] += w * g.x[k][q];
            }
            if (i < a.n_reg) {
#pragma unroll
                for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o.x[k][q] = fmaf(a.reg2, t.x[k][q], o.x[k][q]);
            }
            o.store(dst, gl, d);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sum of squares (two-level, fixed order), axpy
// ---------------------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(int64_t rows, int d, const float* __restrict__ X, int64_t ldx,
                                                            float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    const int64_t n = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float v = X[(e / d) * ldx + (e % d)];
        s = fmaf(v, v, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(int n_partial, const float* __restrict__ partial, float coef,
                                                          int accumulate, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + coef * red[0];
}

__global__ __launch_bounds__(256) void axpy_kernel(int64_t rows, int d, float alpha, const float* __restrict__ alpha_dev,
                                                   const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                   int accumulate) {
    const float a = alpha * (alpha_dev ? alpha_dev[0] : 1.0f);
    const int64_t n = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / d, c = e % d;
        const float v = a * X[r * ldx + c];
        float* y = Y + r * ldy + c;
        *y = accumulate ? (*y + v) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// AdamW
// ---------------------------------------------------------------------------------------------
__global__ void adamw_advance_kernel(float* state, float lr, float b1, float b2) {
    int t = __float_as_int(state[0]) + 1;
    state[0] = __int_as_float(t);
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    state[1] = (float)((double)lr / bc1);
    state[2] = (float)sqrt(bc2);
}

__global__ void timestamp_kernel(uint64_t* slot) { *slot = wall_clock64(); }

__global__ __launch_bounds__(256) void adamw_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ state, float decay_mul, float b1, float b2, float eps) {
    const float step_size = state[1], bc2s = state[2];
    const float w1 = 1.0f - b1, w2 = 1.0f - b2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * decay_mul;                                   // p.mul_(1 - lr * wd)
        const float mi = m[i] + w1 * (gi - m[i]);                      // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = fmaf(w2 * gi, gi, v[i] * b2);                 // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / bc2s + eps;
        pi = pi - step_size * (mi / denom);                            // addcdiv_(exp_avg, denom, -step_size)
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

struct AdamwTensors {
    float* p[LLMREC_ADAMW_MAX_TENSORS];
    const float* g[LLMREC_ADAMW_MAX_TENSORS];
    float* m[LLMREC_ADAMW_MAX_TENSORS];
    float* v[LLMREC_ADAMW_MAX_TENSORS];
    int64_t n[LLMREC_ADAMW_MAX_TENSORS];
    float gscale[LLMREC_ADAMW_MAX_TENSORS];
    float* gout[LLMREC_ADAMW_MAX_TENSORS];
    int32_t block_begin[LLMREC_ADAMW_MAX_TENSORS + 1];
    int32_t n_tensors;
};
constexpr int ADAMW_PER_BLOCK = 256 * 16;

struct ZeroRowJobs {
    const int64_t* ids[LLMREC_ZERO_ROWS_MAX_JOBS];
    float* dst[LLMREC_ZERO_ROWS_MAX_JOBS];
    int64_t ldd[LLMREC_ZERO_ROWS_MAX_JOBS];
    int32_t d[LLMREC_ZERO_ROWS_MAX_JOBS];
    int32_t n_jobs, B_cap, blocks_per_job, first_block;      // blocks >= first_block belong to the clean-up: (job, 16 samples) each
    const int32_t* n_valid;
};

__device__ __forceinline__ void adamw_block(const AdamwTensors& t, const float* __restrict__ state, float decay_mul, float b1, float b2, float eps);

// all parameters of the model in one launch: block -> (tensor, 4096-element chunk)
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamwTensors t, const float* __restrict__ state, float decay_mul,
                                                          float b1, float b2, float eps) {
    adamw_block(t, state, decay_mul, b1, b2, eps);
}

// the same + the row-wise clean-up of the step's scatter targets in the trailing blocks (llmrec_adamw_multi_zero_rows_f32)
__global__ __launch_bounds__(256) void adamw_multi_zero_rows_kernel(AdamwTensors t, const float* __restrict__ state, float decay_mul,
                                                                    float b1, float b2, float eps, ZeroRowJobs z) {
    if ((int)blockIdx.x < z.first_block) { adamw_block(t, state, decay_mul, b1, b2, eps); return; }
    const int rel = (int)blockIdx.x - z.first_block;
    const int job = rel / z.blocks_per_job, blk = rel - job * z.blocks_per_job;
    int B = z.n_valid ? z.n_valid[0] : z.B_cap;
    B = B > z.B_cap ? z.B_cap : (B < 0 ? 0 : B);
    const int gl = threadIdx.x & 15;
    const int b = blk * 16 + (threadIdx.x >> 4);
    if (b >= B) return;
    float* row = z.dst[job] + z.ids[job][b] * z.ldd[job];
    for (int c = gl; c < z.d[job]; c += 16) row[c] = 0.f;
}

__device__ __forceinline__ void adamw_block(const AdamwTensors& t, const float* __restrict__ state, float decay_mul, float b1, float b2, float eps) {
    int k = 0;
    while (k + 1 < t.n_tensors && (int)blockIdx.x >= t.block_begin[k + 1]) ++k;
    const int64_t base = (int64_t)(blockIdx.x - t.block_begin[k]) * ADAMW_PER_BLOCK;
    float* __restrict__ p = t.p[k]; const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
    const int64_t n = t.n[k];
    const float gs = t.gscale[k];
    float* __restrict__ go = t.gout[k];
    const float step_size = state[1], bc2s = state[2];
    const float w1 = 1.0f - b1, w2 = 1.0f - b2;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i >= n) break;
        const float gi = gs == 1.0f ? g[i] : gs * g[i];
        if (go) go[i] = gi;
        float pi = p[i] * decay_mul;
        const float mi = m[i] + w1 * (gi - m[i]);
        const float vi = fmaf(w2 * gi, gi, v[i] * b2);
        const float denom = sqrtf(vi) / bc2s + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

template <typename F4, typename F1>
static int dispatch_rows(int d, bool vec4, F4 f4, F1 f1) {
    if (vec4) {
        if (d <= 64) return f4(std::integral_constant<int, 1>());
        if (d <= 128) return f4(std::integral_constant<int, 2>());
        if (d <= 256) return f4(std::integral_constant<int, 4>());
        if (d <= 512) return f4(std::integral_constant<int, 8>());
    } else {
        if (d <= 16) return f1(std::integral_constant<int, 1>());
        if (d <= 64) return f1(std::integral_constant<int, 4>());
        if (d <= 128) return f1(std::integral_constant<int, 8>());
        if (d <= 256) return f1(std::integral_constant<int, 16>());
    }
    set_error("row kernel: d = %d outside the compiled family (vec4 = %d)", d, (int)vec4);
    return LLMREC_EUNSUPPORTED;
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace llmrec

using namespace llmrec;

struct ZeroTensors {
    float* p[LLMREC_ZERO_MAX_TENSORS];
    int64_t n[LLMREC_ZERO_MAX_TENSORS];
    int32_t block_begin[LLMREC_ZERO_MAX_TENSORS + 1];
    int32_t n_tensors;
};
constexpr int ZERO_PER_BLOCK = 256 * 4 * 8;                 // 8 float4 stores per thread

__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroTensors t) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < LLMREC_ZERO_MAX_TENSORS; ++i) k += (i < t.n_tensors && (int)blockIdx.x >= t.block_begin[i]) ? 1 : 0;
    float* p = t.p[k];
    const int64_t n = t.n[k];
    const int64_t base = (int64_t)(blockIdx.x - t.block_begin[k]) * ZERO_PER_BLOCK;
    const bool vec = (((uintptr_t)p) % 16) == 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + ((int64_t)j * 256 + threadIdx.x) * 4;
        if (vec && i + 4 <= n) *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (int e = 0; e < 4; ++e) if (i + e < n) p[i + e] = 0.f;
    }
}

struct LossWeights { float w[LLMREC_BPR_MAX_PROBLEMS]; };

__global__ void loss_assemble_kernel(int mode, int n_prob, const float* __restrict__ out, LossWeights w, float* scal, float* tail, float inv_world,
                                     double* running) {
    if (threadIdx.x != 0) return;
    if (mode == 0) {
        float s = 0.f;
        for (int p = 0; p < n_prob; ++p) s += out[2 * p] * w.w[p];         // same order as (out[:, 0] * w).sum() over <= 8 terms
        scal[2] = out[0]; scal[3] = out[1];
        scal[1] = s + out[1] + scal[0];
    } else if (mode == 1) {
        for (int p = 0; p < n_prob; ++p) tail[p] = out[2 * p];
        tail[n_prob] = (out[1] + scal[0]) * inv_world;
    } else {
        float s = 0.f;
        for (int p = 0; p < n_prob; ++p) s += tail[p] * w.w[p];
        scal[2] = tail[0]; scal[3] = out[1];
        scal[1] = s + tail[n_prob];
    }
    if (running && mode != 1) { running[0] += (double)scal[1]; running[1] += (double)scal[2]; running[2] += (double)scal[3]; }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(int64_t rows, int d, const float* __restrict__ s, const float* __restrict__ X, int64_t ldx,
                                                         float* __restrict__ Y, int64_t ldy) {
    const int64_t n = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / d; const int c = (int)(e - r * d);
        Y[r * ldy + c] = s[r] * X[r * ldx + c];
    }
}

// dst[ids[j]][0..d) = 0 for j < n (ids[j] < 0 skipped): the row-wise clean-up of a scatter target whose other rows are
// known to be zero already (row-sharded step: B of 10^7 rows), instead of a dense memset of the whole table
__global__ __launch_bounds__(256) void zero_rows_kernel(int64_t n, const int64_t* __restrict__ ids, int d, float* __restrict__ dst, int64_t ldd) {
    const int gl = threadIdx.x & 15;
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= n) return;
    const int64_t row = ids[j];
    if (row < 0) return;
    float* p = dst + row * ldd;
    for (int c = gl; c < d; c += 16) p[c] = 0.f;
}

// dst[list[j]] = src[j] for j < *n (llmrec_scatter_set_rows_f32): one 16-lane group per slot
__global__ __launch_bounds__(256) void scatter_set_rows_kernel(int capacity, const int32_t* __restrict__ list, const int32_t* __restrict__ n_dev, int d,
                                                               const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd) {
    const int gl = threadIdx.x & 15;
    const int j = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= capacity || j >= n_dev[0]) return;
    const float* s_ = src + (int64_t)j * lds_;
    float* o = dst + (int64_t)list[j] * ldd;
    for (int c = gl; c < d; c += 16) o[c] = s_[c];
}

__global__ __launch_bounds__(256) void mark_rows_kernel(int64_t n, const int64_t* __restrict__ ids, uint8_t value, uint8_t* __restrict__ flags) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t row = ids[j];
    if (row >= 0) flags[row] = value;
}

// softmax backward of listed rows (16 lanes per row; d <= 1024)
__global__ __launch_bounds__(256) void softmax_bwd_listed_kernel(int64_t n, const int64_t* __restrict__ ids, int d, float alpha,
                                                                 const float* __restrict__ Y, int64_t ldy, const float* __restrict__ dY, int64_t lddy,
                                                                 const float* __restrict__ post_scale, float* __restrict__ dZ, int64_t lddz) {
    const int gl = threadIdx.x & 15;
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= n) return;
    const int64_t row = ids[j];
    if (row < 0) return;
    const float* y = Y + row * ldy; const float* g = dY + row * lddy;
    float dot = 0.f;
    for (int c = gl; c < d; c += 16) dot = fmaf(y[c], alpha * g[c], dot);
    dot = group_sum<16>(dot);
    const float ps = post_scale ? post_scale[row] : 1.0f;
    float* o = dZ + row * lddz;
    for (int c = gl; c < d; c += 16) o[c] = ps * (y[c] * (alpha * g[c] - dot));
}

// MARK_SPLIT wavefronts per listed row, each takes every MARK_SPLIT-th 64-column piece of the row's adjacency list (a hub row of 10^6
// columns is 2^14 pieces: one wavefront would walk them for a millisecond)
constexpr int MARK_SPLIT = 32;
__global__ __launch_bounds__(256) void mark_neighbours_kernel(int64_t n, const int64_t* __restrict__ ids, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ colidx, uint8_t value, uint8_t* __restrict__ flags) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t j = w / MARK_SPLIT;
    const int part = (int)(w - j * MARK_SPLIT);
    if (j >= n) return;
    const int64_t row = ids[j];
    if (row < 0) return;
    const int32_t s = rowptr[row], e = rowptr[row + 1];
    for (int64_t k = (int64_t)s + part * 64 + (threadIdx.x & 63); k < e; k += 64 * MARK_SPLIT) flags[colidx[k]] = value;
}

// ---------------------------------------------------------------------------------------------
// llmrec_batch_reach_rows: the user rows a batch reaches, as an ascending list. Launch 1: one wavefront per (sample, role) - role 0
// flags the sample's user, roles 1 / 2 every user in the adjacency list of its positive / negative item. Launch 2: ONE block turns
// the byte flags into the ascending list (each thread owns a contiguous 16-aligned run of rows: count, block-wide exclusive scan,
// write) and clears them again, so the scratch is all-zero between calls and nothing needs a stamp.
// ---------------------------------------------------------------------------------------------
constexpr int REACH_SPLIT = 8;                             // wavefronts per (sample, item): a hub item's adjacency list is walked in 8 interleaved parts
__global__ __launch_bounds__(256) void batch_reach_mark_kernel(int B_cap, const int32_t* __restrict__ n_valid, const int64_t* __restrict__ users,
                                                               const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t n_users,
                                                               int64_t n_items, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                               uint8_t* __restrict__ flags) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int nv = n_valid ? *n_valid : B_cap;
    nv = nv < B_cap ? nv : B_cap;
    const int part = w % REACH_SPLIT, job = w / REACH_SPLIT;
    const int b = job / 3, role = job - 3 * b;
    if (b >= nv) return;
    if (role == 0) {
        const int64_t u = users[b];
        if (part == 0 && lane == 0 && u >= 0 && u < n_users) flags[u] = 1;
        return;
    }
    const int64_t it = role == 1 ? pos[b] : neg[b];
    if (it < 0 || it >= n_items) return;
    const int32_t s = rowptr[it], e = rowptr[it + 1];
    for (int32_t k = s + part * 64 + lane; k < e; k += 64 * REACH_SPLIT) flags[colidx[k]] = 1;
}

// ONE block of 1024 threads: thread t owns the 16-aligned run of `per` rows starting at t * per; count, block-wide exclusive scan
// (wave64 shuffles + the 16 wave totals through LDS), write the ids in order, clear the flags.
__global__ __launch_bounds__(1024) void flags_compact_kernel(int64_t n, uint8_t* __restrict__ flags, int32_t* __restrict__ list, int32_t* __restrict__ n_out) {
    __shared__ int32_t wave_total[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = ((n + 1023) / 1024 + 15) / 16 * 16;
    const int64_t r0 = tid * per, r1 = r0 + per < n ? r0 + per : n;
    int32_t c = 0;
    for (int64_t r = r0; r < r1; r += 16) {                  // 16 flags per load (r0 and the allocation are 16-byte aligned; the tail is masked)
        const uint4 v = *reinterpret_cast<const uint4*>(flags + r);
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                c += (r + 4 * q + bb < r1) && ((wd[q] >> (8 * bb)) & 0xffu);
    }
    int32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { const int32_t t = wave_total[w2]; if (w2 < wave) base += t; total += t; }
    int32_t o = base + incl - c;
    if (c) {
        for (int64_t r = r0; r < r1; ++r)
            if (flags[r]) { list[o++] = (int32_t)r; flags[r] = 0; }
    }
    // the entries a 16-wide tile past the end may fetch: defined values
    const int32_t pad_end = (total + 15) / 16 * 16 + 16;
    for (int32_t k = total + tid; k < pad_end; k += 1024) list[k] = 0;
    if (tid == 0) *n_out = total;
}

// ---------------------------------------------------------------------------------------------
// weighted column sums in 64-column groups: out_g[j] (+)= sum_r w[r] * X[r][64 g + j]. The bias gradient of a projection whose
// operand was propagated beforehand (Y = (A F) W^T + (A 1) b^T: db = sum_r (A 1)[r] dY[r]). Two levels, fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int COLSUM_BLOCKS = 128;
constexpr int COLSUM_MAX_D = 64 * LLMREC_COLSUM_MAX_GROUPS;
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t rows, int d, const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ w, float* __restrict__ partial) {
    float acc[COLSUM_MAX_D / 256] = {};
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float wr = w ? w[r] : 1.f;
        const float* row = X + r * ldx;
#pragma unroll
        for (int k = 0; k < COLSUM_MAX_D / 256; ++k) {
            const int c = k * 256 + threadIdx.x;
            if (c < d) acc[k] = fmaf(wr, row[c], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < COLSUM_MAX_D / 256; ++k) {
        const int c = k * 256 + threadIdx.x;
        if (c < d) partial[(int64_t)blockIdx.x * d + c] = acc[k];
    }
}
struct ColsumGroups { float* out[LLMREC_COLSUM_MAX_GROUPS]; int n_groups; };
// one block, one thread per column: the partials of a column are added in four interleaved chains (fixed order; consecutive threads
// read consecutive addresses), then thread j < gw adds the groups that share a destination, in group order
__global__ __launch_bounds__(COLSUM_MAX_D) void colsum_final_kernel(int n_partial, int d, int gw, const float* __restrict__ partial, ColsumGroups g, int accumulate) {
    __shared__ float col[COLSUM_MAX_D];
    const int c = threadIdx.x;
    if (c < d) {
        const float* p = partial + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = 0;
        for (; b + 4 <= n_partial; b += 4) {
            s0 += p[(int64_t)b * d]; s1 += p[(int64_t)(b + 1) * d]; s2 += p[(int64_t)(b + 2) * d]; s3 += p[(int64_t)(b + 3) * d];
        }
        for (; b < n_partial; ++b) s0 += p[(int64_t)b * d];
        col[c] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (c >= gw) return;
    for (int q = 0; q < g.n_groups; ++q) {
        bool first = true;                                       // the first group writing this destination decides overwrite / accumulate
        for (int e = 0; e < q; ++e) first = first && g.out[e] != g.out[q];
        float* o = g.out[q] + c;
        *o = (first && !accumulate) ? col[gw * q + c] : *o + col[gw * q + c];
    }
}

struct GatherTerms { const float* t[LLMREC_MAX_TERMS]; int64_t ld[LLMREC_MAX_TERMS]; int n; };

__global__ __launch_bounds__(256) void gather_mean_kernel(int64_t n, const int64_t* __restrict__ idx, int d, float scale, GatherTerms g,
                                                          float* __restrict__ out, int64_t ldo) {
    const int gl = threadIdx.x & 15;
    const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= n) return;
    const int64_t row = idx[b];
    for (int c = gl; c < d; c += 16) {
        float s = 0.f;
        for (int t = 0; t < g.n; ++t) s += g.t[t][row * g.ld[t] + c];
        out[b * ldo + c] = scale * s;
    }
}

extern "C" {

int llmrec_softmax_rows_fwd_f32(int64_t rows, int32_t d, const float* Z, int64_t ldz, float* Y, int64_t ldy,
                                llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d > 0, "softmax_fwd: bad sizes");
    if (rows == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(Z && Y && ldz >= d && ldy >= d, "softmax_fwd: null pointer or ld < d");
    const bool vec4 = d % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0 && aligned16(Z) && aligned16(Y);
    const int grid = grid_for(rows, ROWS_PER_BLOCK);
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { softmax_fwd_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, Z, ldz, Y, ldy); return 0; },
        [&](auto nc) { softmax_fwd_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, Z, ldz, Y, ldy); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_softmax_rows_bwd_scaled_f32(int64_t rows, int32_t d, float alpha, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                                       float* dZ, int64_t lddz, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d > 0, "softmax_bwd: bad sizes");
    if (rows == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(Y && dY && dZ && ldy >= d && lddy >= d && lddz >= d, "softmax_bwd: null pointer or ld < d");
    const bool vec4 = d % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && lddz % 4 == 0 && aligned16(Y) && aligned16(dY) && aligned16(dZ);
    const int grid = grid_for(rows, ROWS_PER_BLOCK);
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { softmax_bwd_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, Y, ldy, dY, lddy, dZ, lddz, alpha); return 0; },
        [&](auto nc) { softmax_bwd_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, Y, ldy, dY, lddy, dZ, lddz, alpha); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_softmax_rows_bwd_f32(int64_t rows, int32_t d, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                                float* dZ, int64_t lddz, llmrec_stream_t stream_) {
    return llmrec_softmax_rows_bwd_scaled_f32(rows, d, 1.0f, Y, ldy, dY, lddy, dZ, lddz, stream_);
}

int llmrec_fuse_fwd_f32(int64_t rows, int32_t d, float mean_scale,
                        int32_t n_mean, const float* const* mean_terms, const int64_t* mean_ld,
                        int32_t n_norm, const float* const* norm_terms, const int64_t* norm_ld,
                        const float* rates, float* out, int64_t ldo, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d > 0 && n_mean >= 0 && n_norm >= 0, "fuse_fwd: bad sizes");
    LLMREC_CHECK_ARG(n_mean <= LLMREC_MAX_TERMS && n_norm <= LLMREC_MAX_TERMS, "fuse_fwd: more than %d terms", LLMREC_MAX_TERMS);
    if (rows == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(out && ldo >= d, "fuse_fwd: null out or ld < d");
    FuseArgs a = {};
    bool vec4 = d % 4 == 0 && ldo % 4 == 0 && aligned16(out);
    a.n_mean = n_mean; a.n_norm = n_norm; a.mean_scale = mean_scale;
    for (int i = 0; i < n_mean; ++i) {
        LLMREC_CHECK_ARG(mean_terms[i] && mean_ld[i] >= d, "fuse_fwd: bad mean term %d", i);
        a.mean_terms[i] = mean_terms[i]; a.mean_ld[i] = mean_ld[i];
        vec4 = vec4 && mean_ld[i] % 4 == 0 && aligned16(mean_terms[i]);
    }
    for (int i = 0; i < n_norm; ++i) {
        LLMREC_CHECK_ARG(norm_terms[i] && norm_ld[i] >= d, "fuse_fwd: bad norm term %d", i);
        a.norm_terms[i] = norm_terms[i]; a.norm_ld[i] = norm_ld[i]; a.rates[i] = rates[i];
        vec4 = vec4 && norm_ld[i] % 4 == 0 && aligned16(norm_terms[i]);
    }
    const int grid = grid_for(rows, ROWS_PER_BLOCK);
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { fuse_fwd_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, a, out, ldo); return 0; },
        [&](auto nc) { fuse_fwd_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, a, out, ldo); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

static int fuse_bwd_impl(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                         int32_t n_norm, const float* const* norm_terms, const int64_t* norm_ld,
                         const float* rates, float* const* d_terms, const int64_t* d_ld,
                         const float* const* src_terms, const int64_t* src_ld,
                         int32_t accumulate, int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream_);

int llmrec_fuse_bwd_f32(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                        int32_t n_norm, const float* const* norm_terms, const int64_t* norm_ld,
                        const float* rates, float* const* d_terms, const int64_t* d_ld,
                        int32_t accumulate, int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream_) {
    return fuse_bwd_impl(rows, d, dOut, lddo, n_norm, norm_terms, norm_ld, rates, d_terms, d_ld, nullptr, nullptr, accumulate ? 1 : 0,
                         n_reg_terms, reg_two_coef, stream_);
}

int llmrec_fuse_bwd_src_f32(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                            int32_t n_norm, const float* const* norm_terms, const int64_t* norm_ld,
                            const float* rates, float* const* d_terms, const int64_t* d_ld,
                            const float* const* src_terms, const int64_t* src_ld,
                            int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(src_terms && src_ld, "fuse_bwd_src: null source tables");
    return fuse_bwd_impl(rows, d, dOut, lddo, n_norm, norm_terms, norm_ld, rates, d_terms, d_ld, src_terms, src_ld, 2,
                         n_reg_terms, reg_two_coef, stream_);
}

static int fuse_bwd_impl(int64_t rows, int32_t d, const float* dOut, int64_t lddo,
                         int32_t n_norm, const float* const* norm_terms, const int64_t* norm_ld,
                         const float* rates, float* const* d_terms, const int64_t* d_ld,
                         const float* const* src_terms, const int64_t* src_ld,
                         int32_t accumulate, int32_t n_reg_terms, float reg_two_coef, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d > 0 && n_norm >= 0 && n_norm <= LLMREC_MAX_TERMS && n_reg_terms >= 0 && n_reg_terms <= n_norm,
                     "fuse_bwd: bad sizes");
    if (rows == 0 || n_norm == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(dOut && lddo >= d, "fuse_bwd: null dOut or ld < d");
    FuseArgs a = {};
    bool vec4 = d % 4 == 0 && lddo % 4 == 0 && aligned16(dOut);
    a.n_norm = n_norm; a.n_reg = n_reg_terms; a.reg2 = reg_two_coef;
    for (int i = 0; i < n_norm; ++i) {
        LLMREC_CHECK_ARG(norm_terms[i] && d_terms[i] && norm_ld[i] >= d && d_ld[i] >= d, "fuse_bwd: bad term %d", i);
        a.norm_terms[i] = norm_terms[i]; a.norm_ld[i] = norm_ld[i]; a.rates[i] = rates[i];
        a.d_terms[i] = d_terms[i]; a.d_ld[i] = d_ld[i];
        a.s_terms[i] = src_terms ? src_terms[i] : nullptr; a.s_ld[i] = src_terms ? src_ld[i] : 0;
        LLMREC_CHECK_ARG(!a.s_terms[i] || a.s_ld[i] >= d, "fuse_bwd: source term %d has ld < d", i);
        vec4 = vec4 && norm_ld[i] % 4 == 0 && d_ld[i] % 4 == 0 && aligned16(norm_terms[i]) && aligned16(d_terms[i]) &&
               (!a.s_terms[i] || (a.s_ld[i] % 4 == 0 && aligned16(a.s_terms[i])));
    }
    const int grid = grid_for(rows, ROWS_PER_BLOCK);
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { fuse_bwd_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, a, dOut, lddo, accumulate); return 0; },
        [&](auto nc) { fuse_bwd_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(rows, d, a, dOut, lddo, accumulate); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_fuse_fwd_multi_f32(int32_t n_problems, const llmrec_fuse_fwd_problem_t* p, int32_t d, llmrec_stream_t stream_) {
    return llmrec_fuse_fwd_multi_sumsq_f32(n_problems, p, d, 0, nullptr, 0, nullptr, stream_);
}

int llmrec_fuse_fwd_multi_sumsq_f32(int32_t n_problems, const llmrec_fuse_fwd_problem_t* p, int32_t d, int32_t n_sumsq_terms,
                                    float* sumsq_partial, int32_t partial_capacity, int32_t* n_partial_host, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= FUSE_MAX_PROBLEMS && p && d > 0, "fuse_fwd_multi: 1..%d problems", FUSE_MAX_PROBLEMS);
    LLMREC_CHECK_ARG(n_sumsq_terms >= 0 && (n_sumsq_terms == 0 || (sumsq_partial && n_partial_host)), "fuse_fwd_multi: sum of squares without a partial buffer");
    FuseMulti m = {};
    bool vec4 = d % 4 == 0;
    int blocks[FUSE_MAX_PROBLEMS] = {0, 0};
    for (int k = 0; k < n_problems; ++k) {
        const llmrec_fuse_fwd_problem_t& q = p[k];
        LLMREC_CHECK_ARG(q.rows >= 0 && q.n_mean >= 0 && q.n_norm >= 0 && q.n_mean <= LLMREC_MAX_TERMS && q.n_norm <= LLMREC_MAX_TERMS,
                         "fuse_fwd_multi: problem %d has bad sizes", k);
        LLMREC_CHECK_ARG(q.rows == 0 || (q.out && q.ldo >= d), "fuse_fwd_multi: problem %d: null out or ld < d", k);
        FuseArgs& a = m.a[k];
        a.n_mean = q.n_mean; a.n_norm = q.n_norm; a.mean_scale = q.mean_scale;
        vec4 = vec4 && q.ldo % 4 == 0 && aligned16(q.out);
        for (int i = 0; i < q.n_mean; ++i) {
            LLMREC_CHECK_ARG(q.mean_terms[i] && q.mean_ld[i] >= d, "fuse_fwd_multi: problem %d: bad mean term %d", k, i);
            a.mean_terms[i] = q.mean_terms[i]; a.mean_ld[i] = q.mean_ld[i];
            vec4 = vec4 && q.mean_ld[i] % 4 == 0 && aligned16(q.mean_terms[i]);
        }
        for (int i = 0; i < q.n_norm; ++i) {
            LLMREC_CHECK_ARG(q.norm_terms[i] && q.norm_ld[i] >= d, "fuse_fwd_multi: problem %d: bad norm term %d", k, i);
            a.norm_terms[i] = q.norm_terms[i]; a.norm_ld[i] = q.norm_ld[i]; a.rates[i] = q.rates[i];
            vec4 = vec4 && q.norm_ld[i] % 4 == 0 && aligned16(q.norm_terms[i]);
        }
        m.rows[k] = q.rows; m.out[k] = q.out; m.ldo[k] = q.ldo;
        blocks[k] = q.rows > 0 ? grid_for(q.rows, ROWS_PER_BLOCK) : 0;
    }
    m.block_begin = blocks[0];
    const int grid = blocks[0] + blocks[1];
    if (n_sumsq_terms > 0) {
        for (int k = 0; k < n_problems; ++k) LLMREC_CHECK_ARG(n_sumsq_terms <= p[k].n_norm, "fuse_fwd_multi: problem %d has fewer than %d norm terms", k, n_sumsq_terms);
        if (grid > partial_capacity) { set_error("fuse_fwd_multi: %d partial sums > capacity %d", grid, partial_capacity); return LLMREC_EWORKSPACE; }
        *n_partial_host = grid;
        m.n_sumsq = n_sumsq_terms; m.sumsq_partial = sumsq_partial;
    }
    if (grid == 0) return LLMREC_OK;
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { fuse_fwd_multi_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(d, m); return 0; },
        [&](auto nc) { fuse_fwd_multi_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(d, m); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_fuse_bwd_src_multi_f32(int32_t n_problems, const llmrec_fuse_bwd_problem_t* p, int32_t d, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= FUSE_MAX_PROBLEMS && p && d > 0, "fuse_bwd_src_multi: 1..%d problems", FUSE_MAX_PROBLEMS);
    FuseMulti m = {};
    bool vec4 = d % 4 == 0;
    int blocks[FUSE_MAX_PROBLEMS] = {0, 0};
    for (int k = 0; k < n_problems; ++k) {
        const llmrec_fuse_bwd_problem_t& q = p[k];
        LLMREC_CHECK_ARG(q.rows >= 0 && q.n_norm >= 0 && q.n_norm <= LLMREC_MAX_TERMS && q.n_reg_terms >= 0 && q.n_reg_terms <= q.n_norm,
                         "fuse_bwd_src_multi: problem %d has bad sizes", k);
        if (q.rows == 0 || q.n_norm == 0) continue;
        LLMREC_CHECK_ARG(q.dOut && q.lddo >= d && q.src_terms && q.src_ld, "fuse_bwd_src_multi: problem %d: null pointer or ld < d", k);
        FuseArgs& a = m.a[k];
        a.n_norm = q.n_norm; a.n_reg = q.n_reg_terms; a.reg2 = q.reg_two_coef;
        vec4 = vec4 && q.lddo % 4 == 0 && aligned16(q.dOut);
        for (int i = 0; i < q.n_norm; ++i) {
            LLMREC_CHECK_ARG(q.norm_terms[i] && q.d_terms[i] && q.norm_ld[i] >= d && q.d_ld[i] >= d, "fuse_bwd_src_multi: problem %d: bad term %d", k, i);
            a.norm_terms[i] = q.norm_terms[i]; a.norm_ld[i] = q.norm_ld[i]; a.rates[i] = q.rates[i];
            a.d_terms[i] = q.d_terms[i]; a.d_ld[i] = q.d_ld[i];
            a.s_terms[i] = q.src_terms[i]; a.s_ld[i] = q.src_terms[i] ? q.src_ld[i] : 0;
            LLMREC_CHECK_ARG(!a.s_terms[i] || a.s_ld[i] >= d, "fuse_bwd_src_multi: problem %d: source term %d has ld < d", k, i);
            vec4 = vec4 && q.norm_ld[i] % 4 == 0 && q.d_ld[i] % 4 == 0 && aligned16(q.norm_terms[i]) && aligned16(q.d_terms[i]) &&
                   (!a.s_terms[i] || (a.s_ld[i] % 4 == 0 && aligned16(a.s_terms[i])));
        }
        m.rows[k] = q.rows; m.dOut[k] = q.dOut; m.lddo[k] = q.lddo; m.row_flags[k] = q.row_flags; m.row_stamp[k] = q.row_stamp;
        blocks[k] = grid_for(q.rows, ROWS_PER_BLOCK);
    }
    m.block_begin = blocks[0];
    const int grid = blocks[0] + blocks[1];
    if (grid == 0) return LLMREC_OK;
    int rc = dispatch_rows(d, vec4,
        [&](auto nc) { fuse_bwd_src_multi_kernel<4, decltype(nc)::value><<<grid, 256, 0, stream>>>(d, m); return 0; },
        [&](auto nc) { fuse_bwd_src_multi_kernel<1, decltype(nc)::value><<<grid, 256, 0, stream>>>(d, m); return 0; });
    if (rc) return rc;
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int64_t llmrec_sumsq_workspace_bytes(int64_t rows, int32_t d) {
    (void)rows; (void)d;
    return 4 * SUMSQ_BLOCKS;
}

int llmrec_sumsq_f32(int64_t rows, int32_t d, const float* X, int64_t ldx, float coef, int32_t accumulate,
                     float* out, void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d >= 0 && out, "sumsq: bad argument");
    LLMREC_CHECK_ARG(rows * d == 0 || (X && ldx >= d), "sumsq: null X or ld < d");
    if (workspace_bytes < 4 * SUMSQ_BLOCKS || !workspace) { set_error("sumsq: workspace too small"); return LLMREC_EWORKSPACE; }
    const int nb = rows * d == 0 ? 1 : grid_for(rows * d, 256 * 8, SUMSQ_BLOCKS);
    sumsq_partial_kernel<<<nb, 256, 0, stream>>>(rows, d > 0 ? d : 1, X, ldx, (float*)workspace);
    LLMREC_LAUNCH_CHECK();
    sumsq_final_kernel<<<1, 256, 0, stream>>>(nb, (const float*)workspace, coef, accumulate, out);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_axpy_f32(int64_t rows, int32_t d, float alpha, const float* alpha_dev, const float* X, int64_t ldx,
                    float* Y, int64_t ldy, int32_t accumulate, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(rows >= 0 && d >= 0, "axpy: bad sizes");
    if (rows * d == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(X && Y && ldx >= d && ldy >= d, "axpy: null pointer or ld < d");
    axpy_kernel<<<grid_for(rows * d, 256 * 4), 256, 0, stream>>>(rows, d, alpha, alpha_dev, X, ldx, Y, ldy, accumulate);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_adamw_advance(float* state3, float lr, float beta1, float beta2, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(state3, "adamw_advance: null state");
    adamw_advance_kernel<<<1, 1, 0, (hipStream_t)stream_>>>(state3, lr, beta1, beta2);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_adamw_f32(int64_t n, float* p, const float* g, float* m, float* v, const float* state3,
                     float lr, float beta1, float beta2, float eps, float weight_decay, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && state3, "adamw: bad argument");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(p && g && m && v, "adamw: null pointer");
    const float decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    adamw_kernel<<<grid_for(n, 256 * 4), 256, 0, (hipStream_t)stream_>>>(n, p, g, m, v, state3, decay_mul, beta1, beta2, eps);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

static int fill_adamw(AdamwTensors& t, int32_t n_tensors, const llmrec_adamw_tensor_t* tensors_host, const float* state3, int& blocks) {
    LLMREC_CHECK_ARG(n_tensors >= 0 && n_tensors <= LLMREC_ADAMW_MAX_TENSORS && state3 && (n_tensors == 0 || tensors_host),
                     "adamw_multi: bad argument (at most %d tensors)", LLMREC_ADAMW_MAX_TENSORS);
    t.n_tensors = n_tensors;
    blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const llmrec_adamw_tensor_t& x = tensors_host[i];
        LLMREC_CHECK_ARG(x.n >= 0 && (x.n == 0 || (x.p && x.g && x.m && x.v)), "adamw_multi: tensor %d has a null pointer", i);
        LLMREC_CHECK_ARG(x.g_scale != 0.0f, "adamw_multi: tensor %d has g_scale 0 (set 1 for a plain gradient)", i);
        t.p[i] = x.p; t.g[i] = x.g; t.m[i] = x.m; t.v[i] = x.v; t.n[i] = x.n; t.gscale[i] = x.g_scale; t.gout[i] = x.g_out;
        t.block_begin[i] = blocks;
        blocks += (int)ceil_div(x.n, ADAMW_PER_BLOCK);
    }
    for (int i = n_tensors; i <= LLMREC_ADAMW_MAX_TENSORS; ++i) t.block_begin[i] = blocks;
    return LLMREC_OK;
}

int llmrec_adamw_multi_f32(int32_t n_tensors, const llmrec_adamw_tensor_t* tensors_host, const float* state3,
                           float lr, float beta1, float beta2, float eps, float weight_decay, llmrec_stream_t stream_) {
    AdamwTensors t = {};
    int blocks = 0;
    if (int rc = fill_adamw(t, n_tensors, tensors_host, state3, blocks)) return rc;
    if (blocks == 0) return LLMREC_OK;
    const float decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    adamw_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream_>>>(t, state3, decay_mul, beta1, beta2, eps);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_adamw_multi_zero_rows_f32(int32_t n_tensors, const llmrec_adamw_tensor_t* tensors_host, const float* state3,
                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int32_t n_jobs, const llmrec_zero_rows_job_t* jobs_host, int32_t B_cap, const int32_t* n_valid_dev,
                                     llmrec_stream_t stream_) {
    AdamwTensors t = {};
    int blocks = 0;
    if (int rc = fill_adamw(t, n_tensors, tensors_host, state3, blocks)) return rc;
    LLMREC_CHECK_ARG(n_jobs >= 0 && n_jobs <= LLMREC_ZERO_ROWS_MAX_JOBS && B_cap >= 0 && (n_jobs == 0 || jobs_host),
                     "adamw_multi_zero_rows: bad clean-up table (at most %d jobs)", LLMREC_ZERO_ROWS_MAX_JOBS);
    ZeroRowJobs z = {};
    z.n_jobs = n_jobs; z.B_cap = B_cap; z.n_valid = n_valid_dev; z.first_block = blocks;
    z.blocks_per_job = (int)ceil_div(B_cap, 16);
    for (int j = 0; j < n_jobs; ++j) {
        LLMREC_CHECK_ARG(B_cap == 0 || (jobs_host[j].ids && jobs_host[j].dst && jobs_host[j].d > 0 && jobs_host[j].ldd >= jobs_host[j].d),
                         "adamw_multi_zero_rows: job %d: null pointer or ld < d", j);
        z.ids[j] = jobs_host[j].ids; z.dst[j] = jobs_host[j].dst; z.ldd[j] = jobs_host[j].ldd; z.d[j] = jobs_host[j].d;
    }
    const int total = blocks + n_jobs * z.blocks_per_job;
    if (total == 0) return LLMREC_OK;
    const float decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    adamw_multi_zero_rows_kernel<<<total, 256, 0, (hipStream_t)stream_>>>(t, state3, decay_mul, beta1, beta2, eps, z);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_zero_multi_f32(int32_t n_tensors, const llmrec_zero_tensor_t* tensors_host, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_tensors >= 0 && n_tensors <= LLMREC_ZERO_MAX_TENSORS && (n_tensors == 0 || tensors_host),
                     "zero_multi: bad argument (at most %d tensors)", LLMREC_ZERO_MAX_TENSORS);
    ZeroTensors t = {};
    t.n_tensors = n_tensors;
    int64_t blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        LLMREC_CHECK_ARG(tensors_host[i].n >= 0 && (tensors_host[i].n == 0 || tensors_host[i].p), "zero_multi: tensor %d has a null pointer", i);
        t.p[i] = tensors_host[i].p; t.n[i] = tensors_host[i].n;
        t.block_begin[i] = (int32_t)blocks;
        blocks += ceil_div(tensors_host[i].n, ZERO_PER_BLOCK);
        LLMREC_CHECK_ARG(blocks < 0x7fffffffll, "zero_multi: too many elements for one launch");
    }
    for (int i = n_tensors; i <= LLMREC_ZERO_MAX_TENSORS; ++i) t.block_begin[i] = (int32_t)blocks;
    if (blocks == 0) return LLMREC_OK;
    zero_multi_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(t);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_loss_assemble_f32(int32_t mode, int32_t n_problems, const float* bpr_out, const float* w_mf_host,
                             float* scal4, float* tail, float inv_world, double* running_sums3, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(mode >= 0 && mode <= 2 && n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS, "loss_assemble: bad mode / problem count");
    LLMREC_CHECK_ARG(bpr_out && scal4 && (mode == 1 || w_mf_host) && (mode == 0 || tail), "loss_assemble: null pointer");
    LossWeights w = {};
    if (w_mf_host) for (int i = 0; i < n_problems; ++i) w.w[i] = w_mf_host[i];
    loss_assemble_kernel<<<1, 64, 0, (hipStream_t)stream_>>>(mode, n_problems, bpr_out, w, scal4, tail, inv_world, running_sums3);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_scatter_set_rows_f32(int32_t capacity, const int32_t* row_list, const int32_t* n_list_dev, int32_t d, const float* src, int64_t lds_,
                                float* dst, int64_t ldd, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(capacity >= 0 && d > 0, "scatter_set_rows: bad sizes");
    if (capacity == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(row_list && n_list_dev && src && dst && lds_ >= d && ldd >= d, "scatter_set_rows: null pointer or ld < d");
    scatter_set_rows_kernel<<<(unsigned)ceil_div(capacity, 16), 256, 0, (hipStream_t)stream_>>>(capacity, row_list, n_list_dev, d, src, lds_, dst, ldd);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_mark_rows_u8(int64_t n, const int64_t* ids, int32_t value, uint8_t* flags, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && value >= 0 && value <= 255, "mark_rows: bad argument");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(ids && flags, "mark_rows: null pointer");
    LLMREC_CHECK_ARG(n / 256 < 0x7fffffffll, "mark_rows: too many rows for one launch");
    mark_rows_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(n, ids, (uint8_t)value, flags);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_batch_reach_rows(int64_t n_users, int64_t n_items, const int64_t* users, const int64_t* pos, const int64_t* neg, int32_t B_cap,
                            const int32_t* n_valid, const int32_t* item_rowptr, const int32_t* item_colidx, uint8_t* flags,
                            int32_t* row_list, int32_t* n_rows, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_users > 0 && n_users < 0x7fffffffll && n_items > 0 && B_cap >= 0, "batch_reach_rows: bad sizes");
    LLMREC_CHECK_ARG(users && pos && neg && item_rowptr && item_colidx && flags && row_list && n_rows, "batch_reach_rows: null pointer");
    LLMREC_CHECK_ARG((uintptr_t)flags % 16 == 0, "batch_reach_rows: flags must be 16-byte aligned (and readable up to n_users rounded up to 16 bytes)");
    hipStream_t stream = (hipStream_t)stream_;
    if (B_cap > 0) {
        batch_reach_mark_kernel<<<(unsigned)ceil_div(3 * (int64_t)B_cap * REACH_SPLIT, 4), 256, 0, stream>>>(B_cap, n_valid, users, pos, neg, n_users, n_items,
                                                                                                item_rowptr, item_colidx, flags);
        LLMREC_LAUNCH_CHECK();
    }
    flags_compact_kernel<<<1, 1024, 0, stream>>>(n_users, flags, row_list, n_rows);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_softmax_rows_bwd_listed_f32(int64_t n, const int64_t* ids, int32_t d, float alpha, const float* Y, int64_t ldy,
                                       const float* dY, int64_t lddy, const float* post_scale, float* dZ, int64_t lddz,
                                       llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && d > 0, "softmax_bwd_listed: bad sizes");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(ids && Y && dY && dZ && ldy >= d && lddy >= d && lddz >= d, "softmax_bwd_listed: null pointer or ld < d");
    softmax_bwd_listed_kernel<<<(unsigned)ceil_div(n, 16), 256, 0, (hipStream_t)stream_>>>(n, ids, d, alpha, Y, ldy, dY, lddy, post_scale, dZ, lddz);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_mark_neighbours_u8(int64_t n, const int64_t* ids, const int32_t* rowptr, const int32_t* colidx, int32_t value, uint8_t* flags,
                              llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && value >= 0 && value <= 255, "mark_neighbours: bad argument");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(ids && rowptr && colidx && flags, "mark_neighbours: null pointer");
    LLMREC_CHECK_ARG(n * MARK_SPLIT / 4 < 0x7fffffffll, "mark_neighbours: too many rows for one launch");
    mark_neighbours_kernel<<<(unsigned)ceil_div(n * MARK_SPLIT, 4), 256, 0, (hipStream_t)stream_>>>(n, ids, rowptr, colidx, (uint8_t)value, flags);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_scale_rows_f32(int64_t rows, int32_t d, const float* s, const float* X, int64_t ldx, float* Y, int64_t ldy,
                          llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(rows >= 0 && d >= 0, "scale_rows: bad sizes");
    if (rows * d == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(s && X && Y && ldx >= d && ldy >= d, "scale_rows: null pointer or ld < d");
    scale_rows_kernel<<<grid_for(rows * d, 256 * 4), 256, 0, (hipStream_t)stream_>>>(rows, d, s, X, ldx, Y, ldy);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_gather_mean_f32(int64_t n, const int64_t* idx, int32_t d, float scale, int32_t n_terms,
                           const float* const* terms, const int64_t* term_ld, float* out, int64_t ldo, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && d > 0 && n_terms >= 1 && n_terms <= LLMREC_MAX_TERMS, "gather_mean: bad sizes");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(idx && terms && term_ld && out && ldo >= d, "gather_mean: null pointer or ld < d");
    GatherTerms g = {};
    g.n = n_terms;
    for (int t = 0; t < n_terms; ++t) {
        LLMREC_CHECK_ARG(terms[t] && term_ld[t] >= d, "gather_mean: bad term %d", t);
        g.t[t] = terms[t]; g.ld[t] = term_ld[t];
    }
    gather_mean_kernel<<<(unsigned)ceil_div(n, 16), 256, 0, (hipStream_t)stream_>>>(n, idx, d, scale, g, out, ldo);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_zero_rows_f32(int64_t n, const int64_t* ids, int32_t d, float* dst, int64_t ldd, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && d > 0, "zero_rows: bad sizes");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(ids && dst && ldd >= d, "zero_rows: null pointer or ld < d");
    zero_rows_kernel<<<(unsigned)ceil_div(n, 16), 256, 0, (hipStream_t)stream_>>>(n, ids, d, dst, ldd);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int64_t llmrec_weighted_colsum_workspace_bytes(int32_t d) {
    if (d <= 0 || d > COLSUM_MAX_D) return -1;
    return (int64_t)sizeof(float) * COLSUM_BLOCKS * d;
}

int llmrec_weighted_colsum_f32(int64_t rows, int32_t n_groups, int32_t group_width, const float* X, int64_t ldx, const float* w,
                               float* const* group_out_host, int32_t accumulate,
                               void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(rows >= 0 && n_groups >= 1 && n_groups <= LLMREC_COLSUM_MAX_GROUPS && group_width >= 1 && group_width <= 64,
                     "weighted_colsum: 1..%d groups of 1..64 columns", LLMREC_COLSUM_MAX_GROUPS);
    const int32_t d = n_groups * group_width;
    LLMREC_CHECK_ARG(group_out_host && (rows == 0 || (X && ldx >= d)), "weighted_colsum: null pointer or ld < d");
    if (!workspace || workspace_bytes < llmrec_weighted_colsum_workspace_bytes(d)) {
        set_error("weighted_colsum: workspace %lld < %lld", (long long)workspace_bytes, (long long)llmrec_weighted_colsum_workspace_bytes(d));
        return LLMREC_EWORKSPACE;
    }
    ColsumGroups g = {};
    g.n_groups = n_groups;
    for (int q = 0; q < g.n_groups; ++q) {
        LLMREC_CHECK_ARG(group_out_host[q], "weighted_colsum: group %d has no destination", q);
        g.out[q] = group_out_host[q];
    }
    hipStream_t stream = (hipStream_t)stream_;
    float* partial = (float*)workspace;
    const int blocks = (int)(rows < COLSUM_BLOCKS ? (rows > 0 ? rows : 1) : COLSUM_BLOCKS);
    colsum_partial_kernel<<<blocks, 256, 0, stream>>>(rows, d, X, ldx, w, partial);
    LLMREC_LAUNCH_CHECK();
    colsum_final_kernel<<<1, COLSUM_MAX_D, 0, stream>>>(blocks, d, group_width, partial, g, accumulate);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

// measurement aid: the constant-rate counter at the moment the stream reaches this launch
int llmrec_timestamp(uint64_t* slot, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(slot, "timestamp: null slot");
    timestamp_kernel<<<1, 1, 0, (hipStream_t)stream_>>>(slot);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int64_t llmrec_timestamp_rate_hz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return (int64_t)khz * 1000;
}

}  // extern "C"
