// dense.hip - R4: side-feature projection  Y = X W^T + b  and its weight gradient, fp32.
// Replaces nn.Linear forward (aten::addmm) and the autograd weight-grad GEMM (aten::mm) at
// reference Models.py:30-37,145-150. The inputs X are constant feature matrices, so dX never
// exists.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32, a k-ordered fma chain) - at N = 64 the
// projection needs 32 flop per streamed byte, above the 19.7 flop/B ridge of 157 TFLOP/s over
// 8 TB/s, so it is matrix-pipe-bound, not HBM-bound.
//
// Operand feed without LDS: the MFMA A/B operands are ONE float per lane per instruction, and a
// dot product may visit its k's in any order as long as A and B agree. So every lane loads a
// float4 along the contiguous dimension and feeds component s to the s-th of four consecutive
// MFMAs; the wave's 16 lanes x 16 B cover whole 64..256-byte row segments (coalesced), e.g.
//   forward : lane l holds X[row0 + (l&15)][kb + 4*(l>>4) + s] and W[n0 + (l&15)][kb + 4*(l>>4) + s]
//   wgrad   : lane l holds dY[m][4*(l&15) + q] / X[m][k0 + 4*(l&15) + q] with m = m0 + (l>>4) + 4s';
//             component q selects one of four INTERLEAVED 16-wide tiles (n = 4 i + q, k = k0 + 4 j + q).
#include "common.h"

namespace llmrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 load4_guard(const float* row, int k, int K, bool vec_ok) {
    // elements [k, k+4) of a row of length K; zero beyond K
    if (vec_ok && k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) r.x = row[k];
    if (k + 1 < K) r.y = row[k + 1];
    if (k + 2 < K) r.z = row[k + 2];
    if (k + 3 < K) r.w = row[k + 3];
    return r;
}

// Branch-free variant for 16-byte aligned rows with K % 4 == 0: the address is clamped into the row
// and the value zeroed by a select. (A branch around a load makes hipcc fall back to
// s_waitcnt vmcnt(0) inside the k-loop, which serialises the whole software pipeline.)
__device__ __forceinline__ float4 load4_fast(const float* row, int k, int K) {
    const int kc = k < K ? k : K - 4;
    float4 v = *reinterpret_cast<const float4*>(row + kc);
    if (k >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// a GLOBAL-memory pointer known to be the same in every lane, moved into scalar registers (so that loads can use the
// scalar-base + 32-bit-lane-offset form); the address space is kept explicit - a pointer rebuilt from integers would
// otherwise be generic and load through FLAT instructions
typedef const char __attribute__((address_space(1))) * global_cptr;
__device__ __forceinline__ global_cptr uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (global_cptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint64_t uniform_u64(const void* p) {   // the same, as an integer (for buffer descriptors)
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float4 load4_global(global_cptr base, uint32_t byte_off) {
    typedef float raw4 __attribute__((ext_vector_type(4)));
    typedef const raw4 __attribute__((address_space(1))) * graw4;
    const raw4 v = *(graw4)(base + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float comp(const float4& v, int s) {
    return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------------------------
// forward: each wave computes RT*16 rows x NT*16 columns; block = 4 waves
// ---------------------------------------------------------------------------------------------
// SPLITK = false: the block's 4 waves take 4 different row tiles and the whole K range.
// SPLITK = true : the 4 waves share ONE row tile and split K in four; partial tiles are combined
//                 through LDS in wave order (deterministic). Used when M alone gives too few
//                 waves to hide the operand-load latency (M = 17366 -> 1086 row tiles for 1024 SIMDs).
template <int NT, int RT, bool SPLITK>
__global__ __launch_bounds__(256) void linear_fwd_kernel(int64_t M, int N, int K, const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
                                                         int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) float part_lds[];   // SPLITK: [4][RT*16][NT*16]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = SPLITK ? (int64_t)blockIdx.x * (RT * 16) : ((int64_t)blockIdx.x * 4 + wave) * (RT * 16);
    if (!SPLITK && row0 >= M) return;
    const int kq = SPLITK ? (((K + 3) / 4 + 15) / 16) * 16 : K;        // K range of this wave
    const int k_begin = SPLITK ? wave * kq : 0;
    const int k_end = SPLITK ? min(K, k_begin + kq) : K;
    const float* xrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int64_t r = row0 + t * 16 + li;
        if (r > M - 1) r = M - 1;
        xrow[t] = X + r * ldx;
    }
    const float* wrow[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int c = n * 16 + li;
        if (c > N - 1) c = N - 1;
        wrow[n] = W + (int64_t)c * ldw;
    }
    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 xa[RT], wb[NT], xa_n[RT], wb_n[NT];
    const int koff = 4 * lq;
#pragma unroll
    for (int t = 0; t < RT; ++t) xa[t] = load4_guard(xrow[t], k_begin + koff, k_end, vec_ok);
#pragma unroll
    for (int n = 0; n < NT; ++n) wb[n] = load4_guard(wrow[n], k_begin + koff, k_end, vec_ok);
    for (int kb = k_begin; kb < k_end; kb += 16) {
        const int kn = kb + 16 + koff;
        if (kb + 16 < k_end) {
#pragma unroll
            for (int t = 0; t < RT; ++t) xa_n[t] = load4_guard(xrow[t], kn, k_end, vec_ok);
#pragma unroll
            for (int n = 0; n < NT; ++n) wb_n[n] = load4_guard(wrow[n], kn, k_end, vec_ok);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(xa[t], s), comp(wb[n], s), acc[t][n], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) xa[t] = xa_n[t];
#pragma unroll
        for (int n = 0; n < NT; ++n) wb[n] = wb_n[n];
    }
    if (SPLITK) {
        constexpr int R = RT * 16, C = NT * 16;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part_lds[(wave * R + t * 16 + lq * 4 + r) * C + n * 16 + li] = acc[t][n][r];
        __syncthreads();
        for (int e = threadIdx.x; e < R * C; e += 256) {
            const int r = e / C, c = e % C;
            const int64_t row = row0 + r;
            if (row < M && c < N) {
                float v = part_lds[e];
                v += part_lds[R * C + e];
                v += part_lds[2 * R * C + e];
                v += part_lds[3 * R * C + e];
                Y[row * ldy + c] = v + (bias ? bias[c] : 0.f);
            }
        }
        return;
    }
    // D layout: lane holds D[row = lq*4 + r][col = li]
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 16 + li;
        if (c >= N) continue;
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + t * 16 + lq * 4 + r;
                if (row < M) Y[row * ldy + c] = acc[t][n][r] + b;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// grouped forward: up to LLMREC_LINEAR_MAX_PROBLEMS projections (different X, K, W, Y; one N) in
// ONE launch - the reference runs 8 per forward (Models.py:145-150). Work unit = 128 rows of one
// problem; block = 4 waves x (32 rows x N). The W k-slice (N x 32) is shared by the block through a
// double-buffered LDS tile (so W costs one L2 fetch per 128 rows, not per 16), X goes straight from
// global memory to the MFMA A operand with a two-step register prefetch.
// ---------------------------------------------------------------------------------------------
struct LinearGroup {
    const float* X[LLMREC_LINEAR_MAX_PROBLEMS];
    const float* W[LLMREC_LINEAR_MAX_PROBLEMS];
    const float* bias[LLMREC_LINEAR_MAX_PROBLEMS];
    const float* bias_scale[LLMREC_LINEAR_MAX_PROBLEMS];   // [M] or null: Y[r] = X[r] W^T + bias_scale[r] * bias (a pre-propagated operand: (A X) W^T + (A 1) b^T)
    float* Y[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t ldx[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t ldw[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t ldy[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t M[LLMREC_LINEAR_MAX_PROBLEMS];
    int32_t K[LLMREC_LINEAR_MAX_PROBLEMS];
    int32_t vec_ok[LLMREC_LINEAR_MAX_PROBLEMS];
    int32_t unit_begin[LLMREC_LINEAR_MAX_PROBLEMS + 1];
    int32_t n_problems;
};

constexpr int GF_BK = 32;              // k per step
constexpr int GF_WS = GF_BK + 4;       // LDS row stride in floats (144 B: spreads ds_read_b128 over the banks)

template <int NT, int RT, int MODE>
__global__ __launch_bounds__(256, 3) void linear_fwd_grouped_kernel(LinearGroup g, int N) {
    constexpr bool FAST = MODE >= 1;
    __shared__ __attribute__((aligned(16))) float w_lds[2][NT * 16 * GF_WS];
    int prob = 0;
    while (prob + 1 < g.n_problems && (int)blockIdx.x >= g.unit_begin[prob + 1]) ++prob;
    const int64_t M = g.M[prob];
    const int K = g.K[prob];
    const bool vec_ok = g.vec_ok[prob];
    const float* __restrict__ X = g.X[prob];
    const float* __restrict__ W = g.W[prob];
    const int64_t ldx = g.ldx[prob], ldw = g.ldw[prob];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)(blockIdx.x - g.unit_begin[prob]) * (64 * RT) + wave * (16 * RT);

    const float* xrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int64_t r = row0 + t * 16 + li;
        if (r > M - 1) r = M - 1;
        xrow[t] = X + r * ldx;
    }
    // W staging: thread -> (n = tid / 8 (+32), 4 k's = (tid % 8) * 4)
    const int wn = threadIdx.x >> 3, wk = (threadIdx.x & 7) * 4;
    constexpr int WLOADS = (NT * 16 + 31) / 32;                          // float4 per thread per step
    auto load_w = [&](int kb, float4 (&wr)[WLOADS]) {
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            int n = wn + 32 * j;
            const bool in = n < NT * 16;
            if (n > N - 1) n = N - 1;
            if (FAST) {
                float4 v = load4_fast(W + (int64_t)n * ldw, kb + wk, K);
                if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
                wr[j] = v;
            } else {
                wr[j] = in ? load4_guard(W + (int64_t)n * ldw, kb + wk, K, vec_ok) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_w = [&](int buf, const float4 (&wr)[WLOADS]) {
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            const int n = wn + 32 * j;
            if (n < NT * 16) *reinterpret_cast<float4*>(&w_lds[buf][n * GF_WS + wk]) = wr[j];
        }
    };
    auto load_x = [&](int kb, float4 (&xr)[RT][2]) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                xr[t][h] = FAST ? load4_fast(xrow[t], kb + 16 * h + 4 * lq, K) : load4_guard(xrow[t], kb + 16 * h + 4 * lq, K, vec_ok);
    };

    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline, all register stages named statically (loop unrolled by 3):
    //   step s: issue global loads of X(s+2) and W(s+2); multiply X(s) with the W(s) tile in LDS;
    //           write W(s+1) (loaded during step s-1) into the other LDS buffer; barrier.
    // Every load has two full steps to land; a stage is never copied (a copy would wait for it).
    float4 w0[WLOADS], w1[WLOADS], w2[WLOADS];
    float4 x0[RT][2], x1[RT][2], x2[RT][2];
    load_w(0, w0);
    load_x(0, x0);
    load_w(GF_BK, w1);
    load_x(GF_BK, x1);
    store_w(0, w0);
    __syncthreads();
    int cur = 0;
    auto kstep = [&](int kb, const float4 (&xc)[RT][2], float4 (&xl)[RT][2], const float4 (&ws)[WLOADS], float4 (&wl)[WLOADS]) {
        load_w(kb + 2 * GF_BK, wl);                                      // zeros beyond K (guarded)
        load_x(kb + 2 * GF_BK, xl);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 wf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                wf[n] = *reinterpret_cast<const float4*>(&w_lds[cur][(n * 16 + li) * GF_WS + 16 * h + 4 * lq]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(xc[t][h], s), comp(wf[n], s), acc[t][n], 0, 0, 0);
        }
        store_w(cur ^ 1, ws);
        __syncthreads();
        cur ^= 1;
    };
    for (int kb = 0; kb < K; kb += 3 * GF_BK) {                          // steps past K multiply zeros
        kstep(kb, x0, x2, w1, w2);
        kstep(kb + GF_BK, x1, x0, w2, w0);
        kstep(kb + 2 * GF_BK, x2, x1, w0, w1);
    }
    float* __restrict__ Y = g.Y[prob];
    const int64_t ldy = g.ldy[prob];
    const float* bias = g.bias[prob];
    const float* bscale = g.bias_scale[prob];
    float rs[RT][4];                                                       // the bias weight of this lane's rows (1 unless a scale is given)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + t * 16 + lq * 4 + r;
            rs[t][r] = (bscale && row < M) ? bscale[row] : 1.f;
        }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 16 + li;
        if (c >= N) continue;
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + t * 16 + lq * 4 + r;
                if (row < M) Y[row * ldy + c] = fmaf(rs[t][r], b, acc[t][n][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// split-precision variant of the grouped forward ("bf16x3"): every fp32 operand is written as the
// EXACT sum of three bf16 numbers (hi + mid + lo: 3 x 8 significant bits = fp32's 24) and the product
// is evaluated with the six bf16 MFMAs whose terms are >= 2^-24 relative:
//     x w ~= xh wh + xh wm + xm wh + xh wl + xm wm + xl wh          (dropped: 2^-24 |x||w| and below)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (bf16 x bf16 products are exact in fp32). Six MFMAs
// at 16x the fp32-MFMA rate = 3/8 of the matrix time, so the projection becomes HBM-bound on the X
// stream. Error is fp32-roundoff class (tests assert 2e-6 relative); it is NOT the bit-identical fma
// chain of the fp32 kernel, which stays available (LLMREC_GEMM=f32).
// X is split in registers on the fly (6 integer/float VALU ops per element), W when it is staged
// into LDS. A/B operand = 8 consecutive k per lane (lane l: row l&15, k = 8 (l>>4) + j).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 { uint32_t h, m, l; };                       // each: a bf16 value in the HIGH 16 bits (l: low half not cleared)
__device__ __forceinline__ Split3 split3(float x) {
    Split3 r;
    r.h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(r.h);             // exact
    r.m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(r.m);            // exact, <= 8 significant bits
    r.l = __float_as_uint(r2);                             // the pack below keeps only its high half
    return r;
}
// high halves of e0 (-> low half of the result) and e1 (-> high half) in one v_perm_b32: no masking needed
__device__ __forceinline__ uint32_t pack_hi16(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two register-adjacent floats -> their packed (hi, mid, lo) bf16 pairs: 2 + 2 masks, 2 packed subtractions, 3 v_perm_b32
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& H, uint32_t& M, uint32_t& L) {
    const f32x2 x = {x0, x1};
    const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const f32x2 hf = {__uint_as_float(h0), __uint_as_float(h1)};
    const f32x2 r1 = x - hf;                                   // exact
    const uint32_t m0 = __float_as_uint(r1.x) & 0xffff0000u, m1 = __float_as_uint(r1.y) & 0xffff0000u;
    const f32x2 mf = {__uint_as_float(m0), __uint_as_float(m1)};
    const f32x2 r2 = r1 - mf;                                  // exact, <= 8 significant bits
    H = pack_hi16(h0, h1);
    M = pack_hi16(m0, m1);
    L = pack_hi16(__float_as_uint(r2.x), __float_as_uint(r2.y));
}

// 8 floats (two float4, consecutive k of one row) -> three packed bf16x8 fragments
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& H, uint4& M, uint4& L) {
    split2(a.x, a.y, H.x, M.x, L.x);
    split2(a.z, a.w, H.y, M.y, L.y);
    split2(b.x, b.y, H.z, M.z, L.z);
    split2(b.z, b.w, H.w, M.w, L.w);
}
// 8 float4 (rows j = 0..7 of one lane; component c belongs to tile c) -> the three bf16x8 fragments of all four
// tiles. The residual arithmetic runs on the register pairs (x, y) and (z, w) exactly as they were loaded (packed
// subtractions, no moves); only the final v_perm_b32 packing crosses loads. 144 VALU instructions per 32 values.
__device__ __forceinline__ void split32(const float4 (&v)[8], uint4 (&H)[4], uint4 (&M)[4], uint4 (&L)[4]) {
    uint32_t h[8][4], m[8][4], l[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x2 xy = {v[j].x, v[j].y}, zw = {v[j].z, v[j].w};
        h[j][0] = __float_as_uint(xy.x) & 0xffff0000u; h[j][1] = __float_as_uint(xy.y) & 0xffff0000u;
        h[j][2] = __float_as_uint(zw.x) & 0xffff0000u; h[j][3] = __float_as_uint(zw.y) & 0xffff0000u;
        const f32x2 hxy = {__uint_as_float(h[j][0]), __uint_as_float(h[j][1])}, hzw = {__uint_as_float(h[j][2]), __uint_as_float(h[j][3])};
        const f32x2 r1xy = xy - hxy, r1zw = zw - hzw;                       // exact
        m[j][0] = __float_as_uint(r1xy.x) & 0xffff0000u; m[j][1] = __float_as_uint(r1xy.y) & 0xffff0000u;
        m[j][2] = __float_as_uint(r1zw.x) & 0xffff0000u; m[j][3] = __float_as_uint(r1zw.y) & 0xffff0000u;
        const f32x2 mxy = {__uint_as_float(m[j][0]), __uint_as_float(m[j][1])}, mzw = {__uint_as_float(m[j][2]), __uint_as_float(m[j][3])};
        const f32x2 r2xy = r1xy - mxy, r2zw = r1zw - mzw;                   // exact, <= 8 significant bits
        l[j][0] = __float_as_uint(r2xy.x); l[j][1] = __float_as_uint(r2xy.y);
        l[j][2] = __float_as_uint(r2zw.x); l[j][3] = __float_as_uint(r2zw.y);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        H[c] = make_uint4(pack_hi16(h[0][c], h[1][c]), pack_hi16(h[2][c], h[3][c]), pack_hi16(h[4][c], h[5][c]), pack_hi16(h[6][c], h[7][c]));
        M[c] = make_uint4(pack_hi16(m[0][c], m[1][c]), pack_hi16(m[2][c], m[3][c]), pack_hi16(m[4][c], m[5][c]), pack_hi16(m[6][c], m[7][c]));
        L[c] = make_uint4(pack_hi16(l[0][c], l[1][c]), pack_hi16(l[2][c], l[3][c]), pack_hi16(l[4][c], l[5][c]), pack_hi16(l[6][c], l[7][c]));
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


constexpr int GB_WS = 40;              // LDS row stride of one bf16 W tile in 2-byte units (32 k + 8 pad = 80 B)

// MODE 0: guarded loads (any shape); 1: FAST (16-byte aligned rows, K % 4 == 0: branch-free clamped loads with a zero
// select); 2: FAST and every K % 32 == 0 and all byte offsets < 2^32 - a k-step is then valid or invalid for the
// whole wave, so the operand addresses are scalar base + one 32-bit lane offset + a SCALAR k offset (one v_add per
// load, no clamps or selects on X: past K the staged W tile is zero and X is finite), W is zeroed by one select.
template <int NT, int RT, int MODE>
__global__ __launch_bounds__(256, 2) void linear_fwd_grouped_bf16x3_kernel(LinearGroup g, int N) {
    constexpr bool FAST = MODE >= 1;
    constexpr bool K32 = MODE >= 2;
    constexpr bool XPOSE = MODE == 3;
    // [buffer][term h/m/l][n][k] bf16
    __shared__ __attribute__((aligned(16))) uint16_t w_lds[2][3][NT * 16 * GB_WS];
    // MODE 3: each wave's 16 RT rows x 128 B of a k-step, as loaded (full 128-B lines), for the fragment-shaped re-read
    __shared__ float4 x_lds[XPOSE ? 4 : 1][XPOSE ? 16 * RT * 8 : 1];
    int prob = 0;
    while (prob + 1 < g.n_problems && (int)blockIdx.x >= g.unit_begin[prob + 1]) ++prob;
    const int64_t M = g.M[prob];
    const int K = g.K[prob];
    const bool vec_ok = g.vec_ok[prob];
    const float* __restrict__ X = g.X[prob];
    const float* __restrict__ W = g.W[prob];
    const int64_t ldx = g.ldx[prob], ldw = g.ldw[prob];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)(blockIdx.x - g.unit_begin[prob]) * (64 * RT) + wave * (16 * RT);

    const float* xrow[RT];
    uint32_t xoff[RT];                                                     // K32: byte offset of (row, k = 8 lq) from X
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int64_t r = row0 + t * 16 + li;
        if (r > M - 1) r = M - 1;
        xrow[t] = X + r * ldx;
        xoff[t] = (uint32_t)(r * ldx + 8 * lq) * 4u;
    }
    // MODE 3 (XPOSE). A fragment-shaped load touches 16 rows x 4 pieces of 16 B per instruction; the CU's address path
    // serves such an instruction one lane per cycle and the wave sits at load issue for 60 % of its cycles
    // (tools/fwd_prof.hip). Here a k-step's rows are loaded as full 128-B lines - instruction j: lane l reads piece l & 7
    // of row 8 j + (l >> 3) - into the same register stages, and pass through the wave's private 4-KB LDS image on their
    // way to the split: ds_write_b128 as loaded, ds_read_b128 in fragment shape (row li, pieces 2 lq and 2 lq + 1).
    // The image is XOR-swizzled so that both sides are bank-conflict-free: piece p of row r sits at slot
    // 8 r + (p ^ f(r)), f(r) = ((r >> 1) & 7) ^ 2 [(r & 15) in 4..11] (the b128 read groups are the lane sets
    // {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: f maps the 16 (row, piece) pairs of each onto 16 different slots).
    auto swz = [](int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); };
    uint32_t xoffc[2 * RT];                                                // byte offset of (row 8 j + (l >> 3), piece l & 7)
    int wslot[2], rslot[2];
    if (XPOSE) {
#pragma unroll
        for (int j = 0; j < 2 * RT; ++j) {
            int64_t r = row0 + 8 * j + (lane >> 3);
            if (r > M - 1) r = M - 1;
            xoffc[j] = (uint32_t)(r * ldx + 4 * (lane & 7)) * 4u;
        }
#pragma unroll
        for (int bpar = 0; bpar < 2; ++bpar) {                             // rows 8 j + (l >> 3): f depends on j only through j & 1
            const int r = 8 * bpar + (lane >> 3);
            wslot[bpar] = (lane >> 3) * 8 + ((lane & 7) ^ swz(r));         // + 64 j at use
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) rslot[h] = li * 8 + ((2 * lq + h) ^ swz(li));   // + 128 t at use
    }
    const global_cptr Xs = uniform_ptr(X), Ws = uniform_ptr(W);
    const int wn = threadIdx.x >> 3, wk = (threadIdx.x & 7) * 4;
    constexpr int WLOADS = (NT * 16 + 31) / 32;
    auto load_w = [&](int kb, float4 (&wr)[WLOADS]) {
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            int n = wn + 32 * j;
            const bool in = n < NT * 16;
            if (n > N - 1) n = N - 1;
            if (K32) {
                const int kbs = kb < K ? kb : K - 32;                      // wave-uniform: scalar
                float4 v = load4_global(Ws, (uint32_t)(n * (int)ldw + wk) * 4u + (uint32_t)kbs * 4u);
                if (!in || kb >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
                wr[j] = v;
            } else if (FAST) {
                float4 v = load4_fast(W + (int64_t)n * ldw, kb + wk, K);
                if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
                wr[j] = v;
            } else {
                wr[j] = in ? load4_guard(W + (int64_t)n * ldw, kb + wk, K, vec_ok) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_w = [&](int buf, const float4 (&wr)[WLOADS]) {               // split while staging
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            const int n = wn + 32 * j;
            if (n >= NT * 16) continue;
            uint2 wh, wm, wl;
            split2(wr[j].x, wr[j].y, wh.x, wm.x, wl.x);
            split2(wr[j].z, wr[j].w, wh.y, wm.y, wl.y);
            const int off = n * GB_WS + wk;
            *reinterpret_cast<uint2*>(&w_lds[buf][0][off]) = wh;
            *reinterpret_cast<uint2*>(&w_lds[buf][1][off]) = wm;
            *reinterpret_cast<uint2*>(&w_lds[buf][2][off]) = wl;
        }
    };
    auto load_x = [&](int kb, float4 (&xr)[RT][2]) {                         // lane: k = kb + 8 lq + {0..3 | 4..7}
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (XPOSE) {
                    const int kbs = kb < K ? kb : K - 32;
                    xr[t][h] = load4_global(Xs, xoffc[2 * t + h] + (uint32_t)kbs * 4u);      // instruction j = 2 t + h
                } else if (K32) {
                    const int kbs = kb < K ? kb : K - 32;                  // wave-uniform: scalar; past K the W tile is zero
                    xr[t][h] = load4_global(Xs, xoff[t] + (uint32_t)kbs * 4u + 16u * h);
                } else {
                    xr[t][h] = FAST ? load4_fast(xrow[t], kb + 8 * lq + 4 * h, K) : load4_guard(xrow[t], kb + 8 * lq + 4 * h, K, vec_ok);
                }
    };

    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 w0[WLOADS], w1[WLOADS], w2[WLOADS];
    float4 x0[RT][2], x1[RT][2], x2[RT][2];
    load_w(0, w0);
    load_x(0, x0);
    load_w(GF_BK, w1);
    load_x(GF_BK, x1);
    store_w(0, w0);
    __syncthreads();
    int cur = 0;
    auto kstep = [&](int kb, const float4 (&xc)[RT][2], float4 (&xl)[RT][2], const float4 (&ws)[WLOADS], float4 (&wl)[WLOADS]) {
        load_w(kb + 2 * GF_BK, wl);                                        // zeros beyond K (guarded)
        load_x(kb + 2 * GF_BK, xl);
        uint4 ah[RT], am[RT], al[RT];
        if (XPOSE) {
            float4* xw = x_lds[XPOSE ? wave : 0];
#pragma unroll
            for (int j = 0; j < 2 * RT; ++j) xw[wslot[j & 1] + 64 * j] = xc[j >> 1][j & 1];
            float4 xf[RT][2];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) xf[t][h] = xw[rslot[h] + 128 * t];
#pragma unroll
            for (int t = 0; t < RT; ++t) split8(xf[t][0], xf[t][1], ah[t], am[t], al[t]);
        } else {
#pragma unroll
            for (int t = 0; t < RT; ++t) split8(xc[t][0], xc[t][1], ah[t], am[t], al[t]);
        }
        uint4 bh[NT], bm[NT], bl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int off = (n * 16 + li) * GB_WS + 8 * lq;
            bh[n] = *reinterpret_cast<const uint4*>(&w_lds[cur][0][off]);
            bm[n] = *reinterpret_cast<const uint4*>(&w_lds[cur][1][off]);
            bl[n] = *reinterpret_cast<const uint4*>(&w_lds[cur][2][off]);
        }
        // smallest terms first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(al[t], bh[n], acc[t][n]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(ah[t], bl[n], acc[t][n]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(am[t], bm[n], acc[t][n]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(am[t], bh[n], acc[t][n]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(ah[t], bm[n], acc[t][n]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma_bf16(ah[t], bh[n], acc[t][n]);
        store_w(cur ^ 1, ws);
        __syncthreads();
        cur ^= 1;
    };
    for (int kb = 0; kb < K; kb += 3 * GF_BK) {                          // steps past K multiply zeros
        kstep(kb, x0, x2, w1, w2);
        kstep(kb + GF_BK, x1, x0, w2, w0);
        kstep(kb + 2 * GF_BK, x2, x1, w0, w1);
    }
    float* __restrict__ Y = g.Y[prob];
    const int64_t ldy = g.ldy[prob];
    const float* bias = g.bias[prob];
    const float* bscale = g.bias_scale[prob];
    float rs[RT][4];                                                       // the bias weight of this lane's rows (1 unless a scale is given)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + t * 16 + lq * 4 + r;
            rs[t][r] = (bscale && row < M) ? bscale[row] : 1.f;
        }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 16 + li;
        if (c >= N) continue;
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + t * 16 + lq * 4 + r;
                if (row < M) Y[row * ldy + c] = fmaf(rs[t][r], b, acc[t][n][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: wave tile = 64 n (interleaved tiles q) x 64 k (interleaved tiles p), over a
// chunk of MC rows; partial[chunk][n][k] then reduced in chunk order (deterministic).
// ---------------------------------------------------------------------------------------------
// Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2). Blocks that share operand
// rows (all k slabs of one row slab re-read the same dY rows) should share an L2: logical id = the id's position
// within its XCD's stream, so consecutive LOGICAL blocks run on one XCD. (PMC: 1.04 GB fetched per step's weight
// gradients against 0.74 GB algorithmic before - every XCD fetched its own copy of each dY slab.)
__device__ __forceinline__ int xcd_logical_block(int bid, int nblocks) {
    constexpr int XCDS = 8;
    const int per = nblocks / XCDS, rem = nblocks % XCDS;       // XCD x runs per + (x < rem) blocks
    const int x = bid % XCDS, local = bid / XCDS;
    return x * per + (x < rem ? x : rem) + local;
}

struct WgradGroup {
    const float* dY[LLMREC_LINEAR_MAX_PROBLEMS];
    const float* X[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t lddy[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t ldx[LLMREC_LINEAR_MAX_PROBLEMS];
    int64_t M[LLMREC_LINEAR_MAX_PROBLEMS];
    const float* db_w[LLMREC_LINEAR_MAX_PROBLEMS];         // [M] or null: the bias gradient is sum_r db_w[r] dY[r] (v2 body only)
    const int32_t* rows[LLMREC_LINEAR_MAX_PROBLEMS];       // or null: ascending ids of the rows of dY that can be non-zero (v2 body only)
    const int32_t* n_rows[LLMREC_LINEAR_MAX_PROBLEMS];     // device scalar: entries of rows[]
    int32_t vec_ok[LLMREC_LINEAR_MAX_PROBLEMS];
    int32_t chunk_begin[LLMREC_LINEAR_MAX_PROBLEMS + 1];   // first partial slab of each problem
    int32_t n_problems;
};

// One wave = (slab = chunk of MC rows of one problem, 64-wide k slab, 64-wide n block).
// Three register stages rotate STATICALLY (the loop is unrolled by 3), so a tile's loads have two
// full MFMA stages (2 x 2048 cycles) to land before they are consumed.
template <bool FAST>
__global__ __launch_bounds__(256, FAST ? 2 : 1) void linear_wgrad_kernel(WgradGroup g, int N, int K, float* __restrict__ partial,
                                                           float* __restrict__ partial_db, int64_t MC, int n_kslab, int n_slabs) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t wid = (int64_t)xcd_logical_block(blockIdx.x, gridDim.x) * 4 + wave;   // (slab, kslab) flattened, kslab fastest
    const int slab = (int)(wid / n_kslab);
    const int kslab = (int)(wid % n_kslab);
    if (slab >= n_slabs) return;
    int prob = 0;
    while (prob + 1 < g.n_problems && slab >= g.chunk_begin[prob + 1]) ++prob;
    const float* __restrict__ dY = g.dY[prob];
    const float* __restrict__ X = g.X[prob];
    const int64_t lddy = g.lddy[prob], ldx = g.ldx[prob], M = g.M[prob];
    const bool vec_ok = g.vec_ok[prob];
    const int nblk = blockIdx.y;                                 // 64-wide block of output rows n
    const int64_t m_begin = (int64_t)(slab - g.chunk_begin[prob]) * MC;
    const int64_t m_end = (m_begin + MC < M) ? m_begin + MC : M;
    const int n_base = nblk * 64 + 4 * li;                       // this lane's 4 n's: n_base + q
    const int k_base = kslab * 64 + 4 * li;                      // this lane's 4 k's: k_base + p

    f32x4 acc[4][4];                                             // [q][p]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[q][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

    // FAST (chosen on the host): N % 64 == 0, K % 64 == 0, 16-byte aligned rows -> straight-line
    // float4 loads; rows past the slab end are clamped and their dY operand zeroed (a * b = 0).
    const float* pa = dY + n_base;
    const float* pb = X + k_base;
    auto load_tile = [&](int64_t m0, float4 (&aa)[4], float4 (&bb)[4]) {
        if (FAST) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {                        // 4 MFMA k-steps of 4 rows each
                const int64_t m = m0 + 4 * s + lq;
                const int64_t mc = m < m_end ? m : m_end - 1;
                aa[s] = *reinterpret_cast<const float4*>(pa + mc * lddy);   // zeroed at use (mma_tile) if m >= m_end
                bb[s] = *reinterpret_cast<const float4*>(pb + mc * ldx);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t m = m0 + 4 * s + lq;
            if (m < m_end) {
                aa[s] = load4_guard(dY + m * lddy, n_base, N, vec_ok);
                bb[s] = load4_guard(X + m * ldx, k_base, K, vec_ok);
            } else {
                aa[s] = make_float4(0.f, 0.f, 0.f, 0.f);
                bb[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto mma_tile = [&](int64_t m0, float4 (&aa)[4], const float4 (&bb)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (FAST && m0 + 4 * s + lq >= m_end) aa[s] = make_float4(0.f, 0.f, 0.f, 0.f);   // select, at consumption time
            dbs.x += aa[s].x; dbs.y += aa[s].y; dbs.z += aa[s].z; dbs.w += aa[s].w;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    acc[q][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(aa[s], q), comp(bb[s], p), acc[q][p], 0, 0, 0);
        }
    };
    float4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
    load_tile(m_begin, a0, b0);
    load_tile(m_begin + 16, a1, b1);
    // no early exits: slabs are multiples of 48 rows and tiles past m_end load zeros, so the loop
    // body is one straight-line block (early exits made the compiler keep several accumulator sets)
    // sched_barrier pins each stage's loads BEFORE the MFMA block that follows: left alone, the
    // scheduler sinks the loads between the MFMAs (register pressure) and then waits vmcnt(0) on them
    for (int64_t m0 = m_begin; m0 < m_end; m0 += 48) {
        load_tile(m0 + 32, a2, b2);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(m0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(m0 + 48, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(m0 + 16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(m0 + 64, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(m0 + 32, a2, b2);
        __builtin_amdgcn_sched_barrier(0);
    }
    // D[q][p]: lane holds rows i = lq*4 + r (n = nblk*64 + 4 i + q), col j = li (k = kslab*64 + 4 j + p)
    float* pw = partial + (int64_t)slab * N * K;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nblk * 64 + 4 * (lq * 4 + r) + q;
            if (n >= N) continue;
            float* dst = pw + (int64_t)n * K + kslab * 64 + 4 * li;
            const float4 v = make_float4(acc[q][0][r], acc[q][1][r], acc[q][2][r], acc[q][3][r]);
            if (kslab * 64 + 4 * li + 3 < K && (K & 3) == 0) *reinterpret_cast<float4*>(dst) = v;
            else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (kslab * 64 + 4 * li + p < K) dst[p] = acc[q][p][r];
            }
        }
    if (kslab == 0 && partial_db) {
        // column sums of dY over this slab: combine the 4 row groups (lq) in a fixed order
        dbs.x += __shfl_xor(dbs.x, 16, 64); dbs.y += __shfl_xor(dbs.y, 16, 64);
        dbs.z += __shfl_xor(dbs.z, 16, 64); dbs.w += __shfl_xor(dbs.w, 16, 64);
        dbs.x += __shfl_xor(dbs.x, 32, 64); dbs.y += __shfl_xor(dbs.y, 32, 64);
        dbs.z += __shfl_xor(dbs.z, 32, 64); dbs.w += __shfl_xor(dbs.w, 32, 64);
        if (lq == 0) {
            float* pd = partial_db + (int64_t)slab * N;
            if (n_base + 0 < N) pd[n_base + 0] = dbs.x;
            if (n_base + 1 < N) pd[n_base + 1] = dbs.y;
            if (n_base + 2 < N) pd[n_base + 2] = dbs.z;
            if (n_base + 3 < N) pd[n_base + 3] = dbs.w;
        }
    }
}

// Split-precision weight gradient ("bf16x3", see the forward above): same wave tile (64 n x 64 k, interleaved
// 16-wide tiles, coalesced float4 rows of dY and X), 32 rows per step = one v_mfma_f32_16x16x32_bf16 per
// (tile pair, term): lane (li, lq) supplies the rows m0 + 4 j + lq (j = 0..7) of its n / k column to the 8
// contraction slots of its lane group - any row -> slot map works as long as A and B agree. Each fp32 operand is
// split in registers into hi + mid + lo bf16 (exact), six MFMAs per tile pair keep every term >= 2^-24: 96 MFMAs
// x 16 cycles per 32 rows instead of 128 x 32 cycles per 32 rows in the fp32 chain, so the kernel is bound by
// the X stream from HBM instead of the matrix pipe. One wave per SIMD (stages + split need > 256 registers).
// lb = the block's logical index among the launch's (or, in a multi-target launch, the target's) blocks
__device__ __forceinline__ void wgrad_bf16x3_body(const WgradGroup& g, int N, int K, float* __restrict__ partial,
                                                  float* __restrict__ partial_db, int64_t MC, int n_kslab, int n_slabs, int lb) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    // the four waves of a block take four consecutive slabs of ONE k slab and are summed through LDS in wave
    // order before anything is written: a quarter of the partial-slab traffic (all problems feed the same dW)
    __shared__ __attribute__((aligned(16))) float red[3][64 * 64 + 64];
    const int kslab = lb % n_kslab;
    const int group = lb / n_kslab;
    const int slab_raw = group * 4 + wave;
    const bool active = slab_raw < n_slabs;
    const int slab = active ? slab_raw : n_slabs - 1;             // idle waves of the last group recompute a slab and drop it
    int prob = 0;
    while (prob + 1 < g.n_problems && slab >= g.chunk_begin[prob + 1]) ++prob;
    const float* __restrict__ dY = g.dY[prob];
    const float* __restrict__ X = g.X[prob];
    const int64_t lddy = g.lddy[prob], ldx = g.ldx[prob], M = g.M[prob];
    const int nblk = blockIdx.y;
    const int64_t m_begin = (int64_t)(slab - g.chunk_begin[prob]) * MC;
    const int64_t m_end = (m_begin + MC < M) ? m_begin + MC : M;
    const int n_base = nblk * 64 + 4 * li;
    const int k_base = kslab * 64 + 4 * li;

    f32x4 acc[4][4];                                             // [q][p]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[q][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    // Addressing without per-lane 64-bit arithmetic and without branches around the loads (a branch there costs the
    // pipelining): the problem's base pointers are forced into scalar registers, a lane adds one 32-bit BYTE offset
    // (the host falls back to the fp32 kernel when M * ld * 4 >= 2^32), one add per load, and rows past the slab end
    // are clamped with a v_min against one precomputed offset. Full 32-row tiles run in the main loop with no
    // zeroing at all; the ragged tail of a problem's last slab is one separate tile after the loop.
    const global_cptr dYs = uniform_ptr(dY);
    const global_cptr Xs = uniform_ptr(X);
    const uint32_t la = 4u * (uint32_t)lddy, lbx = 4u * (uint32_t)ldx;             // row strides in bytes
    const uint32_t a_last = (uint32_t)(m_end - 1) * la + 4u * (uint32_t)n_base;
    const uint32_t b_last = (uint32_t)(m_end - 1) * lbx + 4u * (uint32_t)k_base;
    auto load_tile = [&](int64_t m0, float4 (&aa)[8], float4 (&bb)[8]) {
        uint32_t oa = (uint32_t)(m0 + lq) * la + 4u * (uint32_t)n_base;
        uint32_t ob = (uint32_t)(m0 + lq) * lbx + 4u * (uint32_t)k_base;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            aa[j] = load4_global(dYs, oa < a_last ? oa : a_last);
            bb[j] = load4_global(Xs, ob < b_last ? ob : b_last);
            oa += 4 * la; ob += 4 * lbx;
        }
    };
    auto mma_tile = [&](float4 (&aa)[8], const float4 (&bb)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { dbs.x += aa[j].x; dbs.y += aa[j].y; dbs.z += aa[j].z; dbs.w += aa[j].w; }
        uint4 bh[4], bm[4], bl[4], ahs[4], ams[4], als[4];
        split32(bb, bh, bm, bl);
        split32(aa, ahs, ams, als);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 ah = ahs[q], am = ams[q], al = als[q];
            // smallest terms first (as in the forward)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(al, bh[p], acc[q][p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(ah, bl[p], acc[q][p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(am, bm[p], acc[q][p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(am, bh[p], acc[q][p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(ah, bm[p], acc[q][p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][p] = mfma_bf16(ah, bh[p], acc[q][p]);
        }
    };
    // two 32-row stages rotate statically over the FULL tiles: the next tile's loads fly during this tile's 96 MFMAs
    const int64_t m_full = m_begin + ((m_end - m_begin) / 32) * 32;
    float4 a0[8], b0[8], a1[8], b1[8];
    load_tile(m_begin, a0, b0);
    for (int64_t m0 = m_begin; m0 < m_full; m0 += 64) {
        load_tile(m0 + 32, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(m0 + 64, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (m0 + 32 < m_full) mma_tile(a1, b1);                 // wave-uniform; no loads inside
        __builtin_amdgcn_sched_barrier(0);
    }
    if (m_full < m_end) {                                        // the ragged tail (at most once per problem): clamped rows zeroed
        load_tile(m_full, a0, b0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (m_full + 4 * j + lq >= m_end) a0[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        mma_tile(a0, b0);
    }
    dbs.x += __shfl_xor(dbs.x, 16, 64); dbs.y += __shfl_xor(dbs.y, 16, 64);
    dbs.z += __shfl_xor(dbs.z, 16, 64); dbs.w += __shfl_xor(dbs.w, 16, 64);
    dbs.x += __shfl_xor(dbs.x, 32, 64); dbs.y += __shfl_xor(dbs.y, 32, 64);
    dbs.z += __shfl_xor(dbs.z, 32, 64); dbs.w += __shfl_xor(dbs.w, 32, 64);
    if (wave > 0) {                                               // tile element (q, p, r) of lane l at [(q * 4 + p) * 4 + r][l]: conflict-free
        float* rw = red[wave - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[((q * 4 + p) * 4 + r) * 64 + lane] = active ? acc[q][p][r] : 0.f;
        if (lq == 0) {
            rw[4096 + 4 * li + 0] = active ? dbs.x : 0.f; rw[4096 + 4 * li + 1] = active ? dbs.y : 0.f;
            rw[4096 + 4 * li + 2] = active ? dbs.z : 0.f; rw[4096 + 4 * li + 3] = active ? dbs.w : 0.f;
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {                                 // fixed order: slab 4g, +1, +2, +3
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][p][r] += red[w][((q * 4 + p) * 4 + r) * 64 + lane];
        dbs.x += red[w][4096 + 4 * li + 0]; dbs.y += red[w][4096 + 4 * li + 1];
        dbs.z += red[w][4096 + 4 * li + 2]; dbs.w += red[w][4096 + 4 * li + 3];
    }
    float* pw = partial + (int64_t)group * N * K;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nblk * 64 + 4 * (lq * 4 + r) + q;
            float* dst = pw + (int64_t)n * K + kslab * 64 + 4 * li;
            *reinterpret_cast<float4*>(dst) = make_float4(acc[q][0][r], acc[q][1][r], acc[q][2][r], acc[q][3][r]);
        }
    if (kslab == 0 && partial_db && lq == 0) {
        float* pd = partial_db + (int64_t)group * N;
        pd[n_base + 0] = dbs.x; pd[n_base + 1] = dbs.y; pd[n_base + 2] = dbs.z; pd[n_base + 3] = dbs.w;
    }
}

template <bool UNUSED = true>
__global__ __launch_bounds__(256, 1) void linear_wgrad_bf16x3_kernel(WgradGroup g, int N, int K, float* __restrict__ partial,
                                                                   float* __restrict__ partial_db, int64_t MC, int n_kslab, int n_slabs) {
    wgrad_bf16x3_body(g, N, K, partial, partial_db, MC, n_kslab, n_slabs, xcd_logical_block(blockIdx.x, gridDim.x));
}

// Several Linears' weight gradients in one launch (llmrec_linear_wgrad_multi_bf16x3): target t owns the logical blocks
// [block_begin[t], block_begin[t + 1]), its own K, slab count and partial buffers; one slab length for all of them, so every
// block of the launch has the same amount of work.
struct WgradMulti {
    WgradGroup g[LLMREC_WGRAD_MAX_TARGETS];
    float* partial[LLMREC_WGRAD_MAX_TARGETS];
    float* partial_db[LLMREC_WGRAD_MAX_TARGETS];
    int32_t K[LLMREC_WGRAD_MAX_TARGETS], n_kslab[LLMREC_WGRAD_MAX_TARGETS], n_slabs[LLMREC_WGRAD_MAX_TARGETS];
    int32_t block_begin[LLMREC_WGRAD_MAX_TARGETS + 1];
    int32_t n_targets;
    int64_t MC;
};
__global__ __launch_bounds__(256, 1) void linear_wgrad_bf16x3_multi_kernel(WgradMulti m, int N) {
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    int t = 0;
    while (t + 1 < m.n_targets && lb >= m.block_begin[t + 1]) ++t;
    wgrad_bf16x3_body(m.g[t], N, m.K[t], m.partial[t], m.partial_db[t], m.MC, m.n_kslab[t], m.n_slabs[t], lb - m.block_begin[t]);
}


// ---------------------------------------------------------------------------------------------
// Weight gradient, second organisation (round 3): v_mfma_f32_32x32x16_bf16, wave tile = 64 n x 128 k over 16-row steps.
//   * the 32x32x16 MFMA holds the VALU for 8 of its 32 cycles (the 16x16x32 one for 8 of 16, profiles/r02_coexec_mfma_valu.txt) and
//     a 128-wide k slab halves the dY loads / splits per X byte (dY is re-read and re-split by every k slab of a row slab);
//   * lane (i = lane & 31, h = lane >> 5) supplies the rows m0 + 8 h + j (j = 0..7) of ITS columns to the contraction slots
//     (h, j) of both operands: X as float4 at k = k0 + 4 i + c (component c -> B tile c: four interleaved 32-column tiles,
//     one instruction = two rows x 512 contiguous bytes), dY as float2 at n = 2 i + t (component t -> A tile t);
//   * D tile (t, c): column = lane & 31 -> k = k0 + 4 (lane & 31) + c, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n = 2 row + t;
//   * the four waves of a block (four consecutive row slabs of one k slab) are summed through LDS by ALL four waves
//     (each adds its quarter of the tile in wave order: deterministic) before one partial slab is written.
// Measured against five other organisations (producer / consumer pairs, loader waves, two waves per SIMD, ...): profiles/experiments/r03_wgrad.md.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma32_bf16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float2 load2_global(global_cptr base, uint32_t byte_off) {
    typedef float raw2 __attribute__((ext_vector_type(2)));
    typedef const raw2 __attribute__((address_space(1))) * graw2;
    const raw2 v = *(graw2)(base + byte_off);
    return make_float2(v.x, v.y);
}
// 8 float2 (rows j = 0..7 of one lane; component t belongs to tile t) -> the three bf16x8 fragments of both tiles
__device__ __forceinline__ void split16(const float2 (&v)[8], uint4 (&H)[2], uint4 (&M)[2], uint4 (&L)[2]) {
    uint32_t h[8][2], m[8][2], l[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x2 xy = {v[j].x, v[j].y};
        h[j][0] = __float_as_uint(xy.x) & 0xffff0000u; h[j][1] = __float_as_uint(xy.y) & 0xffff0000u;
        const f32x2 hxy = {__uint_as_float(h[j][0]), __uint_as_float(h[j][1])};
        const f32x2 r1 = xy - hxy;                                          // exact
        m[j][0] = __float_as_uint(r1.x) & 0xffff0000u; m[j][1] = __float_as_uint(r1.y) & 0xffff0000u;
        const f32x2 mxy = {__uint_as_float(m[j][0]), __uint_as_float(m[j][1])};
        const f32x2 r2 = r1 - mxy;                                          // exact, <= 8 significant bits
        l[j][0] = __float_as_uint(r2.x); l[j][1] = __float_as_uint(r2.y);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        H[t] = make_uint4(pack_hi16(h[0][t], h[1][t]), pack_hi16(h[2][t], h[3][t]), pack_hi16(h[4][t], h[5][t]), pack_hi16(h[6][t], h[7][t]));
        M[t] = make_uint4(pack_hi16(m[0][t], m[1][t]), pack_hi16(m[2][t], m[3][t]), pack_hi16(m[4][t], m[5][t]), pack_hi16(m[6][t], m[7][t]));
        L[t] = make_uint4(pack_hi16(l[0][t], l[1][t]), pack_hi16(l[2][t], l[3][t]), pack_hi16(l[4][t], l[5][t]), pack_hi16(l[6][t], l[7][t]));
    }
}

#ifdef LLMREC_TOOLS_BUILD
// instrumented build (python -m llmrec_amd.build --tools): every wave adds the span of its main loop in shader cycles and in ticks of
// the constant 100 MHz counter -> the EFFECTIVE shader clock of the launch (tools/wgrad_probe.py; profiles/experiments/r03_wgrad.md)
__device__ unsigned long long g_wgrad_clock[3];
#endif

typedef const int32_t __attribute__((address_space(4))) * const_i32p;   // constant address space: uniform loads become SCALAR loads
typedef const float __attribute__((address_space(4))) * const_f32p;
constexpr int W2_KW = 128;                                  // k columns per wave
constexpr int W2_RED_FLOATS = 4 * 128 * 64 + 4 * 64;        // LDS of the block's final reduction (129 KB)

// The accumulation loop of one wave over the rows [m_begin, m_end) of one problem. LISTED: the rows are positions in the problem's
// ascending ROW LIST (llmrec_wgrad_problem_t.row_list: the rows of dY that can be non-zero); position p reads row rows[p] of dY and X.
//   dense:  a tile's buffer descriptor = (address of its first row, bytes up to the slab's end), the lane's eight row offsets are constants;
//   listed: one descriptor per operand over the whole tensor; the 16 row ids of a tile arrive by SCALAR loads (their own counter: they
//           do not queue behind the vector loads) one tile ahead of the tile's operand loads, offset = id * row stride + column.
// Either way no row past the end is read as anything but zero (the buffer's range check), so ragged tiles and prefetches past the end
// need no special case.
template <bool LISTED>
__device__ __forceinline__ void wgrad_v2_accumulate(const WgradGroup& g, const int prob, const int kslab, const int nblk, const int64_t m_begin64,
                                                    const int64_t m_end64, const bool want_db, f32x16 (&acc)[2][4], f32x2& dbs2) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t lddy = g.lddy[prob], ldx = g.ldx[prob], M = g.M[prob];
    // row positions as wave-uniform 32-bit scalars (M < 2^30 is checked by the host)
    const int m_begin = __builtin_amdgcn_readfirstlane((int)m_begin64), m_end = __builtin_amdgcn_readfirstlane((int)m_end64);
    const int n_base = nblk * 64 + 2 * i;                         // this lane's 2 n's: n_base + t
    const int k_base = kslab * W2_KW + 4 * i;                     // this lane's 4 k's: k_base + c
    const uint32_t la = 4u * (uint32_t)lddy, lbx = 4u * (uint32_t)ldx;             // row strides in bytes
    uint32_t offa[8], offb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        offa[j] = (LISTED ? 0u : (uint32_t)(8 * h + j) * la) + 4u * (uint32_t)n_base;
        offb[j] = (LISTED ? 0u : (uint32_t)(8 * h + j) * lbx) + 4u * (uint32_t)k_base;
    }
    const char* dYb = reinterpret_cast<const char*>(uniform_u64(g.dY[prob]));
    const char* Xb = reinterpret_cast<const char*>(uniform_u64(g.X[prob]));
    const const_i32p rl = LISTED ? (const_i32p)uniform_u64(g.rows[prob]) : (const_i32p)0;
    const int n_tiles = (m_end - m_begin + 15) / 16;             // the last tile may be ragged: its missing rows read as zero
    constexpr uint32_t OOB = 0x80000000u;                        // beyond any listed problem's tensor (checked by the host: bytes < 2^31)
    __amdgpu_buffer_rsrc_t ra_all, rb_all;
    if (LISTED) {
        ra_all = __builtin_amdgcn_make_buffer_rsrc((void*)dYb, 0, (int)((uint32_t)M * la), 0x00020000);
        rb_all = __builtin_amdgcn_make_buffer_rsrc((void*)Xb, 0, (int)((uint32_t)M * lbx), 0x00020000);
    }
    // LISTED: the ids of the 16 rows of the tile at position m0 (uniform address: scalar loads); positions past the last tile re-read it
    const int m_last = n_tiles > 0 ? m_begin + 16 * (n_tiles - 1) : m_begin;
    auto load_ids = [&](int m0, int32_t (&ids)[16]) {
        if (LISTED) {
            const const_i32p q = rl + (m0 < m_last ? m0 : m_last);
#pragma unroll
            for (int r = 0; r < 16; ++r) ids[r] = q[r];
        }
    };
    auto load_tile = [&](int m0, const int32_t (&ids)[16], float2 (&aa)[8], float4 (&bb)[8]) {
        if (LISTED) {
            const int left = m_end - m0 - 8 * h;                     // this half's rows j < left are inside the list
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t id = (uint32_t)(h ? ids[8 + j] : ids[j]);
                const bool ok = j < left;
                const u32x2 va = __builtin_amdgcn_raw_buffer_load_b64(ra_all, ok ? id * la + offa[j] : OOB, 0, 0);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rb_all, ok ? id * lbx + offb[j] : OOB, 0, 0);
                aa[j] = make_float2(__uint_as_float(va.x), __uint_as_float(va.y));
                bb[j] = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
            }
        } else {
            const int mc = m0 < m_end ? m0 : m_end;                  // wave-uniform; at m_end the buffers are empty: zeros
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(dYb + (uint64_t)mc * la), 0, (int)((uint32_t)(m_end - mc) * la), 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + (uint64_t)mc * lbx), 0, (int)((uint32_t)(m_end - mc) * lbx), 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x2 va = __builtin_amdgcn_raw_buffer_load_b64(ra, offa[j], 0, 0);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rb, offb[j], 0, 0);
                aa[j] = make_float2(__uint_as_float(va.x), __uint_as_float(va.y));
                bb[j] = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
            }
        }
    };
    // bias gradient: column sums of dY, optionally weighted per row (a pre-propagated operand: db = sum_r (A 1)[r] dY[r]); only the
    // k-slab-0 waves' sums are written, so only they fetch the weights (16 scalar loads per tile, rows clamped into the problem)
    const const_f32p dbw = want_db ? (const_f32p)uniform_u64(g.db_w[prob]) : (const_f32p)0;
    auto mma_tile = [&](int m0, const int32_t (&ids)[16], const float2 (&aa)[8], const float4 (&bb)[8]) {
        if (dbw) {                                                   // wave-uniform
            const int last = (int)M - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r0, r1;
                if (LISTED) {                                        // (a position past the end has a zero dY row: any weight will do)
                    r0 = m0 + j < m_end ? ids[j] : 0; r1 = m0 + 8 + j < m_end ? ids[8 + j] : 0;
                } else {
                    r0 = m0 + j < last ? m0 + j : last; r1 = m0 + 8 + j < last ? m0 + 8 + j : last;
                }
                const float w0 = dbw[r0], w1 = dbw[r1];              // uniform addresses: scalar loads
                const float wj = h ? w1 : w0;
                dbs2.x = fmaf(wj, aa[j].x, dbs2.x); dbs2.y = fmaf(wj, aa[j].y, dbs2.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f32x2 v = {aa[j].x, aa[j].y}; dbs2 += v; }
        }
        uint4 bh[4], bm[4], bl[4], ah[2], am[2], al[2];
        split32(bb, bh, bm, bl);
        split16(aa, ah, am, al);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // smallest terms first (as in the forward)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(al[t], bh[c], acc[t][c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(ah[t], bl[c], acc[t][c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(am[t], bm[c], acc[t][c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(am[t], bh[c], acc[t][c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(ah[t], bm[c], acc[t][c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = mfma32_bf16(ah[t], bh[c], acc[t][c]);
        }
    };
    // two register stages rotate statically; the loop body is straight-line (pairs of tiles), an odd last tile follows it.
    // (LISTED: c0 / c1 = the ids of the tiles held in stages 0 / 1 (the bias weights of mma_tile index by them); nx = the ids of the NEXT tile
    //  to load, fetched right after the previous tile's operand loads were issued, i.e. a whole multiply phase ahead of their use)
    float2 a0[8], a1[8];
    float4 b0[8], b1[8];
    int32_t c0[16] = {}, c1[16] = {}, nx[16] = {};
    auto keep = [&](int32_t (&dst)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = nx[r];
    };
    int m0 = m_begin;
    load_ids(m0, nx);
    load_tile(m0, nx, a0, b0); keep(c0);
    load_ids(m0 + 16, nx);
#ifdef LLMREC_TOOLS_BUILD
    const long long clk0 = clock64(), ref0 = wall_clock64();
#endif
    for (int g2 = n_tiles >> 1; g2 > 0; --g2) {
        load_tile(m0 + 16, nx, a1, b1); keep(c1);
        load_ids(m0 + 32, nx);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(m0, c0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(m0 + 32, nx, a0, b0); keep(c0);
        load_ids(m0 + 48, nx);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(m0 + 16, c1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        m0 += 32;
    }
#ifdef LLMREC_TOOLS_BUILD
    if (lane == 0) { atomicAdd(&g_wgrad_clock[0], (unsigned long long)(clock64() - clk0)); atomicAdd(&g_wgrad_clock[1], (unsigned long long)(wall_clock64() - ref0)); atomicAdd(&g_wgrad_clock[2], 1ull); }
#endif
    if (n_tiles & 1) mma_tile(m0, c0, a0, b0);
}

__device__ __forceinline__ void wgrad_bf16x3_v2_body(const WgradGroup& g, int N, int K, float* __restrict__ partial,
                                                     float* __restrict__ partial_db, int64_t MC, int n_kslab, int n_slabs, int lb, float* red) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int kslab = lb % n_kslab;
    const int group = lb / n_kslab;
    const int slab_raw = group * 4 + wave;
    const bool active = slab_raw < n_slabs;
    const int slab = active ? slab_raw : n_slabs - 1;             // idle waves of the last group recompute a slab and drop it
    int prob = 0;
    while (prob + 1 < g.n_problems && slab >= g.chunk_begin[prob + 1]) ++prob;
    const int nblk = blockIdx.y;
    const bool listed = g.rows[prob] != nullptr;                  // wave-uniform
    // a listed problem's slabs are equal pieces of the ACTUAL list length (read here), whatever length the host laid the launch out for
    int64_t M = g.M[prob], mc = MC;
    if (listed) {
        M = *reinterpret_cast<const int32_t*>(uniform_u64(g.n_rows[prob]));
        M = M < 0 ? 0 : (M > g.M[prob] ? g.M[prob] : M);
        const int64_t ns = g.chunk_begin[prob + 1] - g.chunk_begin[prob];
        mc = ((M + ns - 1) / ns + 15) / 16 * 16;
        if (mc < 16) mc = 16;
    }
    int64_t m_begin = (int64_t)(slab - g.chunk_begin[prob]) * mc;
    if (m_begin > M) m_begin = (M + 15) / 16 * 16;                // an empty slab (16-aligned: the id loads stay aligned and inside the list's padding)
    const int64_t m_end = (m_begin + mc < M) ? m_begin + mc : (m_begin < M ? M : m_begin);

    f32x16 acc[2][4];                                             // [t][c]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
    f32x2 dbs2 = {0.f, 0.f};
    const bool want_db = kslab == 0 && partial_db;
    if (listed) wgrad_v2_accumulate<true>(g, prob, kslab, nblk, m_begin, m_end, want_db, acc, dbs2);
    else wgrad_v2_accumulate<false>(g, prob, kslab, nblk, m_begin, m_end, want_db, acc, dbs2);
    float2 dbs = make_float2(dbs2.x, dbs2.y);
    dbs.x += __shfl_xor(dbs.x, 32, 64); dbs.y += __shfl_xor(dbs.y, 32, 64);
    // element (t, c, r) of lane l at [wave][((t * 4 + c) * 16 + r) * 64 + l]: conflict-free
    float* rw = red + wave * (128 * 64);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) rw[((t * 4 + c) * 16 + r) * 64 + lane] = active ? acc[t][c][r] : 0.f;
    if (h == 0) {
        red[4 * 128 * 64 + wave * 64 + 2 * i + 0] = active ? dbs.x : 0.f;
        red[4 * 128 * 64 + wave * 64 + 2 * i + 1] = active ? dbs.y : 0.f;
    }
    __syncthreads();
    // wave w adds rows t = w >> 1, reg r in [8 (w & 1), +8) of all four waves' tiles, in wave order (slab 4 g, +1, +2, +3)
    float* pw = partial + (int64_t)group * N * K;
    const int t_out = wave >> 1;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * (wave & 1) + rr;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = ((t_out * 4 + c) * 16 + r) * 64 + lane;
            v[c] = ((red[e] + red[128 * 64 + e]) + red[2 * 128 * 64 + e]) + red[3 * 128 * 64 + e];
        }
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = nblk * 64 + 2 * row + t_out;
        *reinterpret_cast<float4*>(pw + (int64_t)n * K + kslab * W2_KW + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (kslab == 0 && partial_db && wave == 0) {
        const float* rd = red + 4 * 128 * 64;
        partial_db[(int64_t)group * N + nblk * 64 + lane] = ((rd[lane] + rd[64 + lane]) + rd[128 + lane]) + rd[192 + lane];
    }
}

__global__ __launch_bounds__(256, 1) void linear_wgrad_bf16x3_v2_multi_kernel(WgradMulti m, int N) {
    __shared__ __attribute__((aligned(16))) float red[W2_RED_FLOATS];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    int t = 0;
    while (t + 1 < m.n_targets && lb >= m.block_begin[t + 1]) ++t;
    wgrad_bf16x3_v2_body(m.g[t], N, m.K[t], m.partial[t], m.partial_db[t], m.MC, m.n_kslab[t], m.n_slabs[t], lb - m.block_begin[t], red);
}

struct ReduceMulti {
    const float* partial[LLMREC_WGRAD_MAX_TARGETS]; const float* partial_b[LLMREC_WGRAD_MAX_TARGETS];
    float* out[LLMREC_WGRAD_MAX_TARGETS]; float* out_b[LLMREC_WGRAD_MAX_TARGETS];
    int64_t ld_out[LLMREC_WGRAD_MAX_TARGETS], n_elem[LLMREC_WGRAD_MAX_TARGETS], n_chunks[LLMREC_WGRAD_MAX_TARGETS];
    int32_t row_len[LLMREC_WGRAD_MAX_TARGETS], n_b[LLMREC_WGRAD_MAX_TARGETS], accumulate[LLMREC_WGRAD_MAX_TARGETS];
    int32_t block_begin[LLMREC_WGRAD_MAX_TARGETS + 1];
    int32_t n_targets;
};
// the optional AdamW update riding in the reduction launch (llmrec_linear_wgrad_multi_adamw_bf16x3)
struct ReduceUpdate {
    float* p[LLMREC_WGRAD_MAX_TARGETS]; float* mo[LLMREC_WGRAD_MAX_TARGETS]; float* vo[LLMREC_WGRAD_MAX_TARGETS];
    float* pb[LLMREC_WGRAD_MAX_TARGETS]; float* mb[LLMREC_WGRAD_MAX_TARGETS]; float* vb[LLMREC_WGRAD_MAX_TARGETS];
    float gscale[LLMREC_WGRAD_MAX_TARGETS];
    const float* state;
    float decay_mul, b1, b2, eps;
};
// reduce_chunks_kernel for every target of a multi-target launch (same chunk order, same four-way summation)
template <bool UPDATE>
__global__ void reduce_chunks_multi_kernel(ReduceMulti m, ReduceUpdate u) {
    int t = 0;
    while (t + 1 < m.n_targets && (int)blockIdx.x >= m.block_begin[t + 1]) ++t;
    const int64_t e = (int64_t)(blockIdx.x - m.block_begin[t]) * blockDim.x + threadIdx.x;
    const int64_t n_elem = m.n_elem[t], n_chunks = m.n_chunks[t];
    const int n_b = m.n_b[t];
    if (e >= n_elem + n_b) return;
    const bool second = e >= n_elem;
    const float* src = second ? m.partial_b[t] + (e - n_elem) : m.partial[t] + e;
    const int64_t stride = second ? n_b : n_elem;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t c = 0;
    for (; c + 4 <= n_chunks; c += 4) {
        s0 += src[c * stride];
        s1 += src[(c + 1) * stride];
        s2 += src[(c + 2) * stride];
        s3 += src[(c + 3) * stride];
    }
    for (; c < n_chunks; ++c) s0 += src[c * stride];
    const float s = (s0 + s1) + (s2 + s3);
    float* o;
    if (second) o = m.out_b[t] + (e - n_elem);
    else { const int64_t r = e / m.row_len[t], col = e % m.row_len[t]; o = m.out[t] + r * m.ld_out[t] + col; }
    const float g = m.accumulate[t] ? (*o + s) : s;
    *o = g;
    if (UPDATE) {                                                // adamw_multi_kernel's arithmetic (rowops.hip) on this element
        const int64_t i = second ? e - n_elem : e;               // (W contiguous: ld_out == row_len)
        float* __restrict__ p = (second ? u.pb[t] : u.p[t]) + i;
        float* __restrict__ mo = (second ? u.mb[t] : u.mo[t]) + i;
        float* __restrict__ vo = (second ? u.vb[t] : u.vo[t]) + i;
        const float gs = u.gscale[t];
        const float step_size = u.state[1], bc2s = u.state[2];
        const float w1 = 1.0f - u.b1, w2 = 1.0f - u.b2;
        const float gi = gs == 1.0f ? g : gs * g;
        float pi = *p * u.decay_mul;
        const float mi = *mo + w1 * (gi - *mo);
        const float vi = fmaf(w2 * gi, gi, *vo * u.b2);
        const float denom = sqrtf(vi) / bc2s + u.eps;
        pi = pi - step_size * (mi / denom);
        *p = pi; *mo = mi; *vo = vi;
    }
}

// out[e] (+)= sum over chunks of partial[chunk][e], chunks in ascending order (deterministic); a second, short
// segment (the bias gradient) rides in the same launch: elements [n_elem, n_elem + n_b) come from partial_b.
__global__ void reduce_chunks_kernel(int64_t n_elem, int64_t n_chunks, const float* __restrict__ partial,
                                     float* __restrict__ out, int64_t ld_out, int row_len, int accumulate,
                                     int n_b, const float* __restrict__ partial_b, float* __restrict__ out_b) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem + n_b; e += (int64_t)gridDim.x * blockDim.x) {
        const bool second = e >= n_elem;
        const float* src = second ? partial_b + (e - n_elem) : partial + e;
        const int64_t stride = second ? n_b : n_elem;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int64_t c = 0;
        for (; c + 4 <= n_chunks; c += 4) {                      // 4 independent loads in flight
            s0 += src[c * stride];
            s1 += src[(c + 1) * stride];
            s2 += src[(c + 2) * stride];
            s3 += src[(c + 3) * stride];
        }
        for (; c < n_chunks; ++c) s0 += src[c * stride];
        const float s = (s0 + s1) + (s2 + s3);
        float* o;
        if (second) o = out_b + (e - n_elem);
        else { const int64_t r = e / row_len, col = e % row_len; o = out + r * ld_out + col; }
        *o = accumulate ? (*o + s) : s;
    }
}

// rows per slab of the fp32 kernel: ~2048 waves over all problems (one round at two waves per SIMD); a multiple of
// its rotation length (48 rows = 3 x 16)
static int64_t wgrad_slab_rows(int64_t M_total, int K, bool bf16x3) {
    const int64_t n_kslab = ceil_div(K, 64);
    const int64_t want_slabs = ceil_div(2048, n_kslab);
    const int64_t mc = align_up(ceil_div(M_total > 0 ? M_total : 1, want_slabs), bf16x3 ? 64 : 48);
    return (!bf16x3 && mc < 144) ? 144 : mc;
}

// rows per slab of the bf16x3 kernel. It runs ONE block (4 waves = 4 slabs of one k slab) per CU at a time, so the
// launch takes ceil(blocks / 256) rounds of (rows per slab + a fixed prologue / LDS-reduction cost): 528 blocks are
// three rounds where 480 are two. Pick the slab length (a multiple of 32, at least half the nominal one so that the
// workspace bound of the fp32 geometry still holds) that minimises rounds x length; ties go to the longer slab
// (fewer partial slabs to reduce).
static int64_t wgrad_slab_rows_bf16(int32_t n_problems, const llmrec_wgrad_problem_t* p, int64_t M_total, int K) {
    const int64_t n_kslab = ceil_div(K, 64);
    const int64_t lo = std::max<int64_t>(64, align_up(wgrad_slab_rows(M_total, K, true) / 2, 32));
    int64_t m_max = 0;
    for (int i = 0; i < n_problems; ++i) m_max = std::max(m_max, p[i].M);
    int64_t best = lo, best_cost = -1;
    for (int64_t mc = lo; mc <= align_up(m_max, 32) + 32; mc += 32) {
        int64_t slabs = 0;
        for (int i = 0; i < n_problems; ++i) slabs += ceil_div(p[i].M, mc);
        const int64_t blocks = ceil_div(slabs, 4) * n_kslab;
        const int64_t cost = ceil_div(blocks, 256) * (mc + 96);
        if (best_cost < 0 || cost <= best_cost) { best = mc; best_cost = cost; }
    }
    return best;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

int llmrec_linear_fwd_f32(int64_t M, int32_t N, int32_t K, const float* X, int64_t ldx,
                          const float* W, int64_t ldw, const float* bias,
                          float* Y, int64_t ldy, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear_fwd: bad sizes");
    if (M == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(X && W && Y && ldx >= K && ldw >= K && ldy >= N, "linear_fwd: null pointer or ld too small");
    if (N > 128) { set_error("linear_fwd: N = %d > 128", N); return LLMREC_EUNSUPPORTED; }
    const int vec_ok = (ldx % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)X | (uintptr_t)W) % 16 == 0);
    const int NT = (N + 15) / 16;
    // enough row tiles to keep >= 2 waves per SIMD busy -> row mode (32-row tiles); otherwise the
    // 4 waves of a block split K over one 16-row tile
    const bool rows_mode = M >= 32 * 2048;
#define LAUNCH_ROWS(NT_, RT_)                                                                         \
    linear_fwd_kernel<NT_, RT_, false><<<(int)ceil_div(M, 4 * 16 * RT_), 256, 0, stream>>>(M, N, K, X, ldx, W, ldw, bias, Y, ldy, vec_ok)
#define LAUNCH_SPLITK(NT_)                                                                            \
    linear_fwd_kernel<NT_, 1, true><<<(int)ceil_div(M, 16), 256, sizeof(float) * 4 * 16 * 16 * NT_, stream>>>(M, N, K, X, ldx, W, ldw, bias, Y, ldy, vec_ok)
    switch (NT) {
        case 1: if (rows_mode) LAUNCH_ROWS(1, 2); else LAUNCH_SPLITK(1); break;
        case 2: if (rows_mode) LAUNCH_ROWS(2, 2); else LAUNCH_SPLITK(2); break;
        case 3: if (rows_mode) LAUNCH_ROWS(3, 2); else LAUNCH_SPLITK(3); break;
        case 4: if (rows_mode) LAUNCH_ROWS(4, 2); else LAUNCH_SPLITK(4); break;
        case 5: if (rows_mode) LAUNCH_ROWS(5, 1); else LAUNCH_SPLITK(5); break;
        case 6: if (rows_mode) LAUNCH_ROWS(6, 1); else LAUNCH_SPLITK(6); break;
        case 7: if (rows_mode) LAUNCH_ROWS(7, 1); else LAUNCH_SPLITK(7); break;
        default: if (rows_mode) LAUNCH_ROWS(8, 1); else LAUNCH_SPLITK(8); break;
    }
#undef LAUNCH_ROWS
#undef LAUNCH_SPLITK
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

static int linear_fwd_grouped_impl(int32_t n_problems, const llmrec_linear_problem_t* p, int32_t N, bool bf16x3, llmrec_stream_t stream_);

int llmrec_linear_fwd_grouped_f32(int32_t n_problems, const llmrec_linear_problem_t* p, int32_t N, llmrec_stream_t stream_) {
    return linear_fwd_grouped_impl(n_problems, p, N, false, stream_);
}

int llmrec_linear_fwd_grouped_bf16x3(int32_t n_problems, const llmrec_linear_problem_t* p, int32_t N, llmrec_stream_t stream_) {
    return linear_fwd_grouped_impl(n_problems, p, N, true, stream_);
}

static int linear_fwd_grouped_impl(int32_t n_problems, const llmrec_linear_problem_t* p, int32_t N, bool bf16x3, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_LINEAR_MAX_PROBLEMS && p && N > 0, "linear_fwd_grouped: bad argument");
    if (N > 64) { set_error("linear_fwd_grouped: N = %d > 64", N); return LLMREC_EUNSUPPORTED; }
    LinearGroup g = {};
    g.n_problems = n_problems;
    int units = 0;
    for (int i = 0; i < n_problems; ++i) {
        LLMREC_CHECK_ARG(p[i].M >= 0 && p[i].K > 0, "linear_fwd_grouped: problem %d has bad sizes", i);
        LLMREC_CHECK_ARG(p[i].M == 0 || (p[i].X && p[i].W && p[i].Y && p[i].ldx >= p[i].K && p[i].ldw >= p[i].K && p[i].ldy >= N),
                         "linear_fwd_grouped: problem %d has a null pointer or a small ld", i);
        g.X[i] = p[i].X; g.W[i] = p[i].W; g.bias[i] = p[i].bias; g.bias_scale[i] = p[i].bias ? p[i].bias_scale : nullptr; g.Y[i] = p[i].Y;
        g.ldx[i] = p[i].ldx; g.ldw[i] = p[i].ldw; g.ldy[i] = p[i].ldy; g.M[i] = p[i].M; g.K[i] = p[i].K;
        g.vec_ok[i] = (p[i].ldx % 4 == 0) && (p[i].ldw % 4 == 0) && (((uintptr_t)p[i].X | (uintptr_t)p[i].W) % 16 == 0);
    }
    // 128-row work units (W tile amortised over 128 rows); 64-row units only when the whole launch
    // would not even give one block per CU (measured: at 1056 units, 64-row units are 3-12 % slower)
    int64_t units128 = 0;
    for (int i = 0; i < n_problems; ++i) units128 += ceil_div(p[i].M, 128);
    const int rows_per_unit = units128 >= 256 ? 128 : 64;
    for (int i = 0; i < n_problems; ++i) {
        g.unit_begin[i] = units;
        units += (int)ceil_div(p[i].M, rows_per_unit);
    }
    for (int i = n_problems; i <= LLMREC_LINEAR_MAX_PROBLEMS; ++i) g.unit_begin[i] = units;
    if (units == 0) return LLMREC_OK;
    bool fast = true;                                  // every problem: 16-byte aligned rows, K % 4 == 0
    for (int i = 0; i < n_problems; ++i) fast = fast && g.vec_ok[i] && (g.K[i] % 4 == 0) && g.K[i] >= 4;
    bool k32 = fast;                                   // scalar-k addressing of the bf16x3 kernel (MODE 2)
    for (int i = 0; i < n_problems; ++i)
        k32 = k32 && (g.K[i] % 32 == 0) && (p[i].M * p[i].ldx + g.K[i]) < (1ll << 30) && ((int64_t)N * p[i].ldw + g.K[i]) < (1ll << 30);
#define GROUPED_LAUNCH(KERNEL, NT_)                                                         \
    do {                                                                                    \
        if (rows_per_unit == 128) {                                                         \
            if (k32) KERNEL<NT_, 2, K32MODE><<<units, 256, 0, stream>>>(g, N);               \
            else if (fast) KERNEL<NT_, 2, 1><<<units, 256, 0, stream>>>(g, N);               \
            else KERNEL<NT_, 2, 0><<<units, 256, 0, stream>>>(g, N);                         \
        } else {                                                                            \
            if (k32) KERNEL<NT_, 1, K32MODE><<<units, 256, 0, stream>>>(g, N);               \
            else if (fast) KERNEL<NT_, 1, 1><<<units, 256, 0, stream>>>(g, N);               \
            else KERNEL<NT_, 1, 0><<<units, 256, 0, stream>>>(g, N);                         \
        }                                                                                   \
    } while (0)
    const int nt = (N + 15) / 16;
    if (bf16x3) {
#define K32MODE 3
        if (nt == 1) GROUPED_LAUNCH(linear_fwd_grouped_bf16x3_kernel, 1);
        else if (nt == 2) GROUPED_LAUNCH(linear_fwd_grouped_bf16x3_kernel, 2);
        else if (nt == 3) GROUPED_LAUNCH(linear_fwd_grouped_bf16x3_kernel, 3);
        else GROUPED_LAUNCH(linear_fwd_grouped_bf16x3_kernel, 4);
#undef K32MODE
    } else {
#define K32MODE 2
        if (nt == 1) GROUPED_LAUNCH(linear_fwd_grouped_kernel, 1);
        else if (nt == 2) GROUPED_LAUNCH(linear_fwd_grouped_kernel, 2);
        else if (nt == 3) GROUPED_LAUNCH(linear_fwd_grouped_kernel, 3);
        else GROUPED_LAUNCH(linear_fwd_grouped_kernel, 4);
#undef K32MODE
    }
#undef GROUPED_LAUNCH
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

static int64_t wgrad_ws_bytes(int64_t n_slabs, int N, int K) {
    return align_up(4 * n_slabs * (int64_t)N * K, 256) + align_up(4 * n_slabs * (int64_t)N, 256);
}

int64_t llmrec_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K) {
    if (M < 0 || N <= 0 || K <= 0) return -1;
    // bound for any split of M rows into <= LLMREC_LINEAR_MAX_PROBLEMS problems
    const int64_t mc = std::min(wgrad_slab_rows(M, K, false), wgrad_slab_rows(M, K, true));   // either kernel
    return wgrad_ws_bytes(ceil_div(M > 0 ? M : 1, mc) + LLMREC_LINEAR_MAX_PROBLEMS, N, K);
}

static int linear_wgrad_grouped_impl(int32_t n_problems, const llmrec_wgrad_problem_t* p, int32_t N, int32_t K,
                                     float* dW, int64_t lddw, float* db, int32_t accumulate,
                                     void* workspace, int64_t workspace_bytes, bool bf16x3, llmrec_stream_t stream_);

int llmrec_linear_wgrad_grouped_f32(int32_t n_problems, const llmrec_wgrad_problem_t* p, int32_t N, int32_t K,
                                    float* dW, int64_t lddw, float* db, int32_t accumulate,
                                    void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    return linear_wgrad_grouped_impl(n_problems, p, N, K, dW, lddw, db, accumulate, workspace, workspace_bytes, false, stream_);
}

int llmrec_linear_wgrad_grouped_bf16x3(int32_t n_problems, const llmrec_wgrad_problem_t* p, int32_t N, int32_t K,
                                       float* dW, int64_t lddw, float* db, int32_t accumulate,
                                       void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    return linear_wgrad_grouped_impl(n_problems, p, N, K, dW, lddw, db, accumulate, workspace, workspace_bytes, true, stream_);
}

static int linear_wgrad_grouped_impl(int32_t n_problems, const llmrec_wgrad_problem_t* p, int32_t N, int32_t K,
                                     float* dW, int64_t lddw, float* db, int32_t accumulate,
                                     void* workspace, int64_t workspace_bytes, bool bf16x3, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_LINEAR_MAX_PROBLEMS && p && N > 0 && K > 0, "linear_wgrad: bad argument");
    LLMREC_CHECK_ARG(dW && lddw >= K, "linear_wgrad: null dW or ld < K");
    int64_t M_total = 0;
    for (int i = 0; i < n_problems; ++i) {
        LLMREC_CHECK_ARG(p[i].M >= 0 && (p[i].M == 0 || (p[i].dY && p[i].X && p[i].lddy >= N && p[i].ldx >= K)),
                         "linear_wgrad: problem %d has a null pointer or a small ld", i);
        M_total += p[i].M;
    }
    if (M_total == 0) {
        if (!accumulate) {
            LLMREC_HIP(hipMemset2DAsync(dW, 4 * lddw, 0, 4 * (size_t)K, N, stream));
            if (db) LLMREC_HIP(hipMemsetAsync(db, 0, 4 * (size_t)N, stream));
        }
        return LLMREC_OK;
    }
    bool fast_shape = (N % 64 == 0) && (K % 64 == 0);
    for (int i = 0; i < n_problems; ++i)
        fast_shape = fast_shape && (p[i].lddy % 4 == 0) && (p[i].ldx % 4 == 0) && (((uintptr_t)p[i].dY | (uintptr_t)p[i].X) % 16 == 0);
    bool small_offsets = true;                                              // the bf16x3 kernel addresses rows with 32-bit byte offsets
    for (int i = 0; i < n_problems; ++i)
        small_offsets = small_offsets && (p[i].M + 256) * std::max(p[i].lddy, p[i].ldx) < (1ll << 30);   // byte offsets < 2^32
    const bool use_bf16 = bf16x3 && fast_shape && small_offsets;
    if (use_bf16 && N == 64 && K % W2_KW == 0) {
        // the 128-wide-k-slab organisation lives in the multi-target launch: one target (its workspace need is below this
        // entry point's bound, checked there)
        bool nonempty = true;
        for (int i = 0; i < n_problems; ++i) nonempty = nonempty && p[i].M > 0;
        if (nonempty) {
            llmrec_wgrad_target_t one = {n_problems, p, K, dW, lddw, db, accumulate};
            return llmrec_linear_wgrad_multi_bf16x3(1, &one, N, workspace, workspace_bytes, stream_);
        }
    }
    const int64_t MC = use_bf16 ? wgrad_slab_rows_bf16(n_problems, p, M_total, K) : wgrad_slab_rows(M_total, K, false);
    WgradGroup g = {};
    g.n_problems = n_problems;
    int n_slabs = 0;
    for (int i = 0; i < n_problems; ++i) {
        g.dY[i] = p[i].dY; g.X[i] = p[i].X; g.lddy[i] = p[i].lddy; g.ldx[i] = p[i].ldx; g.M[i] = p[i].M;
        if (p[i].db_row_weight) { set_error("linear_wgrad: db_row_weight is served by the 128-wide-k-slab bf16x3 organisation only (N = 64, K %% 128 == 0)"); return LLMREC_EUNSUPPORTED; }
        if (p[i].row_list) { set_error("linear_wgrad: row_list is served by llmrec_linear_wgrad_multi_* (bf16x3, N = 64, K %% 128 == 0) only"); return LLMREC_EUNSUPPORTED; }
        g.vec_ok[i] = (p[i].lddy % 4 == 0) && (p[i].ldx % 4 == 0) && (((uintptr_t)p[i].dY | (uintptr_t)p[i].X) % 16 == 0);
        g.chunk_begin[i] = n_slabs;
        n_slabs += (int)ceil_div(p[i].M, MC);
    }
    for (int i = n_problems; i <= LLMREC_LINEAR_MAX_PROBLEMS; ++i) g.chunk_begin[i] = n_slabs;
    if (!workspace || workspace_bytes < wgrad_ws_bytes(n_slabs, N, K)) {
        set_error("linear_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)wgrad_ws_bytes(n_slabs, N, K));
        return LLMREC_EWORKSPACE;
    }
    const int n_kslab = (int)ceil_div(K, 64);
    float* partial = (float*)workspace;
    float* partial_db = (float*)((char*)workspace + align_up(4 * (int64_t)n_slabs * N * K, 256));
    const int64_t n_waves = (int64_t)n_slabs * n_kslab;
    dim3 grid((unsigned)ceil_div(n_waves, 4), (unsigned)ceil_div(N, 64));
    bool fast = (N % 64 == 0) && (K % 64 == 0);
    for (int i = 0; i < n_problems; ++i) fast = fast && g.vec_ok[i];
    int64_t n_chunks = n_slabs;
    if (fast && use_bf16) {
        n_chunks = ceil_div(n_slabs, 4);                                 // the block sums its four slabs before writing
        dim3 grid4((unsigned)(n_chunks * n_kslab), (unsigned)(N / 64));
        linear_wgrad_bf16x3_kernel<true><<<grid4, 256, 0, stream>>>(g, N, K, partial, db ? partial_db : nullptr, MC, n_kslab, n_slabs);
    }
    else if (fast) linear_wgrad_kernel<true><<<grid, 256, 0, stream>>>(g, N, K, partial, db ? partial_db : nullptr, MC, n_kslab, n_slabs);
    else linear_wgrad_kernel<false><<<grid, 256, 0, stream>>>(g, N, K, partial, db ? partial_db : nullptr, MC, n_kslab, n_slabs);
    LLMREC_LAUNCH_CHECK();
    const int64_t ne = (int64_t)N * K;
    reduce_chunks_kernel<<<grid_for(ne + (db ? N : 0), 256), 256, 0, stream>>>(ne, n_chunks, partial, dW, lddw, K, accumulate,
                                                                              db ? N : 0, partial_db, db);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

// rows the launch geometry is laid out for: a listed problem's expected list length (llmrec_wgrad_problem_t.rows_expected), else M
static inline int64_t wgrad_rows_eff(const llmrec_wgrad_problem_t& q) {
    if (!q.row_list || q.rows_expected <= 0) return q.M;
    return std::min<int64_t>(q.M, std::max<int64_t>(q.rows_expected, 16));
}
// which organisation a multi-target launch runs: 2 = wgrad_bf16x3_v2_body (128-wide k slabs: every K % 128 == 0), 1 = wgrad_bf16x3_body
static int wgrad_multi_version(int32_t n_targets, const llmrec_wgrad_target_t* t) {
    for (int i = 0; i < n_targets; ++i)
        if (t[i].K % W2_KW) return 1;
    return 2;
}
// one slab length for every target of a multi-target launch: the launch is ceil(blocks / 256) rounds of (mc + a fixed cost)
static int64_t wgrad_multi_slab_rows(int32_t n_targets, const llmrec_wgrad_target_t* t) {
    const int kw = wgrad_multi_version(n_targets, t) == 1 ? 64 : W2_KW;
    const int64_t budget = (t[0].block_budget > 0 && t[0].block_budget <= 256) ? t[0].block_budget : 256;   // resident blocks per round
    int64_t m_max = 0, work = 0;
    for (int i = 0; i < n_targets; ++i)
        for (int j = 0; j < t[i].n_problems; ++j) {
            const int64_t me = wgrad_rows_eff(t[i].problems[j]);
            m_max = std::max(m_max, me); work += me * ceil_div(t[i].K, kw);
        }
    const int64_t lo = std::max<int64_t>(64, align_up(ceil_div(work, 4 * 2 * budget), 32)); // never more than ~2 rounds of blocks
    int64_t best = lo, best_cost = -1;
    for (int64_t mc = lo; mc <= align_up(m_max, 32) + 32; mc += 32) {
        int64_t blocks = 0;
        for (int i = 0; i < n_targets; ++i) {
            int64_t slabs = 0;
            for (int j = 0; j < t[i].n_problems; ++j) slabs += ceil_div(wgrad_rows_eff(t[i].problems[j]), mc);
            blocks += ceil_div(slabs, 4) * ceil_div(t[i].K, kw);
        }
        const int64_t cost = ceil_div(blocks, budget) * (mc + 96);
        if (best_cost < 0 || cost <= best_cost) { best = mc; best_cost = cost; }
    }
    return best;
}
static bool wgrad_multi_ok(int32_t n_targets, const llmrec_wgrad_target_t* t, int32_t N) {
    if (n_targets < 1 || n_targets > LLMREC_WGRAD_MAX_TARGETS || !t || N <= 0 || N % 64) return false;
    for (int i = 0; i < n_targets; ++i) {
        if (t[i].n_problems < 1 || t[i].n_problems > LLMREC_LINEAR_MAX_PROBLEMS || !t[i].problems || t[i].K <= 0 || t[i].K % 64 || !t[i].dW || t[i].lddw < t[i].K) return false;
        for (int j = 0; j < t[i].n_problems; ++j) {
            const llmrec_wgrad_problem_t& q = t[i].problems[j];
            if (q.M <= 0 || !q.dY || !q.X || q.lddy < N || q.ldx < t[i].K || q.lddy % 4 || q.ldx % 4 || ((uintptr_t)q.dY | (uintptr_t)q.X) % 16) return false;
            if ((q.M + 256) * std::max(q.lddy, q.ldx) >= (1ll << 30)) return false;
            // a row list: the 128-wide organisation only, a device count, byte offsets below 2^31 (the kernel's out-of-range offset)
            if (q.row_list && (!q.n_rows || t[i].K % W2_KW || q.M * std::max(q.lddy, q.ldx) * 4 >= (1ll << 31) || (uintptr_t)q.row_list % 64)) return false;
        }
    }
    return true;
}

int64_t llmrec_linear_wgrad_multi_workspace_bytes(int32_t n_targets, const llmrec_wgrad_target_t* t, int32_t N) {
    if (!wgrad_multi_ok(n_targets, t, N)) return -1;
    const int64_t mc = wgrad_multi_slab_rows(n_targets, t);
    int64_t bytes = 0;
    for (int i = 0; i < n_targets; ++i) {
        int64_t slabs = 0;
        for (int j = 0; j < t[i].n_problems; ++j) slabs += ceil_div(wgrad_rows_eff(t[i].problems[j]), mc);
        bytes += wgrad_ws_bytes(ceil_div(slabs, 4), N, t[i].K);
    }
    return bytes;
}

static int linear_wgrad_multi_impl(int32_t n_targets, const llmrec_wgrad_target_t* t, int32_t N, void* workspace, int64_t workspace_bytes,
                                   const ReduceUpdate* upd, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wgrad_multi_ok(n_targets, t, N)) {
        set_error("linear_wgrad_multi: outside the fast path (1..%d targets, N %% 64 == 0, K %% 64 == 0, non-empty 16-byte aligned problems, 32-bit offsets)",
                  LLMREC_WGRAD_MAX_TARGETS);
        return LLMREC_EUNSUPPORTED;
    }
    const int64_t need = llmrec_linear_wgrad_multi_workspace_bytes(n_targets, t, N);
    if (!workspace || workspace_bytes < need || (uintptr_t)workspace % 16) {
        set_error("linear_wgrad_multi: workspace %lld < %lld (16-byte aligned)", (long long)workspace_bytes, (long long)need);
        return LLMREC_EWORKSPACE;
    }
    WgradMulti m = {};
    ReduceMulti r = {};
    m.n_targets = r.n_targets = n_targets;
    m.MC = wgrad_multi_slab_rows(n_targets, t);
    const int version = wgrad_multi_version(n_targets, t);
    const int kw = version == 1 ? 64 : W2_KW;
    char* ws = (char*)workspace;
    int blocks = 0, rblocks = 0;
    for (int i = 0; i < n_targets; ++i) {
        WgradGroup& g = m.g[i];
        g.n_problems = t[i].n_problems;
        int n_slabs = 0;
        for (int j = 0; j < t[i].n_problems; ++j) {
            const llmrec_wgrad_problem_t& q = t[i].problems[j];
            g.dY[j] = q.dY; g.X[j] = q.X; g.lddy[j] = q.lddy; g.ldx[j] = q.ldx; g.M[j] = q.M; g.vec_ok[j] = 1; g.db_w[j] = q.db_row_weight;
            g.rows[j] = q.row_list; g.n_rows[j] = q.row_list ? q.n_rows : nullptr;
            if (q.row_list && version == 1) { set_error("linear_wgrad_multi: row_list needs every K %% %d == 0", W2_KW); return LLMREC_EUNSUPPORTED; }
            if (q.db_row_weight && version == 1) { set_error("linear_wgrad_multi: db_row_weight needs every K %% %d == 0", W2_KW); return LLMREC_EUNSUPPORTED; }
            g.chunk_begin[j] = n_slabs;
            n_slabs += (int)ceil_div(wgrad_rows_eff(q), m.MC);
        }
        for (int j = t[i].n_problems; j <= LLMREC_LINEAR_MAX_PROBLEMS; ++j) g.chunk_begin[j] = n_slabs;
        const int64_t n_chunks = ceil_div(n_slabs, 4);
        m.K[i] = t[i].K; m.n_kslab[i] = t[i].K / kw; m.n_slabs[i] = n_slabs;
        m.partial[i] = (float*)ws;
        m.partial_db[i] = t[i].db ? (float*)(ws + align_up(4 * n_chunks * N * t[i].K, 256)) : nullptr;
        ws += wgrad_ws_bytes(n_chunks, N, t[i].K);
        m.block_begin[i] = blocks;
        blocks += (int)(n_chunks * m.n_kslab[i]);
        r.partial[i] = m.partial[i]; r.partial_b[i] = m.partial_db[i]; r.out[i] = t[i].dW; r.out_b[i] = t[i].db; r.ld_out[i] = t[i].lddw;
        r.n_elem[i] = (int64_t)N * t[i].K; r.n_chunks[i] = n_chunks; r.row_len[i] = t[i].K; r.n_b[i] = t[i].db ? N : 0; r.accumulate[i] = t[i].accumulate;
        r.block_begin[i] = rblocks;
        rblocks += (int)ceil_div(r.n_elem[i] + r.n_b[i], 256);
    }
    for (int i = n_targets; i <= LLMREC_WGRAD_MAX_TARGETS; ++i) { m.block_begin[i] = blocks; r.block_begin[i] = rblocks; }
    const dim3 grid((unsigned)blocks, (unsigned)(N / 64));
    if (version == 1) linear_wgrad_bf16x3_multi_kernel<<<grid, 256, 0, stream>>>(m, N);
    else linear_wgrad_bf16x3_v2_multi_kernel<<<grid, 256, 0, stream>>>(m, N);
    LLMREC_LAUNCH_CHECK();
    if (upd) reduce_chunks_multi_kernel<true><<<rblocks, 256, 0, stream>>>(r, *upd);
    else reduce_chunks_multi_kernel<false><<<rblocks, 256, 0, stream>>>(r, ReduceUpdate{});
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_linear_wgrad_multi_bf16x3(int32_t n_targets, const llmrec_wgrad_target_t* t, int32_t N,
                                     void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    return linear_wgrad_multi_impl(n_targets, t, N, workspace, workspace_bytes, nullptr, stream_);
}

int llmrec_linear_wgrad_multi_adamw_bf16x3(int32_t n_targets, const llmrec_wgrad_target_t* t, int32_t N,
                                           void* workspace, int64_t workspace_bytes, const llmrec_wgrad_update_t* upd,
                                           const float* state3, float lr, float beta1, float beta2, float eps, float weight_decay,
                                           llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(upd && state3 && n_targets >= 1 && n_targets <= LLMREC_WGRAD_MAX_TARGETS && t, "linear_wgrad_multi_adamw: bad argument");
    ReduceUpdate u = {};
    for (int i = 0; i < n_targets; ++i) {
        LLMREC_CHECK_ARG(upd[i].W && upd[i].m_W && upd[i].v_W, "linear_wgrad_multi_adamw: target %d has a null parameter / moment pointer", i);
        LLMREC_CHECK_ARG(!t[i].db || (upd[i].b && upd[i].m_b && upd[i].v_b), "linear_wgrad_multi_adamw: target %d has db but no bias parameter / moments", i);
        LLMREC_CHECK_ARG(t[i].lddw == t[i].K, "linear_wgrad_multi_adamw: target %d: dW (and W) must be contiguous (lddw == K)", i);
        LLMREC_CHECK_ARG(upd[i].g_scale != 0.0f, "linear_wgrad_multi_adamw: target %d has g_scale 0 (set 1 for a plain gradient)", i);
        u.p[i] = upd[i].W; u.mo[i] = upd[i].m_W; u.vo[i] = upd[i].v_W; u.pb[i] = upd[i].b; u.mb[i] = upd[i].m_b; u.vb[i] = upd[i].v_b;
        u.gscale[i] = upd[i].g_scale;
    }
    u.state = state3; u.decay_mul = (float)(1.0 - (double)lr * (double)weight_decay); u.b1 = beta1; u.b2 = beta2; u.eps = eps;
    return linear_wgrad_multi_impl(n_targets, t, N, workspace, workspace_bytes, &u, stream_);
}

#ifdef LLMREC_TOOLS_BUILD
// tools only: {shader cycles, 100 MHz ticks, waves} summed over the main loops of the v2 weight-gradient waves since the last call (then cleared)
int llmrec_tools_wgrad_clock(unsigned long long* out3_host) {
    if (hipMemcpyFromSymbol(out3_host, HIP_SYMBOL(g_wgrad_clock), sizeof(unsigned long long) * 3) != hipSuccess) return LLMREC_EHIP;
    unsigned long long zero[3] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_clock), zero, sizeof(zero)) != hipSuccess) return LLMREC_EHIP;
    return LLMREC_OK;
}
#endif

int llmrec_linear_wgrad_f32(int64_t M, int32_t N, int32_t K, const float* dY, int64_t lddy,
                            const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                            int32_t accumulate, void* workspace, int64_t workspace_bytes,
                            llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(M >= 0, "linear_wgrad: bad sizes");
    llmrec_wgrad_problem_t one = {dY, lddy, X, ldx, M, nullptr};
    return llmrec_linear_wgrad_grouped_f32(1, &one, N, K, dW, lddw, db, accumulate, workspace, workspace_bytes, stream_);
}

}  // extern "C"
