// topk.hip - R9/R10: full-rank user x item scoring with train-item masking and per-user top-K.
// Replaces torch.matmul(E_u[blk], E_i^T), the D2H copy of each 2048 x I score block and the
// per-user Python set-difference + heapq.nlargest (reference utility/batch_test.py:21-36,83-109,
// 149-157). The U x I score matrix is never written to memory.
//
// The only MFMA user on the path: v_mfma_f32_16x16x4_f32 (exact fp32; a k-ordered fma chain, so
// the scores are reproducible bit for bit by a scalar fmaf loop in the order documented below).
//
// Geometry: block = 4 wavefronts sharing 16 query users (one MFMA row tile, A operand resident in
// registers for the whole launch); wavefront w sweeps the w-th quarter of the items in tiles of 64
// (4 MFMA column tiles) with its own candidate lists, and a bitonic merge of the four lists per
// user ends the block (825 blocks x 4 waves at the Netflix shape instead of 207 x 4). Operands are fed straight from global memory as float4 along k (see
// dense.hip for the k-permutation argument): lane l holds E[row (l&15)][16 c + 4 (l>>4) + s].
// Chain order of k for one score: for c in 0..d/16-1, for s in 0..3, for q in 0..3: k = 16c + 4q + s.
//
// Selection: each wavefront keeps, per user, a sorted 64-slot list (score desc, item id asc) in
// LDS, one slot per lane (a lane only ever touches its own slot column, so no LDS hand-off
// between lanes exists). A score enters the insert path only if it is >= the user's current K-th
// score; all candidates of one user in a tile are inserted with the list held in registers: a
// ballot/popcount position search plus a one-lane DPP shift (v_mov_b32 wave_shr:1) per candidate.
// Train items are removed with a per-user cursor into the user's (ascending) CSR row, turned
// into a 64-bit tile mask.
#include "common.h"
#include <limits.h>

namespace llmrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4g(const float* row, int k, int K, bool vec_ok) {
    if (vec_ok && k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) r.x = row[k];
    if (k + 1 < K) r.y = row[k + 1];
    if (k + 2 < K) r.z = row[k + 2];
    if (k + 3 < K) r.w = row[k + 3];
    return r;
}
__device__ __forceinline__ float cmp4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

struct TopkArgs {
    int n_query;
    const int64_t* query_users;
    const float* Eu; int64_t ldu;
    const float* Ei; int64_t ldi;
    int64_t n_items; int d;
    const int32_t* train_rowptr; const int32_t* train_colidx;
    int K;
    int32_t* out_idx; float* out_score;
    float* S; int64_t lds;
    int vec_ok;
};

template <int DK>
__device__ __forceinline__ void load_items(const TopkArgs& a, int64_t base, int li, int lq, float4 (&b)[4][DK]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        int64_t item = base + 16 * n + li;
        if (item > a.n_items - 1) item = a.n_items - 1;
        const float* row = a.Ei + item * a.ldi;
#pragma unroll
        for (int c = 0; c < DK; ++c) b[n][c] = ld4g(row, 16 * c + 4 * lq, a.d, a.vec_ok);
    }
}

__device__ __forceinline__ float wave_shr1(float x) {     // lane i <- lane i-1 (lane 0 keeps its value)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int wave_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ bool pair_better(float s, int i, float t, int j) { return (s > t) || (s == t && i < j); }

// top-64 of the union of two descending 64-lists held one entry per lane: elementwise best of A and
// reversed B is bitonic and holds the 64 best; six compare-exchange stages sort it (best first).
__device__ __forceinline__ void merge64(float& s, int& i, float bs, int bi, int lane) {
    const float rs = __shfl(bs, 63 - lane, 64);
    const int ri = __shfl(bi, 63 - lane, 64);
    if (pair_better(rs, ri, s, i)) { s = rs; i = ri; }
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
        const float os = __shfl_xor(s, stride, 64);
        const int oi = __shfl_xor(i, stride, 64);
        const bool take_best = (lane & stride) == 0;
        const bool other_better = pair_better(os, oi, s, i);
        if (take_best == other_better) { s = os; i = oi; }
    }
}

// block = 4 wavefronts sharing 16 query users; wave w sweeps the w-th quarter of the item tiles
// and keeps its own lists; a bitonic merge of the four lists per user ends the block.
template <int DK, bool SELECT>
__global__ __launch_bounds__(256) void score_topk_kernel(TopkArgs a) {
    __shared__ float list_s[4][16][64];
    __shared__ int32_t list_i[4][16][64];
    // each wave publishes its current K-th score per user; the global K-th score is >= every wave's
    // local one, so max over the four is a valid (and much tighter) filter for all of them. Reads may
    // be stale - that only lets a few more candidates through.
    __shared__ float thr_pub[4][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int q0 = blockIdx.x * 16;
    if (q0 >= a.n_query) return;                               // block-uniform
    if (threadIdx.x < 64) thr_pub[threadIdx.x >> 4][threadIdx.x & 15] = -INFINITY;
    __syncthreads();

    // A operand: the block's 16 users
    int qa = q0 + li;
    if (qa > a.n_query - 1) qa = a.n_query - 1;
    const int64_t user_a = a.query_users[qa];
    float4 ua[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c) ua[c] = ld4g(a.Eu + user_a * a.ldu, 16 * c + 4 * lq, a.d, a.vec_ok);

    const int64_t tiles_total = (a.n_items + 63) / 64;
    const int64_t tiles_per_wave = (tiles_total + 3) / 4;
    const int64_t t_begin = w * tiles_per_wave;
    const int64_t t_end = t_begin + tiles_per_wave < tiles_total ? t_begin + tiles_per_wave : tiles_total;

    // lanes 0..15 own one user each: train-row cursor and current K-th score of THIS wave's list
    int32_t cur = 0, end = 0;
    float thr = -INFINITY;
    if (SELECT) {
        if (lane < 16 && a.train_rowptr) { cur = a.train_rowptr[user_a]; end = a.train_rowptr[user_a + 1]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { list_s[w][r][lane] = -INFINITY; list_i[w][r][lane] = INT_MAX; }
    }

    for (int64_t t = t_begin; t < t_end; ++t) {
        const int64_t base = t * 64;
        float4 b[4][DK];
        load_items<DK>(a, base, li, lq, b);
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DK; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(cmp4(ua[c], s), cmp4(b[n][c], s), acc[n], 0, 0, 0);

        if (!SELECT) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + lq * 4 + r;
                    const int64_t item = base + 16 * n + li;
                    if (q < a.n_query && item < a.n_items) a.S[(int64_t)q * a.lds + item] = acc[n][r];
                }
            continue;
        }
        // 64-bit mask of this tile's train items, built by the row-owner lanes
        uint32_t mlo = 0, mhi = 0;
        if (lane < 16) {
            while (cur < end) {
                const int64_t c = a.train_colidx[cur];
                if (c >= base + 64) break;
                const int bit = (int)(c - base);
                if (bit >= 0) { if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32); }
                ++cur;                                             // entries before this wave's quarter are skipped
            }
        }
        uint32_t rm_lo[4], rm_hi[4];
        float rthr[4];
        float tshared = thr;
        if (lane < 16) {
            const volatile float* tp = &thr_pub[0][0];
            tshared = fmaxf(fmaxf(tp[lane], tp[16 + lane]), fmaxf(tp[32 + lane], tp[48 + lane]));
            tshared = fmaxf(tshared, thr);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rm_lo[r] = __shfl(mlo, lq * 4 + r, 64);
            rm_hi[r] = __shfl(mhi, lq * 4 + r, 64);
            rthr[r] = __shfl(tshared, lq * 4 + r, 64);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[n][r];
                const int col = 16 * n + li;
                const uint32_t mword = (n < 2) ? rm_lo[r] : rm_hi[r];
                const bool masked = (mword >> (col & 31)) & 1u;
                const bool valid = (base + col < a.n_items) && (q0 + lq * 4 + r < a.n_query) && !masked;
                const unsigned long long bal = __ballot(valid && v >= rthr[r]);
                if (bal == 0) continue;
                // candidates of lane group g all belong to user row g*4 + r: insert them with the
                // row's list held in registers (one LDS round trip per row, not per candidate)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned sub = (unsigned)((bal >> (16 * g)) & 0xffffull);
                    if (sub == 0) continue;
                    const int crow = g * 4 + r;
                    float ls = list_s[w][crow][lane];
                    int32_t lid = list_i[w][crow][lane];
                    do {
                        const int bpos = __ffs((int)sub) - 1;
                        sub &= sub - 1;
                        const float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * g + bpos));
                        const int32_t citem = (int32_t)(base + 16 * n + bpos);
                        const int pos = __popcll(__ballot(pair_better(ls, lid, cv, citem)));
                        if (pos < a.K) {
                            const float ps = wave_shr1(ls);
                            const int32_t pi = wave_shr1(lid);
                            ls = lane < pos ? ls : (lane == pos ? cv : ps);
                            lid = lane < pos ? lid : (lane == pos ? citem : pi);
                        }
                    } while (sub);
                    list_s[w][crow][lane] = ls;
                    list_i[w][crow][lane] = lid;
                    const float nthr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ls), a.K - 1));
                    if (lane == crow) { thr = nthr; *(volatile float*)&thr_pub[w][crow] = nthr; }
                }
            }
        }
    }
    if (SELECT) {
        __syncthreads();
        // wave w merges the four quarter lists of users 4w .. 4w+3
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * w + rr;
            const int q = q0 + r;
            if (q >= a.n_query) break;                             // wave-uniform
            float s0 = list_s[0][r][lane]; int i0 = list_i[0][r][lane];
            float s2 = list_s[2][r][lane]; int i2 = list_i[2][r][lane];
            merge64(s0, i0, list_s[1][r][lane], list_i[1][r][lane], lane);
            merge64(s2, i2, list_s[3][r][lane], list_i[3][r][lane], lane);
            merge64(s0, i0, s2, i2, lane);
            if (lane < a.K) {
                a.out_idx[(int64_t)q * a.K + lane] = i0 == INT_MAX ? -1 : i0;
                a.out_score[(int64_t)q * a.K + lane] = s0;
            }
        }
    }
}

__global__ void topk_hits_kernel(int n_query, const int64_t* __restrict__ query_users, int K,
                                 const int32_t* __restrict__ topk_idx, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ colidx, uint8_t* __restrict__ hits) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_query * K) return;
    const int q = (int)(t / K);
    const int32_t item = topk_idx[t];
    const int64_t u = query_users[q];
    int32_t lo = rowptr[u], hi = rowptr[u + 1];
    const int32_t e = hi;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (colidx[mid] < item) lo = mid + 1; else hi = mid;
    }
    hits[t] = (item >= 0 && lo < e && colidx[lo] == item) ? 1 : 0;
}

template <bool SELECT>
static int launch_topk(const TopkArgs& a, hipStream_t stream) {
    const int DK = (a.d + 15) / 16;
    const int grid = (int)ceil_div(a.n_query, 16);
    switch (DK) {
        case 1: score_topk_kernel<1, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 2: score_topk_kernel<2, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 3: score_topk_kernel<3, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 4: score_topk_kernel<4, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 5: score_topk_kernel<5, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 6: score_topk_kernel<6, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 7: score_topk_kernel<7, SELECT><<<grid, 256, 0, stream>>>(a); break;
        case 8: score_topk_kernel<8, SELECT><<<grid, 256, 0, stream>>>(a); break;
        default: set_error("score_topk: d = %d > 128", a.d); return LLMREC_EUNSUPPORTED;
    }
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

int llmrec_score_topk_f32(int32_t n_query, const int64_t* query_users,
                          const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                          int64_t n_items, int32_t d,
                          const int32_t* train_rowptr, const int32_t* train_colidx,
                          int32_t K, int32_t* out_idx, float* out_score, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && n_items > 0 && d > 0 && K > 0 && K <= LLMREC_TOPK_MAX, "score_topk: bad sizes (K <= %d)", LLMREC_TOPK_MAX);
    if (n_query == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(query_users && Eu && Ei && out_idx && out_score && ldu >= d && ldi >= d, "score_topk: null pointer or ld < d");
    LLMREC_CHECK_ARG((train_rowptr == nullptr) == (train_colidx == nullptr) || train_rowptr, "score_topk: train CSR incomplete");
    LLMREC_CHECK_ARG(n_items < (1ll << 31), "score_topk: n_items exceeds int32 item ids");
    TopkArgs a;
    a.n_query = n_query; a.query_users = query_users; a.Eu = Eu; a.ldu = ldu; a.Ei = Ei; a.ldi = ldi;
    a.n_items = n_items; a.d = d; a.train_rowptr = train_rowptr; a.train_colidx = train_colidx; a.K = K;
    a.out_idx = out_idx; a.out_score = out_score; a.S = nullptr; a.lds = 0;
    a.vec_ok = (ldu % 4 == 0) && (ldi % 4 == 0) && (((uintptr_t)Eu | (uintptr_t)Ei) % 16 == 0);
    return launch_topk<true>(a, (hipStream_t)stream_);
}

int llmrec_scores_f32(int32_t n_query, const int64_t* query_users,
                      const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                      int64_t n_items, int32_t d, float* S, int64_t lds, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && n_items > 0 && d > 0, "scores: bad sizes");
    if (n_query == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(query_users && Eu && Ei && S && ldu >= d && ldi >= d && lds >= n_items, "scores: null pointer or ld too small");
    TopkArgs a;
    a.n_query = n_query; a.query_users = query_users; a.Eu = Eu; a.ldu = ldu; a.Ei = Ei; a.ldi = ldi;
    a.n_items = n_items; a.d = d; a.train_rowptr = nullptr; a.train_colidx = nullptr; a.K = 1;
    a.out_idx = nullptr; a.out_score = nullptr; a.S = S; a.lds = lds;
    a.vec_ok = (ldu % 4 == 0) && (ldi % 4 == 0) && (((uintptr_t)Eu | (uintptr_t)Ei) % 16 == 0);
    return launch_topk<false>(a, (hipStream_t)stream_);
}

int llmrec_topk_hits(int32_t n_query, const int64_t* query_users, int32_t K, const int32_t* topk_idx,
                     const int32_t* test_rowptr, const int32_t* test_colidx, uint8_t* hits, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && K > 0, "topk_hits: bad sizes");
    if (n_query == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(query_users && topk_idx && test_rowptr && test_colidx && hits, "topk_hits: null pointer");
    const int64_t n = (int64_t)n_query * K;
    topk_hits_kernel<<<(int)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(n_query, query_users, K, topk_idx, test_rowptr, test_colidx, hits);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // extern "C"
