// bpr.hip - R7 fused BPR + prune loss (forward and backward) and R11 the on-device sampler.
// Replaces the three gathers, mul/sum, logsigmoid, the D2H argsort of prune_loss and autograd's
// index_put backward (reference main.py:158-165,232-254,330-342), and Data.sample
// (utility/load_data.py:157-195). One launch, no host synchronisation: the reference pays a
// device->host->device round trip per bpr_loss call (8 per step, main.py:159).
#include "common.h"

namespace llmrec {

constexpr int BPR_THREADS = 1024;
constexpr int BPR_GROUPS = BPR_THREADS / 16;

__device__ __forceinline__ float logsigmoid_f(float x) {
    // min(x, 0) - log1p(exp(-|x|)), the form aten::log_sigmoid_forward uses
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// saved layout: [0, B_max) per-sample d(mf)/d(s_b); [B_max + 0..2] = Su, Sp, Sq; [B_max + 3] = k
__global__ __launch_bounds__(BPR_THREADS) void bpr_fwd_kernel(const float* __restrict__ Eu, int64_t ldu,
                                                              const float* __restrict__ Ei, int64_t ldi, int d,
                                                              const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                              const int64_t* __restrict__ neg, int B_max,
                                                              const int32_t* __restrict__ n_valid_dev, double remember_rate,
                                                              float decay, float bsz, float* __restrict__ out2,
                                                              float* __restrict__ saved, const float* __restrict__ global_m,
                                                              int global_B, int my_offset, int scores_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* m_s = reinterpret_cast<float*>(smem);                       // [B_max] logsigmoid values
    float* sg_s = m_s + B_max;                                         // [B_max] sigmoid(-(s + 1e-8))
    float* red = sg_s + B_max;                                         // [BPR_THREADS]
    float* nrm = red + BPR_THREADS;                                    // [BPR_GROUPS * 3]
    int B = n_valid_dev ? n_valid_dev[0] : B_max;
    if (B > B_max) B = B_max;
    if (B < 0) B = 0;
    const int gl = threadIdx.x & 15, grp = threadIdx.x >> 4;

    // phase 1: scores, log-sigmoids, squared norms
    float su = 0.f, sp = 0.f, sq = 0.f;
    for (int b = grp; b < B; b += BPR_GROUPS) {
        const float* u = Eu + users[b] * ldu;
        const float* p = Ei + pos[b] * ldi;
        const float* q = Ei + neg[b] * ldi;
        float dp = 0.f, dn = 0.f, nu = 0.f, np_ = 0.f, nq = 0.f;
        for (int c = gl; c < d; c += 16) {
            const float uu = u[c], pp = p[c], qq = q[c];
            dp = fmaf(uu, pp, dp); dn = fmaf(uu, qq, dn);
            nu = fmaf(uu, uu, nu); np_ = fmaf(pp, pp, np_); nq = fmaf(qq, qq, nq);
        }
        dp = group_sum<16>(dp); dn = group_sum<16>(dn);
        nu = group_sum<16>(nu); np_ = group_sum<16>(np_); nq = group_sum<16>(nq);
        su += nu; sp += np_; sq += nq;
        if (gl == 0) {
            const float x = (dp - dn) + 1e-8f;
            m_s[b] = logsigmoid_f(x);
            sg_s[b] = 1.0f / (1.0f + expf(x));                        // sigmoid(-x) = d logsigmoid / dx
        }
    }
    if (gl == 0) { nrm[grp * 3 + 0] = su; nrm[grp * 3 + 1] = sp; nrm[grp * 3 + 2] = sq; }
    __syncthreads();

    if (scores_only) {
        // sharded batch, pass 1: publish the local log-sigmoids and squared norms; the caller
        // all-gathers them and calls again with global_m (pass 2)
        for (int b = threadIdx.x; b < B_max; b += BPR_THREADS) saved[b] = b < B ? m_s[b] : INFINITY;
        if (threadIdx.x == 0) {
            float Su = 0.f, Sp = 0.f, Sq = 0.f;
            for (int g = 0; g < BPR_GROUPS; ++g) { Su += nrm[g * 3]; Sp += nrm[g * 3 + 1]; Sq += nrm[g * 3 + 2]; }
            saved[B_max + 0] = Su; saved[B_max + 1] = Sp; saved[B_max + 2] = Sq; saved[B_max + 3] = 0.f;
        }
        return;
    }

    // phase 2: keep the k smallest m_b (ties: lower global index first) by rank counting; with a
    // sharded batch the ranking runs against the all-gathered m of every rank
    const int Bg = global_m ? global_B : B;
    const float* mg = global_m ? global_m : m_s;
    const int k = (int)(remember_rate * (double)Bg);                   // int((1 - drop) * len) of main.py:161-162
    float part = 0.f;
    for (int b = threadIdx.x; b < B; b += BPR_THREADS) {
        const float mb = m_s[b];
        bool keep = true;
        if (k < Bg) {
            int rank = 0;
            const int me = my_offset + b;
            for (int j = 0; j < Bg; ++j) {
                const float mj = mg[j];
                rank += (mj < mb) || (mj == mb && j < me);
            }
            keep = rank < k;
        }
        if (keep) part += mb;
        saved[b] = keep ? (-1.0f / (float)k) * sg_s[b] : 0.f;
    }
    for (int b = B + threadIdx.x; b < B_max; b += BPR_THREADS) saved[b] = 0.f;
    red[threadIdx.x] = part;
    __syncthreads();
    for (int off = BPR_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float Su = 0.f, Sp = 0.f, Sq = 0.f;
        for (int g = 0; g < BPR_GROUPS; ++g) { Su += nrm[g * 3]; Sp += nrm[g * 3 + 1]; Sq += nrm[g * 3 + 2]; }
        out2[0] = -(red[0] / (float)k);                                // k == 0 -> nan, as torch's empty mean
        const float reg = 1.0f / (2.0f * Su + 1e-8f) + 1.0f / (2.0f * Sp + 1e-8f) + 1.0f / (2.0f * Sq + 1e-8f);
        out2[1] = decay * (reg / bsz);
        saved[B_max + 0] = Su; saved[B_max + 1] = Sp; saved[B_max + 2] = Sq; saved[B_max + 3] = (float)k;
    }
}

__global__ __launch_bounds__(256) void bpr_bwd_kernel(const float* __restrict__ Eu, int64_t ldu,
                                                      const float* __restrict__ Ei, int64_t ldi, int d,
                                                      const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                      const int64_t* __restrict__ neg, int B_max,
                                                      const int32_t* __restrict__ n_valid_dev, float decay, float bsz,
                                                      const float* __restrict__ saved, const float* __restrict__ grads2,
                                                      float* __restrict__ dEu, int64_t lddu, float* __restrict__ dEi, int64_t lddi) {
    int B = n_valid_dev ? n_valid_dev[0] : B_max;
    if (B > B_max) B = B_max;
    const float g_mf = grads2[0], g_emb = grads2[1];
    const float Su = saved[B_max], Sp = saved[B_max + 1], Sq = saved[B_max + 2];
    // d emb / d X = decay / bsz * (-1 / (2 S + 1e-8)^2) * 4 X
    const float base = -4.0f * decay / bsz * g_emb;
    const float du_ = 2.0f * Su + 1e-8f, dp_ = 2.0f * Sp + 1e-8f, dq_ = 2.0f * Sq + 1e-8f;
    const float cu = base / (du_ * du_), cp = base / (dp_ * dp_), cq = base / (dq_ * dq_);
    const int gl = threadIdx.x & 15;
    const int groups = gridDim.x * (blockDim.x >> 4);
    for (int b = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); b < B; b += groups) {
        const int64_t ui = users[b], pi = pos[b], qi = neg[b];
        const float ds = g_mf * saved[b];
        const float* u = Eu + ui * ldu;
        const float* p = Ei + pi * ldi;
        const float* q = Ei + qi * ldi;
        for (int c = gl; c < d; c += 16) {
            const float uu = u[c], pp = p[c], qq = q[c];
            atomicAdd(dEu + ui * lddu + c, fmaf(ds, pp - qq, cu * uu));
            atomicAdd(dEi + pi * lddi + c, fmaf(ds, uu, cp * pp));
            atomicAdd(dEi + qi * lddi + c, fmaf(-ds, uu, cq * qq));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11)
// ---------------------------------------------------------------------------------------------
struct Philox {
    uint32_t key[2];
    __device__ Philox(uint64_t seed) { key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32); }
    __device__ void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) const {
        uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

__device__ __forceinline__ uint32_t bounded(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

// Keyed bijection of [0, n): 4-round Feistel on 2h bits (2^(2h) >= n) with cycle walking.
__device__ uint64_t keyed_perm(uint64_t x, uint64_t n, int half_bits, const Philox& ph, uint32_t step_lo, uint32_t step_hi) {
    const uint64_t mask = (1ull << half_bits) - 1;
    do {
        uint64_t L = x >> half_bits, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t o[4];
            ph((uint32_t)R, (uint32_t)(R >> 32) ^ (0xA5A50000u + r), step_lo, step_hi, o);
            const uint64_t f = (((uint64_t)o[1] << 32) | o[0]) & mask;
            const uint64_t nL = R, nR = L ^ f;
            L = nL; R = nR;
        }
        x = (L << half_bits) | R;
    } while (x >= n);
    return x;
}

__global__ void sample_bpr_kernel(uint64_t seed, uint64_t step, int64_t n_exist, const int64_t* __restrict__ exist_users,
                                  int64_t n_items, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                  int B, int half_bits, int64_t* __restrict__ users, int64_t* __restrict__ pos,
                                  int64_t* __restrict__ neg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Philox ph(seed);
    const uint32_t slo = (uint32_t)step, shi = (uint32_t)(step >> 32);
    // users: without replacement while B <= n_exist (rd.sample), with replacement otherwise (rd.choice)
    uint64_t slot;
    if ((int64_t)B <= n_exist) {
        slot = keyed_perm((uint64_t)b, (uint64_t)n_exist, half_bits, ph, slo, shi);
    } else {
        uint32_t o[4];
        ph((uint32_t)b, 0x55AA0001u, slo, shi, o);
        slot = ((((uint64_t)o[1] << 32) | o[0]) % (uint64_t)n_exist);
    }
    const int64_t u = exist_users[slot];
    const int32_t s = rowptr[u], e = rowptr[u + 1];
    uint32_t o[4];
    ph((uint32_t)b, 0x55AA0002u, slo, shi, o);
    const int64_t p = colidx[s + (int32_t)bounded(o[0], (uint32_t)(e - s))];
    int64_t q = 0;
    uint32_t ctr = 0;
    int have = 4;
    for (int tries = 0; tries < 4096; ++tries) {
        if (have == 4) { ph((uint32_t)b, 0x55AA0003u + ctr, slo, shi, o); ++ctr; have = 0; }
        const uint32_t r = o[have++];
        q = n_items <= 0xffffffffll ? (int64_t)bounded(r, (uint32_t)n_items) : (int64_t)(r % (uint64_t)n_items);
        int32_t lo = s, hi = e;                                         // binary search in the sorted row
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (colidx[mid] < q) lo = mid + 1; else hi = mid;
        }
        if (!(lo < e && colidx[lo] == q)) break;                        // not a train item: accept
    }
    users[b] = u; pos[b] = p; neg[b] = q;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

int llmrec_bpr_prune_fwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev,
                             double remember_rate, float decay, float batch_size_flag,
                             float* out2, float* saved, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(B_max >= 0 && d > 0 && out2 && saved, "bpr_fwd: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_fwd: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max == 0 || (Eu && Ei && users && pos && neg && ldu >= d && ldi >= d), "bpr_fwd: null pointer or ld < d");
    const size_t shmem = sizeof(float) * ((size_t)2 * B_max + BPR_THREADS + BPR_GROUPS * 3);
    bpr_fwd_kernel<<<1, BPR_THREADS, shmem, stream>>>(Eu, ldu, Ei, ldi, d, users, pos, neg, B_max, n_valid_dev,
                                                     remember_rate, decay, batch_size_flag, out2, saved, nullptr, 0, 0, 0);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_prune_fwd_sharded_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_local, double remember_rate, float decay, float batch_size_flag,
                                     const float* global_m, int32_t global_B, int32_t my_offset, int32_t scores_only,
                                     float* out2, float* saved, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(B_local >= 0 && d > 0 && out2 && saved, "bpr_fwd_sharded: bad argument");
    if (B_local > LLMREC_BPR_MAX_B) { set_error("bpr_fwd_sharded: B_local %d > %d", B_local, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_local == 0 || (Eu && Ei && users && pos && neg && ldu >= d && ldi >= d), "bpr_fwd_sharded: null pointer or ld < d");
    LLMREC_CHECK_ARG(scores_only || (global_m && global_B >= B_local && my_offset >= 0 && my_offset + B_local <= global_B),
                     "bpr_fwd_sharded: pass 2 needs the gathered scores and a valid offset");
    const size_t shmem = sizeof(float) * ((size_t)2 * B_local + BPR_THREADS + BPR_GROUPS * 3);
    bpr_fwd_kernel<<<1, BPR_THREADS, shmem, stream>>>(Eu, ldu, Ei, ldi, d, users, pos, neg, B_local, nullptr,
                                                     remember_rate, decay, batch_size_flag, out2, saved,
                                                     scores_only ? nullptr : global_m, global_B, my_offset, scores_only);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_prune_bwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, float decay, float batch_size_flag,
                             const float* saved, const float* grads2,
                             float* dEu, int64_t lddu, float* dEi, int64_t lddi, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(B_max >= 0 && d > 0 && saved && grads2, "bpr_bwd: bad argument");
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(Eu && Ei && users && pos && neg && dEu && dEi && ldu >= d && ldi >= d && lddu >= d && lddi >= d,
                     "bpr_bwd: null pointer or ld < d");
    bpr_bwd_kernel<<<grid_for(B_max, 16), 256, 0, stream>>>(Eu, ldu, Ei, ldi, d, users, pos, neg, B_max, n_valid_dev, decay,
                                                            batch_size_flag, saved, grads2, dEu, lddu, dEi, lddi);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_sample_bpr(uint64_t seed, uint64_t step, int64_t n_exist_users, const int64_t* exist_users,
                      int64_t n_items, const int32_t* train_rowptr, const int32_t* train_colidx,
                      int32_t B, int64_t* users, int64_t* pos, int64_t* neg, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(B >= 0 && n_exist_users > 0 && n_items > 0, "sample_bpr: bad sizes");
    if (B == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(exist_users && train_rowptr && train_colidx && users && pos && neg, "sample_bpr: null pointer");
    int half_bits = 1;
    while ((1ull << (2 * half_bits)) < (uint64_t)n_exist_users) ++half_bits;
    sample_bpr_kernel<<<(B + 255) / 256, 256, 0, (hipStream_t)stream_>>>(seed, step, n_exist_users, exist_users, n_items,
                                                                          train_rowptr, train_colidx, B, half_bits, users, pos, neg);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // extern "C"
