// bpr.hip - R7 fused BPR + prune loss (forward and backward) and R11 the on-device sampler.
// Replaces the three gathers, mul/sum, logsigmoid, the D2H argsort of prune_loss and autograd's
// index_put backward (reference main.py:158-165,232-254,330-342), and Data.sample
// (utility/load_data.py:157-195). One launch, no host synchronisation: the reference pays a
// device->host->device round trip per bpr_loss call (8 per step, main.py:159).
#include "common.h"

namespace llmrec {

constexpr int BPR_THREADS = 1024;

__device__ __forceinline__ float logsigmoid_f(float x) {
    // min(x, 0) - log1p(exp(-|x|)), the form aten::log_sigmoid_forward uses
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------------------------------------
// Forward = three launches over up to LLMREC_BPR_MAX_PROBLEMS (user table, item table) pairs that
// share one (users, pos, neg) batch - the reference evaluates 8 such losses per step
// (main.py:232-254):
//   bpr_scores_kernel  one 16-lane group per (problem, sample): the three row gathers, both dot
//                      products, log-sigmoid, its derivative and the three squared row norms;
//   bpr_rank_kernel    one thread per (problem, sample): rank counting against the B log-sigmoids
//                      staged in LDS (keep the k smallest, ties by lower index);
//   bpr_reduce_kernel  one block per problem: the kept mean and the norm sums by fixed-order
//                      trees (deterministic).
// saved layout per problem (LLMREC_BPR_SAVED_FLOATS(B) floats):
//   [0, B) d(mf)/d(s_b) | [B..B+2] Su, Sp, Sq | [B+3] k | [B+4 + {0..4} * B + b] m, sg, nu, np, nq
// ---------------------------------------------------------------------------------------------
struct BprTables {
    const float* Eu[LLMREC_BPR_MAX_PROBLEMS];
    const float* Ei[LLMREC_BPR_MAX_PROBLEMS];
    int64_t ldu[LLMREC_BPR_MAX_PROBLEMS];
    int64_t ldi[LLMREC_BPR_MAX_PROBLEMS];
    float* dEu[LLMREC_BPR_MAX_PROBLEMS];
    float* dEi[LLMREC_BPR_MAX_PROBLEMS];
    int64_t lddu[LLMREC_BPR_MAX_PROBLEMS];
    int64_t lddi[LLMREC_BPR_MAX_PROBLEMS];
    float g_mf[LLMREC_BPR_MAX_PROBLEMS];
    float g_emb[LLMREC_BPR_MAX_PROBLEMS];
};

__device__ __forceinline__ int bpr_batch(const int32_t* n_valid_dev, int B_max) {
    int B = n_valid_dev ? n_valid_dev[0] : B_max;
    return B > B_max ? B_max : (B < 0 ? 0 : B);
}

// what "a new step begins" means on the device (one thread of the scores launch): a new row stamp and, when given, AdamW's step
// counter / bias corrections (adamw_advance_kernel's arithmetic, rowops.hip)
struct StepBegin {
    int32_t* row_stamp;
    float* adamw_state;
    float lr, b1, b2;
};

__global__ __launch_bounds__(256) void bpr_scores_kernel(BprTables t, int d, const int64_t* __restrict__ users,
                                                         const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                         int B_max, const int32_t* __restrict__ n_valid_dev,
                                                         float* __restrict__ saved_all, int saved_stride, StepBegin sb) {
    const int B = bpr_batch(n_valid_dev, B_max);
    const int prob = blockIdx.y;
    const int gl = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (sb.row_stamp) sb.row_stamp[0] = (int32_t)((uint32_t)sb.row_stamp[0] + 1u);   // a new step: new stamp
        if (sb.adamw_state) {
            const int tt = __float_as_int(sb.adamw_state[0]) + 1;
            sb.adamw_state[0] = __int_as_float(tt);
            const double bc1 = 1.0 - pow((double)sb.b1, (double)tt);
            const double bc2 = 1.0 - pow((double)sb.b2, (double)tt);
            sb.adamw_state[1] = (float)((double)sb.lr / bc1);
            sb.adamw_state[2] = (float)sqrt(bc2);
        }
    }
    if (b >= B) return;
    float* sc = saved_all + (int64_t)prob * saved_stride + B_max + 4;
    const float* u = t.Eu[prob] + users[b] * t.ldu[prob];
    const float* p = t.Ei[prob] + pos[b] * t.ldi[prob];
    const float* q = t.Ei[prob] + neg[b] * t.ldi[prob];
    float dp = 0.f, dn = 0.f, nu = 0.f, np_ = 0.f, nq = 0.f;
    for (int c = gl; c < d; c += 16) {
        const float uu = u[c], pp = p[c], qq = q[c];
        dp = fmaf(uu, pp, dp); dn = fmaf(uu, qq, dn);
        nu = fmaf(uu, uu, nu); np_ = fmaf(pp, pp, np_); nq = fmaf(qq, qq, nq);
    }
    dp = group_sum<16>(dp); dn = group_sum<16>(dn);
    nu = group_sum<16>(nu); np_ = group_sum<16>(np_); nq = group_sum<16>(nq);
    if (gl == 0) {
        const float x = (dp - dn) + 1e-8f;
        sc[b] = logsigmoid_f(x);
        sc[B_max + b] = 1.0f / (1.0f + expf(x));                      // sigmoid(-x) = d logsigmoid / dx
        sc[2 * B_max + b] = nu; sc[3 * B_max + b] = np_; sc[4 * B_max + b] = nq;
    }
}

__device__ __forceinline__ float block_tree_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = BPR_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// Batch-sharded selection (data-parallel or user-sharded ranks): every rank all-gathers one block of
// LLMREC_BPR_GATHER_FLOATS(P, cap) floats, laid out  [P][cap] m_b | [P][4] Su, Sp, Sq, k | n_valid (as float);
// the global batch is the concatenation of the ranks' valid samples in rank order.
struct BprGather {
    const float* g;        // null: the batch is local
    int n_ranks;           // blocks in g
    int64_t stride;        // floats between two ranks' blocks
    int cap;               // sample capacity per rank (= that rank's B_max)
    int n_prob;            // problems per block (P)
    int my_offset;         // global padded index of this rank's sample 0 (= my_rank * cap)
    int has_tail;          // 0: g is a plain [n_ranks * cap] array of m_b, all valid (single-problem API)
};

__device__ __forceinline__ int gather_valid(const BprGather& ga, int r) {
    if (!ga.has_tail) return ga.cap;
    const int nv = (int)ga.g[r * ga.stride + (int64_t)ga.n_prob * ga.cap + 4 * ga.n_prob];
    return nv > ga.cap ? ga.cap : (nv < 0 ? 0 : nv);
}

__device__ __forceinline__ int gather_batch(const BprGather& ga) {
    int n = 0;
    for (int r = 0; r < ga.n_ranks; ++r) n += gather_valid(ga, r);
    return n;
}

// selection, step 1: rank counting. grid = (ceil(B / 16), problems); each 16-lane group ranks one sample
// against the Bg log-sigmoids staged in LDS and writes d(mf)/d(s_b) (0 when dropped) plus the kept
// m_b into the scratch (slot 1 = sg is consumed here and overwritten with the kept value).
// Padding slots of the gathered layout are staged as +inf: they never precede a real sample.
__global__ __launch_bounds__(256) void bpr_rank_kernel(int B_max, const int32_t* __restrict__ n_valid_dev, double remember_rate,
                                                       float* __restrict__ saved_all, int saved_stride, BprGather ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* m_s = reinterpret_cast<float*>(smem);
    const int B = bpr_batch(n_valid_dev, B_max);
    const int prob = blockIdx.y;
    float* saved = saved_all + (int64_t)prob * saved_stride;
    float* sc = saved + B_max + 4;
    int Bg, slots;
    if (ga.g) {
        slots = ga.n_ranks * ga.cap;
        Bg = gather_batch(ga);
        for (int r = 0; r < ga.n_ranks; ++r) {
            const int nv = gather_valid(ga, r);
            const float* src = ga.g + r * ga.stride + (int64_t)prob * ga.cap;
            for (int j = threadIdx.x; j < ga.cap; j += 256) m_s[r * ga.cap + j] = j < nv ? src[j] : __builtin_inff();
        }
    } else {
        slots = Bg = B;
        for (int j = threadIdx.x; j < B; j += 256) m_s[j] = sc[j];
    }
    __syncthreads();
    const int k = (int)(remember_rate * (double)Bg);                   // int((1 - drop) * len) of main.py:161-162
    // one 16-lane group per sample: the lanes split the comparison loop (integer counts: order-independent)
    const int gl = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= B_max) return;
    if (b >= B) { if (gl == 0) { saved[b] = 0.f; sc[B_max + b] = 0.f; } return; }
    const float mb = sc[b];
    bool keep = true;
    if (k < Bg) {
        int rank = 0;
        const int me = ga.my_offset + b;
        for (int j = gl; j < slots; j += 16) {
            const float mj = m_s[j];
            rank += (mj < mb) || (mj == mb && j < me);
        }
        rank = (int)group_sum<16>((float)rank);                           // < 2^24: exact
        keep = rank < k;
    }
    if (gl == 0) {
        saved[b] = keep ? (-1.0f / (float)k) * sc[B_max + b] : 0.f;
        sc[B_max + b] = keep ? mb : 0.f;
    }
}

// selection, step 2: one block per problem sums the kept log-sigmoids and the three squared norms
// with fixed-order trees (deterministic) and writes the two loss values. With a gathered layout the
// norms (and the batch size) are the sums over the ranks' blocks in rank order - identical on every
// rank - while out[0] stays this rank's share of mf.
__device__ __forceinline__ void bpr_reduce_problem(int prob, int B_max, const int32_t* __restrict__ n_valid_dev,
                                                   double remember_rate, float decay, float bsz,
                                                   float* __restrict__ out_all, float* __restrict__ saved_all,
                                                   int saved_stride, const BprGather& ga, float* red) {
    const int B = bpr_batch(n_valid_dev, B_max);
    float* saved = saved_all + (int64_t)prob * saved_stride;
    const float* sc = saved + B_max + 4;
    const int Bg = ga.g ? gather_batch(ga) : B;
    const int k = (int)(remember_rate * (double)Bg);
    float part = 0.f, su = 0.f, sp = 0.f, sq = 0.f;
    for (int b = threadIdx.x; b < B; b += BPR_THREADS) {
        part += sc[B_max + b];
        su += sc[2 * B_max + b]; sp += sc[3 * B_max + b]; sq += sc[4 * B_max + b];
    }
    const float kept = block_tree_sum(part, red);
    float Su = block_tree_sum(su, red), Sp = block_tree_sum(sp, red), Sq = block_tree_sum(sq, red);
    if (threadIdx.x == 0) {
        if (ga.g && ga.has_tail) {
            Su = Sp = Sq = 0.f;
            for (int r = 0; r < ga.n_ranks; ++r) {
                const float* nb = ga.g + r * ga.stride + (int64_t)ga.n_prob * ga.cap + 4 * prob;
                Su += nb[0]; Sp += nb[1]; Sq += nb[2];
            }
        }
        out_all[prob * 2 + 0] = -(kept / (float)k);                    // k == 0 -> nan, as torch's empty mean
        const float reg = 1.0f / (2.0f * Su + 1e-8f) + 1.0f / (2.0f * Sp + 1e-8f) + 1.0f / (2.0f * Sq + 1e-8f);
        out_all[prob * 2 + 1] = decay * (reg / bsz);
        saved[B_max + 0] = Su; saved[B_max + 1] = Sp; saved[B_max + 2] = Sq; saved[B_max + 3] = (float)k;
    }
}

__global__ __launch_bounds__(BPR_THREADS) void bpr_reduce_kernel(int B_max, const int32_t* __restrict__ n_valid_dev,
                                                                 double remember_rate, float decay, float bsz,
                                                                 float* __restrict__ out_all, float* __restrict__ saved_all,
                                                                 int saved_stride, BprGather ga) {
    __shared__ float red[BPR_THREADS];
    bpr_reduce_problem(blockIdx.x, B_max, n_valid_dev, remember_rate, decay, bsz, out_all, saved_all, saved_stride, ga, red);
}

// The logged scalars of a fused step in ONE single-block launch (llmrec_bpr_multi_losses_assemble_f32): the loss values of every
// problem, the feature regulariser from the fusion launch's per-block partial sums (thread t adds partial[t], partial[t + 1024], ...;
// then the pairwise tree), and the assembly of llmrec_loss_assemble_f32 mode 0.
// The problems are reduced SIDE BY SIDE: 128 threads per problem emulate bpr_reduce_kernel's 1024 threads (slot v = that launch's thread
// v: the sum over b = v, v + 1024, ... in ascending order; then the same pairwise tree red[i] += red[i + off] over 1024 slots), one
// column (kept values, three squared norms) at a time - the bits of out / saved are those of bpr_reduce_kernel, in 40 block barriers
// instead of 8 x 40 (the first version ran the problems one after the other: 39 us on the ID chain's stream).
struct LossW { float w[LLMREC_BPR_MAX_PROBLEMS]; };
constexpr int LA_GROUP = BPR_THREADS / LLMREC_BPR_MAX_PROBLEMS;      // 128 threads per problem
__global__ __launch_bounds__(BPR_THREADS) void bpr_losses_assemble_kernel(int n_prob, int B_max, const int32_t* __restrict__ n_valid_dev,
                                                                          double remember_rate, float decay, float bsz,
                                                                          float* __restrict__ out_all, float* __restrict__ saved_all, int saved_stride,
                                                                          LossW w, const float* __restrict__ partial, int n_partial, float reg_coef,
                                                                          float* __restrict__ scal, double* __restrict__ running) {
    __shared__ float red[LLMREC_BPR_MAX_PROBLEMS][BPR_THREADS];         // 32 KB
    __shared__ float outs[LLMREC_BPR_MAX_PROBLEMS][2];
    const int B = bpr_batch(n_valid_dev, B_max);
    const int prob = threadIdx.x / LA_GROUP, t = threadIdx.x % LA_GROUP;
    const bool live = prob < n_prob;
    float* saved = saved_all + (int64_t)(live ? prob : 0) * saved_stride;
    const float* sc = saved + B_max + 4;
    float tot[4] = {0.f, 0.f, 0.f, 0.f};                               // kept, Su, Sp, Sq (valid in the group's thread 0)
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        const float* src = sc + (int64_t)(col + 1) * B_max;              // slots 1 (kept m_b), 2, 3, 4 (squared norms) of `saved`
        if (live) {
            for (int v = t; v < BPR_THREADS; v += LA_GROUP) {
                float s_ = 0.f;
                for (int b = v; b < B; b += BPR_THREADS) s_ += src[b];
                red[prob][v] = s_;
            }
        }
        __syncthreads();
        for (int off = BPR_THREADS / 2; off > 0; off >>= 1) {
            if (live) for (int i = t; i < off; i += LA_GROUP) red[prob][i] += red[prob][i + off];
            __syncthreads();
        }
        tot[col] = red[live ? prob : 0][0];
        __syncthreads();
    }
    if (live && t == 0) {
        const int k = (int)(remember_rate * (double)B);
        const float mf = -(tot[0] / (float)k);                           // k == 0 -> nan, as torch's empty mean
        const float reg = 1.0f / (2.0f * tot[1] + 1e-8f) + 1.0f / (2.0f * tot[2] + 1e-8f) + 1.0f / (2.0f * tot[3] + 1e-8f);
        const float emb = decay * (reg / bsz);
        out_all[prob * 2 + 0] = mf; out_all[prob * 2 + 1] = emb;
        outs[prob][0] = mf; outs[prob][1] = emb;
        saved[B_max + 0] = tot[1]; saved[B_max + 1] = tot[2]; saved[B_max + 2] = tot[3]; saved[B_max + 3] = (float)k;
    }
    float feat = scal[0];                                              // no partial sums: whatever llmrec_sumsq_f32 left there
    if (partial) {
        float s_ = 0.f;
        for (int i = threadIdx.x; i < n_partial; i += BPR_THREADS) s_ += partial[i];
        feat = reg_coef * block_tree_sum(s_, red[0]);
    }
    __syncthreads();                                                   // the groups' results (LDS) are visible to thread 0
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int p = 0; p < n_prob; ++p) s += outs[p][0] * w.w[p];
    scal[0] = feat;
    scal[2] = outs[0][0]; scal[3] = outs[0][1];
    scal[1] = s + outs[0][1] + feat;
    if (running) { running[0] += (double)scal[1]; running[1] += (double)scal[2]; running[2] += (double)scal[3]; }
}

// pass 1 of the batch-sharded forward: this rank's block of the gathered layout
__global__ __launch_bounds__(256) void bpr_pack_kernel(int n_prob, int B_max, const int32_t* __restrict__ n_valid_dev,
                                                       const float* __restrict__ saved_all, int saved_stride, float* __restrict__ block) {
    const int B = bpr_batch(n_valid_dev, B_max);
    const int total = n_prob * B_max + 4 * n_prob + 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        float v;
        if (i < n_prob * B_max) {
            const int p = i / B_max, b = i - p * B_max;
            v = b < B ? saved_all[(int64_t)p * saved_stride + B_max + 4 + b] : __builtin_inff();
        } else if (i < n_prob * B_max + 4 * n_prob) {
            const int j = i - n_prob * B_max;
            v = saved_all[(int64_t)(j >> 2) * saved_stride + B_max + (j & 3)];
        } else {
            v = (float)B;
        }
        block[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// DETERMINISTIC gradient scatter (round 6). The reference's backward is index_put(accumulate) on CPU: given the seed it adds the
// rows several samples share in one fixed order (main.py:232-254,330-342), and same-seed runs agree. Rounds 1 - 5 added the rows
// with fp32 atomicAdd, so two runs of one seed drifted apart (1.0e-3 vs 3.6e-3 in the final E_i of one trajectory on two boxes).
// Now every destination row has ONE owner that adds the row's contributions in a fixed order, without atomics:
//   bpr_plan_kernel      depends on (users, pos, neg, n_valid) only - the fused step runs it right behind the sampler, off the
//                        critical path. Keys  id << 32 | slot  (distinct), sorted by RANK COUNTING: every block stages its side's keys in
//                        LDS, each 16-lane group counts the keys below its own -> the key's sorted position (no serial sorting network:
//                        one round of LDS reads, ceil(3 B / 16) blocks):
//                          plan[0 .. B_max)          user side, slot b          (id = users[b])
//                          plan[B_max .. 3 B_max)    item side, slot b          (id = pos[b])  /  B_max + b  (id = neg[b])
//                        slots of samples b >= n_valid carry the id 0xffffffff and sort to the end. Behind the keys, as int32:
//                          runlen[3 B_max]           at a run's first position: the number of keys with that id; 0 elsewhere (and for id 0xffffffff)
//   bpr_bwd_runs_kernel  one 16-lane group per sorted position and problem; the HEAD of a run of equal ids owns the destination row
//                        and adds, in ascending problem index (problems whose target POINTER is the same - the five attribute problems
//                        of a step share d prof_u - are summed by the first of them) and ascending slot, what rounds 1 - 5
//                        added with atomics; a dropped sample (coefficient 0) of a problem without a regulariser share is skipped, its
//                        contribution being exactly zero. The (problem, member) pairs of a run are taken 16 at a time: lane l fetches
//                        pair l's slot, coefficient and row ids (ONE round of dependent loads for 16 pairs), the pairs that contribute are
//                        compacted into LDS records, and the group then streams the records' rows four at a time.
// Targets of different problems must be identical (same pointer, same leading dimension) or disjoint.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t PLAN_NO_ID = 0xffffffffu;
__global__ __launch_bounds__(256) void bpr_plan_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                       const int64_t* __restrict__ neg, int B_max,
                                                       const int32_t* __restrict__ n_valid_dev, uint64_t* __restrict__ plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* k = reinterpret_cast<uint64_t*>(smem);
    const int B = bpr_batch(n_valid_dev, B_max);
    const int nbu = (B_max + 15) / 16;
    const bool items = (int)blockIdx.x >= nbu;
    const int n = items ? 2 * B_max : B_max;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = (items && i >= B_max) ? i - B_max : i;
        uint64_t id = PLAN_NO_ID;
        if (b < B) id = (uint64_t)(items ? (i >= B_max ? neg[b] : pos[b]) : users[b]);
        k[i] = (id << 32) | (uint64_t)(uint32_t)i;
    }
    __syncthreads();
    const int gl = threadIdx.x & 15;
    const int i = ((int)blockIdx.x - (items ? nbu : 0)) * 16 + (threadIdx.x >> 4);
    if (i >= n) return;
    const uint64_t me = k[i];
    const uint32_t id = (uint32_t)(me >> 32);
    int below = 0, lower_id = 0, same_id = 0;
    for (int j = gl; j < n; j += 16) {
        const uint64_t kj = k[j];
        const uint32_t idj = (uint32_t)(kj >> 32);
        below += kj < me; lower_id += idj < id; same_id += idj == id;
    }
    below = (int)group_sum<16>((float)below); lower_id = (int)group_sum<16>((float)lower_id); same_id = (int)group_sum<16>((float)same_id);   // < 2^24: exact
    if (gl == 0) {
        const int base = items ? B_max : 0;
        plan[base + below] = me;
        int32_t* runlen = reinterpret_cast<int32_t*>(plan + 3 * (int64_t)B_max);
        runlen[base + below] = (below == lower_id && id != PLAN_NO_ID) ? same_id : 0;
    }
}

constexpr int RUN_MAX_SHARE = LLMREC_BPR_MAX_PROBLEMS;
struct RunShare {                                                      // the problems that scatter into this block's target (LDS)
    const float* Eu[RUN_MAX_SHARE]; const float* Ei[RUN_MAX_SHARE]; const float* saved[RUN_MAX_SHARE];
    int64_t ldu[RUN_MAX_SHARE], ldi[RUN_MAX_SHARE];
    float g_mf[RUN_MAX_SHARE], cu[RUN_MAX_SHARE], cp[RUN_MAX_SHARE], cq[RUN_MAX_SHARE];
    int n, any_reg;
};
struct RunRec { const float* a; const float* b; const float* self; float ds, cr; };   // user row: ds (a - b) + cr self ; item row: ds a + cr self

template <int NV, bool VEC> struct RunAcc;
template <int NV> struct RunAcc<NV, true> {                            // lane gl: float4 chunks gl, gl + 16, ... of a 64 NV-column pass
    float4 v[NV];
    __device__ __forceinline__ void load(const float* row, int c0, int d, int gl) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = c0 + 4 * (gl + 16 * i); v[i] = c < d ? *reinterpret_cast<const float4*>(row + c) : float4{0.f, 0.f, 0.f, 0.f}; }
    }
    __device__ __forceinline__ void store(float* row, int c0, int d, int gl) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = c0 + 4 * (gl + 16 * i); if (c < d) *reinterpret_cast<float4*>(row + c) = v[i]; }
    }
};
template <int NV> struct RunAcc<NV, false> {                           // lane gl: columns gl, gl + 16, ... of a 64 NV-column pass
    float v[4 * NV];
    __device__ __forceinline__ void load(const float* row, int c0, int d, int gl) {
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) { const int c = c0 + gl + 16 * i; v[i] = c < d ? row[c] : 0.f; }
    }
    __device__ __forceinline__ void store(float* row, int c0, int d, int gl) const {
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i) { const int c = c0 + gl + 16 * i; if (c < d) row[c] = v[i]; }
    }
};
template <int NV, bool VEC, bool ITEM, bool REG>
__device__ __forceinline__ void run_add(RunAcc<NV, VEC>& acc, const RunAcc<NV, VEC>& a, const RunAcc<NV, VEC>& b, const RunAcc<NV, VEC>& self,
                                        float ds, float cr) {
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 x = a.v[i], y = ITEM ? float4{0.f, 0.f, 0.f, 0.f} : b.v[i], z = REG ? self.v[i] : float4{0.f, 0.f, 0.f, 0.f};
            acc.v[i].x += fmaf(ds, ITEM ? x.x : x.x - y.x, cr * z.x); acc.v[i].y += fmaf(ds, ITEM ? x.y : x.y - y.y, cr * z.y);
            acc.v[i].z += fmaf(ds, ITEM ? x.z : x.z - y.z, cr * z.z); acc.v[i].w += fmaf(ds, ITEM ? x.w : x.w - y.w, cr * z.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4 * NV; ++i)
            acc.v[i] += fmaf(ds, ITEM ? a.v[i] : a.v[i] - b.v[i], cr * (REG ? self.v[i] : 0.f));
    }
}

// the records [0, n) of one group, four at a time (their row loads are independent: up to 12 rows in flight per group)
template <int NV, bool VEC, bool ITEM, bool REG>
__device__ __forceinline__ void run_stream(RunAcc<NV, VEC>& acc, const RunRec* __restrict__ recs, int n, int c0, int d, int gl) {
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        RunAcc<NV, VEC> a[4], b[4], z[4];
        float ds[4], cr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const RunRec r = recs[i + u];
            ds[u] = r.ds; cr[u] = r.cr;
            a[u].load(r.a, c0, d, gl);
            if (!ITEM) b[u].load(r.b, c0, d, gl);
            if (REG) z[u].load(r.self, c0, d, gl);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) run_add<NV, VEC, ITEM, REG>(acc, a[u], b[u], z[u], ds[u], cr[u]);
    }
    for (; i < n; ++i) {
        const RunRec r = recs[i];
        RunAcc<NV, VEC> a, b, z;
        a.load(r.a, c0, d, gl);
        if (!ITEM) b.load(r.b, c0, d, gl);
        if (REG) z.load(r.self, c0, d, gl);
        run_add<NV, VEC, ITEM, REG>(acc, a, b, z, r.ds, r.cr);
    }
}

// grid = (ceil(B_max / 16) + ceil(2 B_max / 16), problems): a block works on ONE side (its 16 positions are all user-side or all item-side)
template <bool VEC, int NV>                                             // NV: 64-column units per pass (1 when d <= 64)
__global__ __launch_bounds__(256) void bpr_bwd_runs_kernel(BprTables t, int n_prob, int d, const int64_t* __restrict__ users,
                                                           const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                           int B_max, float decay, float bsz, const float* __restrict__ saved_all,
                                                           int saved_stride, const uint64_t* __restrict__ plan,
                                                           const float* __restrict__ grads2_dev) {
    __shared__ RunShare sh;
    __shared__ RunRec recs_s[16][16];
    const int prob = blockIdx.y;
    const int nbu = (B_max + 15) / 16;
    const bool item = (int)blockIdx.x >= nbu;
    float* const dst0 = item ? t.dEi[prob] : t.dEu[prob];
    for (int q = 0; q < prob; ++q)
        if ((item ? t.dEi[q] : t.dEu[q]) == dst0) return;             // an earlier problem owns this target (and sums this one's share): block-uniform
    if (threadIdx.x == 0) {
        int n = 0, any_reg = 0;
        for (int pp = prob; pp < n_prob; ++pp) {
            if ((item ? t.dEi[pp] : t.dEu[pp]) != dst0) continue;
            const float* saved = saved_all + (int64_t)pp * saved_stride;
            const float g_emb = grads2_dev ? grads2_dev[1] : t.g_emb[pp];
            // d emb / d X = decay / bsz * (-1 / (2 S + 1e-8)^2) * 4 X
            const float base = -4.0f * decay / bsz * g_emb;
            const float du_ = 2.0f * saved[B_max] + 1e-8f, dp_ = 2.0f * saved[B_max + 1] + 1e-8f, dq_ = 2.0f * saved[B_max + 2] + 1e-8f;
            sh.Eu[n] = t.Eu[pp]; sh.Ei[n] = t.Ei[pp]; sh.saved[n] = saved; sh.ldu[n] = t.ldu[pp]; sh.ldi[n] = t.ldi[pp];
            sh.g_mf[n] = grads2_dev ? grads2_dev[0] : t.g_mf[pp];
            sh.cu[n] = base / (du_ * du_); sh.cp[n] = base / (dp_ * dp_); sh.cq[n] = base / (dq_ * dq_);
            any_reg |= base != 0.f;
            ++n;
        }
        sh.n = n; sh.any_reg = any_reg;
    }
    __syncthreads();
    const int gl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int side_n = item ? 2 * B_max : B_max, side_base = item ? B_max : 0;
    const int j0 = ((int)blockIdx.x - (item ? nbu : 0)) * 16 + grp;   // position within the side
    if (j0 >= side_n) return;
    const int32_t* runlen = reinterpret_cast<const int32_t*>(plan + 3 * (int64_t)B_max);
    const int len = runlen[side_base + j0];
    if (len <= 0) return;                                              // not the head of a run (or the run of the unused slots)
    const uint64_t* keys = plan + side_base + j0;
    const uint32_t id = (uint32_t)(keys[0] >> 32);
    float* const out = dst0 + (int64_t)id * (item ? t.lddi[prob] : t.lddu[prob]);
    const int total = sh.n * len;
    const bool reg = sh.any_reg != 0;
    RunRec* recs = recs_s[grp];
    const int shift = 16 * (grp & 3);                                  // this group's 16 bits of the wave's ballot
    for (int c0 = 0; c0 < d; c0 += 64 * NV) {
        RunAcc<NV, VEC> acc;
        acc.load(out, c0, d, gl);
        for (int t0 = 0; t0 < total; t0 += 16) {
            const int tt = t0 + gl;
            bool act = false;
            RunRec r = {};
            if (tt < total) {
                const int s_ = tt / len, m_ = tt - s_ * len;
                const uint32_t slot = (uint32_t)keys[m_];
                const bool is_neg = slot >= (uint32_t)B_max;
                const int b = (int)(is_neg ? slot - (uint32_t)B_max : slot);
                const float ds = sh.g_mf[s_] * sh.saved[s_][b];
                if (item) {                                                // d/dEi[p_b] = ds u + cp p ; d/dEi[q_b] = -ds u + cq q
                    r.ds = is_neg ? -ds : ds; r.cr = is_neg ? sh.cq[s_] : sh.cp[s_];
                    act = !(ds == 0.f && r.cr == 0.f);
                    if (act) { r.a = sh.Eu[s_] + users[b] * sh.ldu[s_]; r.b = r.a; r.self = sh.Ei[s_] + (int64_t)id * sh.ldi[s_]; }
                } else {                                                   // d/dEu[u_b] = ds (p - q) + cu u
                    r.ds = ds; r.cr = sh.cu[s_];
                    act = !(ds == 0.f && r.cr == 0.f);
                    if (act) { r.a = sh.Ei[s_] + pos[b] * sh.ldi[s_]; r.b = sh.Ei[s_] + neg[b] * sh.ldi[s_]; r.self = sh.Eu[s_] + (int64_t)id * sh.ldu[s_]; }
                }
            }
            const uint32_t bits = (uint32_t)(__ballot(act) >> shift) & 0xffffu;
            const int n_act = __popc(bits);
            if (act) recs[__popc(bits & ((1u << gl) - 1u))] = r;
            __builtin_amdgcn_wave_barrier();                           // (one wave: its LDS writes and reads execute in order)
            if (item) {
                if (reg) run_stream<NV, VEC, true, true>(acc, recs, n_act, c0, d, gl);
                else run_stream<NV, VEC, true, false>(acc, recs, n_act, c0, d, gl);
            } else {
                if (reg) run_stream<NV, VEC, false, true>(acc, recs, n_act, c0, d, gl);
                else run_stream<NV, VEC, false, false>(acc, recs, n_act, c0, d, gl);
            }
            __builtin_amdgcn_wave_barrier();
        }
        acc.store(out, c0, d, gl);
    }
}

// selection of a LOCAL batch (first launch of llmrec_bpr_multi_select_bwd_f32): grid = (ceil(B_max / 16) + 1, problems).
// Blocks [0, ceil(B_max / 16)): stage the batch's log-sigmoids in LDS and rank their 16 samples as bpr_rank_kernel does (kept
// coefficient -> saved[b], kept value -> slot 1), stamp the rows the batch touches. The LAST block of a problem sums the three
// squared-norm columns with the summation tree of bpr_reduce_kernel - thread t of that 1024-thread launch is emulated by the four
// slots t, t + 256, ... of this block's threads, then the same pairwise tree over 1024 LDS slots - so saved[B_max .. B_max + 2] carry
// the bits the loss launch writes later (the backward launch reads them).
__global__ __launch_bounds__(256) void bpr_select_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ neg, int B_max,
                                                         const int32_t* __restrict__ n_valid_dev, double remember_rate,
                                                         float* __restrict__ saved_all, int saved_stride,
                                                         uint8_t* __restrict__ flag_u, uint8_t* __restrict__ flag_i,
                                                         const int32_t* __restrict__ row_stamp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* m_s = reinterpret_cast<float*>(smem);                      // [B] (rank blocks) / [3][1024] (the norm block)
    const int B = bpr_batch(n_valid_dev, B_max);
    const int prob = blockIdx.y;
    float* saved = saved_all + (int64_t)prob * saved_stride;
    float* sc = saved + B_max + 4;
    const int k = (int)(remember_rate * (double)B);
    if (blockIdx.x == gridDim.x - 1) {
        float (*red)[BPR_THREADS] = reinterpret_cast<float (*)[BPR_THREADS]>(smem);
        for (int v = threadIdx.x; v < BPR_THREADS; v += 256) {        // the partial sums of bpr_reduce_kernel's thread v
            float su = 0.f, sp = 0.f, sq = 0.f;
            for (int b = v; b < B; b += BPR_THREADS) { su += sc[2 * B_max + b]; sp += sc[3 * B_max + b]; sq += sc[4 * B_max + b]; }
            red[0][v] = su; red[1][v] = sp; red[2][v] = sq;
        }
        __syncthreads();
        for (int off = BPR_THREADS / 2; off > 0; off >>= 1) {         // block_tree_sum's tree, three columns at once
            for (int i = threadIdx.x; i < off; i += 256) {
                red[0][i] += red[0][i + off]; red[1][i] += red[1][i + off]; red[2][i] += red[2][i + off];
            }
            __syncthreads();
        }
        if (threadIdx.x < 3) saved[B_max + threadIdx.x] = red[threadIdx.x][0];
        if (threadIdx.x == 3) saved[B_max + 3] = (float)k;
        return;
    }
    for (int j = threadIdx.x; j < B; j += 256) m_s[j] = sc[j];
    __syncthreads();
    const int gl = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= B_max) return;
    if (b >= B) { if (gl == 0) { saved[b] = 0.f; sc[B_max + b] = 0.f; } return; }
    const float mb = m_s[b];
    bool keep = true;
    if (k < B) {
        int rank = 0;
        for (int j = gl; j < B; j += 16) {
            const float mj = m_s[j];
            rank += (mj < mb) || (mj == mb && j < b);
        }
        rank = (int)group_sum<16>((float)rank);                           // < 2^24: exact
        keep = rank < k;
    }
    const float coef = keep ? (-1.0f / (float)k) * sc[B_max + b] : 0.f;   // every lane of the group reads slot 1 before lane 0 rewrites it
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) {
        saved[b] = coef; sc[B_max + b] = keep ? mb : 0.f;
        if (prob == 0) {                                              // rows this batch touches (same rows for every problem)
            const uint8_t stamp = row_stamp ? LLMREC_ROW_STAMP(row_stamp[0]) : (uint8_t)1;
            if (flag_u) flag_u[users[b]] = stamp;
            if (flag_i) { flag_i[pos[b]] = stamp; flag_i[neg[b]] = stamp; }
        }
    }
}

// the rows the loss backward added into are cleared again (one lane group per sample): the scatter targets stay all-zero between steps
// without a dense memset
__global__ __launch_bounds__(256) void bpr_zero_rows_kernel(BprTables t, int d, const int64_t* __restrict__ users,
                                                            const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                            int B_max, const int32_t* __restrict__ n_valid_dev) {
    const int B = bpr_batch(n_valid_dev, B_max);
    const int prob = blockIdx.y;
    const int gl = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= B) return;
    float* du = t.dEu[prob] + users[b] * t.lddu[prob];
    float* dpp = t.dEi[prob] + pos[b] * t.lddi[prob];
    float* dqq = t.dEi[prob] + neg[b] * t.lddi[prob];
    for (int c = gl; c < d; c += 16) { du[c] = 0.f; dpp[c] = 0.f; dqq[c] = 0.f; }
}

// the same gradient as compact rows (row-sharded step): rows3[0][b] = d/dEu[u_b], rows3[1][b] = d/dEi[p_b], rows3[2][b] = d/dEi[q_b]
__global__ __launch_bounds__(256) void bpr_bwd_rows_kernel(const float* __restrict__ Eu, int64_t ldu,
                                                           const float* __restrict__ Ei, int64_t ldi, int d,
                                                           const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                           const int64_t* __restrict__ neg, int B_max,
                                                           const int32_t* __restrict__ n_valid_dev, float decay, float bsz,
                                                           const float* __restrict__ saved, const float* __restrict__ grads2,
                                                           float* __restrict__ rows3) {
    int B = n_valid_dev ? n_valid_dev[0] : B_max;
    if (B > B_max) B = B_max;
    const float g_mf = grads2[0], g_emb = grads2[1];
    const float Su = saved[B_max], Sp = saved[B_max + 1], Sq = saved[B_max + 2];
    const float base = -4.0f * decay / bsz * g_emb;
    const float du_ = 2.0f * Su + 1e-8f, dp_ = 2.0f * Sp + 1e-8f, dq_ = 2.0f * Sq + 1e-8f;
    const float cu = base / (du_ * du_), cp = base / (dp_ * dp_), cq = base / (dq_ * dq_);
    const int gl = threadIdx.x & 15;
    const int groups = gridDim.x * (blockDim.x >> 4);
    const int64_t plane = (int64_t)B_max * d;
    for (int b = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); b < B_max; b += groups) {
        float* ru = rows3 + (int64_t)b * d;
        if (b >= B) {
            for (int c = gl; c < d; c += 16) { ru[c] = 0.f; ru[plane + c] = 0.f; ru[2 * plane + c] = 0.f; }
            continue;
        }
        const float ds = g_mf * saved[b];
        const float* u = Eu + users[b] * ldu;
        const float* p = Ei + pos[b] * ldi;
        const float* q = Ei + neg[b] * ldi;
        for (int c = gl; c < d; c += 16) {
            const float uu = u[c], pp = p[c], qq = q[c];
            ru[c] = fmaf(ds, pp - qq, cu * uu);
            ru[plane + c] = fmaf(ds, uu, cp * pp);
            ru[2 * plane + c] = fmaf(-ds, uu, cq * qq);
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

// one BPR triple of the global batch: user slot b of B (without replacement while B <= n_exist), a uniform
// train item of that user, a uniform non-train item by rejection (binary search in the sorted row)
__device__ __forceinline__ void sample_one(const Philox& ph, uint32_t slo, uint32_t shi, int b, int B, int half_bits,
                                           int64_t n_exist, const int64_t* __restrict__ exist_users, int64_t n_items,
                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                           int64_t& u_out, int64_t& p_out, int64_t& q_out) {
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
    u_out = u; p_out = p; q_out = q;
}

__global__ void sample_bpr_kernel(uint64_t seed, uint64_t step, int64_t n_exist, const int64_t* __restrict__ exist_users,
                                  int64_t n_items, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                  int B, int half_bits, int64_t* __restrict__ users, int64_t* __restrict__ pos,
                                  int64_t* __restrict__ neg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Philox ph(seed);
    sample_one(ph, (uint32_t)step, (uint32_t)(step >> 32), b, B, half_bits, n_exist, exist_users, n_items, rowptr, colidx,
               users[b], pos[b], neg[b]);
}

// The whole mini-batch of one step in ONE single-block launch that a HIP graph can replay: the step counter
// lives on the device and is advanced here. Slots [0, B): this rank's slice [slice_begin, slice_begin + B) of
// the global batch of B_global triples. Slots [B, B + n_aug): the LLM-augmented triples of reference
// main.py:216-224 - n_aug distinct users of the slice (random.sample), their (aug_pos, aug_neg) pair kept
// only if both ids are < n_items; kept pairs first, in draw order, then padding; n_valid = B + kept.
constexpr int SAMPLE_THREADS = 1024;
__global__ __launch_bounds__(SAMPLE_THREADS) void sample_batch_kernel(
    uint64_t seed, unsigned long long* __restrict__ step_dev, int64_t n_exist, const int64_t* __restrict__ exist_users,
    int64_t n_items, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
    int B_global, int half_bits_users, int slice_begin, int B, int n_aug, int half_bits_batch,
    const int64_t* __restrict__ aug_pos, const int64_t* __restrict__ aug_neg,
    int64_t* __restrict__ users, int64_t* __restrict__ pos, int64_t* __restrict__ neg, int32_t* __restrict__ n_valid_dev) {
    __shared__ int wave_tot[SAMPLE_THREADS / 64];
    __shared__ int base_s;
    const unsigned long long step = *step_dev;
    const uint32_t slo = (uint32_t)step, shi = (uint32_t)(step >> 32);
    const Philox ph(seed);
    for (int b = threadIdx.x; b < B; b += SAMPLE_THREADS)
        sample_one(ph, slo, shi, slice_begin + b, B_global, half_bits_users, n_exist, exist_users, n_items, rowptr, colidx,
                   users[b], pos[b], neg[b]);
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();                                                    // the slice is written (block-visible); everyone has read the step
    const Philox pa(seed ^ 0x9E3779B97F4A7C15ull);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int a0 = 0; a0 < n_aug; a0 += SAMPLE_THREADS) {               // block-uniform
        const int a = a0 + threadIdx.x;
        int64_t u = 0, ap = 0, an = 0;
        bool ok = false;
        if (a < n_aug) {
            const uint64_t slot = keyed_perm((uint64_t)a, (uint64_t)B, half_bits_batch, pa, slo, shi);   // distinct slots of the slice
            u = users[slot];
            ap = aug_pos[u]; an = aug_neg[u];
            ok = ap >= 0 && an >= 0 && ap < n_items && an < n_items;
        }
        const unsigned long long bal = __ballot(ok);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int wave_base = 0, chunk_tot = 0;
        for (int k = 0; k < SAMPLE_THREADS / 64; ++k) { const int t = wave_tot[k]; if (k < wv) wave_base += t; chunk_tot += t; }
        const int base = base_s;
        if (a < n_aug && ok) {
            const int o = B + base + wave_base + before;
            users[o] = u; pos[o] = ap; neg[o] = an;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + chunk_tot;
        __syncthreads();
    }
    const int kept = base_s;
    for (int o = B + kept + threadIdx.x; o < B + n_aug; o += SAMPLE_THREADS) { users[o] = 0; pos[o] = 0; neg[o] = 0; }   // padding (never read: beyond n_valid)
    if (threadIdx.x == 0) { n_valid_dev[0] = B + kept; *step_dev = step + 1ull; }
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

static int launch_bpr_fwd(const BprTables& t, int n_prob, int d, const int64_t* users, const int64_t* pos, const int64_t* neg,
                          int B_max, const int32_t* n_valid_dev, double remember_rate, float decay, float bsz,
                          float* out, float* saved, const BprGather& ga, bool do_scores, bool do_select, float* pack_out,
                          hipStream_t stream, StepBegin step_begin = StepBegin{}) {
    const int stride = LLMREC_BPR_SAVED_FLOATS(B_max);
    const BprGather none = {};
    if (do_scores && B_max > 0) {
        dim3 grid((unsigned)ceil_div(B_max, 16), (unsigned)n_prob);
        bpr_scores_kernel<<<grid, 256, 0, stream>>>(t, d, users, pos, neg, B_max, n_valid_dev, saved, stride, step_begin);
        LLMREC_LAUNCH_CHECK();
    }
    if (pack_out) {                                                     // local norm sums, then this rank's gather block
        bpr_reduce_kernel<<<n_prob, BPR_THREADS, 0, stream>>>(B_max, n_valid_dev, remember_rate, decay, bsz, out, saved, stride, none);
        LLMREC_LAUNCH_CHECK();
        bpr_pack_kernel<<<ceil_div(n_prob * B_max + 4 * n_prob + 1, 256), 256, 0, stream>>>(n_prob, B_max, n_valid_dev, saved, stride, pack_out);
        LLMREC_LAUNCH_CHECK();
    }
    if (do_select) {
        const size_t shmem = sizeof(float) * (size_t)(ga.g ? ga.n_ranks * ga.cap : (B_max > 0 ? B_max : 1));
        dim3 grid((unsigned)ceil_div(B_max > 0 ? B_max : 1, 16), (unsigned)n_prob);
        bpr_rank_kernel<<<grid, 256, shmem, stream>>>(B_max, n_valid_dev, remember_rate, saved, stride, ga);
        LLMREC_LAUNCH_CHECK();
        bpr_reduce_kernel<<<n_prob, BPR_THREADS, 0, stream>>>(B_max, n_valid_dev, remember_rate, decay, bsz, out, saved, stride, ga);
        LLMREC_LAUNCH_CHECK();
    }
    return LLMREC_OK;
}

int llmrec_bpr_prune_fwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev,
                             double remember_rate, float decay, float batch_size_flag,
                             float* out2, float* saved, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(B_max >= 0 && d > 0 && out2 && saved, "bpr_fwd: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_fwd: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max == 0 || (Eu && Ei && users && pos && neg && ldu >= d && ldi >= d), "bpr_fwd: null pointer or ld < d");
    BprTables t = {};
    t.Eu[0] = Eu; t.Ei[0] = Ei; t.ldu[0] = ldu; t.ldi[0] = ldi;
    return launch_bpr_fwd(t, 1, d, users, pos, neg, B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out2, saved,
                          BprGather{}, true, true, nullptr, (hipStream_t)stream_);
}

int llmrec_bpr_prune_fwd_sharded_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_local, double remember_rate, float decay, float batch_size_flag,
                                     const float* global_m, int32_t global_B, int32_t my_offset, int32_t scores_only,
                                     float* out2, float* saved, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(B_local >= 0 && d > 0 && out2 && saved, "bpr_fwd_sharded: bad argument");
    if (B_local > LLMREC_BPR_MAX_B || global_B > 3 * LLMREC_BPR_MAX_B) { set_error("bpr_fwd_sharded: batch too large"); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_local == 0 || (Eu && Ei && users && pos && neg && ldu >= d && ldi >= d), "bpr_fwd_sharded: null pointer or ld < d");
    LLMREC_CHECK_ARG(scores_only || (global_m && global_B >= B_local && my_offset >= 0 && my_offset + B_local <= global_B),
                     "bpr_fwd_sharded: pass 2 needs the gathered scores and a valid offset");
    BprTables t = {};
    t.Eu[0] = Eu; t.Ei[0] = Ei; t.ldu[0] = ldu; t.ldi[0] = ldi;
    // pass 1 writes the per-sample scratch (m at saved[B_local + 4 ...)); pass 2 only selects
    BprGather ga = {};
    if (!scores_only) { ga.g = global_m; ga.n_ranks = 1; ga.stride = 0; ga.cap = global_B; ga.n_prob = 1; ga.my_offset = my_offset; ga.has_tail = 0; }
    return launch_bpr_fwd(t, 1, d, users, pos, neg, B_local, nullptr, remember_rate, decay, batch_size_flag, out2, saved,
                          ga, scores_only != 0, scores_only == 0, nullptr, (hipStream_t)stream_);
}

static int fill_tables(BprTables& t, int n, const llmrec_bpr_problem_t* p, int d, bool need_grads) {
    for (int i = 0; i < n; ++i) {
        if (!p[i].Eu || !p[i].Ei || p[i].ldu < d || p[i].ldi < d) return 1;
        if (need_grads && (!p[i].dEu || !p[i].dEi || p[i].lddu < d || p[i].lddi < d)) return 1;
        t.Eu[i] = p[i].Eu; t.Ei[i] = p[i].Ei; t.ldu[i] = p[i].ldu; t.ldi[i] = p[i].ldi;
        t.dEu[i] = p[i].dEu; t.dEi[i] = p[i].dEi; t.lddu[i] = p[i].lddu; t.lddi[i] = p[i].lddi;
        t.g_mf[i] = p[i].g_mf; t.g_emb[i] = p[i].g_emb;
    }
    return 0;
}

int llmrec_bpr_multi_fwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                             float batch_size_flag, float* out, float* saved, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && out && saved,
                     "bpr_multi_fwd: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_fwd: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max == 0 || (users && pos && neg), "bpr_multi_fwd: null index pointer");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, false), "bpr_multi_fwd: bad problem table");
    return launch_bpr_fwd(t, n_problems, d, users, pos, neg, B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out, saved,
                          BprGather{}, true, true, nullptr, (hipStream_t)stream_);
}

int llmrec_bpr_multi_fwd_sharded_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                     float batch_size_flag, int32_t phase, float* gather_block,
                                     const float* gathered, int32_t n_ranks, int64_t rank_stride, int32_t my_rank,
                                     float* out, float* saved, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && out && saved,
                     "bpr_multi_fwd_sharded: bad argument");
    LLMREC_CHECK_ARG(phase == 1 || phase == 2, "bpr_multi_fwd_sharded: phase is 1 (scores + gather block) or 2 (selection)");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_fwd_sharded: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max == 0 || (users && pos && neg), "bpr_multi_fwd_sharded: null index pointer");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, false), "bpr_multi_fwd_sharded: bad problem table");
    if (phase == 1) {
        LLMREC_CHECK_ARG(gather_block, "bpr_multi_fwd_sharded: phase 1 needs the gather block");
        return launch_bpr_fwd(t, n_problems, d, users, pos, neg, B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out, saved,
                              BprGather{}, true, false, gather_block, (hipStream_t)stream_);
    }
    LLMREC_CHECK_ARG(gathered && n_ranks >= 1 && my_rank >= 0 && my_rank < n_ranks &&
                     rank_stride >= LLMREC_BPR_GATHER_FLOATS(n_problems, B_max), "bpr_multi_fwd_sharded: bad gathered layout");
    if ((int64_t)n_ranks * B_max > 4 * LLMREC_BPR_MAX_B) {
        set_error("bpr_multi_fwd_sharded: global batch capacity %lld > %d", (long long)n_ranks * B_max, 4 * LLMREC_BPR_MAX_B);
        return LLMREC_EUNSUPPORTED;
    }
    BprGather ga = {};
    ga.g = gathered; ga.n_ranks = n_ranks; ga.stride = rank_stride; ga.cap = B_max; ga.n_prob = n_problems;
    ga.my_offset = my_rank * B_max; ga.has_tail = 1;
    return launch_bpr_fwd(t, n_problems, d, users, pos, neg, B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out, saved,
                          ga, false, true, nullptr, (hipStream_t)stream_);
}

int llmrec_bpr_scatter_plan(const int64_t* users, const int64_t* pos, const int64_t* neg, int32_t B_max, const int32_t* n_valid_dev,
                            uint64_t* plan, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(B_max >= 0, "bpr_scatter_plan: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_scatter_plan: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(users && pos && neg && plan, "bpr_scatter_plan: null pointer");
    const unsigned blocks = (unsigned)(ceil_div(B_max, 16) + ceil_div(2 * (int64_t)B_max, 16));
    bpr_plan_kernel<<<blocks, 256, sizeof(uint64_t) * 2 * (size_t)B_max, (hipStream_t)stream_>>>(users, pos, neg, B_max, n_valid_dev, plan);   // <= 64 KB of LDS
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

static int launch_bwd_runs(const BprTables& t, int n_prob, int d, const int64_t* users, const int64_t* pos, const int64_t* neg, int B_max,
                           float decay, float bsz, const float* saved, const uint64_t* plan, const float* grads2_dev, hipStream_t stream) {
    dim3 grid((unsigned)(ceil_div(B_max, 16) + ceil_div(2 * (int64_t)B_max, 16)), (unsigned)n_prob);
    bool vec = d % 4 == 0;                                             // float4 rows: every row pointer 16-byte aligned
    for (int i = 0; i < n_prob && vec; ++i)
        vec = t.ldu[i] % 4 == 0 && t.ldi[i] % 4 == 0 && t.lddu[i] % 4 == 0 && t.lddi[i] % 4 == 0 &&
              ((uintptr_t)t.Eu[i] | (uintptr_t)t.Ei[i] | (uintptr_t)t.dEu[i] | (uintptr_t)t.dEi[i]) % 16 == 0;
    const int stride = LLMREC_BPR_SAVED_FLOATS(B_max);
#define RUNS_LAUNCH(V, N) bpr_bwd_runs_kernel<V, N><<<grid, 256, 0, stream>>>(t, n_prob, d, users, pos, neg, B_max, decay, bsz, saved, stride, plan, grads2_dev)
    if (vec) { if (d <= 64) RUNS_LAUNCH(true, 1); else RUNS_LAUNCH(true, 2); }
    else { if (d <= 64) RUNS_LAUNCH(false, 1); else RUNS_LAUNCH(false, 2); }
#undef RUNS_LAUNCH
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_multi_bwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, float decay, float batch_size_flag,
                             const float* saved, const uint64_t* plan, llmrec_stream_t stream_) {
    (void)n_valid_dev;                                                  // (the plan carries it: slots beyond n_valid are not listed)
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && saved,
                     "bpr_multi_bwd: bad argument");
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(users && pos && neg && plan, "bpr_multi_bwd: null index pointer or no scatter plan (llmrec_bpr_scatter_plan)");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, true), "bpr_multi_bwd: bad problem table");
    return launch_bwd_runs(t, n_problems, d, users, pos, neg, B_max, decay, batch_size_flag, saved, plan, nullptr, (hipStream_t)stream_);
}

int llmrec_bpr_multi_scores_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                const int64_t* users, const int64_t* pos, const int64_t* neg,
                                int32_t B_max, const int32_t* n_valid_dev, float* saved, int32_t* row_stamp, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && saved, "bpr_multi_scores: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_scores: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max == 0 || (users && pos && neg), "bpr_multi_scores: null index pointer");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, false), "bpr_multi_scores: bad problem table");
    StepBegin sb = {};
    sb.row_stamp = row_stamp;
    return launch_bpr_fwd(t, n_problems, d, users, pos, neg, B_max, n_valid_dev, 0.0, 0.f, 1.f, nullptr, saved, BprGather{}, true, false,
                          nullptr, (hipStream_t)stream_, sb);
}

int llmrec_bpr_multi_scores_step_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                     const int64_t* users, const int64_t* pos, const int64_t* neg,
                                     int32_t B_max, const int32_t* n_valid_dev, float* saved, int32_t* row_stamp,
                                     float* adamw_state3, float lr, float beta1, float beta2, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && saved, "bpr_multi_scores_step: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_scores_step: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LLMREC_CHECK_ARG(B_max > 0 && users && pos && neg, "bpr_multi_scores_step: empty batch capacity or null index pointer (the launch carries the step's counters)");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, false), "bpr_multi_scores_step: bad problem table");
    StepBegin sb = {};
    sb.row_stamp = row_stamp; sb.adamw_state = adamw_state3; sb.lr = lr; sb.b1 = beta1; sb.b2 = beta2;
    return launch_bpr_fwd(t, n_problems, d, users, pos, neg, B_max, n_valid_dev, 0.0, 0.f, 1.f, nullptr, saved, BprGather{}, true, false,
                          nullptr, (hipStream_t)stream_, sb);
}

int llmrec_bpr_multi_select_bwd_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                    const int64_t* users, const int64_t* pos, const int64_t* neg,
                                    int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                    float batch_size_flag, float* saved, uint8_t* user_row_flags, uint8_t* item_row_flags,
                                    const int32_t* row_stamp, const uint64_t* plan, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0 && saved, "bpr_multi_select_bwd: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_select_bwd: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(users && pos && neg && plan, "bpr_multi_select_bwd: null index pointer or no scatter plan (llmrec_bpr_scatter_plan)");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, true), "bpr_multi_select_bwd: bad problem table");
    dim3 grid((unsigned)ceil_div(B_max, 16) + 1u, (unsigned)n_problems);
    const size_t shmem = sizeof(float) * (size_t)(B_max > 3 * BPR_THREADS ? B_max : 3 * BPR_THREADS);
    bpr_select_kernel<<<grid, 256, shmem, (hipStream_t)stream_>>>(users, pos, neg, B_max, n_valid_dev, remember_rate, saved,
                                                                 LLMREC_BPR_SAVED_FLOATS(B_max), user_row_flags, item_row_flags, row_stamp);
    LLMREC_LAUNCH_CHECK();
    return launch_bwd_runs(t, n_problems, d, users, pos, neg, B_max, decay, batch_size_flag, saved, plan, nullptr, (hipStream_t)stream_);
}

int llmrec_bpr_multi_losses_f32(int32_t n_problems, int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                float batch_size_flag, float* out, float* saved, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && B_max >= 0 && out && saved, "bpr_multi_losses: bad argument");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_losses: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    bpr_reduce_kernel<<<n_problems, BPR_THREADS, 0, (hipStream_t)stream_>>>(B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out, saved,
                                                                           LLMREC_BPR_SAVED_FLOATS(B_max), BprGather{});
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_multi_losses_assemble_f32(int32_t n_problems, int32_t B_max, const int32_t* n_valid_dev, double remember_rate, float decay,
                                         float batch_size_flag, float* out, float* saved, const float* w_mf_host,
                                         const float* sumsq_partial, int32_t n_partial, float feat_reg_coef,
                                         float* scal4, double* running_sums3, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && B_max >= 0 && out && saved && w_mf_host && scal4,
                     "bpr_multi_losses_assemble: bad argument");
    LLMREC_CHECK_ARG(n_partial >= 0 && (n_partial == 0 || sumsq_partial), "bpr_multi_losses_assemble: partial sums without a buffer");
    if (B_max > LLMREC_BPR_MAX_B) { set_error("bpr_multi_losses_assemble: B_max %d > %d", B_max, LLMREC_BPR_MAX_B); return LLMREC_EUNSUPPORTED; }
    LossW w = {};
    for (int i = 0; i < n_problems; ++i) w.w[i] = w_mf_host[i];
    bpr_losses_assemble_kernel<<<1, BPR_THREADS, 0, (hipStream_t)stream_>>>(n_problems, B_max, n_valid_dev, remember_rate, decay, batch_size_flag, out, saved,
                                                                           LLMREC_BPR_SAVED_FLOATS(B_max), w, n_partial > 0 ? sumsq_partial : nullptr,
                                                                           n_partial, feat_reg_coef, scal4, running_sums3);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_multi_zero_rows_f32(int32_t n_problems, const llmrec_bpr_problem_t* problems_host, int32_t d,
                                   const int64_t* users, const int64_t* pos, const int64_t* neg,
                                   int32_t B_max, const int32_t* n_valid_dev, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_problems >= 1 && n_problems <= LLMREC_BPR_MAX_PROBLEMS && problems_host && d > 0, "bpr_multi_zero_rows: bad argument");
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(users && pos && neg, "bpr_multi_zero_rows: null index pointer");
    BprTables t = {};
    LLMREC_CHECK_ARG(!fill_tables(t, n_problems, problems_host, d, true), "bpr_multi_zero_rows: bad problem table");
    dim3 grid((unsigned)ceil_div(B_max, 16), (unsigned)n_problems);
    bpr_zero_rows_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(t, d, users, pos, neg, B_max, n_valid_dev);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_bpr_prune_bwd_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                             const int64_t* users, const int64_t* pos, const int64_t* neg,
                             int32_t B_max, const int32_t* n_valid_dev, float decay, float batch_size_flag,
                             const float* saved, const float* grads2,
                             float* dEu, int64_t lddu, float* dEi, int64_t lddi, const uint64_t* plan, llmrec_stream_t stream_) {
    (void)n_valid_dev;
    LLMREC_CHECK_ARG(B_max >= 0 && d > 0 && saved && grads2, "bpr_bwd: bad argument");
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(Eu && Ei && users && pos && neg && dEu && dEi && ldu >= d && ldi >= d && lddu >= d && lddi >= d,
                     "bpr_bwd: null pointer or ld < d");
    LLMREC_CHECK_ARG(plan, "bpr_bwd: no scatter plan (llmrec_bpr_scatter_plan)");
    LLMREC_CHECK_ARG(dEu != dEi, "bpr_bwd: dEu and dEi must be distinct buffers");
    BprTables t = {};
    t.Eu[0] = Eu; t.Ei[0] = Ei; t.ldu[0] = ldu; t.ldi[0] = ldi;
    t.dEu[0] = dEu; t.dEi[0] = dEi; t.lddu[0] = lddu; t.lddi[0] = lddi;
    return launch_bwd_runs(t, 1, d, users, pos, neg, B_max, decay, batch_size_flag, saved, plan, grads2, (hipStream_t)stream_);
}

int llmrec_bpr_prune_bwd_rows_f32(const float* Eu, int64_t ldu, const float* Ei, int64_t ldi, int32_t d,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg,
                                  int32_t B_max, const int32_t* n_valid_dev, float decay, float batch_size_flag,
                                  const float* saved, const float* grads2, float* rows3, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(B_max >= 0 && d > 0 && saved && grads2, "bpr_bwd_rows: bad argument");
    if (B_max == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(Eu && Ei && users && pos && neg && rows3 && ldu >= d && ldi >= d, "bpr_bwd_rows: null pointer or ld < d");
    bpr_bwd_rows_kernel<<<grid_for(B_max, 16), 256, 0, stream>>>(Eu, ldu, Ei, ldi, d, users, pos, neg, B_max, n_valid_dev, decay,
                                                                 batch_size_flag, saved, grads2, rows3);
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

int llmrec_sample_batch(uint64_t seed, uint64_t* step_dev, int64_t n_exist_users, const int64_t* exist_users,
                        int64_t n_items, const int32_t* train_rowptr, const int32_t* train_colidx,
                        int32_t B_global, int32_t slice_begin, int32_t B, int32_t n_aug,
                        const int64_t* aug_pos, const int64_t* aug_neg,
                        int64_t* users, int64_t* pos, int64_t* neg, int32_t* n_valid_dev, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(B >= 1 && B_global >= B && slice_begin >= 0 && slice_begin + B <= B_global && n_aug >= 0 && n_aug <= B &&
                     n_exist_users > 0 && n_items > 0, "sample_batch: bad sizes");
    LLMREC_CHECK_ARG(step_dev && exist_users && train_rowptr && train_colidx && users && pos && neg && n_valid_dev,
                     "sample_batch: null pointer");
    LLMREC_CHECK_ARG(n_aug == 0 || (aug_pos && aug_neg), "sample_batch: augmented pairs missing");
    int hb_users = 1, hb_batch = 1;
    while ((1ull << (2 * hb_users)) < (uint64_t)n_exist_users) ++hb_users;
    while ((1ull << (2 * hb_batch)) < (uint64_t)B) ++hb_batch;
    sample_batch_kernel<<<1, SAMPLE_THREADS, 0, (hipStream_t)stream_>>>(seed, (unsigned long long*)step_dev, n_exist_users, exist_users,
                                                                       n_items, train_rowptr, train_colidx, B_global, hb_users,
                                                                       slice_begin, B, n_aug, hb_batch, aug_pos, aug_neg,
                                                                       users, pos, neg, n_valid_dev);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // extern "C"
