// topk.hip - R9/R10: full-rank user x item scoring with train-item masking and per-user top-K.
// Replaces torch.matmul(E_u[blk], E_i^T), the D2H copy of each 2048 x I score block and the
// per-user Python set-difference + heapq.nlargest (reference utility/batch_test.py:21-36,83-109,
// 149-157). The U x I score matrix is never written to memory.
//
// The only MFMA user on the path: v_mfma_f32_16x16x4_f32 (exact fp32; a k-ordered fma chain, so
// the scores are reproducible bit for bit by a scalar fmaf loop in the order documented below).
//
// Geometry: block = 4 wavefronts sharing 16 query users (one MFMA row tile, A operand resident in
// registers for the whole launch); wavefront w sweeps the w-th quarter of the items, 32 items (two
// MFMA column tiles) per round (825 blocks x 4 waves at the Netflix shape; <= 128 registers, so all
// of them are resident at once). Operands are fed straight from global memory as float4 along k
// (see dense.hip for the k-permutation argument): lane l holds E[row (l&15)][16 c + 4 (l>>4) + s].
// Chain order of k for one score: for c in 0..d/16-1, for s in 0..3, for q in 0..3: k = 16c + 4q + s.
//
// Selection (order: score desc, item id asc - heapq.nlargest over (score, item) with the
// reference's tie behaviour): a score that reaches the user's current K-th score is APPENDED to
// an unsorted LDS buffer of its (wave, user) - all candidates of a round in parallel, slots from
// ballot/popcount prefix sums, no serial insert. The block keeps ONE sorted list per user (64
// slots, one per lane, in the registers of the wave that owns the user: wave w owns users
// 4w..4w+3). When some buffer may not hold another round (block-uniform flag after the round's
// barrier) the owners drain all buffers: 64 entries at a time, bitonic sort across the lanes, then
// a bitonic merge into the list; the new K-th score is the filter for all four waves - the GLOBAL
// K-th score of everything the block has seen, not a per-quarter one. A sweep needs ~5 drains
// (the filter tightens geometrically) and ~12 sorted vectors per user.
// Train items are removed with a per-user cursor into the user's (ascending) CSR row, turned
// into a 32-bit tile mask.
#include "common.h"
#include <limits.h>
#include <type_traits>

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
    // user tiles from split_from on are swept by n_parts blocks each (a part = a contiguous range of item tiles); their
    // 64-slot lists go to ws_idx / ws_score [(tile - split_from) * n_parts + part][16][64] and are merged by topk_merge_kernel
    int split_from, n_parts;
    int part_major;              // block ids of the split tiles: part-major (all tiles of part 0, then part 1, ...) instead of tile-major
    int32_t* ws_idx; float* ws_score;
    // the item table re-laid in FRAGMENT order by topk_pack_items_kernel (null: the sweep loads Ei itself):
    // packed[((tile * 2 + n) * DK + c) * 64 + lane] = the float4 lane `lane` feeds to column tile n, k chunk c of item tile `tile`
    const float4* packed;
    // mode 1 (bf16 prefilter + exact rescoring): the item table as bf16 (hi, mid) fragments and max_i ||i||^2 (workspace)
    int mode;
    uint4* pk2;
    uint32_t* hdr;               // [1] user tiles flagged for the exact sweep (statistics)
    float* cn;                   // per item: 2^-14 ||item|| rounded up (the item's factor of the score's upper bound)
    uint32_t* fb_word;           // one word per user tile: != 0 = the verification failed, the exact sweep redoes the tile
    const uint32_t* only_flagged; // exact sweep: blocks of tiles whose word is 0 exit at once (NULL: every tile)
    // users with long train rows: a block builds bitmaps (one word per item tile) of up to TK_HEAVY_PER_BLOCK of its users in its own
    // slice of this area and reads one word per round instead of walking the row (NULL: every row is walked)
    uint32_t* heavy_bm; int heavy_words;
    // heavy_all != 0 (bf16 sweep): EVERY row of the block is a bitmap and the block's slice is tile-major - word [tile][16 users], one 64-byte line per
    // item tile - so a lane reads the masks of its four rows with 16 bytes and no row is walked (tk_rows_setup)
    int heavy_all;
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

// value of lane (lane ^ J): DPP inside a 16-lane row (no LDS round trip), ds_swizzle for 16, ds_bpermute for 32
template <int J>
__device__ __forceinline__ int xor_lane_i(int x, int lane) {
    if (J == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);           // quad_perm [1,0,3,2]
    if (J == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);           // quad_perm [2,3,0,1]
    if (J == 4) {
        const int up = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0xf, true);          // row_shl:4  lane i <- i + 4
        const int dn = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);          // row_shr:4  lane i <- i - 4
        return (lane & 4) ? dn : up;
    }
    if (J == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true);           // row_ror:8
    if (J == 16) return __builtin_amdgcn_ds_swizzle(x, 0x401F);                            // bit mode: and 0x1f, xor 0x10
    return __shfl_xor(x, J, 64);
}
// ---- bitonic networks over 64 lanes on ONE 64-bit key per entry ----
// key = (order-preserving bits of the score) << 32 | ~item: a larger key is the better entry - higher score, then the smaller item id: the
// reference's rank rule (heapq.nlargest over (score, item) pairs, utility/batch_test.py:21-36, ties by item id asc) -; the empty slot
// (-inf, INT_MAX) is the smallest key a list can hold. One 64-bit compare per stage instead of three 32-bit ones (score >, score ==, id <):
// a stage is 2 DPP moves + 1 compare + 2 selects - the drains were ~45 % of the bf16 sweep's vector instructions at the Netflix width with the
// two-register form. A strict total order: a compare-exchange never duplicates or loses an entry (NaN scores never reach a list: the filter
// compare rejects them). The exact sweep's lists, the bf16 sweep's lists (by upper bound: there the order among equal bounds cannot change the
// output - the lists are re-ranked by the exact score and verified) and the final exact re-ranking all use it.
__device__ __forceinline__ uint32_t tk_ord(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float tk_unord(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

__device__ __forceinline__ uint64_t tk_key(float ub, int32_t id) { return ((uint64_t)tk_ord(ub) << 32) | (uint32_t)~id; }
__device__ __forceinline__ float tk_key_ub(uint64_t k) { return tk_unord((uint32_t)(k >> 32)); }
__device__ __forceinline__ int32_t tk_key_id(uint64_t k) { return (int32_t)~(uint32_t)k; }
template <int J>
__device__ __forceinline__ uint64_t xor_lane_k(uint64_t k, int lane) {
    const uint32_t ol = (uint32_t)xor_lane_i<J>((int)(uint32_t)k, lane), oh = (uint32_t)xor_lane_i<J>((int)(uint32_t)(k >> 32), lane);
    return ((uint64_t)oh << 32) | ol;
}
template <int J>
__device__ __forceinline__ void cmpxk(uint64_t& k, int lane, bool keep_better) {
    const uint64_t o = xor_lane_k<J>(k, lane);
    k = (keep_better == (o > k)) ? o : k;
}
template <int FROM>
__device__ __forceinline__ void bitonic_netk(uint64_t& k, int lane, int dirbit) {
    const bool up = (lane & dirbit) == 0;
    if (FROM >= 32) cmpxk<32>(k, lane, ((lane & 32) == 0) == up);
    if (FROM >= 16) cmpxk<16>(k, lane, ((lane & 16) == 0) == up);
    if (FROM >= 8) cmpxk<8>(k, lane, ((lane & 8) == 0) == up);
    if (FROM >= 4) cmpxk<4>(k, lane, ((lane & 4) == 0) == up);
    if (FROM >= 2) cmpxk<2>(k, lane, ((lane & 2) == 0) == up);
    cmpxk<1>(k, lane, ((lane & 1) == 0) == up);
}
__device__ __forceinline__ void merge64k(uint64_t& k, uint64_t b, int lane) {
    const uint32_t rl = (uint32_t)__shfl((int)(uint32_t)b, 63 - lane, 64), rh = (uint32_t)__shfl((int)(uint32_t)(b >> 32), 63 - lane, 64);
    const uint64_t r = ((uint64_t)rh << 32) | rl;
    k = r > k ? r : k;
    bitonic_netk<32>(k, lane, 0);
}
__device__ __forceinline__ void sort64k(uint64_t& k, int lane) {
    bitonic_netk<1>(k, lane, 2);
    bitonic_netk<2>(k, lane, 4);
    bitonic_netk<4>(k, lane, 8);
    bitonic_netk<8>(k, lane, 16);
    bitonic_netk<16>(k, lane, 32);
    bitonic_netk<32>(k, lane, 0);
}
constexpr uint64_t TK_KEY_EMPTY = 0x007FFFFF80000000ull;   // tk_key(-inf, INT_MAX)

// (The cycle accounting and the one-part-removed ablation builds of round 2 - profiles/r02_topk_ablation.txt - were tools-only
// variants of this kernel; they are no longer part of the source.)

constexpr int TK_TILE = 32;   // items per wave per round
constexpr int TK_TRAIN_STAGE = 16;   // train items staged in LDS per (wave, user): the sweep's rounds never wait for a train-row load
// the train items [cur, cur + TK_TRAIN_STAGE) of one user's row -> this lane's LDS slots (INT_MAX past the row's end). The loads are issued
// four at a time (clamped index, no branch around them) and waited for HERE: at the start of the sweep, and again only when a (wave, user)
// pair has consumed all TK_TRAIN_STAGE staged items. (A load per consumed item inside the sweep stalled the wave for a memory latency
// at every train item - vmcnt counts loads in order, the next tile's fragments queue behind it: 0.07 ms of 0.37 at the Netflix shape.)
__device__ __forceinline__ void tk_train_stage(const int32_t* __restrict__ colidx, int32_t cur, int32_t end, int32_t* slot) {
    if (cur >= end) {
#pragma unroll
        for (int j = 0; j < TK_TRAIN_STAGE; ++j) slot[j] = INT_MAX;
        return;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < TK_TRAIN_STAGE; j0 += 4) {           // four loads in flight at a time (rolled: the sweep has no registers to spare)
        int32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = colidx[cur + j0 + j < end ? cur + j0 + j : end - 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) slot[j0 + j] = cur + j0 + j < end ? v[j] : INT_MAX;
    }
}
// inside the sweep: the users in `need` (bit = row-owner lane) have consumed their staged window; their next TK_TRAIN_STAGE items are loaded by
// the wave's first lanes, one coalesced load per user (scalar row bounds: no registers beyond one value per lane)
__device__ __forceinline__ void tk_train_refill(const int32_t* __restrict__ colidx, unsigned need, int32_t cur, int32_t end,
                                                int32_t (*rows)[TK_TRAIN_STAGE], int lane) {
    while (need) {                                             // wave-uniform
        const int u = __builtin_ctz(need);
        need &= need - 1u;
        const int32_t cu = __builtin_amdgcn_readlane(cur, u), eu = __builtin_amdgcn_readlane(end, u);   // cu: a multiple of TK_TRAIN_STAGE
        if (lane < TK_TRAIN_STAGE) rows[u][lane] = cu + lane < eu ? colidx[cu + lane] : INT_MAX;
    }
}
// ---- long train rows ----
// The row walk above costs an LDS round trip per train item and a memory latency per TK_TRAIN_STAGE of them, in ONE wavefront: the bench's
// heaviest user (2 012 train items) kept its block sweeping 35 us after every other block had finished. A block therefore turns the rows of
// up to TK_HEAVY_PER_BLOCK of its users with more than TK_HEAVY_DEG train items into bitmaps - one 32-bit word per item tile, in the
// block's own slice of the workspace: no allocation, no cross-block traffic - and those users' row-owner lanes read one word per round (with
// the tile prefetch) instead. Further long rows of the same block are walked. Called by all 256 threads (every wave holds the same 16 rows
// in its lanes 0..15); returns this lane's word offset inside the block's slice, -1: walk the row.
constexpr int TK_HEAVY_PER_BLOCK = 2, TK_HEAVY_DEG = 48;
// rows of heavy_words words a block's slice holds: 16 where the workspace is laid out for tk_rows_setup (a block may still choose the two-row form)
__device__ __forceinline__ int tk_heavy_stride(const TopkArgs& a) { return a.heavy_all ? 16 : TK_HEAVY_PER_BLOCK; }
__device__ __forceinline__ int tk_heavy_setup(const TopkArgs& a, int lane, bool row_valid, int32_t row_begin, int32_t row_end) {
    if (!a.heavy_bm) return -1;                                // uniform
    unsigned hb = (unsigned)__ballot(lane < 16 && row_valid && row_end - row_begin > TK_HEAVY_DEG) & 0xffffu;
    if (hb == 0u) return -1;                                   // block-uniform: every wave sees the same rows
    uint32_t* const slice = a.heavy_bm + (size_t)blockIdx.x * tk_heavy_stride(a) * a.heavy_words;
    int mine = -1;
    for (int s = 0; s < TK_HEAVY_PER_BLOCK && hb != 0u; ++s) {
        const int u = __builtin_ctz(hb);
        hb &= hb - 1u;
        const int32_t rb = __builtin_amdgcn_readlane(row_begin, u), re = __builtin_amdgcn_readlane(row_end, u);
        uint32_t* const row = slice + (size_t)s * a.heavy_words;
        for (int i = threadIdx.x; i < a.heavy_words; i += 256) row[i] = 0u;
        // the slice is this block's own and lives in this XCD's L2: no agent-scope fence (an L2 write-back per block: measured 0.02 ms of the
        // sweep) - but the zeroes must have reached the L2 before another wave's atomics: the workgroup-scope fence emits no wait on gfx950
        // (one CU, one L1), so wait for this wave's stores explicitly
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int e = rb + (int)threadIdx.x; e < re; e += 256) {
            const int32_t it = a.train_colidx[e];
            if ((uint32_t)it < (uint32_t)a.n_items) atomicOr(&row[it >> 5], 1u << (it & 31));   // (an id outside the table never matches in the walk either)
        }
        if (lane == u) mine = s * a.heavy_words;
        if (a.hdr && threadIdx.x == 0) atomicAdd(&a.hdr[2], 1u);      // (statistics: train rows swept as bitmaps, counted per block)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the atomics are performed at the L2 before any wave reads a word back)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // (the L1 drops what it holds: the words are read with plain loads)
    return mine;
}
// (a plain load: the set-up ended with an acquire fence - this CU's L1 holds no line of the slice from before the atomics - and the slice is written by
//  this block alone. The agent-scope atomic load used until round 6 is served beyond the XCD's L2 on this part: ~0.5 us per round, measured as + 70 us per
//  sweep when every block read two such words per round.)
__device__ __forceinline__ uint32_t tk_heavy_word(const TopkArgs& a, int off, int64_t tile) {
    return a.heavy_bm[(size_t)blockIdx.x * tk_heavy_stride(a) * a.heavy_words + off + tile];
}
// ---- every train row of the block as a bitmap, tile-major (bf16 sweep, tables of up to 131 072 items while the slices fit 64 MB) ----
// The row walk of the sweep - compare the row's next item with the tile, step the cursor through the LDS stage, exchange the masks through LDS - was
// ~45 of the ~200 instructions a wave issues per round, and the sweep is bound by exactly that count. Here the block turns ALL its rows into bitmaps
// once (a few hundred train items, one LDS atomicOr each) in a slice of its own laid out [item tile][16 users]: the masks a lane needs for a round - the
// words of its four rows - are 16 consecutive bytes, ONE load issued with the tile's fragments one round ahead. Called by all 256 threads.
__device__ __forceinline__ void tk_rows_setup(const TopkArgs& a, int lane, bool row_valid, int32_t row_begin, int32_t row_end, uint32_t* lds, int lds_words,
                                              int blk_t0, int blk_t1, int wave_t0, int wave_t1) {
    // Built in LDS (the candidate pools' 32 KB, unused until the sweep starts), lds_words / 16 tiles of all 16 rows at a time, and copied out with
    // 16-byte stores: zeroing the slice in global memory and one global atomicOr per train item - the first form - cost 49 us of a 0.31 ms sweep
    // (all blocks start together: 36 MB of zeroes, then 0.66 M atomics at the L2). Only the tiles the block sweeps [blk_t0, blk_t1) are built, and every
    // wave copies out the tiles of ITS quarter [wave_t0, wave_t1): it is the only reader of those words, so it waits for its own stores only (no
    // barrier and no L1 invalidate behind the copy: the words were never read before and the L1 is write-through).
    uint32_t* const slice = a.heavy_bm + (size_t)blockIdx.x * 16 * a.heavy_words;
    // the first 256 items of every row: 16 independent loads per thread, in flight together (one row after the other - load, wait, atomic - was a chain
    // of 16 memory latencies per pass: 28 us of set-up); what a row holds beyond 256 items is read in the passes
    int32_t first[16];
    unsigned long_rows = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int32_t rb = __builtin_amdgcn_readlane(row_begin, u), re = __builtin_amdgcn_readlane(row_valid ? row_end : row_begin, u);
        const int e = rb + (int)threadIdx.x;
        first[u] = e < re ? a.train_colidx[e] : -1;
        if (re - rb > 256) long_rows |= 1u << u;
    }
    const int stage = lds_words / 16;                           // item tiles per pass
    for (int t0 = blk_t0; t0 < blk_t1; t0 += stage) {
        const int nt = blk_t1 - t0 < stage ? blk_t1 - t0 : stage;
        for (int i = threadIdx.x; i < nt * 4; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int32_t it = first[u];
            const int t = (it >> 5) - t0;
            if ((uint32_t)it < (uint32_t)a.n_items && (uint32_t)t < (uint32_t)nt) atomicOr(&lds[t * 16 + u], 1u << (it & 31));
        }
        for (unsigned r = long_rows; r != 0u; r &= r - 1u) {      // (block-uniform)
            const int u = __builtin_ctz(r);
            const int32_t rb = __builtin_amdgcn_readlane(row_begin, u), re = __builtin_amdgcn_readlane(row_end, u);
            for (int e = rb + 256 + (int)threadIdx.x; e < re; e += 256) {
                const int32_t it = a.train_colidx[e];
                const int t = (it >> 5) - t0;
                if ((uint32_t)it < (uint32_t)a.n_items && (uint32_t)t < (uint32_t)nt) atomicOr(&lds[t * 16 + u], 1u << (it & 31));
            }
        }
        __syncthreads();
        const int c0 = wave_t0 > t0 ? wave_t0 : t0, c1 = wave_t1 < t0 + nt ? wave_t1 : t0 + nt;      // this wave's tiles of the pass
        for (int i = (c0 - t0) * 4 + lane; i < (c1 - t0) * 4; i += 64)
            reinterpret_cast<uint4*>(slice + (size_t)t0 * 16)[i] = reinterpret_cast<const uint4*>(lds)[i];
        __syncthreads();                                       // (the stage is reused - by the next pass or by the sweep's pools)
    }
    if (a.hdr && threadIdx.x == 0) atomicAdd(&a.hdr[2], 16u);     // (statistics: train rows swept as bitmaps)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wave's own stores only: no barrier, no L1 invalidate - it wrote what it will read)
}
// the words of rows 4 lq .. 4 lq + 3 for one item tile: 16 bytes, one load (plain: see tk_heavy_word)
__device__ __forceinline__ uint4 tk_rows_words(const TopkArgs& a, int lq, int64_t tile) {
    return *reinterpret_cast<const uint4*>(a.heavy_bm + ((size_t)blockIdx.x * a.heavy_words + tile) * 16 + lq * 4);
}
#ifndef TK_ROWS_BM_MIN_N
#define TK_ROWS_BM_MIN_N (16 * 12)
#endif
constexpr int TK_ROWS_BM_MIN = TK_ROWS_BM_MIN_N;   // train items of a block's 16 rows from which tk_rows_setup pays
constexpr int TK_CAP = 64;    // buffer slots per (wave, user): drained before a round could overflow it

// scores only (llmrec_scores_f32): S[q][item], same MFMA chain as the selection kernel
template <int DK>
__global__ __launch_bounds__(256) void scores_kernel(TopkArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int q0 = blockIdx.x * 16;
    if (q0 >= a.n_query) return;
    int qa = q0 + li;
    if (qa > a.n_query - 1) qa = a.n_query - 1;
    const int64_t user_a = a.query_users[qa];
    float4 ua[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c) ua[c] = ld4g(a.Eu + user_a * a.ldu, 16 * c + 4 * lq, a.d, a.vec_ok);
    const int64_t tiles_total = (a.n_items + 63) / 64;
    for (int64_t t = w; t < tiles_total; t += 4) {
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
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + lq * 4 + r;
                const int64_t item = base + 16 * n + li;
                if (q < a.n_query && item < a.n_items) a.S[(int64_t)q * a.lds + item] = acc[n][r];
            }
    }
}

// The sweep's B fragments are 16 rows x 64 bytes per load instruction when taken from the row-major table: 64 different cache lines per
// wave instruction, served one lane at a time by the CU's address path (round 2's ablation: 31 % of the sweep is fragment-load issue, and
// the exact-fp32 MFMA hides none of it). One pass over the (L2-resident) item table per call re-lays it in fragment order, so that every
// load of the sweep is 64 lanes x 16 contiguous bytes. Same values, same MFMA order: the scores do not change by a bit.
__global__ __launch_bounds__(256) void topk_pack_items_kernel(TopkArgs a, int DK, float4* __restrict__ packed, int64_t n_vec) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;     // ((tile * 2 + n) * DK + c) * 64 + lane
    if (v >= n_vec) return;
    const int lane = (int)(v & 63), li = lane & 15, lq = lane >> 4;
    const int64_t f = v >> 6;
    const int c = (int)(f % DK);
    const int64_t tn = f / DK;                                     // tile * 2 + n
    int64_t item = tn * 16 + li;
    if (item > a.n_items - 1) item = a.n_items - 1;                // the partial last tile repeats the last item (masked by the range check)
    packed[v] = ld4g(a.Ei + item * a.ldi, 16 * c + 4 * lq, a.d, a.vec_ok);
}

// FAST: d == 16 DK and 16-byte aligned rows - plain float4 loads. (A branch around a global load makes
// hipcc wait for each load before issuing the next: the guarded loader costs ~8 exposed L2 latencies per round.)
// PACKED: the item tiles come from a.packed (fragment order, one contiguous KB per load instruction).
template <int DK, bool FAST, bool PACKED>
__global__ __launch_bounds__(256, (DK <= 4 ? 4 : 2)) void score_topk_kernel(TopkArgs a) {
    __shared__ float buf_s[4][16][TK_CAP];
    __shared__ int32_t buf_i[4][16][TK_CAP];
    __shared__ __attribute__((aligned(16))) int32_t cnt_s[4][16];     // fill of buffer (wave, user): LDS atomics (the appends are lane-local)
    __shared__ float thr_s[16];
    __shared__ int32_t flag_s[2];
    __shared__ int32_t train_s[4][16][TK_TRAIN_STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t mask_s[4][16];   // train masks of an event round (zero between rounds)
    // readfirstlane: tells the compiler the wave index is uniform, so the per-wave round counters and
    // branches live in SGPRs / scalar branches instead of 64-bit VGPR arithmetic under exec masks
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    int tile = blockIdx.x, part = 0, n_parts = 1;
    if ((int)blockIdx.x >= a.split_from) {                     // block-uniform
        n_parts = a.n_parts;
        const int r_ = (int)blockIdx.x - a.split_from, nst_ = (a.n_query + 15) / 16 - a.split_from;
        tile = a.split_from + (a.part_major ? r_ % nst_ : r_ / n_parts);
        part = a.part_major ? r_ / nst_ : r_ % n_parts;
    }
    const int q0 = tile * 16;
    if (q0 >= a.n_query) return;                               // block-uniform
    if (a.only_flagged && a.only_flagged[tile] == 0u) return;  // (the second launch of the bf16 mode: only the tiles its verification flagged)
    if (threadIdx.x < 16) thr_s[threadIdx.x] = -INFINITY;
    if (threadIdx.x < 64) { (&cnt_s[0][0])[threadIdx.x] = 0; (&mask_s[0][0])[threadIdx.x] = 0u; }
    if (threadIdx.x < 2) flag_s[threadIdx.x] = 0;             // [0] drain requested, [1] waves that finished their quarter
    __syncthreads();

    // A operand: the block's 16 users
    int qa = q0 + li;
    if (qa > a.n_query - 1) qa = a.n_query - 1;
    const int64_t user_a = a.query_users[qa];
    float4 ua[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c)
        ua[c] = FAST ? *reinterpret_cast<const float4*>(a.Eu + user_a * a.ldu + 16 * c + 4 * lq)
                     : ld4g(a.Eu + user_a * a.ldu, 16 * c + 4 * lq, a.d, a.vec_ok);

    const int64_t tiles_all = (a.n_items + TK_TILE - 1) / TK_TILE;
    const int64_t part_begin = tiles_all * part / n_parts, tiles_total = tiles_all * (part + 1) / n_parts;   // this block's item tiles
    const int64_t tiles_per_wave = (tiles_total - part_begin + 3) / 4;
    const int64_t t_begin = part_begin + w * tiles_per_wave < tiles_total ? part_begin + w * tiles_per_wave : tiles_total;
    const int64_t t_end = t_begin + tiles_per_wave < tiles_total ? t_begin + tiles_per_wave : tiles_total;

    // lanes 0..15 own one user each for the train-row cursor: position of the first train item inside this
    // wave's quarter (binary search) and that item's id in a register - a round touches memory only when
    // it consumes a train item (a per-round peek would put a dependent global load, and a vmcnt(0) that
    // also waits for the prefetched tile, on every round's critical path).
    int32_t cur = 0, end = 0, nxt = INT_MAX;                           // cur: nxt's position in colidx; its LDS slot is cur % TK_TRAIN_STAGE
    int32_t row_begin = 0;
    if (lane < 16 && a.train_rowptr) {
        cur = a.train_rowptr[user_a]; end = a.train_rowptr[user_a + 1];
        row_begin = cur;
        const int64_t first = t_begin * TK_TILE;
        int32_t lo = cur, hi = end;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (a.train_colidx[mid] < first) lo = mid + 1; else hi = mid;
        }
        cur = lo;
        tk_train_stage(a.train_colidx, cur & ~(TK_TRAIN_STAGE - 1), end, &train_s[w][lane][0]);   // (the window is aligned in colidx positions:
        nxt = train_s[w][lane][cur & (TK_TRAIN_STAGE - 1)];                                       //  slots before cur are never read)
    }
    const int hoff = tk_heavy_setup(a, lane, q0 + lane < a.n_query, row_begin, end);               // >= 0: this lane's row is a bitmap
    if (hoff >= 0) nxt = INT_MAX;
    uint32_t hm = 0u;                                                                              // the bitmap word of the round's tile
    uint64_t lk[4];                                            // the block's lists of users 4 w + rr as 64-bit keys (score, ~item), slot = lane
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) lk[rr] = TK_KEY_EMPTY;

    // The waves of a block run decoupled: a wave that needs a drain (or has finished its quarter) raises
    // the flag / waits at the rendezvous, the others join at their next round boundary.
    const int64_t my_rounds = t_end > t_begin ? t_end - t_begin : 0;
    float4 b[2][DK];
    // this lane's two item rows of the current tile; full tiles advance the pointers by one tile stride,
    // only the sweep's single partial tile (the end of the item table) takes the clamped path
    const float* row0 = a.Ei + (t_begin * TK_TILE + li) * a.ldi;
    const float* row1 = row0 + 16 * a.ldi;
    const int64_t tile_stride = (int64_t)TK_TILE * a.ldi;
    const float4* pk = PACKED ? a.packed + (t_begin * 2 * DK) * 64 + lane : nullptr;
    auto load_tile = [&](int64_t t) {
        if (PACKED) {
#pragma unroll
            for (int c = 0; c < DK; ++c) { b[0][c] = pk[c * 64]; b[1][c] = pk[(DK + c) * 64]; }
            pk += 2 * DK * 64;
            return;
        }
        const float* r0 = row0; const float* r1 = row1;
        if ((t + 1) * TK_TILE > a.n_items) {                   // wave-uniform
            int64_t i0 = t * TK_TILE + li, i1 = i0 + 16;
            if (i0 > a.n_items - 1) i0 = a.n_items - 1;
            if (i1 > a.n_items - 1) i1 = a.n_items - 1;
            r0 = a.Ei + i0 * a.ldi; r1 = a.Ei + i1 * a.ldi;
        }
#pragma unroll
        for (int c = 0; c < DK; ++c) {
            b[0][c] = FAST ? *reinterpret_cast<const float4*>(r0 + 16 * c + 4 * lq) : ld4g(r0, 16 * c + 4 * lq, a.d, a.vec_ok);
            b[1][c] = FAST ? *reinterpret_cast<const float4*>(r1 + 16 * c + 4 * lq) : ld4g(r1, 16 * c + 4 * lq, a.d, a.vec_ok);
        }
        row0 += tile_stride; row1 += tile_stride;
    };
    if (my_rounds > 0) { load_tile(t_begin); if (hoff >= 0) hm = tk_heavy_word(a, hoff, t_begin); }
    int64_t round = 0;
    bool counted = false;
    for (;;) {
        const bool fin = round >= my_rounds;
        // a buffer must keep room for the next round's TK_TILE candidates: this wave's four counters of the lane's user group, straight from LDS
        const int4 c4 = *reinterpret_cast<const int4*>(&cnt_s[w][lq * 4]);
        const int fill = max(max(c4.x, c4.y), max(c4.z, c4.w));
        bool drain = fin || __ballot(fill > TK_CAP - TK_TILE) != 0ull;
        // (an atomic load, not a volatile one: volatile accesses to LDS are compiled as FLAT loads with a vmcnt(0)
        //  wait, which would also wait for the prefetched tile)
        if (!drain) drain = __hip_atomic_load(&flag_s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        if (drain) {                                           // wave-uniform; every wave of the block gets here
            if (fin && !counted) { counted = true; if (lane == 0) atomicAdd(&flag_s[1], 1); }
            if (!fin && lane == 0) __hip_atomic_store(&flag_s[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();                                   // (every wave's appends - LDS atomics and stores - are complete and visible)
            // snapshot of the finished-quarter count, taken BETWEEN the drain's barriers: every increment of this cycle
            // happened before the first barrier and the next cycle's cannot happen before the last one - reading it after
            // the last barrier would race with a faster wave that has already finished its final round
            const int done_quarters = flag_s[1];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int u = 4 * w + rr;
                const int p1 = cnt_s[0][u], p2 = p1 + cnt_s[1][u], p3 = p2 + cnt_s[2][u], total = p3 + cnt_s[3][u];
                uint64_t k1 = lk[rr];
                for (int j0 = 0; j0 < total; j0 += 64) {       // wave-uniform
                    const int j = j0 + lane;
                    const int ww = (j >= p1) + (j >= p2) + (j >= p3);
                    const int start = ww == 0 ? 0 : (ww == 1 ? p1 : (ww == 2 ? p2 : p3));
                    uint64_t bk = TK_KEY_EMPTY;
                    if (j < total) bk = tk_key(buf_s[ww][u][j - start], buf_i[ww][u][j - start]);
                    sort64k(bk, lane);
                    merge64k(k1, bk, lane);
                }
                lk[rr] = k1;
                if (total > 0) {
                    const float nthr = tk_unord((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k1 >> 32), a.K - 1));
                    if (lane == 0) thr_s[u] = nthr;
                }
            }
            __syncthreads();                                   // (every owner has read the counters)
            if (threadIdx.x < 64) (&cnt_s[0][0])[threadIdx.x] = 0;
            if (threadIdx.x == 0) flag_s[0] = 0;
            __syncthreads();
            if (done_quarters == 4) break;    // all four quarters swept and drained
            continue;
        }
        const int64_t base = (t_begin + round) * TK_TILE;
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
        for (int c = 0; c < DK; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(cmp4(ua[c], s), cmp4(b[n][c], s), acc[n], 0, 0, 0);
        const uint32_t hm_now = hm;                                    // (this tile's bitmap word: the prefetch below overwrites hm)
        // the next tile's operands go into the registers the MFMAs have just read: the loads fly during the selection (the train-mask
        // code below touches LDS only - a refill of the staged window, once per 16 items of a walked row, is the one wait)
        __builtin_amdgcn_sched_barrier(0);
        if (round + 1 < my_rounds) { load_tile(t_begin + round + 1); if (hoff >= 0) hm = tk_heavy_word(a, hoff, t_begin + round + 1); }
        __builtin_amdgcn_sched_barrier(0);
        // 32-bit mask of this tile's train items, by the row-owner lanes; the items come from the LDS stage (tk_train_stage)
        uint32_t m = hm_now;
        for (;;) {
            const int32_t base32 = (int32_t)base;
            bool need = false;
            while ((uint32_t)(nxt - base32) < (uint32_t)TK_TILE) {      // lanes >= 16 hold INT_MAX; nxt >= base (sorted rows)
                m |= 1u << (nxt - base32);
                if ((++cur & (TK_TRAIN_STAGE - 1)) == 0) { need = true; break; }    // window consumed
                nxt = train_s[w][lane & 15][cur & (TK_TRAIN_STAGE - 1)];
            }
            const unsigned nb = (unsigned)__ballot(need);
            if (nb == 0u) break;                                       // (wave-uniform; the usual case)
            tk_train_refill(a.train_colidx, nb, cur, end, train_s[w], lane);
            if (need) nxt = train_s[w][lane & 15][0];
        }

        // first level: one compare per score against the user's filter (the compare's lane mask IS the ballot);
        // range / train-mask checks only for the few (row, column-tile) pairs that have a candidate at all.
        // Rows past n_query carry a +inf filter.
        // event rounds only (wave-uniform): the owners' masks go through LDS - one 16-byte read gives a lane the masks of its four rows
        uint32_t rm4[4] = {0u, 0u, 0u, 0u};
        if (__ballot(m != 0u) != 0ull) {
            if (m != 0u) mask_s[w][lane & 15] = m;
            __builtin_amdgcn_wave_barrier();
            const uint4 t = *reinterpret_cast<const uint4*>(&mask_s[w][lq * 4]);
            rm4[0] = t.x; rm4[1] = t.y; rm4[2] = t.z; rm4[3] = t.w;
            __builtin_amdgcn_wave_barrier();
            if (m != 0u) mask_s[w][lane & 15] = 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rthr = (q0 + lq * 4 + r < a.n_query) ? thr_s[lq * 4 + r] : INFINITY;
            const uint32_t rm = rm4[r];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float v = acc[n][r];
                if (__ballot(v >= rthr) == 0ull) continue;
                const int col = 16 * n + li;
                const bool pass = (v >= rthr) && (base + col < a.n_items) && !((rm >> col) & 1u);
                if (pass) {                                    // lane-local append: a slot from the buffer's LDS counter (the order inside a
                    const int off = atomicAdd(&cnt_s[w][lq * 4 + r], 1);   // buffer is irrelevant: the drain sorts it by (score, item))
                    buf_s[w][lq * 4 + r][off] = v;
                    buf_i[w][lq * 4 + r][off] = (int32_t)(base + col);
                }
            }
        }
        ++round;
    }
    if (n_parts > 1) {                                         // a part's lists: all 64 slots, merged by topk_merge_kernel
        const int64_t base = ((int64_t)(tile - a.split_from) * n_parts + part) * 16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            a.ws_idx[(base + 4 * w + rr) * 64 + lane] = tk_key_id(lk[rr]);
            a.ws_score[(base + 4 * w + rr) * 64 + lane] = tk_key_ub(lk[rr]);
        }
        return;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int q = q0 + 4 * w + rr;
        if (q < a.n_query && lane < a.K) {
            const int32_t id = tk_key_id(lk[rr]);
            a.out_idx[(int64_t)q * a.K + lane] = id == INT_MAX ? -1 : id;
            a.out_score[(int64_t)q * a.K + lane] = tk_key_ub(lk[rr]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: bf16 SWEEP + EXACT VERIFICATION. The sweep above spends 1024 matrix cycles per 16 users x 32 items on
// the exact-fp32 MFMA. Here the sweep runs on v_mfma_f32_16x16x32_bf16 with both operands cut into two bf16 numbers (x = h + m + r,
// round-to-nearest splits: |r| <= 2^-16 |x|):
//     s' = <u_h, i_h> + <u_h, i_m> + <u_m, i_h>          (three MFMAs per 32 k: 192 matrix cycles per 16 x 32 tile at d = 64)
//     s - s' = <u_h, i_r> + <u_m, i_m> + <u_m, i_r> + <u_r, i>   =>   |s - s'| <= (3.02 x 2^-16 + fp32 accumulation of 192 terms) sum_k |u_k| |i_k|
//                                                                            <=  2^-14 ||u|| ||i||                     (Cauchy-Schwarz)
// so  ub(u, i) := s' + 2^-14 ||u|| ||i||  is an UPPER BOUND of the exact score, item by item (a cold item with a tiny embedding gets a tiny
// slack: thousands of near-identical cold rows do not blur the boundary the way one max-norm slack for all items did - the first form of the
// verification sent 34 % of the Netflix-shaped bench's user tiles to the exact sweep for exactly that reason).
// The sweep keeps, per user, the 64 best items BY ub (the exact kernel's buffers, drains, bitonic merges; the filter is the list's 64th ub).
// At the end the 64 kept items are scored EXACTLY - one lane per item runs the k-ordered fp32 fma chain of v_mfma_f32_16x16x4_f32 (for c, for
// s, for q: k = 16 c + 4 q + s; a VALU v_fma_f32 chain gives the MFMA's bits) - and re-ranked by (exact score desc, id asc).
// VERIFICATION: every item x outside the list has s(x) <= ub(x) <= ub_64. If ub_64 < e_K (the K-th exact score of the list) no outside item
// can reach the top K: the list's top K IS the exact top K, bit for bit what score_topk_kernel returns. Otherwise (more than 64 - K items
// within the slack of the boundary: massive exact ties) the user tile is FLAGGED and the exact sweep runs for the flagged tiles in a second
// launch (blocks of unflagged tiles exit at once). The first version of the mode rescored every candidate at every drain: bit-identical too,
// but 5 - 20 % SLOWER than the exact sweep (profiles/experiments/r05_topk.md).
// ---------------------------------------------------------------------------------------------------------------------------
typedef __bf16 tk_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 tk_mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tk_bf16x8, a), __builtin_bit_cast(tk_bf16x8, b), c, 0, 0, 0);
}
// round-to-nearest-even bf16 of a finite float, as a float (low 16 bits zero)
__device__ __forceinline__ uint32_t tk_rn_bf16_bits(float x) {
    const uint32_t b = __float_as_uint(x);
    return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
}
// two floats -> packed (hi, mid) bf16 pairs, round-to-nearest splits (h = RN(x); the residual x - h is exact in fp32; m = RN(x - h))
__device__ __forceinline__ void tk_split2(float x0, float x1, uint32_t& H, uint32_t& M) {
    const uint32_t h0 = tk_rn_bf16_bits(x0), h1 = tk_rn_bf16_bits(x1);
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const uint32_t m0 = tk_rn_bf16_bits(r0), m1 = tk_rn_bf16_bits(r1);
    H = __builtin_amdgcn_perm(h1, h0, 0x07060302u);                      // high halves: e0 -> low 16 bits, e1 -> high 16 bits
    M = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
}
__device__ __forceinline__ void tk_split8(const float (&x)[8], uint4& H, uint4& M) {
    tk_split2(x[0], x[1], H.x, M.x); tk_split2(x[2], x[3], H.y, M.y);
    tk_split2(x[4], x[5], H.z, M.z); tk_split2(x[6], x[7], H.w, M.w);
}
// ub = s' + slack ||u|| ||i||  (both norms rounded UP by 2^-10). slack >= 3.02 * 2^-16 (the dropped <u_m, i_m> term and the two split residuals)
// + n * 2^-24 (fp32 accumulation of the n = 3 d products): d <= 64 -> 3.02 + 0.75 < 4 = 2^-14 * 2^16; d <= 128 -> 3.02 + 1.5 > 4, so the
// bound doubles there (the d = 64 constant was used for every d).
__device__ __host__ __forceinline__ float tk_pre_slack(int d) { return d <= 64 ? 0x1p-14f : 0x1p-13f; }
constexpr float TK_NORM_UP = 1.0f + 0x1p-10f;

// workspace header of the mode (first 256 bytes of the fragment area): [0] drains of the bf16 sweep (all blocks), [1] user tiles flagged for the exact sweep, [2] train rows swept as bitmaps
// the item table as bf16 (hi, mid) MFMA fragments: pk2[(((tile * 2 + n) * DK32 + c) * 2 + hm) * 64 + lane], lane = 16 (k group) + item-in-tile;
// one thread per (item, 8 consecutive k). Also clears the header and the fallback flags (the norm kernel and the sweep follow on the stream).
__global__ __launch_bounds__(256) void topk_pack_items_bf16_kernel(TopkArgs a, int DK32, uint4* __restrict__ pk2) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < 3) a.hdr[v] = 0u;
    const int n_tiles_u = (a.n_query + 15) / 16;
    if (v < n_tiles_u) a.fb_word[v] = 0u;
    const int G = DK32 * 4;                                            // 8-float groups per (padded) row
    const int64_t n_pad = (a.n_items + TK_TILE - 1) / TK_TILE * TK_TILE;
    if (v >= n_pad * G) return;
    const int64_t item_p = v / G;
    const int g = (int)(v - item_p * G);
    const int64_t item = item_p < a.n_items ? item_p : a.n_items - 1;  // the partial last tile repeats the last item (masked by the range check)
    const float* row = a.Ei + item * a.ldi;
    float x[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int k = 8 * g + j; x[j] = k < a.d ? row[k] : 0.f; ss = fmaf(x[j], x[j], ss); }
    if (a.cn && (G & (G - 1)) == 0) {                                  // G = 4, 8, 16: an item's threads are G aligned neighbouring lanes - its
        for (int off = G >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);   // squared norm by a butterfly (fixed tree); G = 12: topk_item_norm_kernel
        if (g == 0) a.cn[item_p] = item_p < a.n_items ? tk_pre_slack(a.d) * (sqrtf(ss) * TK_NORM_UP) : __builtin_nanf("");   // (NaN: a padded slot's bound never passes a filter)
    }
    uint4 H, M;
    tk_split8(x, H, M);
    const int64_t tile = item_p / TK_TILE;
    const int within = (int)(item_p - tile * TK_TILE), n = within >> 4, li = within & 15;
    const int c = g >> 2, lq = g & 3;
    const int64_t base = (((tile * 2 + n) * DK32 + c) * 2) * 64 + (lq * 16 + li);
    pk2[base] = H; pk2[base + 64] = M;
}

// cn[item] = tk_pre_slack(d) * ||item|| rounded up: the item's factor of the score's upper bound. One 16-lane group per item; padded slots get NaN.
__global__ __launch_bounds__(256) void topk_item_norm_kernel(TopkArgs a, float* __restrict__ cn, int64_t n_pad) {
    const int gl = threadIdx.x & 15;
    for (int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); item < n_pad; item += (int64_t)gridDim.x * 16) {
        float ss = 0.f;
        if (item < a.n_items) {
            const float* row = a.Ei + item * a.ldi;
            for (int k = gl; k < a.d; k += 16) ss = fmaf(row[k], row[k], ss);
        }
        ss = group_sum<16>(ss);
        if (gl == 0) cn[item] = item < a.n_items ? tk_pre_slack(a.d) * (sqrtf(ss) * TK_NORM_UP) : __builtin_nanf("");
    }
}

// One user's 64 kept items (lane = slot, sorted by approximate score; id INT_MAX = empty slot) -> the exact ranking, the verification, the
// output. `urow` = the user's fp32 row (LDS or global). Wave-uniform control flow.
// BATCH: 16-column chunks of the item rows loaded together (64 B per lane and chunk). 1: one chunk at a time - a chain of d / 16 dependent memory round
// trips per user, but only 16 registers; 4: d <= 64 in ONE round trip (the sweep's tail: its fragment registers are dead by then - 28 us of a 0.243 ms
// sweep were these chains, four users per wave one after the other). ulen: length of urow (the staged LDS copy is zero-padded to whole chunks);
// u_vec: urow + 4 j is 16-byte aligned for every j.
template <int BATCH>
__device__ __forceinline__ void tk_finalize_user(const TopkArgs& a, const float* urow, int ulen, bool u_vec, int q, int tile, float ub, int32_t id, int lane) {
    const float ub64 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ub), 63));   // bounds every item outside the list (-inf: list not full)
    float e = -INFINITY;
    {
        const int64_t it = id == INT_MAX ? 0 : id;
        const float* ir = a.Ei + it * a.ldi;
        float acc = 0.f;
        const int nc = (a.d + 15) / 16;
#pragma unroll 1
        for (int c0 = 0; c0 < nc; c0 += BATCH) {                         // the k-ordered chain: c, then s, then q; k = 16 c + 4 q + s
            float4 iv[BATCH][4];
#pragma unroll
            for (int b = 0; b < BATCH; ++b)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) iv[b][qq] = ld4g(ir, 16 * (c0 + b) + 4 * qq, a.d, a.vec_ok);   // (zeros past d)
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                float4 uv[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) uv[qq] = ld4g(urow, 16 * (c0 + b) + 4 * qq, ulen, u_vec);
                if (c0 + b < nc) {                                       // (wave-uniform; chunks past d stay out of the chain: fma(0, 0, -0.0f) is +0.0f)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) acc = __builtin_fmaf(cmp4(uv[qq], s_), cmp4(iv[b][qq], s_), acc);
                }
            }
        }
        if (id != INT_MAX) e = acc;
    }
    uint64_t ek = tk_key(e, id);                                      // (exact score desc, item id asc): the exact sweep's order
    sort64k(ek, lane);
    const float es = tk_key_ub(ek);
    const int32_t ei = tk_key_id(ek);
    const float eK = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(es), a.K - 1));
    const bool ok = (ub64 == -INFINITY) || (ub64 < eK);               // (NaN anywhere -> not ok -> the exact sweep decides)
    if (q < a.n_query && lane < a.K) {
        a.out_idx[(int64_t)q * a.K + lane] = ei == INT_MAX ? -1 : ei;
        a.out_score[(int64_t)q * a.K + lane] = es;
    }
    if (!ok && q < a.n_query && lane == 0) {
        if (atomicExch(reinterpret_cast<unsigned int*>(&a.fb_word[tile]), 1u) == 0u) atomicAdd(&a.hdr[1], 1u);
    }
}

// ---- the bf16 sweep's candidate pools ----
// One UNSORTED pool of TK_POOL (ub, item) entries per user, shared by the block's four wavefronts (slots from one LDS counter per user). A round appends
// every score that reaches the user's filter. When some pool may not hold another round of all four waves (4 x TK_TILE appends), the block
// drains: the owner wave of a user finds the 64th largest ub of the pool by BISECTION over the 32 bits of the order-preserving key - one compare per
// held entry and bit, the counts from the compares' lane masks on the scalar unit - keeps the 64 entries at or above it (entries above it first, then
// entries equal to it in slot order: which of several EQUAL bounds stays cannot change the output, the list is re-ranked by exact score and verified against the
// bound), and that value is the user's new filter. ~170 vector instructions per user and drain whatever the pool holds - the sorted lists of
// rounds 2 - 5 (bitonic sort of every 64 buffered entries + merge into a register list: ~200 instructions per 64 entries, ~12 x per user and sweep) were
// ~40 % of the sweep's vector instructions. The ONE sort per user happens after the sweep, on the 64 survivors.
constexpr int TK_POOL = 256;

// A separator of the 64 largest of the wave's 4 x 64 keys e[j] (0 = no entry, below every key; every entry >= lo, the user's filter so far; at least 64
// entries): the largest P the search reaches with #{e >= P} >= 64 - it stops at the first P with EXACTLY 64 keys at or above it (P is then a lower bound
// of the 64th largest key, which is all a filter needs), else it ends at the 64th largest key itself (ties). The bits above the highest bit in which the
// largest key differs from lo are common to every key and skipped: ~10 steps instead of 32 on scores of one binade. Wave-uniform.
template <int NJ>                                              // NJ: registers of e that hold entries (the pool's fill / 64, rounded up)
__device__ __forceinline__ uint32_t tk_select64(const uint32_t (&e)[4], uint32_t lo) {
    uint32_t hi = max(e[0], e[1]);
    if (NJ > 2) hi = max(hi, e[2]);
    if (NJ > 3) hi = max(hi, e[3]);
    hi = max(hi, (uint32_t)xor_lane_i<1>((int)hi, 0)); hi = max(hi, (uint32_t)xor_lane_i<2>((int)hi, 0));
    hi = max(hi, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x124, 0xf, 0xf, true));      // row_ror:4
    hi = max(hi, (uint32_t)xor_lane_i<8>((int)hi, 0));
    hi = max(hi, (uint32_t)xor_lane_i<16>((int)hi, 0));
    hi = max((uint32_t)__builtin_amdgcn_readlane((int)hi, 0), (uint32_t)__builtin_amdgcn_readlane((int)hi, 32));
    const uint32_t diff = hi ^ lo;
    if (diff == 0u) return hi;                                 // every key equal
    const int top = 31 - __builtin_clz(diff);
    uint32_t P = top == 31 ? 0u : hi & ~((2u << top) - 1u);
#pragma unroll 1
    for (int b = top; b >= 0; --b) {
        const uint32_t c = P | (1u << b);
        int n = __popcll(__ballot(e[0] >= c)) + __popcll(__ballot(e[1] >= c));
        if (NJ > 2) n += __popcll(__ballot(e[2] >= c));
        if (NJ > 3) n += __popcll(__ballot(e[3] >= c));
        P = n >= 64 ? c : P;
        if (n == 64) break;
    }
    return P;
}
__device__ __forceinline__ int tk_mbcnt(uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

template <int DK32>
__global__ __launch_bounds__(256, (DK32 <= 2 ? 4 : 2)) void score_topk_pre_kernel(TopkArgs a, const uint4* __restrict__ pk2, const float* __restrict__ cn) {
    __shared__ __attribute__((aligned(16))) float2 pool[16][TK_POOL];   // the user's candidates: (upper bound ub, item id as bits) - one 8-byte store per append
    __shared__ __attribute__((aligned(16))) int32_t cnt_s[16];        // fill of the user's pool: LDS atomics (the appends are lane-local)
    __shared__ __attribute__((aligned(16))) float thr_s[16];          // the filter: the 64th ub of the user's pool at the last drain (+inf: no such user)
    __shared__ int32_t flag_s[1];                                      // waves that finished their quarter
    __shared__ int32_t train_s[4][16][TK_TRAIN_STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t mask_s[4][16];   // train masks of an event round (zero between rounds)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    int tile = blockIdx.x, part = 0, n_parts = 1;
    if ((int)blockIdx.x >= a.split_from) {
        n_parts = a.n_parts;
        const int r_ = (int)blockIdx.x - a.split_from, nst_ = (a.n_query + 15) / 16 - a.split_from;
        tile = a.split_from + (a.part_major ? r_ % nst_ : r_ / n_parts);
        part = a.part_major ? r_ / nst_ : r_ % n_parts;
    }
    const int q0 = tile * 16;
    if (q0 >= a.n_query) return;
    if (threadIdx.x < 16) { thr_s[threadIdx.x] = q0 + (int)threadIdx.x < a.n_query ? -INFINITY : INFINITY; cnt_s[threadIdx.x] = 0; }
    if (threadIdx.x < 64) (&mask_s[0][0])[threadIdx.x] = 0u;
    if (threadIdx.x == 0) flag_s[0] = 0;
    __syncthreads();

    // A operand: row li of the block's users, k = 32 c + 8 lq + j -> (hi, mid) fragments
    int qa = q0 + li;
    if (qa > a.n_query - 1) qa = a.n_query - 1;
    const int64_t user_a = a.query_users[qa];
    uint4 uH[DK32], uM[DK32];
    float un2 = 0.f;
#pragma unroll
    for (int c = 0; c < DK32; ++c) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 32 * c + 8 * lq + j; x[j] = k < a.d ? a.Eu[user_a * a.ldu + k] : 0.f; un2 = fmaf(x[j], x[j], un2); }
        tk_split8(x, uH[c], uM[c]);
    }
    un2 += __shfl_xor(un2, 16, 64); un2 += __shfl_xor(un2, 32, 64);    // ||u||^2 of row li, in every lane with that li
    const float un_row = sqrtf(un2) * TK_NORM_UP;
    float unr[4];                                                      // ||u|| (rounded up) of the four rows this lane's accumulators belong to
#pragma unroll
    for (int r = 0; r < 4; ++r) unr[r] = __shfl(un_row, lq * 4 + r, 64);

    const int64_t tiles_all = (a.n_items + TK_TILE - 1) / TK_TILE;
    const int64_t part_begin = tiles_all * part / n_parts, tiles_total = tiles_all * (part + 1) / n_parts;
    const int64_t tiles_per_wave = (tiles_total - part_begin + 3) / 4;
    const int64_t t_begin = part_begin + w * tiles_per_wave < tiles_total ? part_begin + w * tiles_per_wave : tiles_total;
    const int64_t t_end = t_begin + tiles_per_wave < tiles_total ? t_begin + tiles_per_wave : tiles_total;

    int32_t cur = 0, end = 0, nxt = INT_MAX;                           // cur: nxt's position in colidx; its LDS slot is cur % TK_TRAIN_STAGE
    int32_t row_begin = 0;
    if (lane < 16 && a.train_rowptr) {
        cur = a.train_rowptr[user_a]; end = a.train_rowptr[user_a + 1];
        row_begin = cur;
    }
    // (block-uniform: every wave holds the same 16 rows) every train row of the block as a bitmap, no row walked - when the rows are long enough to pay
    // for the slice's set-up and the two loads per round: measured at the Netflix shape, 4.2 train items per user: walking 0.283 ms, bitmaps 0.296; ~50 per
    // user: 0.381 / 0.321. (All 16 rows as bitmaps whenever ONE row is long, instead of the two-row form beside walked rows: 0.2645 against 0.262 - no gain.)
    int row_items = lane < 16 && q0 + lane < a.n_query ? end - row_begin : 0;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) row_items += __shfl_xor(row_items, off, 64);
    const bool all_bm = a.heavy_all != 0 && __builtin_amdgcn_readfirstlane(row_items) > TK_ROWS_BM_MIN;
    if (lane < 16 && a.train_rowptr && !all_bm) {
        const int64_t first = t_begin * TK_TILE;
        int32_t lo = cur, hi = end;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (a.train_colidx[mid] < first) lo = mid + 1; else hi = mid;
        }
        cur = lo;
        tk_train_stage(a.train_colidx, cur & ~(TK_TRAIN_STAGE - 1), end, &train_s[w][lane][0]);   // (the window is aligned in colidx positions:
        nxt = train_s[w][lane][cur & (TK_TRAIN_STAGE - 1)];                                       //  slots before cur are never read)
    }
    int hoff = -1;
    if (all_bm) tk_rows_setup(a, lane, q0 + lane < a.n_query, row_begin, end, reinterpret_cast<uint32_t*>(&pool[0][0]), 16 * TK_POOL * 2,
                              (int)part_begin, (int)tiles_total, (int)t_begin, (int)t_end);
    else hoff = tk_heavy_setup(a, lane, q0 + lane < a.n_query, row_begin, end);                    // >= 0: this lane's row is a bitmap
    if (hoff >= 0) nxt = INT_MAX;
    uint4 rmw = make_uint4(0u, 0u, 0u, 0u);                                                        // all_bm: the words of the lane's four rows, one round ahead
    uint32_t hm = 0u;                                                                              // the bitmap word of the round's tile

    const int64_t my_rounds = t_end > t_begin ? t_end - t_begin : 0;
    uint4 bH[2][DK32], bM[2][DK32];
    float cnv[2];                                                      // the slack factors of this lane's two items of the tile
    const uint4* pk = pk2 + (t_begin * 2 * DK32 * 2) * 64 + lane;
    const float* cnp = cn + t_begin * TK_TILE + li;
    auto load_tile = [&]() {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int c = 0; c < DK32; ++c) { bH[n][c] = pk[((n * DK32 + c) * 2) * 64]; bM[n][c] = pk[((n * DK32 + c) * 2 + 1) * 64]; }
        cnv[0] = cnp[0]; cnv[1] = cnp[16];
        pk += 2 * DK32 * 2 * 64; cnp += TK_TILE;
    };
    if (my_rounds > 0) { load_tile(); if (hoff >= 0) hm = tk_heavy_word(a, hoff, t_begin); if (all_bm) rmw = tk_rows_words(a, lq, t_begin); }
    // One drain of the block's pools (every wave of the block calls it at the same point of its program: a barrier on each side). Returns the
    // number of waves that have finished their quarter of the items.
    auto drain_pools = [&]() -> int {
        __syncthreads();                                   // (every wave's appends - LDS atomics and stores - are complete and visible)
        const int done_quarters = flag_s[0];
        if (a.hdr && threadIdx.x == 0) atomicAdd(&a.hdr[0], 1u);   // (statistics: drains, over all blocks)
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int u = 4 * w + rr;
            const int n = __builtin_amdgcn_readfirstlane(cnt_s[u]);
            if (n <= 64) continue;                         // (wave-uniform) nothing to drop yet / nothing new since the last drain
            // (wave-uniform) the pool's entries in NJ registers per lane, slot = lane + 64 j: most drains find 130 - 190 entries, i.e. three
            auto drain_user = [&](auto nj_tag) -> uint32_t {
                constexpr int NJ = decltype(nj_tag)::value;
                int32_t id[4]; uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int idx = lane + 64 * j;
                    const float2 en = pool[u][idx];
                    id[j] = __float_as_int(en.y);
                    e[j] = idx < n ? tk_ord(en.x) : 0u;        // (every held key is >= tk_ord(-inf) > 0)
                }
                const uint32_t P = tk_select64<NJ>(e, tk_ord(thr_s[u]));
                // (LDS operations of one wave are performed in order: every slot was read above)
                int base = 0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const uint64_t mg = __ballot(e[j] > P);
                    if (e[j] > P) { const int pos = base + tk_mbcnt(mg); pool[u][pos] = make_float2(tk_unord(e[j]), __int_as_float(id[j])); }
                    base += __popcll(mg);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (base >= 64) break;                     // (wave-uniform)
                    const uint64_t me = __ballot(e[j] == P);
                    const int pos = base + tk_mbcnt(me);
                    if (e[j] == P && pos < 64) { pool[u][pos] = make_float2(tk_unord(P), __int_as_float(id[j])); }
                    base += __popcll(me);
                }
                return P;
            };
            const uint32_t P = n <= 128 ? drain_user(std::integral_constant<int, 2>()) : n <= 192 ? drain_user(std::integral_constant<int, 3>())
                                                                                                  : drain_user(std::integral_constant<int, 4>());
            if (lane == 0) { cnt_s[u] = 64; thr_s[u] = tk_unord(P); }
        }
        __syncthreads();
        return done_quarters;
    };
    const int n_rounds = (int)my_rounds;
    for (int round = 0; round < n_rounds; ++round) {
        // the round's view of the block state, ONE LDS round trip: the fills of the pools of the lane's user group (a pool must keep room for the next
        // round's 4 x TK_TILE candidates) and the four rows' filters (they change at drains only). The pools and their counters are shared by the four
        // waves and the counters only grow between drains: a wave that finds a pool too full waits at the barrier, and every other wave finds the
        // same at the top of its next round - no drain-request flag (the per-wave buffers of rounds 2 - 5 needed one, and a second LDS round trip).
        const int4 c4 = *reinterpret_cast<const int4*>(&cnt_s[lq * 4]);
        float4 t4 = *reinterpret_cast<const float4*>(&thr_s[lq * 4]);
        asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w));      // (both reads in flight together: the compiler sank this one to its use - a second round trip)
        const int fill = max(max(c4.x, c4.y), max(c4.z, c4.w));
        if (__ballot(fill > TK_POOL - 4 * TK_TILE) != 0ull) {
            drain_pools();
            t4 = *reinterpret_cast<const float4*>(&thr_s[lq * 4]);
        }
        const int32_t base = ((int32_t)t_begin + round) * TK_TILE;      // (item ids are int32)
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
        for (int c = 0; c < DK32; ++c)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[n] = tk_mfma_bf16(uM[c], bH[n][c], acc[n]);
                acc[n] = tk_mfma_bf16(uH[c], bM[n][c], acc[n]);
                acc[n] = tk_mfma_bf16(uH[c], bH[n][c], acc[n]);
            }
        const float cn_now[2] = {cnv[0], cnv[1]};                       // (this tile's factors: the prefetch below overwrites cnv)
        const uint32_t hm_now = hm;
        const uint4 rmw_now = rmw;
        __builtin_amdgcn_sched_barrier(0);
        if (round + 1 < n_rounds) { load_tile(); if (hoff >= 0) hm = tk_heavy_word(a, hoff, t_begin + round + 1); if (all_bm) rmw = tk_rows_words(a, lq, t_begin + round + 1); }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t rm4[4] = {rmw_now.x, rmw_now.y, rmw_now.z, rmw_now.w};
        // rows that are walked: only in rounds whose tile holds a train item of some row (or a set bitmap word) - wave-uniform, one compare otherwise
        if (!all_bm && __ballot((uint32_t)(nxt - base) < (uint32_t)TK_TILE || hm_now != 0u) != 0ull) {
            __builtin_amdgcn_sched_barrier(0);
            // 32-bit mask of this tile's train items, by the row-owner lanes; the items come from the LDS stage (tk_train_stage)
            uint32_t m = hm_now;
            for (;;) {
                bool need = false;
                while ((uint32_t)(nxt - base) < (uint32_t)TK_TILE) {        // lanes >= 16 hold INT_MAX; nxt >= base (sorted rows)
                    m |= 1u << (nxt - base);
                    if ((++cur & (TK_TRAIN_STAGE - 1)) == 0) { need = true; break; }    // window consumed
                    nxt = train_s[w][lane & 15][cur & (TK_TRAIN_STAGE - 1)];
                }
                const unsigned nb = (unsigned)__ballot(need);
                if (nb == 0u) break;                                       // (wave-uniform; the usual case)
                tk_train_refill(a.train_colidx, nb, cur, end, train_s[w], lane);
                if (need) nxt = train_s[w][lane & 15][0];
            }
            // the owners' masks go through LDS - one 16-byte read gives a lane the masks of its four rows
            if (m != 0u) mask_s[w][lane & 15] = m;
            __builtin_amdgcn_wave_barrier();
            const uint4 t = *reinterpret_cast<const uint4*>(&mask_s[w][lq * 4]);
            rm4[0] = t.x; rm4[1] = t.y; rm4[2] = t.z; rm4[3] = t.w;
            __builtin_amdgcn_wave_barrier();
            if (m != 0u) mask_s[w][lane & 15] = 0u;
        }
        // MASKED = false: no row of the wave has a train item in this tile (the usual round of sparse rows) - the append skips the mask test
        auto select_round = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rthr = cmp4(t4, r);
                const uint32_t rm = rm4[r];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float v = fmaf(unr[r], cn_now[n], acc[n][r]);    // ub = s' + 2^-14 ||u|| ||i||
                    // the common case is ONE compare and one scalar branch: the sweep is bound by the instructions it issues (~1 per 4 cycles and SIMD,
                    // every type counted: profiles/experiments/r06_topk_pool.md), and the compiler folded the train-mask test and the exec juggling of
                    // the append into the straight line (11 instructions per (row, column tile) instead of 3) until the scheduling barrier pinned them here
                    if (__ballot(v >= rthr) == 0ull) continue;
                    __builtin_amdgcn_sched_barrier(0);
                    const int col = 16 * n + li;
                    const bool pass = (v >= rthr) && !(MASKED && ((rm >> col) & 1u));   // (slots past the table's end: cn = NaN, so v is NaN and never passes)
                    if (pass) {                                // lane-local append: a slot from the pool's LDS counter (the order inside a
                        const int off = atomicAdd(&cnt_s[lq * 4 + r], 1);  // pool is irrelevant)
                        pool[lq * 4 + r][off] = make_float2(v, __int_as_float(base + col));
                    }
                }
            }
        };
        if (__ballot((rm4[0] | rm4[1] | rm4[2] | rm4[3]) != 0u) != 0ull) select_round(std::true_type());
        else select_round(std::false_type());
    }
    // this wave's quarter is swept: it keeps draining with the block until all four are (the last drain leaves <= 64 entries per pool)
    if (lane == 0) atomicAdd(&flag_s[0], 1);
    while (drain_pools() != 4) {}
    // the users' survivors (<= 64 each: the last drain ran after every wave's last round) -> ONE sorted list per user, slot = lane, as 64-bit keys
    uint64_t kk[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int u = 4 * w + rr;
        const int n = cnt_s[u];
        const float2 en = pool[u][lane];
        kk[rr] = lane < n ? tk_key(en.x, __float_as_int(en.y)) : TK_KEY_EMPTY;
        sort64k(kk[rr], lane);
    }
    if (n_parts > 1) {                                         // a part's lists (approximate scores): merged and finalised by topk_merge_pre_kernel
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = ((int64_t)(tile - a.split_from) * n_parts + part) * 16 + 4 * w + rr;
            a.ws_idx[row * 64 + lane] = tk_key_id(kk[rr]);
            a.ws_score[row * 64 + lane] = tk_key_ub(kk[rr]);
        }
        return;
    }
    // exact re-ranking + verification. The wave's four user rows are staged in LDS (the pools are read: their area is free once every wave is here), so
    // the item rows of a user's 64 candidates can be loaded four 16-column chunks at a time without the user's row taking registers beside them
    __syncthreads();
    float* const urows = reinterpret_cast<float*>(&pool[0][0]);
    const int ulen = ((a.d + 15) / 16) * 16;                   // <= 128 floats per row
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        int qq = q0 + 4 * w + rr;
        if (qq > a.n_query - 1) qq = a.n_query - 1;
        const float* src = a.Eu + a.query_users[qq] * a.ldu;
        for (int k0 = lane; k0 < ulen; k0 += 64) urows[(4 * w + rr) * ulen + k0] = k0 < a.d ? src[k0] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();                           // (this wave's rows only: LDS operations of one wave are performed in order)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int q = q0 + 4 * w + rr;
        if (q >= a.n_query) continue;                          // wave-uniform
        tk_finalize_user<4>(a, urows + (4 * w + rr) * ulen, ulen, true, q, tile, tk_key_ub(kk[rr]), tk_key_id(kk[rr]), lane);
    }
}

// the lists of a split tile's parts (by upper bound) -> the tile's 64 best by upper bound -> exact ranking + verification.
// 16 wavefronts per block, ONE user each: the exact re-ranking is a chain of dependent L2 round trips per user, and this launch runs alone
// behind the sweep (four users per wave in sequence made it 33 us for 57 tiles).
__global__ __launch_bounds__(1024) void topk_merge_pre_kernel(TopkArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;     // w = the user's slot in the tile (0..15)
    const int tile = a.split_from + blockIdx.x;
    const int q = tile * 16 + w;
    if (q >= a.n_query) return;                                // wave-uniform
    uint64_t k1 = TK_KEY_EMPTY;
    for (int p = 0; p < a.n_parts; ++p) {
        const int64_t row = ((int64_t)blockIdx.x * a.n_parts + p) * 16 + w;
        merge64k(k1, tk_key(a.ws_score[row * 64 + lane], a.ws_idx[row * 64 + lane]), lane);
    }
    tk_finalize_user<4>(a, a.Eu + a.query_users[q] * a.ldu, a.d, a.vec_ok != 0, q, tile, tk_key_ub(k1), tk_key_id(k1), lane);
}

// the lists of a split tile's parts -> the tile's top K (one wave per four users, as in the sweep)
__global__ __launch_bounds__(256) void topk_merge_kernel(TopkArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tile = a.split_from + blockIdx.x;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int q = tile * 16 + 4 * w + rr;
        if (q >= a.n_query) continue;                          // wave-uniform
        uint64_t k1 = TK_KEY_EMPTY;
        for (int p = 0; p < a.n_parts; ++p) {
            const int64_t row = ((int64_t)blockIdx.x * a.n_parts + p) * 16 + 4 * w + rr;
            merge64k(k1, tk_key(a.ws_score[row * 64 + lane], a.ws_idx[row * 64 + lane]), lane);
        }
        if (lane < a.K) {
            const int32_t id = tk_key_id(k1);
            a.out_idx[(int64_t)q * a.K + lane] = id == INT_MAX ? -1 : id;
            a.out_score[(int64_t)q * a.K + lane] = tk_key_ub(k1);
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

// R10: per-user precision / recall / ndcg / hit-ratio at every K of Ks from the hit matrix of the ranked lists
// (reference utility/metrics.py:8-18,43-87 via batch_test.py:70-80), in double like numpy. One thread per user.
// ndcg: IDCG from the retrieved hit vector itself, sorted descending (the reference's definition): the ideal
// list is min(#hits in the top-Kmax list, K) ones. precision: mean over the ranked items that exist (a user with
// fewer than K candidates has a shorter list). out: [n_query][4][n_ks] = precision, recall, ndcg, hit_ratio.
struct MetricKs { int32_t k[8]; int32_t n; };
__global__ void topk_metrics_kernel(int n_query, const int64_t* __restrict__ query_users, int K,
                                    const uint8_t* __restrict__ hits, const int32_t* __restrict__ topk_idx,
                                    const int32_t* __restrict__ test_rowptr, MetricKs ks, double* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_query) return;
    const int64_t u = query_users[q];
    const double n_pos = (double)(test_rowptr[u + 1] - test_rowptr[u]);
    const uint8_t* h = hits + (int64_t)q * K;
    const int32_t* id = topk_idx + (int64_t)q * K;
    int total_hits = 0, list_len = 0;
    for (int j = 0; j < K; ++j) { total_hits += h[j]; list_len += id[j] >= 0; }
    double* o = out + (int64_t)q * 4 * ks.n;
    for (int t = 0; t < ks.n; ++t) {
        const int kk = ks.k[t] < K ? ks.k[t] : K;
        double s = 0.0, dcg = 0.0, idcg = 0.0;
        for (int j = 0; j < kk; ++j) {
            const double disc = 1.0 / log2((double)(j + 2));
            if (h[j]) { s += 1.0; dcg += disc; }
            if (j < total_hits) idcg += disc;
        }
        const int denom = list_len < kk ? list_len : kk;
        o[0 * ks.n + t] = s / (double)(denom > 0 ? denom : 1);
        o[1 * ks.n + t] = n_pos > 0.0 ? s / n_pos : 0.0;
        o[2 * ks.n + t] = idcg > 0.0 ? dcg / idcg : 0.0;
        o[3 * ks.n + t] = s > 0.0 ? 1.0 : 0.0;
    }
}

// R10 in two launches for a whole evaluation (round 6): hits + per-user metrics + their sums over the users, nothing per user leaves the
// kernel. Launch 1: one thread per user (the binary searches of topk_hits_kernel, the arithmetic of topk_metrics_kernel, in double), then the
// block's 128 users are added by a fixed pairwise tree in LDS -> partial[block][4 n_ks]. Launch 2: one block adds the partials in block order
// (thread t: partial[t], partial[t + 256], ...; then the pairwise tree) -> out[4 n_ks], which may be mapped host memory. Deterministic.
constexpr int ES_THREADS = 128;
__global__ __launch_bounds__(ES_THREADS) void topk_eval_sums_kernel(int n_query, const int64_t* __restrict__ query_users, int K,
                                                                    const int32_t* __restrict__ topk_idx, const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ colidx, MetricKs ks, double* __restrict__ partial) {
    __shared__ double red[ES_THREADS];
    const int q = blockIdx.x * ES_THREADS + threadIdx.x;
    double val[4][8];                                                  // [metric][cut-off], statically indexed (registers)
    const int nv = 4 * ks.n;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 8; ++t) val[m][t] = 0.0;
    if (q < n_query) {
        const int64_t u = query_users[q];
        const int32_t r0 = rowptr[u], r1 = rowptr[u + 1];
        const double n_pos = (double)(r1 - r0);
        const int32_t* id = topk_idx + (int64_t)q * K;
        unsigned long long hit_lo = 0ull, hit_hi = 0ull;              // K <= 128 hit flags
        int total_hits = 0, list_len = 0;
        for (int j = 0; j < K; ++j) {
            const int32_t item = id[j];
            int32_t lo = r0, hi = r1;
            while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (colidx[mid] < item) lo = mid + 1; else hi = mid; }
            const bool h = item >= 0 && lo < r1 && colidx[lo] == item;
            if (h) { if (j < 64) hit_lo |= 1ull << j; else hit_hi |= 1ull << (j - 64); }
            total_hits += h; list_len += item >= 0;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < ks.n) {
                const int kk = ks.k[t] < K ? ks.k[t] : K;
                double s = 0.0, dcg = 0.0, idcg = 0.0;
                for (int j = 0; j < kk; ++j) {
                    const double disc = 1.0 / log2((double)(j + 2));
                    const bool h = j < 64 ? ((hit_lo >> j) & 1ull) != 0ull : ((hit_hi >> (j - 64)) & 1ull) != 0ull;
                    if (h) { s += 1.0; dcg += disc; }
                    if (j < total_hits) idcg += disc;
                }
                const int denom = list_len < kk ? list_len : kk;
                val[0][t] = s / (double)(denom > 0 ? denom : 1);
                val[1][t] = n_pos > 0.0 ? s / n_pos : 0.0;
                val[2][t] = idcg > 0.0 ? dcg / idcg : 0.0;
                val[3][t] = s > 0.0 ? 1.0 : 0.0;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < ks.n) {                                            // (block-uniform)
                red[threadIdx.x] = val[m][t];
                __syncthreads();
                for (int off = ES_THREADS / 2; off > 0; off >>= 1) {
                    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                    __syncthreads();
                }
                if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * nv + m * ks.n + t] = red[0];
                __syncthreads();
            }
        }
    }
}
__global__ __launch_bounds__(256) void topk_eval_sums_finish_kernel(int n_blocks, int nv, const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double red[256];
    for (int i = 0; i < nv; ++i) {
        double s = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += 256) s += partial[(int64_t)b * nv + i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[i] = red[0];
        __syncthreads();
    }
}

static int device_cus() {
    static const int n = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
                              return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }();
    return n;
}
// The user tiles beyond the last full round of one tile per CU leave most CUs one block short (825 tiles on 256 CUs: 57 CUs
// sweep four blocks, 199 three, and the kernel lasts as long as the four). Those left-over tiles are cut into parts along
// the items, one block per part, so that every CU gets at most one short extra block.
static void plan_split(int n_query, int64_t n_items, int* split_from, int* n_parts) {
    const int n_tiles = (int)ceil_div(n_query, 16), cus = device_cus();
    const int full = n_tiles / cus * cus, left = n_tiles - full;
    int parts = left > 0 ? cus / left : 1;
    if (parts > 8) parts = 8;
    const int64_t item_tiles = (n_items + TK_TILE - 1) / TK_TILE;
    while (parts > 1 && item_tiles / parts < 64) --parts;      // at least 16 rounds per wave of a part
    *split_from = parts > 1 ? full : n_tiles; *n_parts = parts > 1 ? parts : 1;
}
// ITEM PARTS for tables beyond the L2 (round 6; bf16 sweep only). At 10^6 items every 16-user tile streamed the whole
// fragment table (256 MB at d = 64) through the L2-miss path: 4096 tiles x 256 MB at the ~7.4 TB/s of that path = 0.145 s, slower than the
// exact sweep. Here EVERY user tile is cut into parts of TK_PART_ITEMS items - a part's fragments fit the 4 MB L2 of an XCD - and the block
// ids are PART-MAJOR: the ~1000 blocks resident at any moment sweep the same part, so one block's fetch serves the others on its XCD from
// the L2 instead of 4096 fetches from the fabric. The parts' 64-slot lists are merged, re-scored exactly and verified by
// topk_merge_pre_kernel, as for the left-over tiles of the small-table plan. g_part_items: 0 = the policy below; tools set it
// (llmrec_topk_set_part_items) to sweep the part size.
constexpr int64_t TK_PARTS_FROM_ITEMS = 131072;                // tables up to here: the small-table plan (L2 / MALL-resident fragments)
constexpr int64_t TK_PART_BYTES = 4 << 20;                      // fragments per part: one XCD's L2. Measured with the pool sweep (65 536 users x 10^6 items, d = 64;
constexpr int TK_PART_ITEMS_MIN = 8192;                         // exact sweep 112.3 ms, bf16 sweep without parts 129.2): parts of 8 192 / 16 384 / 32 768 / 65 536 items
static int g_part_items = 0;                                    // 59.5 / 44.4 / 58.3 / 98.2 ms; d = 128, 16 384 users (exact 51.1, no parts 60.7): 23.7 / 31.0 / 50.8 / 61.2.
                                                                // (The sorted-list sweep of rounds 2 - 5 had its optimum at 8 MB: 76.9 / 56.7 / 50.7 / 91.8 and 27.8 / 29.8 / 50.2 / 61.9.)
// d < 0: the smallest part any width gets (workspace sizing: the layout must not depend on d)
static bool plan_parts(int n_query, int64_t n_items, int d, int* n_parts) {
    if (g_part_items < 0 || (g_part_items == 0 && n_items <= TK_PARTS_FROM_ITEMS)) return false;
    int64_t per = g_part_items;
    if (per == 0) {
        per = d < 0 ? TK_PART_ITEMS_MIN : TK_PART_BYTES / ((int64_t)ceil_div(d, 32) * 32 * 4);      // hi + mid bf16 = 4 bytes per (padded) column
        int64_t p2 = TK_PART_ITEMS_MIN;
        while (p2 * 2 <= per && p2 < 65536) p2 *= 2;
        per = p2;
    }
    const int64_t parts = ceil_div(n_items, per);
    if (parts < 2 || parts > 4096 || ceil_div(n_query, 16) * parts > 0x7fffffffll) return false;
    *n_parts = (int)parts;
    return true;
}

template <bool SELECT>
static int launch_topk(const TopkArgs& a, hipStream_t stream) {
    const int DK = (a.d + 15) / 16;
    const int n_tiles = (int)ceil_div(a.n_query, 16);
    const int grid = (int)(a.split_from + (int64_t)(n_tiles - a.split_from) * a.n_parts);
    const bool fast = a.vec_ok && a.d == 16 * DK;
    if (SELECT && a.mode == 1) {
        const int DK32 = (a.d + 31) / 32;
        int64_t n_thr = ceil_div(a.n_items, TK_TILE) * TK_TILE * DK32 * 4;
        if (n_thr < n_tiles) n_thr = n_tiles;                  // (the pack launch also clears one flag word per user tile)
        topk_pack_items_bf16_kernel<<<(unsigned)ceil_div(n_thr, 256), 256, 0, stream>>>(a, DK32, a.pk2);
        LLMREC_LAUNCH_CHECK();
        if ((DK32 * 4) & (DK32 * 4 - 1)) {                     // d in (64, 96]: 12 threads per item in the pack launch - the norms get their own
            const int64_t n_pad = ceil_div(a.n_items, TK_TILE) * TK_TILE;
            topk_item_norm_kernel<<<grid_for(n_pad, 16), 256, 0, stream>>>(a, a.cn, n_pad);
            LLMREC_LAUNCH_CHECK();
        }
        switch (DK32) {
            case 1: score_topk_pre_kernel<1><<<grid, 256, 0, stream>>>(a, a.pk2, a.cn); break;
            case 2: score_topk_pre_kernel<2><<<grid, 256, 0, stream>>>(a, a.pk2, a.cn); break;
            case 3: score_topk_pre_kernel<3><<<grid, 256, 0, stream>>>(a, a.pk2, a.cn); break;
            case 4: score_topk_pre_kernel<4><<<grid, 256, 0, stream>>>(a, a.pk2, a.cn); break;
            default: set_error("score_topk: d = %d > 128", a.d); return LLMREC_EUNSUPPORTED;
        }
        LLMREC_LAUNCH_CHECK();
        if (a.n_parts > 1) {
            topk_merge_pre_kernel<<<n_tiles - a.split_from, 1024, 0, stream>>>(a);
            LLMREC_LAUNCH_CHECK();
        }
        // second launch: the exact sweep for the tiles whose verification failed (every other block exits at once); unsplit, fragments
        // straight from Ei (the workspace's fragment area holds the bf16 table)
        TopkArgs e = a;
        e.mode = 0; e.only_flagged = a.fb_word; e.split_from = n_tiles; e.n_parts = 1; e.packed = nullptr;
#define LLMREC_TOPK_FB(D) case D: \
        if (fast) score_topk_kernel<D, true, false><<<n_tiles, 256, 0, stream>>>(e); \
        else score_topk_kernel<D, false, false><<<n_tiles, 256, 0, stream>>>(e); \
        break;
        switch (DK) {
            LLMREC_TOPK_FB(1) LLMREC_TOPK_FB(2) LLMREC_TOPK_FB(3) LLMREC_TOPK_FB(4)
            LLMREC_TOPK_FB(5) LLMREC_TOPK_FB(6) LLMREC_TOPK_FB(7) LLMREC_TOPK_FB(8)
            default: set_error("score_topk: d = %d > 128", a.d); return LLMREC_EUNSUPPORTED;
        }
#undef LLMREC_TOPK_FB
        LLMREC_LAUNCH_CHECK();
        return LLMREC_OK;
    }
    if (SELECT && a.packed) {
        const int64_t n_vec = ceil_div(a.n_items, TK_TILE) * 2 * DK * 64;
        topk_pack_items_kernel<<<(unsigned)ceil_div(n_vec, 256), 256, 0, stream>>>(a, DK, const_cast<float4*>(a.packed), n_vec);
        LLMREC_LAUNCH_CHECK();
    }
#define LLMREC_TOPK_CASE(D) case D: \
        if (!SELECT) scores_kernel<D><<<grid, 256, 0, stream>>>(a); \
        else if (a.packed) score_topk_kernel<D, true, true><<<grid, 256, 0, stream>>>(a); \
        else if (fast) score_topk_kernel<D, true, false><<<grid, 256, 0, stream>>>(a); \
        else score_topk_kernel<D, false, false><<<grid, 256, 0, stream>>>(a); \
        break;
    switch (DK) {
        LLMREC_TOPK_CASE(1) LLMREC_TOPK_CASE(2) LLMREC_TOPK_CASE(3) LLMREC_TOPK_CASE(4)
        LLMREC_TOPK_CASE(5) LLMREC_TOPK_CASE(6) LLMREC_TOPK_CASE(7) LLMREC_TOPK_CASE(8)
        default: set_error("score_topk: d = %d > 128", a.d); return LLMREC_EUNSUPPORTED;
    }
#undef LLMREC_TOPK_CASE
    LLMREC_LAUNCH_CHECK();
    if (SELECT && a.n_parts > 1) {
        topk_merge_kernel<<<n_tiles - a.split_from, 256, 0, stream>>>(a);
        LLMREC_LAUNCH_CHECK();
    }
    return LLMREC_OK;
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

static int64_t topk_split_bytes(int32_t n_query, int64_t n_items) {
    int split_from = 0, n_parts = 1, wide_parts = 0;
    plan_split(n_query, n_items, &split_from, &n_parts);
    int64_t bytes = n_parts == 1 ? 0 : ((int64_t)ceil_div(n_query, 16) - split_from) * n_parts * 16 * 64 * 8;
    if (plan_parts(n_query, n_items, -1, &wide_parts)) {       // (the bf16 sweep's item parts: every tile is split; sized for the smallest part)
        const int64_t wide = (int64_t)ceil_div(n_query, 16) * wide_parts * 16 * 64 * 8;
        if (wide > bytes) bytes = wide;
    }
    return bytes;
}
static int64_t topk_packed_bytes(int64_t n_items, int32_t d) {
    const int64_t exact = ceil_div(n_items, TK_TILE) * 2 * ceil_div(d, 16) * 64 * 16;      // fp32 fragments (rows padded to whole tiles, d to 16)
    const int64_t pre = ceil_div(n_items, TK_TILE) * 2 * ceil_div(d, 32) * 2 * 64 * 16;   // bf16 (hi, mid) fragments (d padded to 32)
    return (exact > pre ? exact : pre) + 256 + align_up(4 * ceil_div(n_items, TK_TILE) * TK_TILE, 256);   // + the bf16 mode's header and per-item factors
}
static int64_t topk_flag_bytes(int32_t n_query) { return align_up(4 * ceil_div(n_query, 16), 256); }   // one word per user tile (bf16 mode)
// bitmaps of long train rows: TK_HEAVY_PER_BLOCK rows of one word per item tile for every block of the sweep; 0 = off (item tables beyond
// 131 072 items, or more than 64 MB of slices)
static int64_t topk_heavy_bytes(int32_t n_query, int64_t n_items, int* all_rows = nullptr) {
    const int64_t words = ceil_div(n_items, TK_TILE);
    int split_from = 0, n_parts = 1;
    plan_split(n_query, n_items, &split_from, &n_parts);
    const int64_t n_tiles = ceil_div(n_query, 16), grid = split_from + (n_tiles - split_from) * n_parts;
    // every row of a block as a bitmap (tile-major slices, bf16 sweep: tk_rows_setup) while 16 rows per block fit the budget, else the long rows only
    const int64_t all = grid * 16 * words * 4, bytes = grid * TK_HEAVY_PER_BLOCK * words * 4;
    if (all_rows) *all_rows = 0;
    if (words > 4096) return 0;
    if (all <= (64ll << 20)) { if (all_rows) *all_rows = 1; return align_up(all, 256); }
    return bytes > (64ll << 20) ? 0 : align_up(bytes, 256);
}

int llmrec_topk_set_part_items(int32_t items) {
    LLMREC_CHECK_ARG(items == -1 || items == 0 || (items >= 1024 && items % TK_TILE == 0), "topk_set_part_items: -1 (off), 0 (policy) or a multiple of %d >= 1024", TK_TILE);
    g_part_items = items;
    return LLMREC_OK;
}

int64_t llmrec_score_topk_workspace_bytes(int32_t n_query, int64_t n_items, int32_t d) {
    if (n_query < 0 || n_items <= 0 || d <= 0) return -1;
    return align_up(topk_split_bytes(n_query, n_items), 256) + topk_packed_bytes(n_items, d) + topk_flag_bytes(n_query) + topk_heavy_bytes(n_query, n_items);
}

int64_t llmrec_score_topk_stats_offset(int32_t n_query, int64_t n_items) {
    if (n_query < 0 || n_items <= 0) return -1;
    return align_up(topk_split_bytes(n_query, n_items), 256);
}

int llmrec_score_topk_f32(int32_t n_query, const int64_t* query_users,
                          const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                          int64_t n_items, int32_t d,
                          const int32_t* train_rowptr, const int32_t* train_colidx,
                          int32_t K, int32_t* out_idx, float* out_score, llmrec_stream_t stream_) {
    return llmrec_score_topk_ws_f32(n_query, query_users, Eu, ldu, Ei, ldi, n_items, d, train_rowptr, train_colidx, K, out_idx, out_score,
                                    nullptr, 0, stream_);
}

int llmrec_score_topk_ws_f32(int32_t n_query, const int64_t* query_users,
                             const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                             int64_t n_items, int32_t d,
                             const int32_t* train_rowptr, const int32_t* train_colidx,
                             int32_t K, int32_t* out_idx, float* out_score,
                             void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    return llmrec_score_topk_mode_f32(n_query, query_users, Eu, ldu, Ei, ldi, n_items, d, train_rowptr, train_colidx, K, out_idx, out_score,
                                      workspace, workspace_bytes, LLMREC_TOPK_MODE_EXACT_SWEEP, stream_);
}

int llmrec_score_topk_mode_f32(int32_t n_query, const int64_t* query_users,
                               const float* Eu, int64_t ldu, const float* Ei, int64_t ldi,
                               int64_t n_items, int32_t d,
                               const int32_t* train_rowptr, const int32_t* train_colidx,
                               int32_t K, int32_t* out_idx, float* out_score,
                               void* workspace, int64_t workspace_bytes, int32_t mode, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && n_items > 0 && d > 0 && K > 0 && K <= LLMREC_TOPK_MAX, "score_topk: bad sizes (K <= %d)", LLMREC_TOPK_MAX);
    LLMREC_CHECK_ARG(mode == LLMREC_TOPK_MODE_EXACT_SWEEP || mode == LLMREC_TOPK_MODE_PREFILTER, "score_topk: unknown mode %d", mode);
    LLMREC_CHECK_ARG(mode != LLMREC_TOPK_MODE_PREFILTER || workspace, "score_topk: the prefilter mode needs the workspace");
    if (n_query == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(query_users && Eu && Ei && out_idx && out_score && ldu >= d && ldi >= d, "score_topk: null pointer or ld < d");
    LLMREC_CHECK_ARG((train_rowptr == nullptr) == (train_colidx == nullptr) || train_rowptr, "score_topk: train CSR incomplete");
    LLMREC_CHECK_ARG(n_items < (1ll << 31), "score_topk: n_items exceeds int32 item ids");
    TopkArgs a;
    a.n_query = n_query; a.query_users = query_users; a.Eu = Eu; a.ldu = ldu; a.Ei = Ei; a.ldi = ldi;
    a.n_items = n_items; a.d = d; a.train_rowptr = train_rowptr; a.train_colidx = train_colidx; a.K = K;
    a.out_idx = out_idx; a.out_score = out_score; a.S = nullptr; a.lds = 0;
    a.vec_ok = (ldu % 4 == 0) && (ldi % 4 == 0) && (((uintptr_t)Eu | (uintptr_t)Ei) % 16 == 0);
    a.split_from = (int)ceil_div(n_query, 16); a.n_parts = 1; a.part_major = 0; a.ws_idx = nullptr; a.ws_score = nullptr; a.packed = nullptr;
    if (mode == LLMREC_TOPK_MODE_PREFILTER && K > LLMREC_TOPK_PREFILTER_MAX_K) mode = LLMREC_TOPK_MODE_EXACT_SWEEP;   // (no room to verify in 64 slots)
    a.mode = mode; a.pk2 = nullptr; a.hdr = nullptr; a.cn = nullptr; a.fb_word = nullptr; a.only_flagged = nullptr;
    a.heavy_bm = nullptr; a.heavy_words = 0; a.heavy_all = 0;
    if (workspace) {                                           // without a workspace: one block per user tile, fragments straight from Ei
        const int64_t need = llmrec_score_topk_workspace_bytes(n_query, n_items, d), split = topk_split_bytes(n_query, n_items);
        LLMREC_CHECK_ARG(workspace_bytes >= need && (uintptr_t)workspace % 16 == 0, "score_topk: workspace of %lld bytes needed (16-byte aligned)", (long long)need);
        if (split > 0) {
            int wide_parts = 0;
            if (mode == LLMREC_TOPK_MODE_PREFILTER && plan_parts(n_query, n_items, d, &wide_parts)) {
                a.split_from = 0; a.n_parts = wide_parts; a.part_major = 1;
            } else {
                plan_split(n_query, n_items, &a.split_from, &a.n_parts);
            }
            if (a.n_parts > 1) {
                a.ws_score = (float*)workspace;
                a.ws_idx = (int32_t*)((char*)workspace + split / 2);
            }
        }
        char* frag = (char*)workspace + align_up(split, 256);
        a.packed = (const float4*)frag;
        int all_rows = 0;
        if (train_rowptr && !a.part_major && topk_heavy_bytes(n_query, n_items, &all_rows) > 0) {      // (item parts: a block walks 1 / n_parts of a row)
            a.heavy_bm = (uint32_t*)(frag + topk_packed_bytes(n_items, d) + topk_flag_bytes(n_query));
            a.heavy_words = (int)ceil_div(n_items, TK_TILE);
            a.heavy_all = all_rows;                            // (the slices' stride; the exact sweep keeps to the two-row form inside them)
        }
        if (mode == LLMREC_TOPK_MODE_PREFILTER) {
            a.hdr = (uint32_t*)frag;                           // (the first 256 bytes of the fragment area)
            a.pk2 = (uint4*)(frag + 256);
            a.cn = (float*)(frag + topk_packed_bytes(n_items, d) - align_up(4 * ceil_div(n_items, TK_TILE) * TK_TILE, 256));
            a.fb_word = (uint32_t*)(frag + topk_packed_bytes(n_items, d));
        }
    }
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
    a.split_from = (int)ceil_div(n_query, 16); a.n_parts = 1; a.part_major = 0; a.ws_idx = nullptr; a.ws_score = nullptr; a.packed = nullptr;
    a.mode = 0; a.pk2 = nullptr; a.hdr = nullptr; a.cn = nullptr; a.fb_word = nullptr; a.only_flagged = nullptr;
    a.heavy_bm = nullptr; a.heavy_words = 0; a.heavy_all = 0;
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

int64_t llmrec_topk_eval_sums_workspace_bytes(int32_t n_query, int32_t n_ks) {
    if (n_query < 0 || n_ks < 1 || n_ks > 8) return -1;
    return align_up((int64_t)ceil_div(n_query > 0 ? n_query : 1, ES_THREADS) * 4 * n_ks * (int64_t)sizeof(double), 256);
}

int llmrec_topk_eval_sums(int32_t n_query, const int64_t* query_users, int32_t K, const int32_t* topk_idx, const int32_t* test_rowptr,
                          const int32_t* test_colidx, int32_t n_ks, const int32_t* ks_host, void* workspace, int64_t workspace_bytes,
                          double* out, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && K > 0 && K <= 128 && n_ks >= 1 && n_ks <= 8 && ks_host && out, "topk_eval_sums: bad sizes (K <= 128, at most 8 cut-offs)");
    LLMREC_CHECK_ARG(n_query == 0 || (query_users && topk_idx && test_rowptr && test_colidx), "topk_eval_sums: null pointer");
    if (!workspace || workspace_bytes < llmrec_topk_eval_sums_workspace_bytes(n_query, n_ks)) {
        set_error("topk_eval_sums: workspace %lld < %lld", (long long)workspace_bytes, (long long)llmrec_topk_eval_sums_workspace_bytes(n_query, n_ks));
        return LLMREC_EWORKSPACE;
    }
    MetricKs ks = {};
    ks.n = n_ks;
    for (int i = 0; i < n_ks; ++i) { LLMREC_CHECK_ARG(ks_host[i] > 0, "topk_eval_sums: cut-offs must be positive"); ks.k[i] = ks_host[i]; }
    const int n_blocks = (int)ceil_div(n_query, ES_THREADS);
    if (n_blocks > 0) {
        topk_eval_sums_kernel<<<n_blocks, ES_THREADS, 0, (hipStream_t)stream_>>>(n_query, query_users, K, topk_idx, test_rowptr, test_colidx, ks, (double*)workspace);
        LLMREC_LAUNCH_CHECK();
    }
    topk_eval_sums_finish_kernel<<<1, 256, 0, (hipStream_t)stream_>>>(n_blocks, 4 * n_ks, (const double*)workspace, out);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_topk_metrics(int32_t n_query, const int64_t* query_users, int32_t K, const uint8_t* hits, const int32_t* topk_idx,
                        const int32_t* test_rowptr, int32_t n_ks, const int32_t* ks_host, double* out, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_query >= 0 && K > 0 && n_ks >= 1 && n_ks <= 8 && ks_host, "topk_metrics: bad sizes (at most 8 cut-offs)");
    if (n_query == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(query_users && hits && topk_idx && test_rowptr && out, "topk_metrics: null pointer");
    MetricKs ks = {};
    ks.n = n_ks;
    for (int i = 0; i < n_ks; ++i) { LLMREC_CHECK_ARG(ks_host[i] > 0, "topk_metrics: cut-offs must be positive"); ks.k[i] = ks_host[i]; }
    topk_metrics_kernel<<<(int)ceil_div(n_query, 128), 128, 0, (hipStream_t)stream_>>>(n_query, query_users, K, hits, topk_idx, test_rowptr, ks, out);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // extern "C"
