/* Host-side helper of utility/load_data.py::Data.sample() (reference utility/load_data.py:157-195): the per-user draws of one batch -
 * one positive from the user's train list, one negative rejected while it is in that list - replayed on a block of RAW 32-bit words of
 * numpy's global MT19937 stream, exactly as RandomState.randint(0, high, size=1) would consume them: rng = high - 1; no word when
 * rng == 0; else words & mask (mask = the next 2^k - 1 >= rng) until one is <= rng. Plain C (gcc), loaded with ctypes; without it the
 * same loop runs in Python (Data._draw_items_fast). Returns the number of words consumed, or -1 when the block is too short. */
#include "../../include/llmrec_host.h"

static inline uint32_t mask_of(uint32_t rng) {
    uint32_t m = rng;
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    return m;
}

int64_t llmrec_host_draw_items(int64_t n, const int64_t* users, const int64_t* list_ptr, const int64_t* list_items, int64_t n_items,
                               const uint32_t* raw, int64_t n_raw, int64_t* pos_out, int64_t* neg_out) {
    int64_t k = 0;
    const uint32_t n_rng = (uint32_t)(n_items - 1), n_mask = mask_of((uint32_t)(n_items - 1));
    for (int64_t j = 0; j < n; ++j) {
        const int64_t u = users[j];
        const int64_t* mine = list_items + list_ptr[u];
        const int64_t len = list_ptr[u + 1] - list_ptr[u];
        const uint32_t rng = (uint32_t)(len - 1);
        uint32_t v = 0;
        if (rng != 0) {
            const uint32_t m = mask_of(rng);
            for (;;) {
                if (k >= n_raw) return -1;
                v = raw[k++] & m;
                if (v <= rng) break;
            }
        }
        pos_out[j] = mine[v];
        for (;;) {
            uint32_t c = 0;
            if (n_rng != 0) {
                for (;;) {
                    if (k >= n_raw) return -1;
                    c = raw[k++] & n_mask;
                    if (c <= n_rng) break;
                }
            }
            int seen = 0;
            for (int64_t t = 0; t < len; ++t)
                if (mine[t] == (int64_t)c) { seen = 1; break; }
            if (!seen) { neg_out[j] = (int64_t)c; break; }
            if (n_rng == 0) return -2;                       /* a single item that the user already has: the reference would loop forever */
        }
    }
    return k;
}


/* CPython's random.sample(population, k) (Lib/random.py, 3.10) on CPython's own MT19937 state: the POSITIONS it selects, in its order.
 * state[0..623] = the generator's words, state[624] = its index (random.getstate()[1]); updated in place (random.setstate). Each
 * _randbelow(n) is getrandbits(n.bit_length()) = one 32-bit word >> (32 - bits), repeated while >= n. use_pool = the branch random.sample
 * takes for (n, k) (n <= 21 + 4 ** ceil(log(3 k, 4)) for k > 5: the caller evaluates it with Python's own float arithmetic).
 * scratch: n int64 (pool branch) or (n + 63) / 64 uint64 words (set branch, as a bitmap). Returns 0, or -1 for bad arguments. */
static inline uint32_t mt_next(uint32_t* mt, uint32_t* idx) {
    if (*idx >= 624) {
        static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u]; }
        for (; kk < 623; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u]; }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        *idx = 0;
    }
    uint32_t y = mt[(*idx)++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
static inline int bit_length(uint64_t n) { int b = 0; while (n) { ++b; n >>= 1; } return b; }
static inline int64_t randbelow(uint32_t* mt, uint32_t* idx, int64_t n) {          /* 0 < n <= 2^32 */
    const int bits = bit_length((uint64_t)n);
    for (;;) {
        const int64_t r = (int64_t)(mt_next(mt, idx) >> (32 - bits));
        if (r < n) return r;
    }
}
int32_t llmrec_host_py_sample(uint32_t* state625, int64_t n, int64_t k, int32_t use_pool, int64_t* scratch, int64_t* out_pos) {
    if (!state625 || n <= 0 || k < 0 || k > n || n > 0x7fffffffll || !scratch || !out_pos) return -1;
    uint32_t* mt = state625;
    uint32_t idx = state625[624];
    if (use_pool) {
        for (int64_t i = 0; i < n; ++i) scratch[i] = i;
        for (int64_t i = 0; i < k; ++i) {
            const int64_t j = randbelow(mt, &idx, n - i);
            out_pos[i] = scratch[j];
            scratch[j] = scratch[n - i - 1];
        }
    } else {
        uint64_t* seen = (uint64_t*)scratch;
        for (int64_t i = 0; i < (n + 63) / 64; ++i) seen[i] = 0;
        for (int64_t i = 0; i < k; ++i) {
            int64_t j = randbelow(mt, &idx, n);
            while (seen[j >> 6] >> (j & 63) & 1u) j = randbelow(mt, &idx, n);
            seen[j >> 6] |= (uint64_t)1 << (j & 63);
            out_pos[i] = j;
        }
    }
    state625[624] = idx;
    return 0;
}
