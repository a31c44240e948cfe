/* libllmrec_host.so - the plain-C HOST helper of the drop-in's sampler (no GPU code; built with gcc by llmrec_amd/build.py::build_host).
 * Optional: utility/load_data.py runs the same loop in Python when the library is absent.
 *
 * Replaces (reference utility/load_data.py:163-186, inside Data.sample()): for every user of the batch one
 * `np.random.randint(0, len(train_items[u]), size=1)` (the positive, by position in the user's list) and `np.random.randint(0, n_items, size=1)`
 * repeated while the drawn id is in the user's train list (the negative) - ~2.1 k scalar numpy calls per batch of 1024. The helper replays exactly
 * those draws on a block of RAW words of the same MT19937 stream (RandomState.randint with the default int64 dtype is masked rejection over
 * 32-bit words; no word is consumed when the range has one element), so the caller can fetch the block with one numpy call and advance the global
 * stream by the number of words the helper reports: same seed -> the reference's batches, bit for bit (tests/test_host_cpu.py). */
#ifndef LLMREC_HOST_H
#define LLMREC_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* users[n]: the batch's user ids; list_ptr[n_users + 1] / list_items: every user's train items in FILE order (positives are drawn by position);
 * raw[n_raw]: the next words of the stream. Writes pos_out[n], neg_out[n]; returns the number of words consumed (>= 0), -1 if raw was too
 * short (call again with a longer block of the SAME stream position), -2 if a user owns the only item (the reference would never return). */
int64_t llmrec_host_draw_items(int64_t n, const int64_t* users, const int64_t* list_ptr, const int64_t* list_items, int64_t n_items,
                               const uint32_t* raw, int64_t n_raw, int64_t* pos_out, int64_t* neg_out);
/* CPython's random.sample(population, k) (reference utility/load_data.py:159: `rd.sample(self.exist_users, self.batch_size)`) replayed on CPython's
 * own MT19937 state: writes the k selected POSITIONS (population[pos] is what random.sample returns) in its order and advances the state in place.
 * state625 = random.getstate()[1] as 625 uint32 (624 words + index); use_pool = which of random.sample's two branches applies to (n, k) - evaluated
 * by the caller with Python's own arithmetic (n <= 21 + 4 ** ceil(log(3 k, 4)) when k > 5, n <= 21 otherwise); scratch: max(n, (n + 63) / 64) int64. */
int32_t llmrec_host_py_sample(uint32_t* state625, int64_t n, int64_t k, int32_t use_pool, int64_t* scratch, int64_t* out_pos);
#ifdef __cplusplus
}
#endif
#endif
