// common.h - shared host/device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/llmrec_hip.h"

namespace llmrec {

void set_error(const char* fmt, ...);

#define LLMREC_CHECK_ARG(cond, ...)                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            ::llmrec::set_error(__VA_ARGS__);                         \
            return LLMREC_EINVAL;                                     \
        }                                                             \
    } while (0)

#define LLMREC_HIP(call)                                                                   \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::llmrec::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return LLMREC_EHIP;                                                            \
        }                                                                                  \
    } while (0)

// after a kernel launch: catches bad launch configurations without synchronising
#define LLMREC_LAUNCH_CHECK()                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            ::llmrec::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return LLMREC_EHIP;                                                            \
        }                                                                                  \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// grid size for a grid-stride kernel: enough blocks to fill 256 CUs x 8, never more than needed
static inline int grid_for(int64_t work_items, int per_block, int max_blocks = 256 * 8) {
    int64_t b = ceil_div(work_items, per_block);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

#ifdef __HIPCC__
// butterfly sum over the `width` lanes of an aligned lane group (width = power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
#endif

}  // namespace llmrec
