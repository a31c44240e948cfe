// Microbenchmark: HBM streaming rate of the MFMA-fragment access pattern (16 rows x 64 B per wave
// instruction) as a function of the row stride, vs a plain coalesced stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
// each block: 128 rows (4 waves x 32 rows), sweeps `kbytes` bytes per row in 64-B pieces per instruction
__global__ void frag_kernel(const float* __restrict__ X, int64_t row_stride_f, int64_t tile_stride_f, int kfloats, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const float* base = X + (int64_t)blockIdx.x * tile_stride_f;
    const float* r0 = base + (int64_t)(wave * 32 + li) * row_stride_f + 4 * lq;
    const float* r1 = r0 + 16 * row_stride_f;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int k = 0; k < kfloats; k += 64) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(r0 + k + 16 * j); v[4 + j] = *reinterpret_cast<const float4*>(r1 + k + 16 * j); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
__global__ void stream_kernel(const float4* __restrict__ X, int64_t n4, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = X[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
int run_frag(const float* X, int64_t row_stride_f, int64_t tile_stride_f, int kfloats, int blocks, float* out, void* stream) {
    frag_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(X, row_stride_f, tile_stride_f, kfloats, out);
    return (int)hipGetLastError();
}
int run_stream(const float* X, int64_t n4, int blocks, float* out, void* stream) {
    stream_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4*)X, n4, out);
    return (int)hipGetLastError();
}
}
