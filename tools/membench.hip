// Microbenchmark: HBM streaming rate of the MFMA-fragment access pattern (16 rows x 64 B per wave
// instruction) as a function of the row stride, vs a plain coalesced stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
// The grouped projection's X stream exactly: 128-row blocks, 32 k per step, lane (li, lq) loads the 16-B pieces
// k = kb + 8 lq + 4 h (h = 0, 1) of rows t * 16 + li (t = 0, 1); three register stages, optional barrier per
// step. CONTIG = 1: pieces k = kb + 16 h + 4 lq instead (64 contiguous bytes of a row per instruction).
template <int CONTIG, int BARRIER>
__global__ __launch_bounds__(256, 2) void gemmx_kernel(const float* __restrict__ X, int64_t ldx, int K, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const float* row[2];
    for (int t = 0; t < 2; ++t) row[t] = X + ((int64_t)blockIdx.x * 128 + wave * 32 + t * 16 + li) * ldx;
    float4 acc = make_float4(0, 0, 0, 0);
    auto load = [&](int kb, float4 (&x)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int k = CONTIG ? kb + 16 * h + 4 * lq : kb + 8 * lq + 4 * h;
                if (k > K - 4) k = K - 4;
                x[t][h] = *reinterpret_cast<const float4*>(row[t] + k);
            }
    };
    auto use = [&](const float4 (&x)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) { acc.x += x[t][h].x; acc.y += x[t][h].y; acc.z += x[t][h].z; acc.w += x[t][h].w; }
    };
    float4 x0[2][2], x1[2][2], x2[2][2];
    load(0, x0); load(32, x1);
    for (int kb = 0; kb < K; kb += 96) {
        load(kb + 64, x2); use(x0); if (BARRIER) __syncthreads();
        load(kb + 96, x0); use(x1); if (BARRIER) __syncthreads();
        load(kb + 128, x1); use(x2); if (BARRIER) __syncthreads();
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
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
int run_gemmx(const float* X, int64_t ldx, int K, int blocks, int contig, int barrier, float* out, void* stream) {
    if (contig && barrier) gemmx_kernel<1, 1><<<blocks, 256, 0, (hipStream_t)stream>>>(X, ldx, K, out);
    else if (contig) gemmx_kernel<1, 0><<<blocks, 256, 0, (hipStream_t)stream>>>(X, ldx, K, out);
    else if (barrier) gemmx_kernel<0, 1><<<blocks, 256, 0, (hipStream_t)stream>>>(X, ldx, K, out);
    else gemmx_kernel<0, 0><<<blocks, 256, 0, (hipStream_t)stream>>>(X, ldx, K, out);
    return (int)hipGetLastError();
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

// The weight-gradient kernel's X stream alone: one wave per SIMD (a 100 KB LDS array forces one block per CU), each wave
// walks a slab of rows for one 64-column k slab, 32 rows per tile, lane (li, lq) loads the float4 at row m0 + 4 j + lq,
// column k0 + 4 li (j = 0..7); STAGES tiles in flight. What this reaches is the kernel's memory-latency ceiling.
template <int STAGES>
__global__ __launch_bounds__(256, 1) void wgradx_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int n_kslab, int64_t MC, float* out) {
    __shared__ float pad[25000];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int kslab = blockIdx.x % n_kslab, group = blockIdx.x / n_kslab;
    const int64_t m_begin = (int64_t)(group * 4 + wave) * MC;
    int64_t m_end = m_begin + MC; if (m_end > M) m_end = M;
    if (threadIdx.x == 0 && M < 0) pad[0] = 1.f;
    const float* base = X + kslab * 64 + 4 * li;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 st[STAGES][8];
    auto load = [&](int64_t m0, float4 (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { int64_t r = m0 + 4 * j + lq; if (r > m_end - 1) r = m_end - 1; v[j] = *reinterpret_cast<const float4*>(base + r * ldx); }
    };
    auto use = [&](const float4 (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    };
    if (m_begin >= M) return;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) load(m_begin + 32 * s, st[s]);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += 32 * STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            load(m0 + 32 * (s + STAGES - 1), st[(s + STAGES - 1) % STAGES]);
            __builtin_amdgcn_sched_barrier(0);
            if (m0 + 32 * s < m_end) use(st[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x + pad[1];
}
extern "C" int run_wgradx(const float* X, int64_t ldx, int64_t M, int K, int64_t MC, int stages, float* out, void* stream) {
    const int n_kslab = K / 64;
    const int groups = (int)((M + 4 * MC - 1) / (4 * MC));
    dim3 grid(n_kslab * groups);
    if (stages == 2) wgradx_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, M, n_kslab, MC, out);
    else if (stages == 3) wgradx_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, M, n_kslab, MC, out);
    else if (stages == 4) wgradx_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, M, n_kslab, MC, out);
    else wgradx_kernel<6><<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, M, n_kslab, MC, out);
    return (int)hipGetLastError();
}
