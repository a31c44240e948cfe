// Does the matrix pipe co-execute with the vector ALU on gfx950 for v_mfma_f32_16x16x4_f32 (the exact-fp32 MFMA of topk.hip)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coexec.hip -o tools/coexec.bin && tools/coexec.bin
// A: two waves per SIMD, one issuing MFMAs and one issuing VALU fmas, alone and together.
// B: one wave per SIMD interleaving 1 MFMA with V independent VALU ops in program order.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MF>   // MF = 0: fp32 16x16x4, 1: bf16 16x16x32
__device__ __forceinline__ void mfma_loop(int n, float seed, float* out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MF == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[j & 3], 0, 0, 0);
            else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[j & 3], 0, 0, 0);
        }
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) *out = 1.f;
}
__device__ int g_kind = 0;   // 0 fma f32, 1 int add/xor, 2 compare + select, 3 DPP move + max, 4 LDS reads
__device__ __forceinline__ void valu_loop(int n, float seed, float* out) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const int kind = g_kind;
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed;
    if (kind == 0) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        }
    } else if (kind == 1) {
        int u[8]; for (int i = 0; i < 8; ++i) u[i] = __float_as_int(v[i]);
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) u[j & 7] = (u[j & 7] + 12345) ^ u[(j + 1) & 7];
        }
        for (int i = 0; i < 8; ++i) v[i] = __int_as_float(u[i]);
    } else if (kind == 2) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j & 7] = v[j & 7] > v[(j + 3) & 7] ? v[(j + 1) & 7] : v[(j + 2) & 7];
        }
    } else if (kind == 3) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) { const int x = __float_as_int(v[j & 7]); v[j & 7] = fmaxf(v[(j + 1) & 7], __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false))); }
        }
    } else {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j & 7] += lds[(threadIdx.x + 64 * j + it) & 1023];
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) *out = 1.f;
}
// mode bit 0: waves 0..3 run MFMAs; bit 1: waves 4..7 run VALU
template <int MF>
__global__ __launch_bounds__(512) void coexec_a(int mode, int n_mfma, int n_valu, float seed, float* out) {
    const int w = threadIdx.x >> 6;
    if (w < 4) { if (mode & 1) mfma_loop<MF>(n_mfma, seed, out); }
    else { if (mode & 2) valu_loop(n_valu, seed, out); }
}
// one wave: per MFMA, V independent VALU fmas in program order
template <int V, int MF>
__global__ __launch_bounds__(256) void coexec_b(int n, float seed, float* out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)seed;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MF == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[j & 3], 0, 0, 0);
            else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[j & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < V; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0]; for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) *out = 1.f;
}
template <class F> static float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, nullptr); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b, nullptr); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
// mfma_mask / valu_mask: bit w set = wave w of the 8-wave block runs that loop
template <int MF>
__global__ __launch_bounds__(512) void coexec_c(int mfma_mask, int valu_mask, int n_mfma, int n_valu, float seed, float* out) {
    const int w = threadIdx.x >> 6;
    if ((mfma_mask >> w) & 1) mfma_loop<MF>(n_mfma, seed, out);
    else if ((valu_mask >> w) & 1) valu_loop(n_valu, seed, out);
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    const int blocks = 256;
    const int nm = 4000, nv = 2000;
    {
        int kind = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kind), &kind, 4);
        struct { int mm, vm; const char* what; } cases[] = {
            {0x01, 0x00, "mfma on wave 0"}, {0x00, 0x10, "valu on wave 4"}, {0x00, 0x02, "valu on wave 1"},
            {0x01, 0x10, "mfma wave 0 + valu wave 4"}, {0x01, 0x02, "mfma wave 0 + valu wave 1"}, {0x01, 0x20, "mfma wave 0 + valu wave 5"},
            {0x11, 0x00, "mfma waves 0,4"}, {0x03, 0x00, "mfma waves 0,1"}, {0xff, 0x00, "mfma on all 8"}, {0x00, 0xff, "valu on all 8"}, {0x0f, 0xf0, "mfma 0-3 + valu 4-7"},
        };
        for (auto& c : cases) {
            float t = timeit([&] { coexec_c<0><<<blocks, 512>>>(c.mm, c.vm, nm, 8000, 1.f, out); });
            printf("C %-32s %.3f ms\n", c.what, t);
        }
    }   // 64000 MFMAs (x32 cycles = 2.05 M cycles); 128000 VALU (x4 = 0.5 M cycles)
    for (int mf = 0; mf < 2; ++mf) {
        printf("== %s\n", mf == 0 ? "v_mfma_f32_16x16x4_f32 (8 passes)" : "v_mfma_f32_16x16x32_bf16 (4 passes... 8 on gfx950)");
        for (int kind = 0; kind < 5; ++kind) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kind), &kind, 4);
            const int nvv = 4000;
            float t1 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(1, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(1, nm, nvv, 1.f, out); });
            float t2 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(2, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(2, nm, nvv, 1.f, out); });
            float t3 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(3, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(3, nm, nvv, 1.f, out); });
            const char* kn[5] = {"fma f32", "int add/xor", "compare+select", "DPP + max", "LDS reads"};
            printf("A kind %-15s: mfma alone %.3f ms, other wave alone %.3f ms, together %.3f ms (sum %.3f, max %.3f)\n", kn[kind], t1, t2, t3, t1 + t2, t1 > t2 ? t1 : t2);
        }
        { int kind = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kind), &kind, 4); }
        for (int nvv : {2000, 8000}) {
            float t1 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(1, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(1, nm, nvv, 1.f, out); });
            float t2 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(2, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(2, nm, nvv, 1.f, out); });
            float t3 = timeit([&] { if (mf == 0) coexec_a<0><<<blocks, 512>>>(3, nm, nvv, 1.f, out); else coexec_a<1><<<blocks, 512>>>(3, nm, nvv, 1.f, out); });
            printf("A two waves/SIMD: mfma alone %.3f ms, valu alone (%d x 64 fma) %.3f ms, together %.3f ms (sum %.3f, max %.3f)\n", t1, nvv, t2, t3, t1 + t2, t1 > t2 ? t1 : t2);
        }
        float b0 = timeit([&] { if (mf == 0) coexec_b<0, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<0, 1><<<blocks, 256>>>(nm, 1.f, out); });
        float b2 = timeit([&] { if (mf == 0) coexec_b<2, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<2, 1><<<blocks, 256>>>(nm, 1.f, out); });
        float b4 = timeit([&] { if (mf == 0) coexec_b<4, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<4, 1><<<blocks, 256>>>(nm, 1.f, out); });
        float b6 = timeit([&] { if (mf == 0) coexec_b<6, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<6, 1><<<blocks, 256>>>(nm, 1.f, out); });
        float b8 = timeit([&] { if (mf == 0) coexec_b<8, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<8, 1><<<blocks, 256>>>(nm, 1.f, out); });
        float b12 = timeit([&] { if (mf == 0) coexec_b<12, 0><<<blocks, 256>>>(nm, 1.f, out); else coexec_b<12, 1><<<blocks, 256>>>(nm, 1.f, out); });
        printf("B one wave/SIMD, V VALU ops after each MFMA: V=0 %.3f  V=2 %.3f  V=4 %.3f  V=6 %.3f  V=8 %.3f  V=12 %.3f ms\n", b0, b2, b4, b6, b8, b12);
    }
    return 0;
}
