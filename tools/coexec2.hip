// Follow-up to tools/coexec.hip with the instruction order pinned by inline asm.
//   D: two waves on one SIMD; the MFMA wave paces its MFMAs with s_nop (so it is never blocked AT an MFMA); partner = dense VALU
//   E: one wave: MFMA followed by V independent v_add_u32 in program order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PACE>
__device__ __forceinline__ void mfma_paced(int n, float seed, float* out) {
    f32x4 a0 = {seed, seed, seed, seed}, a1 = a0, a2 = a0, a3 = a0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (PACE == 0)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            else
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n s_nop 15\n s_nop 9\n v_mfma_f32_16x16x4_f32 %1, %4, %4, %1\n s_nop 15\n s_nop 9\n"
                             "v_mfma_f32_16x16x4_f32 %2, %4, %4, %2\n s_nop 15\n s_nop 9\n v_mfma_f32_16x16x4_f32 %3, %4, %4, %3\n s_nop 15\n s_nop 9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        }
    }
    if (a0[0] + a1[0] + a2[0] + a3[0] == 12345.f) *out = 1.f;
}
__device__ __forceinline__ void valu_dense(int n, int seed, float* out) {
    int x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(seed));
    }
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) *out = 1.f;
}
template <int PACE>
__global__ __launch_bounds__(512) void kd(int mfma_mask, int valu_mask, int nm, int nv, float seed, float* out) {
    const int w = threadIdx.x >> 6;
    if ((mfma_mask >> w) & 1) mfma_paced<PACE>(nm, seed, out);
    else if ((valu_mask >> w) & 1) valu_dense(nv, (int)seed, out);
}
// E: one wave, V independent VALU after each MFMA
template <int V>
__global__ __launch_bounds__(256) void ke(int n, float seed, float* out) {
    f32x4 a0 = {seed, seed, seed, seed}, a1 = a0;
    int x0 = 1, x1 = 2, x2 = 3, x3 = 4, x4 = 5, x5 = 6, x6 = 7, x7 = 8; const int one = (int)seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(a0) : "v"(seed));
            if (V >= 4) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(one));
            if (V >= 8) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(one));
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(a1) : "v"(seed));
            if (V >= 4) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(one));
            if (V >= 8) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(one));
        }
    }
    if (a0[0] + a1[0] + (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7) == 12345.f) *out = 1.f;
}
// F: other MFMA shapes, two waves on one SIMD (wave 0 MFMA back to back, wave 4 dense v_add)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__device__ __forceinline__ void mfma_shape(int n, float seed, float* out) {
    f32x16 c0, c1; for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = seed; }
    f32x4 d0 = {seed, seed, seed, seed}, d1 = d0;
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)seed;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    s16x4 ab4 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
    f16x8 hh; for (int i = 0; i < 8; ++i) hh[i] = (_Float16)seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SHAPE == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, c1, 0, 0, 0); }
            if (SHAPE == 1) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c1, 0, 0, 0); }
            if (SHAPE == 2) { d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, seed, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, seed, d1, 0, 0, 0); }
            if (SHAPE == 3) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d1, 0, 0, 0); }
            if (SHAPE == 4) { d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab4, ab4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab4, ab4, d1, 0, 0, 0); }
            if (SHAPE == 5) { c0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab4, ab4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab4, ab4, c1, 0, 0, 0); }
            if (SHAPE == 6) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, hh, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, hh, d1, 0, 0, 0); }
            if (SHAPE == 7) { d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, d1, 0, 0, 0); }
        }
    }
    float s = d0[0] + d1[0]; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.f) *out = 1.f;
}
template <int SHAPE>
__global__ __launch_bounds__(512) void kf(int mfma_mask, int valu_mask, int nm, int nv, float seed, float* out) {
    const int w = threadIdx.x >> 6;
    if ((mfma_mask >> w) & 1) mfma_shape<SHAPE>(nm, seed, out);
    else if ((valu_mask >> w) & 1) valu_dense(nv, (int)seed, out);
}
template <class F> static float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, nullptr); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b, nullptr); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    const int blocks = 256, nm = 4000, nv = 4000;   // 64000 MFMAs per wave; 256000 v_add per wave
    float t;
    t = timeit([&] { kd<0><<<blocks, 512>>>(0x01, 0, nm, nv, 1.f, out); }); printf("D mfma back-to-back, wave 0 alone          %.3f ms\n", t);
    t = timeit([&] { kd<1><<<blocks, 512>>>(0x01, 0, nm, nv, 1.f, out); }); printf("D mfma paced by s_nop, wave 0 alone          %.3f ms\n", t);
    t = timeit([&] { kd<0><<<blocks, 512>>>(0, 0x10, nm, nv, 1.f, out); }); printf("D dense v_add, wave 4 alone                  %.3f ms\n", t);
    t = timeit([&] { kd<0><<<blocks, 512>>>(0x01, 0x10, nm, nv, 1.f, out); }); printf("D back-to-back mfma (w0) + v_add (w4)        %.3f ms\n", t);
    t = timeit([&] { kd<1><<<blocks, 512>>>(0x01, 0x10, nm, nv, 1.f, out); }); printf("D paced mfma (w0) + v_add (w4)               %.3f ms\n", t);
    t = timeit([&] { kd<1><<<blocks, 512>>>(0x10, 0x01, nm, nv, 1.f, out); }); printf("D paced mfma (w4) + v_add (w0)               %.3f ms\n", t);
    t = timeit([&] { kd<0><<<blocks, 512>>>(0x10, 0x01, nm, nv, 1.f, out); }); printf("D back-to-back mfma (w4) + v_add (w0)        %.3f ms\n", t);
    t = timeit([&] { ke<0><<<blocks, 256>>>(nm, 1.f, out); }); printf("E one wave, mfma only                        %.3f ms\n", t);
    t = timeit([&] { ke<4><<<blocks, 256>>>(nm, 1.f, out); }); printf("E one wave, mfma + 4 v_add each              %.3f ms\n", t);
    t = timeit([&] { ke<8><<<blocks, 256>>>(nm, 1.f, out); }); printf("E one wave, mfma + 8 v_add each              %.3f ms\n", t);
    const char* names[8] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16 (1k)", "v_mfma_f32_32x32x8_bf16 (1k)", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32"};
    for (int sh = 0; sh < 8; ++sh) {
        auto run = [&](int mm, int vm) { return timeit([&] {
            switch (sh) { case 0: kf<0><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; case 1: kf<1><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; case 2: kf<2><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break;
                case 3: kf<3><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; case 4: kf<4><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; case 5: kf<5><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break;
                case 6: kf<6><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; default: kf<7><<<blocks, 512>>>(mm, vm, nm, nv, 1.f, out); break; } }); };
        const float a = run(0x01, 0), b = run(0, 0x10), c = run(0x01, 0x10);
        printf("F %-26s: mfma (w0) alone %.3f, v_add (w4) alone %.3f, together %.3f ms (sum %.3f)\n", names[sh], a, b, c, a + b);
    }
    return 0;
}
