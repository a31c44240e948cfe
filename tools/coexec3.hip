// Which VALU instructions of the bf16x3 split co-execute with which bf16 MFMA shape (two waves on one SIMD: wave 0 = MFMAs back to back,
// wave 4 = a dense stream of one VALU opcode, 8 independent registers)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: 16x16x32 bf16, 2 chains; 1: 16x16x32 bf16, 4 chains; 2: 32x32x16 bf16, 2 chains; 3: 16x16x4 f32
__device__ __forceinline__ void mfma_stream(int n, float seed, float* out) {
    f32x16 c0, c1; for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = seed; }
    f32x4 d0 = {seed, seed, seed, seed}, d1 = d0, d2 = d0, d3 = d0;
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (SHAPE == 0) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d1, 0, 0, 0);
                              d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d1, 0, 0, 0); }
            if (SHAPE == 1) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d1, 0, 0, 0);
                              d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, d3, 0, 0, 0); }
            if (SHAPE == 2) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c1, 0, 0, 0); }
            if (SHAPE == 3) { d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, d1, 0, 0, 0); }
        }
    }
    float s = d0[0] + d1[0] + d2[0] + d3[0]; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.f) *out = 1.f;
}
#define OP8(INSTR) asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n" INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(k))
#define I_ADD(r) "v_add_u32 " #r ", " #r ", %8"
#define I_SUBF(r) "v_sub_f32 " #r ", " #r ", %8"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %8"
#define I_CVT(r) "v_cvt_pk_bf16_f32 " #r ", " #r ", %8"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 16, " #r
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %8"
#define I_MOV(r) "v_mov_b32 " #r ", %8"
template <int KIND>
__device__ __forceinline__ void valu_stream(int n, int k, float* out) {
    int x0 = k, x1 = k + 1, x2 = k + 2, x3 = k + 3, x4 = k + 4, x5 = k + 5, x6 = k + 6, x7 = k + 7;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) OP8(I_ADD); if (KIND == 1) OP8(I_SUBF); if (KIND == 2) OP8(I_FMA); if (KIND == 3) OP8(I_CVT);
            if (KIND == 4) OP8(I_AND); if (KIND == 5) OP8(I_LSHL); if (KIND == 6) OP8(I_PERM); if (KIND == 7) OP8(I_MOV);
        }
    }
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) *out = 1.f;
}
// memory-instruction partners: KIND 8 = global_load_dwordx4 (a 16 KB window, L1/L2 hits), KIND 9 = ds_read_b128, KIND 10 = global_load_lds_dwordx4
__device__ float4 g_win[1024];
template <int KIND>
__device__ __forceinline__ void mem_stream(int n, float* out) {
    __shared__ float4 lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x & 1023] = make_float4(1.f, 2.f, 3.f, 4.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gv[8];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 8) { const float4* p = &g_win[(lane + 64 * j + it) & 1023]; asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(gv[j]) : "v"(p) : "memory"); }
            if (KIND == 9) { float4 v = lds[(lane + 64 * j + it) & 1023]; asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
            if (KIND == 10) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)&g_win[(lane + 64 * j + it) & 1023], (__attribute__((address_space(3))) void*)&lds[64 * j], 16, 0, 0);
        }
        if (KIND == 10 || KIND == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == 8) { for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(gv[j].x), "v"(gv[j].w)); }
    }
    if (acc.x == 12345.f) *out = 1.f;
}
template <int SHAPE, int KIND>
__global__ __launch_bounds__(512) void kk(int mode, int nm, int nv, float seed, float* out) {
    const int w = threadIdx.x >> 6;
    if (w == 0 && (mode & 1)) mfma_stream<SHAPE>(nm, seed, out);
    if (w == 4 && (mode & 2)) { if (KIND < 8) valu_stream<KIND < 8 ? KIND : 0>(nv, (int)seed, out); else mem_stream<KIND>(nv / 4, out); }
}
template <class F> static float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, nullptr); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b, nullptr); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
template <int SHAPE, int KIND> static void run(float* out, const char* sn, const char* kn) {
    const int blocks = 256, nm = 4000, nv = 4000;
    const float a = timeit([&] { kk<SHAPE, KIND><<<blocks, 512>>>(1, nm, nv, 1.f, out); });
    const float b = timeit([&] { kk<SHAPE, KIND><<<blocks, 512>>>(2, nm, nv, 1.f, out); });
    const float c = timeit([&] { kk<SHAPE, KIND><<<blocks, 512>>>(3, nm, nv, 1.f, out); });
    printf("%-24s + %-18s: mfma %.3f  valu %.3f  together %.3f  (sum %.3f)  overlap %.0f %%\n", sn, kn, a, b, c, a + b, 100.0 * (a + b - c) / (a < b ? a : b));
}
template <int SHAPE> static void run_shape(float* out, const char* sn) {
    run<SHAPE, 0>(out, sn, "v_add_u32"); run<SHAPE, 1>(out, sn, "v_sub_f32"); run<SHAPE, 2>(out, sn, "v_fma_f32"); run<SHAPE, 3>(out, sn, "v_cvt_pk_bf16_f32");
    run<SHAPE, 4>(out, sn, "v_and_b32"); run<SHAPE, 5>(out, sn, "v_lshlrev_b32"); run<SHAPE, 6>(out, sn, "v_perm_b32"); run<SHAPE, 7>(out, sn, "v_mov_b32");
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    run<3, 8>(out, "16x16x4 f32, 2 chains", "global_load_dwordx4"); run<3, 9>(out, "16x16x4 f32, 2 chains", "ds_read_b128"); run<3, 10>(out, "16x16x4 f32, 2 chains", "global_load_lds x4");
    run<2, 8>(out, "32x32x16 bf16, 2 chains", "global_load_dwordx4"); run<2, 9>(out, "32x32x16 bf16, 2 chains", "ds_read_b128"); run<2, 10>(out, "32x32x16 bf16, 2 chains", "global_load_lds x4");
    run<0, 8>(out, "16x16x32 bf16, 2 chains", "global_load_dwordx4"); run<0, 9>(out, "16x16x32 bf16, 2 chains", "ds_read_b128"); run<0, 10>(out, "16x16x32 bf16, 2 chains", "global_load_lds x4");
    run_shape<0>(out, "16x16x32 bf16, 2 chains"); run_shape<1>(out, "16x16x32 bf16, 4 chains"); run_shape<2>(out, "32x32x16 bf16, 2 chains"); run_shape<3>(out, "16x16x4 f32, 2 chains");
    return 0;
}
