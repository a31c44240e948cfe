// Cycle accounting of linear_fwd_grouped_bf16x3_kernel (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLLMREC_FWD_PROFILE tools/fwd_prof.hip -o tools/fwd_prof.bin && tools/fwd_prof.bin
// Prints, per wave and k-step on average, clock64 ticks spent in each segment of a k-step.
#include "../llmrec_amd/csrc/dense.hip"
#include <cstdio>
#include <cstdarg>
namespace llmrec { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc(10, stderr); } }
#include <vector>

int main() {
    const int I = 17366, U = 13187, d = 64;
    const int64_t Ms[8] = {I, I, I, I, I, U, I, I};
    const int Ks[8] = {1536, 1536, 1536, 1536, 1536, 1536, 768, 512};
    llmrec_linear_problem_t p[8];
    double ksteps = 0;
    for (int i = 0; i < 8; ++i) {
        float *X, *W, *b, *Y;
        hipMalloc(&X, (size_t)Ms[i] * Ks[i] * 4); hipMalloc(&W, (size_t)d * Ks[i] * 4); hipMalloc(&b, d * 4); hipMalloc(&Y, (size_t)Ms[i] * d * 4);
        hipMemset(X, 0x3c, (size_t)Ms[i] * Ks[i] * 4); hipMemset(W, 0x3c, (size_t)d * Ks[i] * 4); hipMemset(b, 0, d * 4);
        p[i] = {X, Ks[i], Ms[i], Ks[i], W, Ks[i], b, Y, d};
        const int steps = ((Ks[i] + 95) / 96) * 3;
        ksteps += (double)((Ms[i] + 127) / 128) * 4 * steps;
    }
    for (int it = 0; it < 3; ++it) llmrec_linear_fwd_grouped_bf16x3(8, p, d, nullptr);
    hipDeviceSynchronize();
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(llmrec::g_fwd_prof), zero, sizeof(zero));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) llmrec_linear_fwd_grouped_bf16x3(8, p, d, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long q[8];
    hipMemcpyFromSymbol(q, HIP_SYMBOL(llmrec::g_fwd_prof), sizeof(q));
    const char* names[8] = {"issue loads of step k+2", "wait for this step's X", "split X (VALU)", "W fragments from LDS", "48 MFMAs", "W(k+1): wait + split + LDS stores",
                            "block barrier", "prologue + rest"};
    printf("kernel %.4f ms (with the instrumentation), %.0f wave-k-steps per launch\n", ms / iters, ksteps);
    double tot = 0; for (int i = 0; i < 7; ++i) tot += q[i];
    for (int i = 0; i < 8; ++i) printf("%-36s %8.0f ticks per wave-k-step  %5.1f %%\n", names[i], q[i] / (ksteps * iters), 100.0 * q[i] / tot);
    return 0;
}
