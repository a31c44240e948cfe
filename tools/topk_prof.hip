// Cycle accounting of score_topk_kernel (dev tool): hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLLMREC_TOPK_PROFILE
//   -Iinclude tools/topk_prof.hip -o gpurun_out/topk_prof && gpurun_out/topk_prof
// Prints, per wave on average, s_memtime cycles spent in: whole sweep | rounds | drains | rendezvous wait | sort+merge | MFMA part of rounds.
#include "../llmrec_amd/csrc/topk.hip"
#include <cstdio>
#include <cstdarg>
namespace llmrec { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc(10, stderr); } }
#include <vector>
#include <random>

int main() {
    const int U = 13187, I = 17366, d = 64, K = 50;
    std::mt19937 rng(1); std::normal_distribution<float> nd;
    std::vector<float> eu((size_t)U * d), ei((size_t)I * d);
    for (auto& x : eu) x = nd(rng);
    for (auto& x : ei) x = nd(rng);
    std::vector<int64_t> q(U); for (int i = 0; i < U; ++i) q[i] = i;
    std::vector<int32_t> rp(U + 1), ci;
    for (int u = 0; u < U; ++u) { rp[u] = (int32_t)ci.size(); int n = 1 + rng() % 7; int32_t c = rng() % 1000; for (int j = 0; j < n; ++j) { ci.push_back(c); c += 1 + rng() % 2000; if (c >= I) break; } }
    rp[U] = (int32_t)ci.size();
    float *dEu, *dEi, *dS; int64_t* dq; int32_t *drp, *dci, *dI;
    hipMalloc(&dEu, eu.size() * 4); hipMalloc(&dEi, ei.size() * 4); hipMalloc(&dq, U * 8); hipMalloc(&drp, rp.size() * 4);
    hipMalloc(&dci, ci.size() * 4); hipMalloc(&dI, (size_t)U * K * 4); hipMalloc(&dS, (size_t)U * K * 4);
    hipMemcpy(dEu, eu.data(), eu.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dEi, ei.data(), ei.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, q.data(), U * 8, hipMemcpyHostToDevice); hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) llmrec_score_topk_f32(U, dq, dEu, d, dEi, d, I, d, drp, dci, K, dI, dS, nullptr);
    hipDeviceSynchronize();
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(llmrec::g_topk_prof), zero, sizeof(zero));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) llmrec_score_topk_f32(U, dq, dEu, d, dEi, d, I, d, drp, dci, K, dI, dS, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long p[8];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(llmrec::g_topk_prof), sizeof(p));
    const double waves = (double)((U + 15) / 16) * 4 * iters;
    const char* names[8] = {"sweep", "rounds", "drains", "rendezvous wait", "sort+merge", "round: tile wait", "round: mfma", "round: mask+select"};
    printf("kernel %.4f ms (with the instrumentation)\n", ms / iters);
    for (int i = 0; i < 8; ++i) printf("%-22s %10.0f memtime ticks per wave\n", names[i], p[i] / waves);
    return 0;
}
