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
#ifdef LLMREC_TOPK_PROFILE
    unsigned long long zero[10] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(llmrec::g_topk_prof), zero, sizeof(zero));
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) llmrec_score_topk_f32(U, dq, dEu, d, dEi, d, I, d, drp, dci, K, dI, dS, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.4f ms%s\n", ms / iters, 
#ifdef LLMREC_TOPK_PROFILE
        " (with the instrumentation)");
#else
        "");
#endif
#ifdef LLMREC_TOPK_PROFILE
    unsigned long long p[10];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(llmrec::g_topk_prof), sizeof(p));
    const double waves = (double)((U + 15) / 16) * 4 * iters;
    const char* names[8] = {"sweep", "rounds", "drains", "rendezvous wait", "sort+merge", "round: tile wait", "round: mfma", "round: mask+select"};
    for (int i = 0; i < 8; ++i) printf("%-22s %10.0f memtime ticks per wave\n", names[i], p[i] / waves);
    printf("sweep, wall clock      %10.1f us per wave (100 MHz counter) -> %.3f memtime ticks per ns\n", p[8] / waves / 100.0, (p[0] / waves) / (p[8] / waves * 10.0));
    {
        static unsigned long long place[4096][4];
        hipMemcpyFromSymbol(place, HIP_SYMBOL(llmrec::g_topk_place), sizeof(place));
        const int nb = (U + 15) / 16;
        unsigned long long t0 = ~0ull; for (int b = 0; b < nb; ++b) if (place[b][2] < t0) t0 = place[b][2];
        // blocks per CU (key = xcc, se, sh, cu) and the span of each block of the LAST launch
        std::vector<int> per_cu(8 * 8 * 2 * 16, 0);
        for (int b = 0; b < nb; ++b) { const unsigned h = (unsigned)place[b][0]; const int cu = (h >> 8) & 15, sh = (h >> 12) & 1, se = (h >> 13) & 7, xcc = (int)place[b][1] & 7; per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++; }
        int hist[16] = {0}; for (int c : per_cu) hist[c < 15 ? c : 15]++;
        printf("CUs by number of blocks placed on them:"); for (int i = 0; i < 16; ++i) if (hist[i]) printf("  %d blocks: %d CUs", i, hist[i]); printf("\n");
        double span_by[16] = {0}; int n_by[16] = {0}; double start_max = 0, end_max = 0;
        for (int b = 0; b < nb; ++b) { const unsigned h = (unsigned)place[b][0]; const int cu = (h >> 8) & 15, sh = (h >> 12) & 1, se = (h >> 13) & 7, xcc = (int)place[b][1] & 7;
            const int c = per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]; span_by[c < 15 ? c : 15] += (place[b][3] - place[b][2]) / 100.0; n_by[c < 15 ? c : 15]++;
            if ((place[b][2] - t0) / 100.0 > start_max) start_max = (place[b][2] - t0) / 100.0; if ((place[b][3] - t0) / 100.0 > end_max) end_max = (place[b][3] - t0) / 100.0; }
        for (int i = 0; i < 16; ++i) if (n_by[i]) printf("  blocks on a CU with %d blocks: mean span %.1f us (%d blocks)\n", i, span_by[i] / n_by[i], n_by[i]);
        printf("latest block start %.1f us, latest end %.1f us after the first start\n", start_max, end_max);
    }
    printf("longest sweep          %10.0f memtime ticks\n", (double)p[9]);
#endif
    return 0;
}
