// micro-benchmark: k_dec_b (ConvT3 + final conv + sigmoid + per-image reduction) alone on synthetic data.
// dev tool, not part of the product.   usage: dec_b_bench [rows] [dbg]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/ubench/variants tools/ubench/dec_b_bench.hip -o tools/ubench/dec_b_bench
#include "decoder.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace efe;

__global__ void k_fill(float* p, size_t n, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = scale * (float)((i * 2654435761u) >> 20 & 1023) * (1.f / 1024.f);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 16384;
    const int dbg = argc > 2 ? atoi(argv[2]) : 0;
    const float yscale = argc > 3 ? (float)atof(argv[3]) : 1.f;     // 0 = all-zero activations (switching-power experiment)
    float *y, *w3, *w4, *b, *val;
    const size_t wn = 9 * 1 * 8 * 64 * 4;
    hipMalloc(&y, (size_t)rows * 65536 * 4); hipMalloc(&w3, wn * 4); hipMalloc(&w4, 9 * 32 * 4); hipMalloc(&b, 128 * 4); hipMalloc(&val, (size_t)rows * 4);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, y, (size_t)rows * 65536, yscale);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w3, wn, 0.02f);
    hipLaunchKernelGGL(k_fill, dim3(2), dim3(256), 0, 0, w4, (size_t)9 * 32, 0.05f);
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(128), 0, 0, b, (size_t)128, 0.01f);
    DecBArgs a{};
    a.y2 = y; a.w3 = w3; a.b3 = b; a.w4 = w4; a.b4 = 0.01f; a.rows = rows; a.m0 = 0; a.rows_per_group = rows;
    a.gm = GroupMap{}; a.gm.per_stage = 1; a.gm.S = 1; a.reward0 = 0; a.store0 = 0; a.val = val; a.po = nullptr; a.dbg = dbg; a.tl = nullptr;
    int* queue; hipMalloc(&queue, 4); hipMemset(queue, 0, 4); a.queue = queue;
    launch_dec_b(a, 0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#if defined(EFE_PHASE_CLK) || defined(EFE_CLK_RATE)
    {
        long long* tl; hipMalloc(&tl, (size_t)rows * 8 * 8); hipMemset(tl, 0, (size_t)rows * 8 * 8);
        a.tl = tl; hipMemset(queue, 0, 4); launch_dec_b(a, 0); hipDeviceSynchronize(); a.tl = nullptr;
        std::vector<long long> h((size_t)rows * 8); hipMemcpy(h.data(), tl, (size_t)rows * 8 * 8, hipMemcpyDeviceToHost);
        const char* nm[8] = {"prologue", "stage+barrier", "prefetch+tap loops", "T mfma + T writes", "barrier", "gather", "final", "-"};
        double tot = 0; printf("cycles per image (wave 0):");
        for (int i = 0; i < 7; ++i) { double sacc = 0; for (int r = 0; r < rows; ++r) sacc += (double)h[(size_t)r * 8 + i]; sacc /= rows; tot += sacc; printf("  %s %.0f", nm[i], sacc); }
        printf("  | total %.0f (MFMA demand of a pair: 2 x 163840)\n", tot);
        {   // effective shader clock of the workgroups: cycle-counter ticks / wall time (phs[7] = 100 MHz wall ticks per workgroup)
            double cyc = 0, wall = 0; int nwg = 0;
            for (int r = 0; r < rows; ++r) { double c = 0; for (int i = 0; i < 7; ++i) c += (double)h[(size_t)r * 8 + i]; if (h[(size_t)r * 8 + 7] > 0) { cyc += c; wall += (double)h[(size_t)r * 8 + 7]; ++nwg; } }
            printf("workgroups that reported: %d, mean wall per workgroup %.1f us, cycle counter rate %.3f GHz\n", nwg, wall / nwg / 100.0, cyc / (wall / 100e6) / 1e9);
        }
    }
#endif
    float best = 1e9f, sum = 0.f;
#ifdef EFE_CLK_RATE
    long long* tl2; hipMalloc(&tl2, (size_t)rows * 8 * 8); a.tl = tl2;       // every timed launch also reports its cycle-counter rate
#endif
    for (int rep = 0; rep < 12; ++rep) {
        hipMemset(queue, 0, 4);
#ifdef EFE_CLK_RATE
        hipMemset(tl2, 0, (size_t)rows * 8 * 8);
#endif
        hipEventRecord(e0); launch_dec_b(a, 0); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        sum += ms;
#ifdef EFE_CLK_RATE
        {
            std::vector<long long> h2((size_t)rows * 8); hipMemcpy(h2.data(), tl2, (size_t)rows * 8 * 8, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0; int n = 0;
            for (int r = 0; r < rows; ++r) if (h2[(size_t)r * 8 + 7] > 0) { cyc += (double)h2[(size_t)r * 8]; wall += (double)h2[(size_t)r * 8 + 7]; ++n; }
            printf("  rep %2d: %.3f ms, %d workgroups, %.2f M cycles per workgroup, cycle counter rate %.3f GHz\n", rep, ms, n, cyc / n / 1e6, cyc / (wall / 100e6) / 1e9);
        }
#endif
    }
    const double flops = (double)rows * (2.0 * 9.0 * 64 * 32 * 1024 + 2.0 * 9 * 32 * 4096);
    printf("k_dec_b rows %d dbg %d: best of 12 %.3f ms (mean %.3f)  %.1f TFLOP/s  frac %.3f\n", rows, dbg, best, sum / 12, flops / best / 1e9, flops / best / 1e9 / 157.3);
    return 0;
}
