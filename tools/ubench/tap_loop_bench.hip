// micro-benchmark: MFMA issue efficiency of the software-pipelined tap loop (mfma_pipe.h) in isolation, i.e. the
// CT1 inner loop of k_dec_a without staging / epilogues / stores.  dev tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/ubench/variants tools/ubench/tap_loop_bench.hip -o tools/ubench/tap_loop_bench
#include "mfma_pipe.h"
#include <cstdio>
#include <vector>
using namespace efe;

template <int MODE, int MTT, int NTT, int NWV>
__global__ void __launch_bounds__(64 * NWV, NWV / 2) k_loop(const float* w1, float* out, long long* clk, int reps, int ntaps) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int j = lane & 31, h = lane >> 5;
    const int pcol = j & 15, prow0 = 4 * w + (j >> 4);
    for (int i = tid; i < 257 * 16; i += 64 * NWV) sm[i] = make_float4(1e-3f * (i & 7), 0.f, 1e-3f, 0.f);
    __syncthreads();
    const float4* W1 = reinterpret_cast<const float4*>(w1);             // uniform base (TapPipe adds the lane)
#ifdef UB_AGPR
    { float z = 0.f; asm volatile("; force the AGPR form of the MFMAs" : : "a"(z)); }
#endif
    f32x16 acc[MTT][NTT];
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        tap_loop<MTT, NTT>(acc, ntaps, W1, sm, h, [&](int t, int (&bs)[NTT], int (&sw)[NTT], int& wt) {
            const int kh = t / 3, kw = t - kh * 3;
            wt = t;
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                const int sy = prow0 + 2 * nt + 1 - kh, sx = pcol + 1 - kw;
                const bool ok = sy >= 0 && sy < 16 && sx >= 0 && sx < 16;
                const int sp = ok ? sy * 16 + sx : 256;
                bs[nt] = sp * 16; sw[nt] = sp & 15;
            }
        }, ConvWIdx{});
        if (MODE == 1) {        // the in-place epilogue + barriers of the real kernel
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                const int pix = 64 * w + 32 * nt + j;
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int c4 = mt * 8 + 2 * g4 + h;
                        float4 v;
                        v.x = fmaxf(acc[mt][nt][4 * g4 + 0], 0.f) * 1e-3f; v.y = fmaxf(acc[mt][nt][4 * g4 + 1], 0.f) * 1e-3f;
                        v.z = fmaxf(acc[mt][nt][4 * g4 + 2], 0.f) * 1e-3f; v.w = fmaxf(acc[mt][nt][4 * g4 + 3], 0.f) * 1e-3f;
                        sm[swz(pix, c4)] = v;
                    }
            }
            __syncthreads();
        }
    }
    long long c1 = clock64(), w1c = wall_clock64();
    float s = 0;
#pragma unroll
    for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[mt][nt][e];
    out[blockIdx.x * 64 * NWV + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1c - w0; }
}

template <int MODE, int MTT, int NTT, int NWV>
static void run(const char* name, const float* w, float* out, long long* clk, int wrate) {
    const size_t lds = (257 * 16 + 32) * sizeof(float4);
    (void)hipFuncSetAttribute((const void*)(k_loop<MODE, MTT, NTT, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int grid = 256 * per_cu, reps = 200;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_loop<MODE, MTT, NTT, NWV>), dim3(grid), dim3(64 * NWV), lds, 0, w, out, clk, 10, 9); hipDeviceSynchronize();
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL((k_loop<MODE, MTT, NTT, NWV>), dim3(grid), dim3(64 * NWV), lds, 0, w, out, clk, reps, 9); hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            const double mfmas = (double)reps * 9 * 8 * 4 * MTT * NTT;                    // per wave
            const double ghz = (double)hc[0] / ((double)hc[1] / (wrate * 1e3)) / 1e9;
            const double flops = (double)grid * NWV * mfmas * 4096.0;
            printf("%-22s tile %dx%d waves/WG %d WGs/CU %d: %.2f ms  %.1f TFLOP/s  clk %.2f GHz  cycles/MFMA/wave %.1f  pipe busy %.3f\n", name, MTT, NTT, NWV, per_cu, ms,
                   flops / ms / 1e9, ghz, (double)hc[0] / mfmas, mfmas * 64.0 * per_cu * (NWV / 4) / (double)hc[0]);
        }
    }
}

int main() {
    float *w, *out; long long* clk;
    const size_t wn = 9 * 2 * 8 * 64 * 4;
    hipMalloc(&w, wn * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&clk, 16);
    std::vector<float> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = 1e-3f * (float)((i * 7) % 13);
    hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice);
    int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
    run<0, 2, 2, 4>("tap loop only", w, out, clk, wrate);
    run<0, 1, 2, 4>("tap loop only", w, out, clk, wrate);
    run<0, 1, 1, 4>("tap loop only", w, out, clk, wrate);
    run<0, 1, 1, 8>("tap loop only", w, out, clk, wrate);
    run<0, 2, 1, 8>("tap loop only", w, out, clk, wrate);
    return 0;
}
