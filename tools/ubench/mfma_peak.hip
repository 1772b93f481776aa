// micro-benchmark: sustained fp32 MFMA rate and shader clock on this GPU (dev tool, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 2) k_mfma(float* out, long long* clk, int iters) {
    f32x16 a0, a1, a2, a3;
    for (int e = 0; e < 16; ++e) { a0[e] = 0; a1[e] = 0; a2[e] = 0; a3[e] = 0; }
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 1.0f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a3, 0, 0, 0);
        }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0; for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 16);
    int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
    for (int waves = 1; waves <= 2; ++waves) {
        int grid = 256 * waves, iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, clk, 1000); hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, clk, iters); hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double flops = (double)grid * 4 * iters * 32.0 * 4096.0;
            printf("WGs/CU %d: %.1f ms  %.1f TFLOP/s  shader clk %.3f GHz (clock64/wall %.0f kHz)\n", waves, ms, flops / ms / 1e9,
                   (double)h[0] / ((double)h[1] / (wrate * 1e3)) / 1e9, (double)wrate);
        }
    }
    return 0;
}
