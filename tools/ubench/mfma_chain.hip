// micro-benchmark: issue rate of fp32 MFMAs as a function of how many independent accumulator chains a wave rotates through
// (1 = every MFMA depends on the previous one) and of the waves per SIMD.  dev tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH, int FORM>
__global__ void __launch_bounds__(256, 2) k_chain(float* out, long long* clk, int iters) {
    f32x16 a[4]; f32x4 c[4];
    for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) a[i][e] = 0; for (int e = 0; e < 4; ++e) c[i][e] = 0; }
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 1.0f;
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (FORM == 0) a[u % NCH] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[u % NCH], 0, 0, 0);
            if (FORM == 1) c[u % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c[u % NCH], 0, 0, 0);
            if (FORM == 2) c[u % NCH] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, c[u % NCH], 0, 0, 0);
        }
    }
    long long c1 = clock64();
    float s = 0; for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) s += a[i][e]; for (int e = 0; e < 4; ++e) s += c[i][e]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
template <int NCH, int FORM>
void run(const char* name, float* out, long long* clk) {
    for (int waves = 1; waves <= 2; ++waves) {
        int grid = 256 * waves, iters = 4000;
        hipLaunchKernelGGL((k_chain<NCH, FORM>), dim3(grid), dim3(256), 0, 0, out, clk, 100); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL((k_chain<NCH, FORM>), dim3(grid), dim3(256), 0, 0, out, clk, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        const double macs = FORM == 0 ? 2048.0 : FORM == 1 ? 1024.0 : 256.0;
        printf("%s chains %d waves/SIMD %d: %.1f shader cycles per MFMA per wave (%.1f per SIMD), %.1f TFLOP/s by the event clock\n", name, NCH, waves,
               (double)h / (iters * 32.0), (double)h / (iters * 32.0) / waves, (double)grid * 4 * iters * 32.0 * macs * 2 / ms / 1e9);
    }
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 16);
    run<1, 0>("32x32x2", out, clk); run<2, 0>("32x32x2", out, clk); run<4, 0>("32x32x2", out, clk);
    run<1, 1>("16x16x4", out, clk); run<2, 1>("16x16x4", out, clk); run<4, 1>("16x16x4", out, clk);
    run<1, 2>("4x4x1  ", out, clk); run<2, 2>("4x4x1  ", out, clk); run<4, 2>("4x4x1  ", out, clk);
    return 0;
}
