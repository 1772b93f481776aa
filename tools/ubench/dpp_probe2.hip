// probe: wave-wide DPP shifts on gfx950 (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const int s = __builtin_bit_cast(int, (float)l);
    out[0 * 64 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, s, 0x138, 0xf, 0xf, true));   // wave_shr:1
    out[1 * 64 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, s, 0x130, 0xf, 0xf, true));   // wave_shl:1
}
int main() {
    float* d; hipMalloc(&d, 2 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int c = 0; c < 2; ++c) { printf("%s:", c ? "wave_shl:1" : "wave_shr:1"); for (int l = 0; l < 64; ++l) printf(" %g", h[c * 64 + l]); printf("\n"); }
    return 0;
}
