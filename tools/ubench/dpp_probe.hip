// probe: semantics of the DPP row shifts / rotates used by k_dec_b's horizontal presum (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float dppf(float old, float src, int ctrl_sel) {
    int o = __builtin_bit_cast(int, old), s = __builtin_bit_cast(int, src), r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(o, s, 0x111, 0xf, 0xf, false); break;   // row_shr:1, keep old where no source
        case 1: r = __builtin_amdgcn_update_dpp(o, s, 0x101, 0xf, 0xf, false); break;   // row_shl:1
        case 2: r = __builtin_amdgcn_update_dpp(o, s, 0x121, 0xf, 0xf, false); break;   // row_ror:1
        case 3: r = __builtin_amdgcn_update_dpp(o, s, 0x12f, 0xf, 0xf, false); break;   // row_ror:15
        case 4: r = __builtin_amdgcn_update_dpp(o, s, 0x111, 0xf, 0xf, true); break;    // row_shr:1 bound_ctrl (zero)
        default: r = __builtin_amdgcn_update_dpp(o, s, 0x101, 0xf, 0xf, true); break;   // row_shl:1 bound_ctrl
    }
    return __builtin_bit_cast(float, r);
}
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float v = (float)l, old = -1.0f;
    for (int c = 0; c < 6; ++c) out[c * 64 + l] = dppf(old, v, c);
}
int main() {
    float* d; hipMalloc(&d, 6 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[6] = {"row_shr:1 old", "row_shl:1 old", "row_ror:1", "row_ror:15", "row_shr:1 bc", "row_shl:1 bc"};
    for (int c = 0; c < 6; ++c) { printf("%-14s:", nm[c]); for (int l = 0; l < 20; ++l) printf(" %g", h[c * 64 + l]); printf(" ... l31=%g l32=%g l47=%g l48=%g\n", h[c*64+31], h[c*64+32], h[c*64+47], h[c*64+48]); }
    return 0;
}
