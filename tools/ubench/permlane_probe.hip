// probe: semantics of v_permlane32_swap_b32 as __builtin_amdgcn_permlane32_swap(a, b, fi, bc) returns them (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    u32x2 r = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    out[2 * l] = r.x; out[2 * l + 1] = r.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: x = %3u  y = %3u\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
