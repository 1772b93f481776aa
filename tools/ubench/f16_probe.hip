// probe (gfx950): does v_mfma_f32_32x32x16_f16 keep f16 DENORMAL inputs (a flushed low plane would cost the fp16 x 2 operand split its
// accuracy for every activation below 2^-3), how does float -> half conversion round, and does it produce denormals?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x16 c = (f32x16)(0.f);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // conversion: packed, RNE?  1 + 2^-11 (tie -> even = 1), 1 + 3 * 2^-11 (tie -> 1 + 2^-9... even), 2^-20 (denormal)
    f32x2 v; v.x = 1.0f + 0.00048828125f; v.y = 1.0f + 3 * 0.00048828125f;
    f16x2 hcv = __builtin_convertvector(v, f16x2);
    f32x2 w; w.x = 9.5367431640625e-07f; w.y = 1.0f + 0.0007f;
    f16x2 hd = __builtin_convertvector(w, f16x2);
    if (threadIdx.x == 0) { out[1] = (float)hcv.x; out[2] = (float)hcv.y; out[3] = (float)hd.x; out[4] = (float)hd.y; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    float h[8];
    const float tests[4][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /* 2^-20: f16 denormal */, 1024.0f}, {3.0517578125e-05f /* 2^-15: denormal */, 4.0f}, {6.103515625e-05f /* 2^-14: min normal */, 4.0f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("mfma f16: a = %g  b = %g  ->  c[0] = %.9g   (exact: %.9g)\n", t[0], t[1], h[0], 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
    printf("cvt: 1 + 2^-11 -> %.10g (RNE: 1)   1 + 3 * 2^-11 -> %.10g (RNE: 1.001953125)   2^-20 -> %.10g (denormal kept: 9.5367e-07)   1.0007 -> %.10g\n", h[1], h[2], h[3], h[4]);
    return 0;
}
