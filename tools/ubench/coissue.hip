// micro-benchmark: how fast does a VALU / LDS / store instruction stream of one wave run while the OTHER wave on the same
// SIMD streams fp32 MFMAs (and what does it cost the MFMA stream)?  512 workgroups x 256 threads, 66 KB LDS each: WG i and
// i+256 share a CU (tools/ubench/hwid_probe.hip), one wave of each per SIMD.  dev tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// role: 0 idle, 1 mfma, 2 independent VALU fma chains, 3 LDS reads, 4 global stores, 5 VALU transcendental mix
__global__ void __launch_bounds__(256, 2) k_co(int roleA, int roleB, int iters, float* out, long long* clk) {
    extern __shared__ float sm[];
    const int role = blockIdx.x < 256 ? roleA : roleB;
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) sm[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    float x = tid * 1e-3f + 1.f, y = 1.0001f;
    float r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3, r4 = x + 4, r5 = x + 5, r6 = x + 6, r7 = x + 7;
    f32x16 a0, a1, a2, a3;
    for (int e = 0; e < 16; ++e) { a0[e] = 0; a1[e] = 0; a2[e] = 0; a3[e] = 0; }
    const long long c0 = clock64();
    if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a3, 0, 0, 0);
            }
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {       // 64 independent-ish fmas
                r0 = fmaf(r0, y, 1e-3f); r1 = fmaf(r1, y, 1e-3f); r2 = fmaf(r2, y, 1e-3f); r3 = fmaf(r3, y, 1e-3f);
                r4 = fmaf(r4, y, 1e-3f); r5 = fmaf(r5, y, 1e-3f); r6 = fmaf(r6, y, 1e-3f); r7 = fmaf(r7, y, 1e-3f);
            }
        }
    } else if (role == 3) {
        int idx = tid;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { r0 += sm[(idx + u * 64) & 4095]; }
            idx = (idx + 7) & 4095;
        }
    } else if (role == 4) {
        float4* o4 = reinterpret_cast<float4*>(out) + (size_t)blockIdx.x * 256 * 16 * 64;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) o4[((size_t)(i & 63) * 16 + u) * 256 + tid] = make_float4(r0, r1, r2, r3);
            r0 += 1.f;
        }
    } else if (role == 5) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float p = 1.0f / (1.0f + expf(-r0));
                r1 += -(1.0f - p) * logf(1.00001f - p) - p * logf(0.00001f + p);
                r0 = r0 * 0.999f + 1e-3f;
            }
        }
    }
    const long long c1 = clock64();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    if (role != 4) out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = c1 - c0;
}

static double avg(const std::vector<long long>& v, int lo) { double s = 0; for (int i = lo; i < lo + 256; ++i) s += (double)v[i]; return s / 256; }

int main() {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)512 * 256 * 16 * 64 * 16); hipMalloc(&clk, 512 * 8);
    const size_t lds = 66 * 1024;
    hipFuncSetAttribute((const void*)k_co, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const char* nm[6] = {"idle", "mfma", "valu fma", "lds read", "global store", "valu exp/log"};
    struct Cfg { int a, b, ia, ib; const char* unit; double per_iter; };
    // instruction counts per iteration of role B: fma 64, lds 16, store 16, transcendental 4 outputs
    const int IM = 4000;
    for (int rb = 2; rb <= 5; ++rb) {
        const int ib = rb == 2 ? 4000 : rb == 3 ? 8000 : rb == 4 ? 2000 : 2000;
        const double per = rb == 2 ? 64 : rb == 3 ? 16 : rb == 4 ? 16 : 4;
        for (int ra = 0; ra <= 1; ++ra) {
            // size the MFMA stream so it outlasts the other role
            hipLaunchKernelGGL(k_co, dim3(512), dim3(256), lds, 0, ra, rb, ra ? IM * 8 : 0, out, clk);   // warm
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_co, dim3(512), dim3(256), lds, 0, ra, rb, 0, out, clk);
            std::vector<long long> h(512);
            // role A iters and role B iters differ: launch with per-role iteration counts folded into one argument is not possible, so run B's count
            hipLaunchKernelGGL(k_co, dim3(512), dim3(256), lds, 0, ra, rb, ib, out, clk); hipDeviceSynchronize();
            hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
            const double ca = avg(h, 0), cb = avg(h, 256);
            printf("A=%-5s B=%-13s: B %.1f cycles per %s", nm[ra], nm[rb], cb / (ib * per), rb == 5 ? "output (exp+div+2 log)" : "instruction");
            if (ra) printf("   | A: %.1f cycles per MFMA (64 = full rate)", ca / (ib * 16.0));
            printf("\n");
        }
    }
    return 0;
}
