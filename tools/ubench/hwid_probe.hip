// dev tool: which workgroups of a 512-WG persistent grid (256 threads, 66 KB LDS: 2 per CU) share a CU on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 2) k_probe(unsigned* out) {
    extern __shared__ float4 sm[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        sm[0] = make_float4(0, 0, 0, 0);
    }
    long long t0 = clock64();
    while (clock64() - t0 < 2000000) __builtin_amdgcn_s_sleep(64);     // keep every WG resident so all 512 coexist
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 8);
    const size_t lds = (257 * 16 + 32) * 16;
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_probe, dim3(512), dim3(256), lds, 0, d); hipDeviceSynchronize();
    std::vector<unsigned> h(1024); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(b);
        if (b < 24 || (b >= 256 && b < 272)) printf("block %3d: xcc %u se %u sh %u cu %2u wave %u simd %u\n", b, xcc, se, sh, cuid, hw & 0xf, (hw >> 4) & 3);
    }
    printf("%zu distinct CUs\n", cu.size());
    int shown = 0; std::map<int, int> diffs, cnt;
    for (auto& kv : cu) {
        cnt[(int)kv.second.size()]++;
        if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
        if (shown++ < 6) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    for (auto& kv : cnt) printf("CUs with %d WGs: %d\n", kv.first, kv.second);
    for (auto& kv : diffs) printf("partner distance %d: %d CUs\n", kv.first, kv.second);
    return 0;
}
