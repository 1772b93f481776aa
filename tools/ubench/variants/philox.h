// Counter-based noise for MC-dropout masks, reparameterisation normals and action sampling.
//
// Replaces the reference's use of torch's global generator (nn.Dropout(0.5) at
// /root/reference/src/torchmodel.py:44,47,50,96,99,102,109,112,115,118; torch.randn_like at :55,131;
// torch.multinomial at :364,379).  Every draw is addressed by its logical identity, so results do not
// depend on batching, launch geometry or GPU count:
//
//   key     = (seed & 0xffffffff, seed >> 32)
//   counter = (blk | tag << 16, global_row, pass << 16 | sample, stage)
//
//   dropout : feature f of a row is KEPT iff bit (f & 31) of word ((f >> 5) & 3) of counter blk = f >> 7
//   normals : element k = Box-Muller lane (k & 3) of counter blk = k >> 2
//   uniform : word 0 of blk 0
//
// The CPU mirror used by the parity tests is oracle/philox.py (checked against Random123 known answers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace efe {

enum : uint32_t { TAG_MID = 0x10, TAG_DEC = 0x20, TAG_ENC = 0x30, TAG_EPS = 0x40, TAG_ACT = 0x50 };
enum : uint32_t { PASS_T1 = 0, PASS_D1 = 1, PASS_E1 = 2, PASS_T2 = 3, PASS_D2A = 4, PASS_D2B = 5,
                  PASS_ROOT = 6, PASS_HABIT = 7, PASS_SIM = 8 };

__host__ __device__ inline uint32_t stream_id(uint32_t pass, uint32_t sample) {
    return ((pass & 0xFFFFu) << 16) | (sample & 0xFFFFu);
}

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += W0;
        k1 += W1;
    }
    return c;
}

__device__ __forceinline__ uint4 noise_words(uint32_t k0, uint32_t k1, uint32_t tag, uint32_t blk, uint32_t row,
                                             uint32_t stream, uint32_t stage) {
    return philox4x32_10(make_uint4(blk | (tag << 16), row, stream, stage), k0, k1);
}

__device__ __forceinline__ float u01(uint32_t x) {
    return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f;   // 2^-24, result in (0,1)
}

// element k (0..) of the normal vector of one row
__device__ __forceinline__ float normal_elem(uint32_t k0, uint32_t k1, uint32_t row, uint32_t stream, uint32_t stage, int k) {
    const uint4 w = noise_words(k0, k1, TAG_EPS, (uint32_t)(k >> 2), row, stream, stage);
    const int lane = k & 3;
    const uint32_t a = (lane & 2) ? w.z : w.x;
    const uint32_t b = (lane & 2) ? w.w : w.y;
    const float r = sqrtf(-2.0f * logf(u01(a)));
    const float th = 6.283185307179586f * u01(b);
    return (lane & 1) ? r * sinf(th) : r * cosf(th);
}

}  // namespace efe
