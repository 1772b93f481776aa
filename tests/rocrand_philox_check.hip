// Test program (not product code): the engine's noise words (csrc/philox.h) against rocRAND's own Philox4x32-10 engine
// (rocrand/rocrand_philox4x32_10.h: rocrand_state_philox4x32_10) for the same key and counter.
//
// north_star names rocRAND for the MC-dropout masks; the engine owns its generator instead (the CPU oracle has to
// mirror it bit for bit and every draw has to be addressable by its logical identity).  This shows it is the SAME generator:
//   rocrand_init(seed, subsequence, offset)  puts  offset / 4  into counter words (x, y)  and  subsequence  into (z, w), key = seed
// so   noise_words(k0, k1, tag, blk, row, stream, stage)  ==  rocrand4( state(seed = k0 | k1 << 32,
//                                                                            subsequence = stream | stage << 32,
//                                                                            offset = 4 * ((blk | tag << 16) | row << 32)) )
// (rows < 2^30 through this API: the offset is a count of 32-bit numbers in 64 bits; the engine's counter takes the full 32 bits).
//
//   rocrand_philox_check            device: efe::noise_words vs the rocRAND device engine vs the rocRAND host engine, exit 0 / 1
//   rocrand_philox_check --host     no GPU: prints "c0 c1 c2 c3 k0 k1 w0 w1 w2 w3" of the rocRAND HOST engine for the same counter list
//                                   (tests/test_noise_statistics.py compares them with oracle/philox.py)
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "philox.h"

struct Draw { uint32_t tag, blk, row, stream, stage, k0, k1; };

static std::vector<Draw> draws() {
    std::vector<Draw> v;
    // the Random123 known-answer counters first (all zero / all ones / pi digits), then draws shaped like the engine's
    v.push_back({0, 0, 0, 0, 0, 0, 0});
    v.push_back({0xFFFF, 0xFFFF, 0x3FFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF});
    uint64_t s = 0x243F6A8885A308D3ull;
    auto next = [&s]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    const uint32_t tags[5] = {efe::TAG_MID, efe::TAG_DEC, efe::TAG_ENC, efe::TAG_EPS, efe::TAG_ACT};
    for (int i = 0; i < 4094; ++i) {
        Draw d;
        d.tag = tags[next() % 5] + next() % 4;
        d.blk = next() % 221;                       // 28224 features / 128 = the largest mask of BASELINE configs[4]
        d.row = next() % (1u << 30);
        d.stream = efe::stream_id(next() % 9, next() % 64);
        d.stage = next();
        d.k0 = next() * 2u + (next() & 1u);
        d.k1 = (i & 1) ? 0u : next();
        v.push_back(d);
    }
    return v;
}

__host__ __device__ static inline uint4 rocrand_words(const Draw d) {
    const unsigned long long seed = (unsigned long long)d.k0 | ((unsigned long long)d.k1 << 32);
    const unsigned long long subsequence = (unsigned long long)d.stream | ((unsigned long long)d.stage << 32);
    const unsigned long long counter_lo = (unsigned long long)(d.blk | (d.tag << 16)) | ((unsigned long long)d.row << 32);
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subsequence, 4ull * counter_lo, &st);
    return rocrand4(&st);
}

__global__ void k_check(const Draw* d, int n, uint4* ours, uint4* theirs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ours[i] = efe::noise_words(d[i].k0, d[i].k1, d[i].tag, d[i].blk, d[i].row, d[i].stream, d[i].stage);
    theirs[i] = rocrand_words(d[i]);
}

int main(int argc, char** argv) {
    std::vector<Draw> v = draws();
    const int n = (int)v.size();
    if (argc > 1 && !strcmp(argv[1], "--host")) {
        for (const Draw& d : v) {
            const uint4 w = rocrand_words(d);
            printf("%u %u %u %u %u %u %u %u %u %u\n", d.blk | (d.tag << 16), d.row, d.stream, d.stage, d.k0, d.k1, w.x, w.y, w.z, w.w);
        }
        return 0;
    }
    Draw* dd; uint4 *ours, *theirs;
    if (hipMalloc(&dd, n * sizeof(Draw)) != hipSuccess || hipMalloc(&ours, n * sizeof(uint4)) != hipSuccess ||
        hipMalloc(&theirs, n * sizeof(uint4)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    if (hipMemcpy(dd, v.data(), n * sizeof(Draw), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); return 2; }
    k_check<<<(n + 255) / 256, 256>>>(dd, n, ours, theirs);
    std::vector<uint4> a(n), b(n);
    if (hipMemcpy(a.data(), ours, n * sizeof(uint4), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(b.data(), theirs, n * sizeof(uint4), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const uint4 h = rocrand_words(v[i]);
        const bool ok = a[i].x == b[i].x && a[i].y == b[i].y && a[i].z == b[i].z && a[i].w == b[i].w &&
                        a[i].x == h.x && a[i].y == h.y && a[i].z == h.z && a[i].w == h.w;
        if (!ok && bad++ < 5)
            fprintf(stderr, "draw %d: engine %08x %08x %08x %08x  rocrand(device) %08x %08x %08x %08x  rocrand(host) %08x %08x %08x %08x\n", i,
                    a[i].x, a[i].y, a[i].z, a[i].w, b[i].x, b[i].y, b[i].z, b[i].w, h.x, h.y, h.z, h.w);
    }
    // Random123 known answer: counter 0, key 0
    if (!(a[0].x == 0x6627e8d5u && a[0].y == 0xe169c58du && a[0].z == 0xbc57ac4cu && a[0].w == 0x9b00dbd8u)) { fprintf(stderr, "known answer (0, 0) differs\n"); ++bad; }
    if (bad) { fprintf(stderr, "%d of %d draws differ\n", bad, n); return 1; }
    printf("rocrand_philox_check OK: %d draws, engine == rocRAND device engine == rocRAND host engine\n", n);
    return 0;
}
