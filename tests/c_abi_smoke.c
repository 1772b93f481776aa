/* Plain-C consumer of include/efe_engine.h: no Python, no torch -- the drop-in boundary is a C ABI.
 * Build:  gcc tests/c_abi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L<pkg> -lefe_mi355x -L/opt/rocm/lib -lamdhip64 -lm
 * Loads random weights of the reference architecture (state_dict key names of SURVEY 8b), runs ModelDown.decoder,
 * calculate_G and the action posterior on device buffers, checks ranges / determinism.
 *
 *   c_abi_smoke <weights.bin> <tests/golden/calcG_m4s1_g115.bin>
 * additionally loads a weight file (records: key\0, int32 ndim, int64 shape[ndim], float32 data -- written by the test from the synthetic weights
 * the fixture was captured with) and compares efe_calculate_g, with the fixture's injected normals, against the REFERENCE's values in the committed
 * blob (tests/golden/make_c_blob.py: derived from calcG_m4s1_g115.npz, i.e. from /root/reference/src/torchmodel.py:270-300 itself). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "efe_engine.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "FAIL %s -> %d (%s)\n", #x, rc_, efe_last_error(ctx)); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned long long s_ = 88172645463325252ULL;
static float urand(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (float)((s_ >> 40) / 16777216.0) * 2.f - 1.f; }

static int set_w(efe_ctx* ctx, const char* key, int64_t d0, int64_t d1, int64_t d2, int64_t d3, float bound) {
    int64_t shape[4] = {d0, d1, d2, d3};
    int nd = d3 ? 4 : d2 ? 3 : d1 ? 2 : 1;
    size_t n = 1;
    for (int i = 0; i < nd; ++i) n *= (size_t)shape[i];
    float* h = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) h[i] = urand() * bound;
    int rc = efe_set_weight(ctx, key, h, shape, nd);
    free(h);
    return rc;
}

static int lin(efe_ctx* ctx, const char* name, int out, int in) {
    char k[96];
    snprintf(k, sizeof k, "%s.weight", name);
    if (set_w(ctx, k, out, in, 0, 0, sqrtf(3.f / in))) return 1;
    snprintf(k, sizeof k, "%s.bias", name);
    return set_w(ctx, k, out, 0, 0, 0, 0.1f);
}

static int conv(efe_ctx* ctx, const char* name, int d0, int d1, int nbias, float fan) {
    char k[96];
    snprintf(k, sizeof k, "%s.weight", name);
    if (set_w(ctx, k, d0, d1, 3, 3, sqrtf(3.f / fan))) return 1;
    snprintf(k, sizeof k, "%s.bias", name);
    return set_w(ctx, k, nbias, 0, 0, 0, 0.1f);
}

/* the reference-captured fixture through the plain C ABI: 0 = equal within the tolerances of tests/test_gpu_parity.py */
static int fixture_check(efe_ctx* ctx, const char* wpath, const char* bpath) {
    FILE* f = fopen(wpath, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", wpath); return 1; }
    int nw = 0;
    for (;;) {
        char key[128]; int k = 0, ch;
        while ((ch = fgetc(f)) != EOF && ch != 0 && k < 127) key[k++] = (char)ch;
        if (ch == EOF) break;
        key[k] = 0;
        int32_t nd; int64_t shape[4]; size_t n = 1;
        if (fread(&nd, 4, 1, f) != 1 || nd < 1 || nd > 4 || fread(shape, 8, (size_t)nd, f) != (size_t)nd) { fprintf(stderr, "bad weight record %s\n", key); return 1; }
        for (int i = 0; i < nd; ++i) n *= (size_t)shape[i];
        float* h = (float*)malloc(n * sizeof(float));
        if (fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short weight record %s\n", key); return 1; }
        CHECK(efe_set_weight(ctx, key, h, shape, nd));
        free(h); ++nw;
    }
    fclose(f);
    CHECK(efe_commit_weights(ctx));
    f = fopen(bpath, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", bpath); return 1; }
    int32_t hdr[4]; uint64_t seed;                       /* M, S, stage, reserved; noise seed */
    if (fread(hdr, 4, 4, f) != 4 || fread(&seed, 8, 1, f) != 1) return 1;
    const int M = hdr[0], S = hdr[1];
    const size_t n_in = (size_t)M * 10 + (size_t)M * 4 + (size_t)3 * S * M * 10, n_out = (size_t)7 * M;      /* G, t0, t1, t2, t2_1, t2_2, (unused) */
    float* in = (float*)malloc(n_in * 4); float* ex = (float*)malloc(n_out * 4);
    if (fread(in, 4, n_in, f) != n_in || fread(ex, 4, 6 * (size_t)M, f) != 6 * (size_t)M) { fprintf(stderr, "short blob\n"); return 1; }
    fclose(f);
    float *ds, *dpi, *deps, *dG, *dT, *dmean, *dparts;
    HIP(hipMalloc((void**)&ds, M * 40)); HIP(hipMalloc((void**)&dpi, M * 16)); HIP(hipMalloc((void**)&deps, (size_t)3 * S * M * 40));
    HIP(hipMalloc((void**)&dG, M * 4)); HIP(hipMalloc((void**)&dT, 3 * M * 4)); HIP(hipMalloc((void**)&dmean, M * 40)); HIP(hipMalloc((void**)&dparts, 2 * M * 4));
    HIP(hipMemcpy(ds, in, M * 40, hipMemcpyHostToDevice)); HIP(hipMemcpy(dpi, in + M * 10, M * 16, hipMemcpyHostToDevice));
    HIP(hipMemcpy(deps, in + M * 14, (size_t)3 * S * M * 40, hipMemcpyHostToDevice));
    efe_noise nz = {seed, (uint32_t)hdr[2], 0u, 0u, 0u};
    CHECK(efe_calculate_g(ctx, ds, dpi, M, S, 0, &nz, deps, dG, dT, NULL, dmean, NULL, dparts, NULL));
    float got[7 * 64];
    HIP(hipMemcpy(got, dG, M * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(got + M, dT, 3 * M * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(got + 4 * M, dparts, 2 * M * 4, hipMemcpyDeviceToHost));
    float t21 = 1.f;
    for (int i = 0; i < M; ++i) if (fabsf(ex[4 * M + i]) > t21) t21 = fabsf(ex[4 * M + i]);
    const float gtol = 1e-6f * t21 + 5e-4f;               /* tests/test_gpu_parity.py::gtol */
    const char* names[6] = {"G", "term0", "term1", "term2", "term2_1", "term2_2"};
    for (int q = 0; q < 6; ++q)
        for (int i = 0; i < M; ++i) {
            const float e = ex[q * M + i], g = got[q * M + i];
            const float tol = (q == 0 || q == 3) ? gtol : (q == 1 || q == 2) ? 1e-4f : 1e-6f * fabsf(e) + 1e-3f;
            if (!(fabsf(g - e) <= tol)) { fprintf(stderr, "fixture: %s[%d] = %.7g, reference %.7g (tol %g)\n", names[q], i, g, e, tol); return 1; }
        }
    printf("c_abi_smoke fixture OK: %d weight tensors, calculate_G(M=%d, S=%d) == reference capture, G[0]=%.6f (reference %.6f)\n", nw, M, S, got[0], ex[0]);
    hipFree(ds); hipFree(dpi); hipFree(deps); hipFree(dG); hipFree(dT); hipFree(dmean); hipFree(dparts); free(in); free(ex);
    return 0;
}

int main(int argc, char** argv) {
    efe_ctx* ctx = NULL;
    if (efe_abi_version() != 6) { fprintf(stderr, "abi version\n"); return 1; }
    if (efe_create(&ctx, 0)) { fprintf(stderr, "efe_create failed (no HIP device?)\n"); return 2; }
    CHECK(lin(ctx, "top.qpi_net.0", 128, 10)); CHECK(lin(ctx, "top.qpi_net.2", 128, 128)); CHECK(lin(ctx, "top.qpi_net.4", 4, 128));
    CHECK(lin(ctx, "mid.ps_net.0", 512, 14)); CHECK(lin(ctx, "mid.ps_net.3", 512, 512)); CHECK(lin(ctx, "mid.ps_net.6", 512, 512));
    CHECK(lin(ctx, "mid.ps_net.9", 20, 512));
    CHECK(conv(ctx, "down.qs_net.0", 32, 1, 32, 9)); CHECK(conv(ctx, "down.qs_net.2", 32, 32, 32, 288));
    CHECK(conv(ctx, "down.qs_net.4", 64, 32, 64, 288)); CHECK(conv(ctx, "down.qs_net.6", 64, 64, 64, 576));
    CHECK(lin(ctx, "down.qs_net.9", 256, 576)); CHECK(lin(ctx, "down.qs_net.12", 256, 256)); CHECK(lin(ctx, "down.qs_net.15", 256, 256));
    CHECK(lin(ctx, "down.qs_net.18", 20, 256));
    CHECK(lin(ctx, "down.po_net.0", 256, 10)); CHECK(lin(ctx, "down.po_net.3", 256, 256)); CHECK(lin(ctx, "down.po_net.6", 256, 256));
    CHECK(lin(ctx, "down.po_net.9", 16384, 256));
    CHECK(conv(ctx, "down.po_net.13", 64, 64, 64, 576)); CHECK(conv(ctx, "down.po_net.15", 64, 64, 64, 144));
    CHECK(conv(ctx, "down.po_net.17", 64, 32, 32, 144)); CHECK(conv(ctx, "down.po_net.19", 32, 1, 1, 288));
    CHECK(efe_commit_weights(ctx));

    enum { M = 8, S = 3 };
    float hs[M * 10], hpi[M * 4];
    for (int i = 0; i < M * 10; ++i) hs[i] = urand();
    memset(hpi, 0, sizeof hpi);
    for (int i = 0; i < M; ++i) hpi[i * 4 + (i & 3)] = 1.f;
    float *ds, *dpi, *dpo, *dG, *dT, *dps1, *dmean, *dpo1, *dP, *dlogP;
    HIP(hipMalloc((void**)&ds, sizeof hs)); HIP(hipMalloc((void**)&dpi, sizeof hpi));
    HIP(hipMalloc((void**)&dpo, M * 4096 * 4)); HIP(hipMalloc((void**)&dG, M * 4)); HIP(hipMalloc((void**)&dT, 3 * M * 4));
    HIP(hipMalloc((void**)&dps1, M * 40)); HIP(hipMalloc((void**)&dmean, M * 40)); HIP(hipMalloc((void**)&dpo1, M * 4096 * 4));
    HIP(hipMalloc((void**)&dP, M * 4)); HIP(hipMalloc((void**)&dlogP, M * 4));
    HIP(hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice)); HIP(hipMemcpy(dpi, hpi, sizeof hpi, hipMemcpyHostToDevice));

    efe_noise nz = {1234u, 0u, 1u, 0u, 0u};
    CHECK(efe_decoder(ctx, ds, M, &nz, dpo, NULL));
    float* hpo = (float*)malloc(M * 4096 * 4);
    HIP(hipMemcpy(hpo, dpo, M * 4096 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < M * 4096; ++i) if (!(hpo[i] >= 0.f && hpo[i] <= 1.f)) { fprintf(stderr, "decoder output out of [0,1]\n"); return 1; }

    float hG[2][M];
    for (int rep = 0; rep < 2; ++rep) {          /* same keys -> bit-identical results */
        CHECK(efe_calculate_g(ctx, ds, dpi, M, S, 0, &nz, NULL, dG, dT, dps1, dmean, dpo1, NULL, NULL));
        HIP(hipMemcpy(hG[rep], dG, M * 4, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < M; ++i) {
        if (!isfinite(hG[0][i]) || hG[0][i] != hG[1][i]) { fprintf(stderr, "calculate_G not finite/deterministic\n"); return 1; }
    }
    {   /* liveness mask: entry 0 (rows 0..3) dead, the other rows must not change */
        unsigned char hmask[M / 4], *dmask;
        for (int g = 0; g < M / 4; ++g) hmask[g] = g != 0;
        HIP(hipMalloc((void**)&dmask, M / 4));
        HIP(hipMemcpy(dmask, hmask, M / 4, hipMemcpyHostToDevice));
        efe_rows rows = {dmask, NULL, 4, M / 4};    /* the mask is an ARGUMENT of the call (ABI 4); n_total = its length (ABI 5) ... */
        CHECK(efe_calculate_g_rows(ctx, ds, dpi, M, S, 0, &nz, NULL, &rows, dG, dT, dps1, dmean, dpo1, NULL, NULL));
        float hGm[M];
        HIP(hipMemcpy(hGm, dG, M * 4, hipMemcpyDeviceToHost));
        for (int i = 4; i < M; ++i) if (hGm[i] != hG[0][i]) { fprintf(stderr, "row mask changed a live row\n"); return 1; }
        /* ... so a plain call on the same context right after it sees no mask: row 0 is computed again */
        CHECK(efe_calculate_g(ctx, ds, dpi, M, S, 0, &nz, NULL, dG, dT, dps1, dmean, dpo1, NULL, NULL));
        HIP(hipMemcpy(hGm, dG, M * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < M; ++i) if (hGm[i] != hG[0][i]) { fprintf(stderr, "an unmasked call saw the previous call's mask\n"); return 1; }
        {   /* a compacted call: entries 1 .. M/4-1 as a dense batch that still draws the noise of the rows it holds (efe_rows.ids) */
            int hids[M / 4], *dids;
            for (int g = 1; g < M / 4; ++g) hids[g - 1] = g;
            HIP(hipMalloc((void**)&dids, sizeof(hids)));
            HIP(hipMemcpy(dids, hids, sizeof(hids), hipMemcpyHostToDevice));
            efe_rows crows = {NULL, dids, 4, M / 4};
            CHECK(efe_calculate_g_rows(ctx, ds + 4 * 10, dpi + 4 * 4, M - 4, S, 0, &nz, NULL, &crows, dG, dT, dps1, dmean, dpo1, NULL, NULL));
            HIP(hipMemcpy(hGm, dG, (M - 4) * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < M - 4; ++i) if (hGm[i] != hG[0][i + 4]) { fprintf(stderr, "compacted rows differ from the full batch\n"); return 1; }
            /* n_total is the bound of the ids: with the development option check_rows a stale id is an ERROR, not an out-of-bounds read */
            efe_rows brows = {NULL, dids, 4, 1};          /* ids hold entry 1 (= M/4 - 1 for M = 8): outside [0, 1) */
            CHECK(efe_set_option(ctx, "check_rows", 1));
            if (efe_calculate_g_rows(ctx, ds + 4 * 10, dpi + 4 * 4, M - 4, S, 0, &nz, NULL, &brows, dG, dT, dps1, dmean, dpo1, NULL, NULL) == 0 ||
                !strstr(efe_last_error(ctx), "outside [0, n_total")) { fprintf(stderr, "check_rows accepted an id >= n_total (%s)\n", efe_last_error(ctx)); return 1; }
            CHECK(efe_calculate_g_rows(ctx, ds + 4 * 10, dpi + 4 * 4, M - 4, S, 0, &nz, NULL, &crows, dG, dT, dps1, dmean, dpo1, NULL, NULL));
            CHECK(efe_set_option(ctx, "check_rows", 0));
            efe_rows trows = {dmask, NULL, 4, 1};         /* more entries in the call than the batch is said to have */
            if (efe_calculate_g_rows(ctx, ds, dpi, M, S, 0, &nz, NULL, &trows, dG, dT, dps1, dmean, dpo1, NULL, NULL) == 0) { fprintf(stderr, "n_total < entries accepted\n"); return 1; }
            HIP(hipFree(dids));
        }
        /* ABI 6: the context-state shim efe_set_row_mask is gone -- the masked call above left no state behind: a plain call is complete */
        /* ... and a handle that is not a live context is refused by every entry point instead of dereferenced (the registry behind efe_ctx_alive) */
        if (!efe_ctx_alive(ctx) || efe_ctx_alive(NULL) || efe_ctx_alive((const efe_ctx*)hG)) { fprintf(stderr, "efe_ctx_alive wrong\n"); return 1; }
        if (efe_calculate_g((efe_ctx*)hG, ds, dpi, M, S, 0, &nz, NULL, dG, dT, dps1, dmean, dpo1, NULL, NULL) == 0) { fprintf(stderr, "a made-up handle was accepted\n"); return 1; }
        CHECK(efe_calculate_g(ctx, ds, dpi, M, S, 0, &nz, NULL, dG, dT, dps1, dmean, dpo1, NULL, NULL));     /* dG complete again for the posterior */
        HIP(hipFree(dmask));
    }
    CHECK(efe_action_posterior(ctx, dG, M / 4, 4, 10.0f, dP, dlogP, NULL));
    float hP[M];
    HIP(hipMemcpy(hP, dP, M * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < M / 4; ++g) {
        float t = hP[4 * g] + hP[4 * g + 1] + hP[4 * g + 2] + hP[4 * g + 3];
        if (fabsf(t - 1.f) > 1e-5f) { fprintf(stderr, "posterior does not sum to 1\n"); return 1; }
    }
    /* error path: bad argument -> non-zero + message, context still usable */
    if (efe_decoder(ctx, ds, 0, &nz, dpo, NULL) == 0) { fprintf(stderr, "M = 0 accepted\n"); return 1; }
    if (strlen(efe_last_error(ctx)) == 0) { fprintf(stderr, "no error message\n"); return 1; }
    CHECK(efe_decoder(ctx, ds, M, &nz, dpo, NULL));
    printf("c_abi_smoke OK  G[0]=%g  macs(last call)=%lld\n", hG[0][0], (long long)efe_last_call_macs(ctx));
    if (argc > 2 && fixture_check(ctx, argv[1], argv[2])) return 1;
    efe_destroy(ctx);
    return 0;
}
