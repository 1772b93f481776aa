# timing experiment (WRONG RESULTS on purpose: k_dec_bg still reads NHWC): k_convt_12's layer-2 passes with channels as the MFMA rows and y2 written
# parity / channel-group blocked, [ph][pw][8 channel groups][input position][8 channels] -- every store instruction writes 1 KiB contiguous (16 per
# strip and wave instead of 64 dword stores).  Bounds what the store side of a blocked y2 could gain before k_dec_bg's fetch is re-mapped.
PATCH = {'generic_dec.hip': [
    ("""        f32x16 ac[NAC];
#pragma unroll
        for (int p = 0; p < NAC; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) ac[p][e] = bias;
        auto step = [&](float4 (&av)[NMP], float4 (&bv)[NVP], int kc) {
#pragma unroll
            for (int m = 0; m < NMP; ++m) {
                const float4 b = bv[pvw[P][m]];
                f32x16& c = ac[pac[P][m]];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, av[m].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, av[m].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, av[m].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, av[m].w, c, 0, 0, 0);""",
     """        f32x16 ac[NAC];
#pragma unroll
        for (int p = 0; p < NAC; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) ac[p][e] = P == 2 ? bias : bias16[e];
        auto step = [&](float4 (&av)[NMP], float4 (&bv)[NVP], int kc) {
#pragma unroll
            for (int m = 0; m < NMP; ++m) {
                const float4 b = bv[pvw[P][m]];
                f32x16& c = ac[pac[P][m]];
                if (P == 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, av[m].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, av[m].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, av[m].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, av[m].w, c, 0, 0, 0);
                } else {
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, b.x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, b.y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, b.z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, b.w, c, 0, 0, 0);
                }"""),
    ("""    const float bias1 = a.b1[co], bias2 = a.b2[co];""",
     """    const float bias1 = a.b1[co], bias2 = a.b2[co];
    float bias16[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bias16[e] = a.b2[mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];"""),
    ("""                    const unsigned sbase = (unsigned)(s * strip_floats + co) * 4u;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int4 off = *reinterpret_cast<const int4*>(cl_off + nt * 32 + 8 * g4 + 4 * h);
                        const unsigned offs[4] = {(unsigned)off.x, (unsigned)off.y, (unsigned)off.z, (unsigned)off.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned o = offs[k] + sbase;
#pragma unroll
                            for (int pw = 0; pw < 2; ++pw)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(ac[pw][4 * g4 + k], 0.0f)), yr, o, (unsigned)((P * Wout + pw) * 64) * 4u, 0);
                        }
                    }""",
     """                    // [ph = P][pw][cg = 4 mt + g4][pos = s SPX + q][8]: lane (j, h) writes 16 bytes at position q, channel half h
                    const unsigned o = qv && s * SPX + q < npix_img ? (unsigned)(((s * SPX + q) * 8 + 4 * h) * 4) : 0x80000000u;
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            u32x4g v;
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(unsigned, fmaxf(ac[pw][4 * g4 + k], 0.0f));
                            __builtin_amdgcn_raw_buffer_store_b128(v, yr, o, (unsigned)((((P * 2 + pw) * 8 + 4 * mt + g4) * npix_img) * 32), 0);
                        }"""),
]}
