# A/B build (correct results): k_dec_b4 / k_dec_a raise the wave priority (s_setprio 2) for their contraction loops and drop it for the staging / epilogue parts
PATCH = {'decoder.hip': [
    ("""        f32x16 acc[4], bias16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = sb3[2 * g4 + h];""", """        f32x16 acc[4], bias16;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = sb3[2 * g4 + h];"""),
    ("""        __syncthreads();
        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }     // the deferred rows' pixels (stores behind the prefetch loads)""", """        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }     // the deferred rows' pixels (stores behind the prefetch loads)"""),
]}
