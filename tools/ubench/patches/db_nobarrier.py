# timing experiment (WRONG RESULTS on purpose): k_dec_b4 without the two barriers of a strip
PATCH = {'decoder.hip': [
    ("        float4 a0 = wf(4, 0), a1 = wf(5, 0), a2 = wf(7, 0), a3 = wf(8, 0);      // the strip's first weight fragments: in flight across the barrier\n        __syncthreads();",
     "        float4 a0 = wf(4, 0), a1 = wf(5, 0), a2 = wf(7, 0), a3 = wf(8, 0);      // the strip's first weight fragments: in flight across the barrier"),
    ("        __syncthreads();\n        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }", "        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }"),
]}
