# timing experiment (WRONG RESULTS on purpose): k_dec_b4 without the 32 -> 1 tap contraction (no 4x4x1 MFMAs; the H planes get the ReLU'd accumulators' first registers)
PATCH = {'decoder.hip': [
    ("                    Tq[0][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[2 * ph][e], Tq[0][kh], 0, 0, 0);\n                    Tq[1][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[2 * ph + 1][e], Tq[1][kh], 0, 0, 0);",
     "                    if (e < 4) { Tq[0][kh][e] = acc[2 * ph][e + 4 * kh] * w4g[kh][e]; Tq[1][kh][e] = acc[2 * ph + 1][e + 4 * kh] * w4g[kh][e]; }"),
]}
