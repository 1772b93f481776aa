# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the 4x4x1 tap contraction
PATCH = {'bf16x3.hip': [
    ("                        Tq[0][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[0][e], Tq[0][kh], 0, 0, 0);\n                        Tq[1][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[1][e], Tq[1][kh], 0, 0, 0);",
     "                        if (e == 0) { Tq[0][kh][0] += acc[0][kh]; Tq[1][kh][1] += acc[1][kh + 4]; }"),
]}
