# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the fragment reads of steps (fragments read once per kernel)
PATCH = {'bf16x3.hip': [
    ("        const int k = l / 3, p = l % 3;\n        if (k == 0) axn[p]", "        const int k = l / 3, p = l % 3;\n        if (bv[0] != nullptr) return;\n        if (k == 0) axn[p]"),
    ("    float4 ax[2][3], ay[2][3], bx[2][3], by[2][3];", "    float4 ax[2][3] = {}, ay[2][3] = {}, bx[2][3] = {}, by[2][3] = {};"),
]}
