# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 with one of the six products per tap (same fragment reads)
PATCH = {'bf16x3.hip': [
    ("            const int pr = m >> 1;\n            if (!sameT) {", "            const int pr = m >> 1;\n            if (pr < 5) { if (i + 1 < NP) ld(i + 1, m, ax[n], ay[n], bx[n], by[n]); continue; }\n            if (!sameT) {"),
]}
