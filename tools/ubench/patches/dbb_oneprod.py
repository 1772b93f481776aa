# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 with one product per tap (same fragment reads)
PATCH = {'bf16x3.hip': [
    ("            const int pr = m >> 1;\n            if (!sameT) {", "            const int pr = m >> 1;\n            if (pr < SC::NPR - 1) {\n            } else if (!sameT) {"),
]}
