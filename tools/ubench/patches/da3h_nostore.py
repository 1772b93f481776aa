# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 without the y2 stores
PATCH = {'bf16x3.hip': [("                            yp[(mt * 4 + g4) * 512] = v;", "                            if (v.x == 123.456f) yp[(mt * 4 + g4) * 512] = v;")]}
