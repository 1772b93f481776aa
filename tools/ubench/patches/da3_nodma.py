# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 without the per-tap weight DMA (only the image's first slab is copied)
PATCH = {'bf16x3.hip': [("                        if (next != nullptr && (q == 3 || q == 9)) {", "                        if (next != nullptr && T < 0 && (q == 3 || q == 9)) {")]}
