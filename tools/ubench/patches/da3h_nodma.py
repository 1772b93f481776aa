# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 without the per-tap weight DMA (only the image's first slab is copied)
PATCH = {'bf16x3.hip': [("                        if (next != nullptr) {                  // DMA piece p of the next slab: behind MFMA MF / 2 + p / 3 of step p % 3", "                        if (next != nullptr && T < 0) {")]}
