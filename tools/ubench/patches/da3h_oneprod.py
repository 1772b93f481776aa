# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 with one product per step (every LDS read and DMA piece kept)
PATCH = {'bf16x3.hip': [("                    const int pr = m / (2 * NTW), mt = (m / NTW) & 1, nt = m % NTW;\n                    acc[mt][nt] = SC::mfma(",
                         "                    const int pr = m / (2 * NTW), mt = (m / NTW) & 1, nt = m % NTW;\n                    if (pr == NPR - 1) acc[mt][nt] = SC::mfma(")]}
