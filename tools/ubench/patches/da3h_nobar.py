# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 without the barrier (and the DMA drain) in front of every tap
PATCH = {'bf16x3.hip': [("            glds_drain();                                 // this wave's pieces of slab T have landed ...\n            __syncthreads();                              // ... and everybody's; nobody reads buffer (T + 1) & 1 any more\n", "")]}
