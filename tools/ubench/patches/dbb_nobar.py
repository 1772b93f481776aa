# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 with the slab barrier only in front of step 0
PATCH = {'bf16x3.hip': [
    ("                glds_drain();                                     // this wave's pieces of slab ks have landed ...\n                __syncthreads();                                  // ... and everybody's (ks = 0: the staged strip too); nobody reads the other buffer any more",
     "                if (ks == 0) { glds_drain(); __syncthreads(); }"),
]}
