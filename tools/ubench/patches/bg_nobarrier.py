# timing experiment (WRONG RESULTS on purpose): k_dec_bg without the two strip barriers
PATCH = {'generic_dec.hip': [
    ("        __syncthreads();                                   // barrier A:", "        // barrier A:"),
    ("        __syncthreads();                                   // barrier B:", "        // barrier B:"),
]}
