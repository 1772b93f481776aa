# timing experiment (WRONG RESULTS on purpose): k_fc4_b3 without the weight-fragment stream from L2 (the first step's fragments are reused)
PATCH = {'bf16x3.hip': [
    ("return wfrag(wr, ln, (size_t)(((mt0 + mt) * 16 + ks) * 3 + p) * 64); };", "return wfrag(wr, ln, (size_t)(((mt0 + mt) * 16 + (ks & 0)) * 3 + p) * 64); };"),
]}
