# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the deferred gather
PATCH = {'bf16x3.hip': [
    ("                if (s > 0) {                                      // the previous strip's gather, a piece per step",
     "                if (s > 100) {"),
    ("            if (s > 0) g_store(oh0);", "            if (s > 100) g_store(oh0);"),
]}
