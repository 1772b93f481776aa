# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the deferred gather
PATCH = {'bf16x3.hip': [
    ("                if (s > 0 && ph == 0) {\n                    if (ks == 1)", "                if (s > 100 && ph == 0) {\n                    if (ks == 1)"),
    ("            if (s > 0 && ph == 0) { g_store(oh0, 0); g_store(oh1, 1); }", "            if (s > 100 && ph == 0) { g_store(oh0, 0); g_store(oh1, 1); }"),
]}
