# timing experiment (WRONG RESULTS on purpose): k_dec_bg without the deferred gather pieces
PATCH = {'generic_dec.hip': [
    ("                if (s > 0) {\n                    if (kc == 0) g_load(0, r0 - TH, hbp);", "                if (false) {\n                    if (kc == 0) g_load(0, r0 - TH, hbp);"),
    ("                g_store(0); g_store(1);                    // the previous strip's pixels (sigmoid pieces: view A section)", ""),
]}
