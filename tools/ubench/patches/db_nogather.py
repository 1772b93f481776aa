# timing experiment (WRONG RESULTS on purpose): k_dec_b4 without the deferred gather (no H-plane reads, sigmoid, entropy / reward terms, image stores)
PATCH = {'decoder.hip': [
    ("                if (gq) {\n                    if (kc == 0) { g_load(oh0, 0, 2 * SR * (s - 1), hbp); g_load(oh1, 1, 2 * SR * (s - 1), hbp); }", "                if (false) {\n                    if (kc == 0) { g_load(oh0, 0, 2 * SR * (s - 1), hbp); g_load(oh1, 1, 2 * SR * (s - 1), hbp); }"),
    ("                if (kc == 1 && gq) g_term(oh1, 1);", "                if (kc == 1 && gq) part += acc[1][0] * 1e-30f;"),
    ("        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }     // the deferred rows' pixels (stores behind the prefetch loads)", "        // (no deferred stores)"),
    ("                g_load(oh, 0, 2 * SR * s, hb); g_sig(oh, 0); g_term(oh, 0); g_store(oh, 0);", "                part += (float)oh * 1e-30f;"),
]}
