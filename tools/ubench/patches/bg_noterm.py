# timing experiment (WRONG RESULTS on purpose): k_dec_bg without the sigmoid / entropy pieces of the deferred gather (loads and adds stay)
PATCH = {'generic_dec.hip': [
    ("                    if (kc == 4) g_sig();\n                    if (kc == 5) g_term();", "                    if (kc == 5) part += gv[0].x + gv[0].y + gv[C - 1].x;"),
]}
