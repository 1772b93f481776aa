# timing experiment (WRONG RESULTS on purpose): k_dec_b4 with the input strip staged ONCE per image (no per-strip HBM prefetch, no ds_write of the strip)
PATCH = {'decoder.hip': [
    ("            smv[(rl * 32 + ix) * DB_PS + c4] = (SR * s + rl < 32) ? pf[it] : (f32x4)(0.f);", "            if (s == s_lo) smv[(rl * 32 + ix) * DB_PS + c4] = (SR * s + rl < 32) ? pf[it] : (f32x4)(0.f);"),
    ("                if (kc == 6) prefetch((s < NS - 1) ? s + 1 : NS - 1, tl);       // the next strip's input, behind this strip's last weight-fragment request", "                // (no prefetch)"),
]}
