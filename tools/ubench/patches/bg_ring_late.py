# timing experiment: k_dec_bg with all eight ring writes behind the last tap MFMA group
PATCH = {'generic_dec.hip': [
    ("                    for (int i = 0; i < 4; ++i) sm[wrap1(nb + ppt + 16 * i, RPa) * PS4 + c4] = pf[i];", "                    ;"),
    ("                    for (int i = 4; i < 8; ++i) sm[wrap1(nb + ppt + 16 * i, RPa) * PS4 + c4] = pf[i];", "                    for (int i = 0; i < 8; ++i) sm[wrap1(nb + ppt + 16 * i, RPa) * PS4 + c4] = pf[i];"),
]}
