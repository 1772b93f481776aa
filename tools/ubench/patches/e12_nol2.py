# timing experiment (WRONG RESULTS on purpose): k_conv_e12 without its L2 phase contraction (one tap instead of nine)
PATCH = {'generic_enc.hip': [
    ("""            tap_loop_kc_pd<1, 1, 4, E12_PD>(acc1, 9, Wl, sx, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
                wt = t; bs[0] = base_of(t); sw[0] = 0;
            }, PackedWIdx{1, 4, 0});
            if (pvx) {
                float* op = dst + ((size_t)(y0 + yl) * W2 + x2) * 32 + 4 * h;""", """            tap_loop_kc_pd<1, 1, 4, E12_PD>(acc1, 1, Wl, sx, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
                wt = t; bs[0] = base_of(t); sw[0] = 0;
            }, PackedWIdx{1, 4, 0});
            if (pvx) {
                float* op = dst + ((size_t)(y0 + yl) * W2 + x2) * 32 + 4 * h;"""),
]}
