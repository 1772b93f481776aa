# timing experiment (WRONG RESULTS on purpose): k_dec_b4 without the horizontal pre-sums and H-plane writes (tap MFMAs kept, results folded into one register)
PATCH = {'decoder.hip': [
    ("""                float l = wave_shr1(Tq[1][kh][2]), r = wave_shl1(Tq[0][kh][0]);
                l = (j == 0) ? 0.f : l;                       // lane 32 received lane 31 (the other channel half's pixel 31): image edge
                r = (j == 31) ? 0.f : r;
                float2 eo;
                eo.x = (Tq[1][kh][0] + Tq[0][kh][1]) + l;
                eo.y = (r + Tq[1][kh][1]) + Tq[0][kh][2];
                *reinterpret_cast<float2*>(hp + kh * 64) = eo;""",
     """                part += (Tq[1][kh][0] + Tq[0][kh][1]) * 1e-30f; (void)hp;"""),
]}
