# timing experiment (WRONG RESULTS on purpose): k_convt_p without its output stores (one store per lane and pass kept so the accumulators stay live)
PATCH = {'generic_dec.hip': [
    ("""                        for (int i = 0; i < 4; ++i) {
                            const unsigned o = offs[i] + sbase;""", """                        for (int i = 0; i < 4; ++i) {
                            if (g4 != 3 || i != 3) { part_ += ac[0][4 * g4 + i] + ac[NAC - 1][4 * g4 + i]; continue; }
                            const unsigned o = offs[i] + sbase;"""),
    ("                if (co < a.Cout) {\n                    const unsigned sbase = (unsigned)(s * strip_floats + co) * 4u;", "                float part_ = 0.f;\n                if (co < a.Cout) {\n                    const unsigned sbase = (unsigned)(s * strip_floats + co) * 4u;"),
    ("                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, o, P == 2 ? 0u : (unsigned)((P * a.Wout + pw) * a.ldo) * 4u, 0);",
     "                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + part_ * 1e-30f), yr, o, P == 2 ? 0u : (unsigned)((P * a.Wout + pw) * a.ldo) * 4u, 0);"),
]}
