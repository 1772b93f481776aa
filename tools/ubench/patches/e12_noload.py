# timing experiment (WRONG RESULTS on purpose): k_conv_e12 with ONE conv1 operand request per tile instead of nine (the rest reuse it)
PATCH = {'generic_enc.hip': [
    ("""            const float2 q = *reinterpret_cast<const float2*>(r0 + (size_t)kh * rowb + (off + (unsigned)(kw * GEN_IMG_LD * 4)));
            v[2 * t] = q.x; v[2 * t + 1] = q.y;""",
     """            if (t == 0) { const float2 q = *reinterpret_cast<const float2*>(r0 + (size_t)kh * rowb + (off + (unsigned)(kw * GEN_IMG_LD * 4))); v[0] = q.x; v[1] = q.y; }
            else { v[2 * t] = v[0] + (float)t; v[2 * t + 1] = v[1] - (float)t; }"""),
]}
