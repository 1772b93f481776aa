# timing experiment (WRONG RESULTS on purpose): k_conv_e12 with ONE conv1 operand request per tile instead of 18 (the rest reuse it)
PATCH = {'generic_enc.hip': [
    ("""            v[2 * t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off + (unsigned)(kw * GEN_IMG_LD * 4), so, 0));
            v[2 * t + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off + (unsigned)(kw * GEN_IMG_LD * 4 + 8), so, 0));""",
     """            if (t == 0) v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off + (unsigned)(kw * GEN_IMG_LD * 4), so, 0));
            else v[2 * t] = v[0] + (float)t;
            v[2 * t + 1] = v[0] - (float)t;"""),
]}
