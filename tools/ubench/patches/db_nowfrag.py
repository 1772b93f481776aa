# timing experiment (WRONG RESULTS on purpose): k_dec_b4 with ONE weight fragment per tap group (no L2 fragment stream: the first chunk's fragments are reused)
PATCH = {'decoder.hip': [
    ("    auto wf = [&](int tap, int kc) -> float4 { return wfrag(wr, ln, (size_t)(tap * 8 + kc) * 64); };", "    const float4 wf0_ = wfrag(wr, ln, 0);\n    auto wf = [&](int tap, int kc) -> float4 { float4 q = wf0_; q.x += (float)(tap + kc) * 1e-30f; return q; };"),
]}
