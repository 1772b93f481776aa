# timing experiment (WRONG RESULTS on purpose): k_fc4_b3 without the per-step LDS fragment reads (step 0's fragments are reused)
PATCH = {'bf16x3.hip': [
    ("return *reinterpret_cast<const float4*>(brow[nt] + p * 512 + ks * 32); };", "return *reinterpret_cast<const float4*>(brow[nt] + p * 512 + (ks & 0) * 32); };"),
]}
