# timing experiment (correct results): k_dec_bg with its strip-0 fetch (12 float4 per thread from HBM into the input ring) executed TWICE:
# the difference to the product build bounds what the image prologue's exposed memory latency costs
PATCH = {'generic_dec.hip': [
    ("""    {   // strip 0: rows 0 .. TH -> slots 0 .. RP - 1 (RP <= 192 pixels = 12 float4 per thread, all requested before the first is written:
        // one HBM round trip per image instead of three)
        float4 v[12];""", """    for (int rep_ = 0; rep_ < 2; ++rep_) {
        __syncthreads();
        float4 v[12];"""),
]}
