# timing experiment (correct results): k_dec_b4 with its per-image prologue loads (first strip from HBM) and its per-image epilogue (the last strip's own
# gather + the quarter reductions) executed TWICE: the difference to the product build bounds what overlapping them with a neighbouring image could save
PATCH = {'decoder.hip': [
    ("""#pragma unroll
    for (int it = 0; it < NPF; ++it) { const int idx = it * NTHR + tid; pf[it] = Xv[y2_at(min(SR * s_lo + (idx >> 9), 31), idx)]; }
""", """#pragma unroll
    for (int it = 0; it < NPF; ++it) { const int idx = it * NTHR + tid; pf[it] = Xv[y2_at(min(SR * s_lo + (idx >> 9), 31), idx)]; }
    { float t_ = 0.f;
#pragma unroll
      for (int it = 0; it < NPF; ++it) t_ += pf[it][0];
      if (t_ == 1.2345e-31f) part = t_;          // consumes the first request (never true)
      __syncthreads(); }
#pragma unroll
    for (int it = 0; it < NPF; ++it) { const int idx = it * NTHR + tid; pf[it] = Xv[y2_at(min(SR * s_lo + (idx >> 9), 31), idx) ^ 0]; asm volatile("" : "+v"(pf[it])); }
"""),
    ("""    fold(parts == 1 ? 3 : 0);
    __syncthreads();""", """    fold(parts == 1 ? 3 : 0);
    __syncthreads();
    for (int rep_ = 0; rep_ < 1; ++rep_) {
        if (w < (parts == 1 ? 4 : 1)) {
            const float* qk = sq + w * NTHR + lane;
            float v = (qk[0] + qk[64]) + (qk[128] + qk[192]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) sQ[w] = v;
        }
        __syncthreads();
    }"""),
]}
