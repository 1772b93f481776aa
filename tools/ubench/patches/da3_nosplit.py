# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 without the three-way split of staged / written-back activations (the fp32 bits are stored instead)
PATCH = {'bf16x3.hip': [("        split3_pk(v0, v1, hi[0], mid[0], lo[0]); split3_pk(v2, v3, hi[1], mid[1], lo[1]);\n",
 "        hi[0] = __builtin_bit_cast(uint32_t, v0) >> 16 | (__builtin_bit_cast(uint32_t, v1) & 0xffff0000u); hi[1] = __builtin_bit_cast(uint32_t, v2) >> 16 | (__builtin_bit_cast(uint32_t, v3) & 0xffff0000u);\n        mid[0] = hi[0]; mid[1] = hi[1]; lo[0] = hi[0]; lo[1] = hi[1];\n")]}
