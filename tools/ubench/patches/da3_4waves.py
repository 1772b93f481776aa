# A/B build: k_dec_a_b3 with FOUR waves per workgroup (one per SIMD, 2 x 2 register tiles) instead of eight
PATCH = {'bf16x3.hip': [("#define EFE_DA3_NTW 1", "#define EFE_DA3_NTW 2")]}
