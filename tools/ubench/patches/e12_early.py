# A/B build (correct results): k_conv_e12 requests a block's conv1 operands BEFORE the previous strip's L2 phase instead of behind its barrier
PATCH = {'generic_enc.hip': [
    ("#define EFE_E12_EARLY 0", "#define EFE_E12_EARLY 1"),
]}
