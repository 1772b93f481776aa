# A/B build (correct results): k_conv_e12's L2 phase with its weight fragments two channel blocks ahead (the shared default) instead of four
PATCH = {'generic_enc.hip': [
    ("#define EFE_E12_PD 4", "#define EFE_E12_PD 2"),
]}
