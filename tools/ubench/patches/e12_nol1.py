# timing experiment (WRONG RESULTS on purpose): k_conv_e12 without its L1 phases (one tile per wave and block instead of all)
PATCH = {'generic_enc.hip': [
    ("constexpr int E12_ROUNDS = 5;", "constexpr int E12_ROUNDS = 1;"),
]}
