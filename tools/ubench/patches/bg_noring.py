# timing experiment (WRONG RESULTS on purpose): k_dec_bg without the next strip's fetch and ring writes (every strip contracts the first rows)
PATCH = {'generic_dec.hip': [
    ("                if (kc == 6 && more) {       // the next strip's new rows", "                if (false) {       // the next strip's new rows"),
    ("                if (more) {\n                    // the new rows take the slots behind the halo row;", "                if (false) {\n                    // the new rows take the slots behind the halo row;"),
    ("                if (more) {\n#pragma unroll\n                    for (int i = 4; i < 8; ++i)", "                if (false) {\n#pragma unroll\n                    for (int i = 4; i < 8; ++i)"),
]}
