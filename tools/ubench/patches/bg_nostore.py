# timing experiment (WRONG RESULTS on purpose): k_dec_bg without the image stores of the deferred gather
PATCH = {'generic_dec.hip': [
    ("                g_store();                                 // the previous strip's pixels (sigmoid piece: view A section)", ""),
]}
