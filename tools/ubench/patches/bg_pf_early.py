# timing experiment: k_dec_bg with the next strip's rows requested three channel blocks earlier (in front of later weight-fragment waits)
PATCH = {'generic_dec.hip': [
    ("                if (kc == 6 && more) {       // the next strip's new rows", "                if (kc == 3 && more) {       // the next strip's new rows"),
]}
