# timing experiment (WRONG RESULTS on purpose): k_conv_e12 without the two barriers of a strip
PATCH = {'generic_enc.hip': [
    ("""        __syncthreads();                                   // every wave is done reading the rows that are replaced
        if (!E12_EARLY) l1_req_all(nfirst, nnew);""", """        if (!E12_EARLY) l1_req_all(nfirst, nnew);"""),
    ("""        rs0 = rs0 >= NR ? rs0 - NR : rs0;
        __syncthreads();
    }
}

static int conv_e_ty""", """        rs0 = rs0 >= NR ? rs0 - NR : rs0;
    }
}

static int conv_e_ty"""),
]}
