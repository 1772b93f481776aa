# timing experiment (correct results): k_convt_12 with its image prologue (x-ring fetch of rows -1 .. TH, zero region, offset table) executed TWICE:
# the difference to the product build is what the prologue costs per image
PATCH = {'generic_dec.hip': [
    ("""    // round 0's x rows -1 .. TH -> slots 0 .. RP - 1; the zero region
    for (int pp0 = ppt; pp0 < RP; pp0 += 8 * pstep) {""", """    for (int rep_ = 0; rep_ < 2; ++rep_) {
    __syncthreads();
    for (int pp0 = ppt; pp0 < RP; pp0 += 8 * pstep) {"""),
    ("""    const int nt = wave & 1, mt = wave >> 1;
    // (the wave's feature tile is part of the resource base""", """    }
    const int nt = wave & 1, mt = wave >> 1;
    // (the wave's feature tile is part of the resource base"""),
]}
