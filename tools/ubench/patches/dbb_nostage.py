# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the strip staging (split + LDS writes)
PATCH = {'bf16x3.hip': [
    ("                put_planes(smb + (size_t)(it * 32 + ix) * DBB_PXB, 4 * c4, in ? pf[it][0] : 0.f, in ? pf[it][1] : 0.f, in ? pf[it][2] : 0.f, in ? pf[it][3] : 0.f);",
     "                if (s == 0 && nimgs == 0) put_planes(smb + (size_t)(it * 32 + ix) * DBB_PXB, 4 * c4, in ? pf[it][0] : 0.f, in ? pf[it][1] : 0.f, in ? pf[it][2] : 0.f, in ? pf[it][3] : 0.f);"),
]}
