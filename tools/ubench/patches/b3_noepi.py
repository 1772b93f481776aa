# timing experiment (WRONG RESULTS on purpose): k_fc4_b3 without its epilogue (one store per lane and step)
PATCH = {'bf16x3.hip': [
    ("            if (!rv[nt]) continue;\n            const uint4 rnd", "            if (nt == 0 && rv[0]) a.Y[(size_t)(row0 + j) * a.ldy + mt0 * 32 + 4 * g] = acc[0][0][0] + acc[1][1][5] + acc[0][1][3] + acc[1][0][7];\n            if (true) continue;\n            const uint4 rnd"),
]}
