# timing experiment (WRONG RESULTS on purpose): k_dec_b_b3 without the weight slab copies after the first
PATCH = {'bf16x3.hip': [
    ("                if (ks < 3 || !last || nimg < a.rows) slab_dma((ks + 1) & 3, (ks + 1) & 1);", "                if (s > 100) slab_dma((ks + 1) & 3, (ks + 1) & 1);"),
]}
