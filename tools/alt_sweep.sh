#!/bin/bash
# dev tool (GPU box): one bench workload under several experiment builds (tools/ubench/alt/<name>), product build first and last.
#   tools/alt_sweep.sh <tag> "<alt names>" [bench args...]
TAG=$1; ALTS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { timeout 300 python bench.py --no-cpu --min-seconds 2 "$@" ; }
run "$@" > $OUT/00_product.json 2> $OUT/00_product.err
for a in $ALTS; do EFE_LIB_PATH=tools/ubench/alt/$a/libefe_mi355x.so run "$@" > $OUT/alt_$a.json 2> $OUT/alt_$a.err; done
run "$@" > $OUT/zz_product.json 2> $OUT/zz_product.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d.get("kernels_one_step", {})
        print(f.split('/')[-1], round(d["value"], 1), d["roofline"]["frac"], {n: v["ms"] for n, v in k.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
