#!/bin/bash
# dev tool (GPU box): same-box A/B of two engine builds on one bench workload, alternating runs.
#   tools/ab_bench.sh <tag> <alt-lib-name> <reps> [bench args...]     e.g.  tools/ab_bench.sh r4a r3 2 --workload animalai
TAG=$1; ALT=$2; REPS=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $REPS); do
  timeout 300 python bench.py --no-cpu --min-seconds 2 "$@" > $OUT/new_$i.json 2> $OUT/new_$i.err
  EFE_LIB_PATH=tools/ubench/alt/$ALT/libefe_mi355x.so timeout 300 python bench.py --no-cpu --min-seconds 2 "$@" > $OUT/old_$i.json 2> $OUT/old_$i.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d.get("kernels_one_step", {})
        print(f, round(d["value"], 1), d["roofline"]["frac"], {n: v["ms"] for n, v in k.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
