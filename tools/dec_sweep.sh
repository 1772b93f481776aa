#!/bin/bash
# dev tool (GPU box): the decoder kernel classes of a stand-alone 19 200-image decoder call (tools/dec_only.py) under several experiment
# builds (tools/ubench/alt/<name>), product build first and last.      tools/dec_sweep.sh <tag> "<alt names>"      (EFE_ENGINE_OPTS passes through)
TAG=$1; ALTS=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python tools/dec_only.py 19200 4 > $OUT/00_product.txt 2>&1
for a in $ALTS; do EFE_LIB_PATH=tools/ubench/alt/$a/libefe_mi355x.so timeout 200 python tools/dec_only.py 19200 4 > $OUT/alt_$a.txt 2>&1; done
timeout 200 python tools/dec_only.py 19200 4 > $OUT/zz_product.txt 2>&1
for f in $OUT/*.txt; do echo "== $f"; grep block $f | tail -2; done
