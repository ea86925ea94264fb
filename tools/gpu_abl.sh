#!/bin/bash
# timing ablations of the opt kernels (results are wrong under FRESCO_OPT_ABL != 0): usage bash tools/gpu_abl.sh <tag> "<abl values>"
TAG=${1:-a}
OUT=$PWD/gpurun_out
mkdir -p $OUT
for a in ${2:-0 1 2 4 8 16 32 64 128 48 176}; do
  FRESCO_OPT_ABL=$a FRESCO_OPT_SPLIT=0 BENCH_OPT_LAYERS=${LAYERS_SEL:-3} timeout 300 python tools/bench_opt.py 20 --no-baselines > $OUT/abl_${TAG}_$a.log 2>&1
  echo "abl=$a $(grep '^layer' $OUT/abl_${TAG}_$a.log | sed -e "s/'gram_roofline.*//")"
done
