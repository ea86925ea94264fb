#!/bin/bash
# One GPU-box visit: per-kernel times of every ablation library in tools/abl/ (tools/opt_ablate.py) at the layers named
# by BENCH_OPT_LAYERS (default: the largest), base library first and last.  usage: bash tools/gpu_ablate.sh <tag>
TAG=${1:-abl}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export BENCH_OPT_LAYERS=${BENCH_OPT_LAYERS:-3} AB_SWITCHES=NONE BENCH_OPT_NOCHECK=1 AB_ITERS=${AB_ITERS:-4} FRESCO_OPT_SPLIT=0
run() { echo "== $1"; timeout 120 python tools/ab_opt.py 1 2>&1 | grep "^round" | cut -c1-600; }
( run base
  for f in tools/abl/libfresco_hip_*.so; do
    n=${f#tools/abl/libfresco_hip_}; n=${n%.so}
    FRESCO_HIP_LIB=$PWD/$f run $n
  done
  run base ) | tee $OUT/ablate_$TAG.log
