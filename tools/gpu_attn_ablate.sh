#!/bin/bash
# times the variants built by tools/attnp_ablate.py (one line per variant and kernel form) -> gpurun_out/attnp_abl_<tag>.txt
TAG=${1:-r}
OUT=$PWD/gpurun_out/attnp_abl_$TAG.txt
mkdir -p gpurun_out; : > $OUT
echo "== product library, FRESCO_ATTN_PIPE=0 (ping-pong kernel, reference of this box)" >> $OUT
FRESCO_ATTN_PIPE=0 timeout 120 python tools/bench_flash.py 20 0.3 spatial40 >> $OUT 2>&1
for so in tools/abl/libfresco_hip_*.so; do
  v=$(basename $so .so); v=${v#libfresco_hip_}
  for mode in 1 2; do
    echo "== $v mode $mode" >> $OUT
    FRESCO_HIP_LIB=$PWD/$so FRESCO_ATTN_PIPE=$mode timeout 120 python tools/bench_flash.py 20 0.3 spatial40 >> $OUT 2>&1
  done
done
grep -E "^==|HW=" $OUT
