#!/bin/bash
# One GPU-box visit for the launch-form switches of opt_fast.hip: parity tests of the opt path, then the A/B timings.
# usage (repo root on the box): bash tools/gpu_ab_opt.sh <tag>
TAG=${1:-ab}
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_opt.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -s --tb=short -p no:cacheprovider -k "opt or closure or adam or gram or sharded or forms" > $OUT/pytest_opt_$TAG.log 2>&1
tail -40 $OUT/pytest_opt_$TAG.log
timeout 600 python tools/ab_opt.py 2 > $OUT/ab_opt_$TAG.log 2>&1
cat $OUT/ab_opt_$TAG.log | cut -c1-700
[ -x tools/bin/ubench_mfma_i8 ] && timeout 60 tools/bin/ubench_mfma_i8 > $OUT/ubench_mfma_i8_$TAG.txt 2>&1 && cat $OUT/ubench_mfma_i8_$TAG.txt
