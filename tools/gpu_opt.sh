#!/bin/bash
# One GPU-box visit for the feature-optimisation path: parity tests of the opt kernels, then cfg3 timings per
# launch mode.  usage (repo root on the box): bash tools/gpu_opt.sh <tag>
TAG=${1:-o}
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_opt.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -s --tb=short -p no:cacheprovider -k "opt or closure or adam or gram or sharded" > $OUT/pytest_opt_$TAG.log 2>&1
tail -30 $OUT/pytest_opt_$TAG.log
for m in ${MODES:-0 1 2}; do
  echo "== FRESCO_OPT_SPLIT=$m"
  FRESCO_OPT_SPLIT=$m timeout 600 python tools/bench_opt.py 20 --no-baselines > $OUT/bench_opt_${TAG}_split$m.log 2>&1
  grep "^layer\|cfg3 extra" $OUT/bench_opt_${TAG}_split$m.log
done
