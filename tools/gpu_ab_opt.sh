#!/bin/bash
# One GPU-box visit for the feature-optimisation kernels: parity tests of the opt path, then same-box timings of the
# current library against an older build (tools/abl/libfresco_hip_old.so, if present) and of the launch-form switches.
# usage (repo root on the box): bash tools/gpu_ab_opt.sh <tag> [pytest -k expression]
TAG=${1:-ab}
KEXPR=${2:-"opt or closure or adam or gram or sharded or forms"}
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_opt.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -s --tb=short -p no:cacheprovider -k "$KEXPR" > $OUT/pytest_opt_$TAG.log 2>&1
tail -30 $OUT/pytest_opt_$TAG.log
for rep in 1 2; do
  if [ -f tools/abl/libfresco_hip_old.so ]; then
    echo "== old library (${OLD_ENV})"
    env $OLD_ENV FRESCO_HIP_LIB=$PWD/tools/abl/libfresco_hip_old.so AB_SWITCHES=NONE timeout 300 python tools/ab_opt.py 1 2>&1 | grep "^round" | cut -c1-900 | tee -a $OUT/ab_opt_$TAG.log
  fi
  echo "== current library"
  timeout 300 python tools/ab_opt.py 1 2>&1 | grep "^round" | cut -c1-900 | tee -a $OUT/ab_opt_$TAG.log
done
