#!/bin/bash
# One GPU-box visit for the flash kernel: parity (test-suite, fuzz, full-size), then timing in both logit regimes.
# usage (repo root, on the box): bash tools/gpu_attn_ab.sh <tag> [quick]
TAG=${1:-r}
OUT=$PWD/gpurun_out/attn_ab_$TAG.txt
mkdir -p gpurun_out
: > $OUT
run() { echo "== $*" >> $OUT; timeout 400 "$@" >> $OUT 2>&1; echo "rc=$?" >> $OUT; }
run python -m pytest tests/test_gpu_attention.py -q -x -p no:cacheprovider --tb=short
run python tools/fuzz_attn.py 60 1
if [ "$2" != quick ]; then
  run python -m pytest tests/test_gpu_fullsize.py -q -x -p no:cacheprovider --tb=short -k "processor or attention"
fi
for gain in 0.3 1.0; do
  run python tools/bench_flash.py 20 $gain
done
grep -E "^==|^--|^rc=|passed|failed|HW=4096|all .* cases ok|FAIL|Error|error" $OUT | cut -c1-200
