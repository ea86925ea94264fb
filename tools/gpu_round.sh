#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel trace of the bench.
# usage (from the repo root on the box): bash tools/gpu_round.sh <tag>
TAG=${1:-r}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
tail -25 $OUT/pytest_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -4 $OUT/smoke_$TAG.log
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $REPO/bench.py --no-cpu-baseline --no-aux > $OUT/prof_$TAG.log 2>&1
cd $REPO
find $OUT/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f"
