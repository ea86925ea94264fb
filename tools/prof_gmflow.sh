# kernel-level profile of one GMFlow forward batch (f3): bash tools/prof_gmflow.sh <tag>
TAG=${1:-a}; OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && BENCH_GMFLOW_NO_LIBRARY_LEG=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profgm_$TAG -- python $REPO/tools/bench_gmflow.py > $OUT/profgm_$TAG.log 2>&1 )
f=$(find $OUT/profgm_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/gmflow_kernel_stats_$TAG.csv && head -40 $f | cut -c1-200
find $OUT/profgm_$TAG -name "*kernel_trace.csv" -delete
tail -c 300 $OUT/profgm_$TAG.log
