#!/bin/bash
# Round-6 GPU-box visits.  usage (repo root on the box): bash tools/gpu_r6.sh <tag> <stages...>
#   stages: tests smoke bench gloo2 stall prof profopt pmcopt pmcattn fullstep
TAG=${1:-a}; shift
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
for S in "$@"; do
  echo "=== stage $S ($(date +%T))"
  case $S in
  tests)
    timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider ${PYTEST_ARGS} > $OUT/pytest_$TAG.log 2>&1
    tail -12 $OUT/pytest_$TAG.log; grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head -20 ;;
  newtests)
    timeout 1200 python -m pytest tests/test_gpu_cfg45.py tests/test_gpu_latent_delta.py tests/test_gpu_attn32.py -m gpu -q -s --tb=short -p no:cacheprovider > $OUT/pytest_new_$TAG.log 2>&1
    grep -E "opt N=|warp_tensor N=|cfg5 L2|latent delta|passed|failed|FAILED|ERROR|Error" $OUT/pytest_new_$TAG.log | head -60 ;;
  r6new)
    timeout 2400 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_flownet.py tests/test_gmflow.py tests/test_gpu_cfg45.py -m gpu -q -s --tb=short -p no:cacheprovider -k "sharded_optimize or range or side_operands or out_of_range or full_batch or cfg5_full" > $OUT/pytest_r6new_$TAG.log 2>&1
    grep -E "cfg5|opt FULL|passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest_r6new_$TAG.log | head -60 ;;
  kvtests)
    timeout 1800 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py tests/test_gmflow.py tests/test_gpu_integration.py -m gpu -q -s --tb=short -p no:cacheprovider -k "${KEXPR:-fused or processor or sharded_optimize or out_of_range or integration}" > $OUT/pytest_kv_$TAG.log 2>&1
    grep -E "fused K|passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest_kv_$TAG.log | head -60 ;;
  qbench)
    timeout 900 python bench.py --no-cpu-baseline --no-aux > $OUT/qbench_$TAG.json 2> $OUT/qbench_$TAG.err; python - <<PYEOF
import json
r=json.loads(open("$OUT/qbench_$TAG.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["timing"]["ms_per_step_all"], r["roofline"]["avg_launch_us"], r["host_issue_ms_per_step"])
for k,v in r["kernel_avg_us"].items(): print("  ", k, v)
PYEOF
    tail -3 $OUT/qbench_$TAG.err ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -4 $OUT/smoke_$TAG.log ;;
  bench)
    timeout 1200 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -c 3000 $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err ;;
  gloo2)
    # functional run of the multi-process path on ONE GPU (gloo, host-staged collectives): census / parity / exchange timing code
    FRESCO_BENCH_BACKEND=gloo FRESCO_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 > $OUT/bench_gloo2_$TAG.json 2> $OUT/bench_gloo2_$TAG.err
    tail -c 2500 $OUT/bench_gloo2_$TAG.json; tail -5 $OUT/bench_gloo2_$TAG.err ;;
  stall)
    for L in 3 0; do timeout 600 python tools/stall_hunt.py 200 $L > $OUT/stall_${TAG}_L$L.json 2> $OUT/stall_${TAG}_L$L.err; cat $OUT/stall_${TAG}_L$L.json | cut -c1-1800; tail -2 $OUT/stall_${TAG}_L$L.err; done
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/stalltrace_$TAG -- python $REPO/tools/stall_hunt.py 100 3 > $OUT/stalltrace_$TAG.json 2> $OUT/stalltrace_$TAG.err )
    cut -c1-1200 $OUT/stalltrace_$TAG.json
    python tools/trace_gaps.py $OUT/stalltrace_$TAG 12 > $OUT/stallgaps_$TAG.txt 2>&1; cat $OUT/stallgaps_$TAG.txt
    find $OUT/stalltrace_$TAG -name "*.csv" -size +8M -delete ;;
  prof)
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $REPO/bench.py --no-cpu-baseline --no-aux > $OUT/prof_$TAG.log 2>&1 )
    f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats_$TAG.csv && head -16 $f
    tail -c 600 $OUT/prof_$TAG.log
    find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete ;;
  profopt)
    ( cd /tmp && export TMPDIR=/tmp && FRESCO_OPT_SPLIT=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profopt_$TAG -- python $REPO/tools/bench_opt.py 20 --no-baselines > $OUT/profopt_$TAG.log 2>&1 )
    f=$(find $OUT/profopt_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/opt_kernel_stats_$TAG.csv && head -24 $f
    find $OUT/profopt_$TAG -name "*kernel_trace.csv" -delete ;;
  pmcopt)
    LAYER=3 bash tools/pmc_opt.sh ${TAG}_L3 > $OUT/pmcopt_${TAG}_L3.txt 2>&1; cat $OUT/pmcopt_${TAG}_L3.txt | head -70
    LAYER=2 bash tools/pmc_opt.sh ${TAG}_L2 > $OUT/pmcopt_${TAG}_L2.txt 2>&1; head -12 $OUT/pmcopt_${TAG}_L2.txt ;;
  pmcattn)
    PASSES=5 bash tools/pmc_attn.sh $TAG > $OUT/pmcattn_$TAG.txt 2>&1; tail -40 $OUT/pmcattn_$TAG.txt ;;
  fullstep)
    timeout 900 python tools/bench_full_step.py > $OUT/full_step_$TAG.json 2> $OUT/full_step_$TAG.err; cat $OUT/full_step_$TAG.json; tail -5 $OUT/full_step_$TAG.err ;;
  esac
done
