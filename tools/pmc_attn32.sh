#!/bin/bash
# PMC counter passes over fresco_attn_f32 at one shape: bash tools/pmc_attn32.sh <tag> B L D Dv
TAG=${1:-p}; shift
SHAPE="${@:-64 1024 128 128}"
REPO=$PWD
OUT=$PWD/gpurun_out/pmc32_$TAG
mkdir -p $OUT
python tools/run_attn32_only.py 20 $SHAPE | tee $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/tools/run_attn32_only.py 10 $SHAPE > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep fresco $f | cut -c1-160
find $OUT/stats -name "*kernel_trace.csv" -delete
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
i=0
for P in "$P1" "$P2" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python $REPO/tools/run_attn32_only.py 3 $SHAPE > $OUT/pass$i.log 2>&1
done
cd $REPO
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/pmc32_" + os.environ["TAG"]
rows = []
for f in sorted(glob.glob(base + "/pass*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "fresco" in k:
            for c, v in d.items():
                rows.append((k, c, n[(k, c)], v, v / n[(k, c)]))
with open(base + "/summary.csv", "w") as f:
    f.write("kernel,counter,dispatches,sum,per_dispatch\n")
    for r in rows:
        f.write("%s,%s,%d,%.0f,%.1f\n" % r)
for r in rows:
    if "f32p" in r[0]:
        print(r[0][:60], r[1], round(r[4]))
PY
find $OUT -name "*counter_collection.csv" -size +4M -delete
