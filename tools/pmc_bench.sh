#!/bin/bash
# PMC counter passes over the whole bench workload (separate passes, kernel-trace only): HBM bytes and
# MFMA / wave-time counters for every fresco kernel of a cfg2 step.  usage: bash tools/pmc_bench.sh <tag>
TAG=${1:-p}
REPO=$PWD
OUT=$PWD/gpurun_out/pmcb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
P4="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python $REPO/bench.py --no-cpu-baseline --no-aux --steps 15 --warmup 1 > $OUT/pass$i.log 2>&1
done
cd $REPO
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/pmcb_" + os.environ["TAG"]
rows = []
for f in sorted(glob.glob(base + "/pass*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60] + "|grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "fresco" in k:
            for c, v in d.items():
                rows.append((k, c, n[(k, c)], v, v / n[(k, c)]))
with open(base + "/summary.csv", "w") as f:
    f.write("kernel|grid,counter,dispatches,sum,per_dispatch\n")
    for r in rows:
        f.write("%s,%s,%d,%.0f,%.1f\n" % r)
for r in rows:
    if r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
        print(r[0][12:80], r[1], r[2], round(r[4]))
PY
