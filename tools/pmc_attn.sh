#!/bin/bash
# PMC counter passes over the cross-frame attention kernels (separate passes, kernel-trace only).
TAG=${1:-p}
GAIN=${2:-1.0}
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
python tools/run_attn_only.py 20 $GAIN | tee $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"   # L2 hit rate and fabric-side read requests
i=0
PASSES=${PASSES:-5}
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  [ $i -ge $PASSES ] && break
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python $REPO/tools/run_attn_only.py 3 $GAIN > $OUT/pass$i.log 2>&1
done
cd $REPO
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/pmc_" + os.environ["TAG"]
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
    if "flash" in r[0]:
        print(r[0][20:52], r[1], round(r[4]))
PY
