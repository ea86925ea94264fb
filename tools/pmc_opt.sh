#!/bin/bash
# PMC counter passes over optimize_feature (config 3) at one decoder layer (LAYER=3: C = 640, 64 x 64 (default); 2: C = 1280,
# 32 x 32; 1: 16 x 16; 0: 8 x 8; 8 frames): HBM-side bytes, matrix-pipe / LDS counters per launch of every opt kernel
# (separate passes, kernel-trace only).  usage: LAYER=3 bash tools/pmc_opt.sh <tag>
TAG=${1:-p}
REPO=$PWD
OUT=$PWD/gpurun_out/pmco_$TAG
mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tools")
import bench_opt, fresco_amd
from fresco_amd import ops
g = torch.Generator().manual_seed(0)
N, R, dev = 8, 512, "cuda"
flows, occs, sal = bench_opt._inputs(N, R, dev, g)
import os
C, h = bench_opt.LAYERS[int(os.environ.get("LAYER", "3"))]
x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
for _ in range(2):
    fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=5)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
export FRESCO_OPT_SPLIT=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  [ $i -ge ${PASSES:-5} ] && break
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python $OUT/run.py $REPO > $OUT/pass$i.log 2>&1
done
cd $REPO
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/pmco_" + os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in sorted(glob.glob(base + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "fresco" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(base + "/summary.csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,HBM_MB_per_launch(2*FETCH+WRITE)\n")
    for k, d in sorted(agg.items()):
        fe = d.get("FETCH_SIZE", 0) / max(n[(k, "FETCH_SIZE")], 1); wr = d.get("WRITE_SIZE", 0) / max(n[(k, "WRITE_SIZE")], 1)
        line = "%s,%d,%.0f,%.0f,%.1f" % (k, n[(k, "FETCH_SIZE")], fe, wr, (2 * fe + wr) * 1024 / 1e6)
        f.write(line + "\n"); print(line)
    f.write("kernel,counter,per_launch\n")
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            if c in ("FETCH_SIZE", "WRITE_SIZE"): continue
            line = "%s,%s,%.4g" % (k, c, v / max(n[(k, c)], 1))
            f.write(line + "\n"); print(line)
PY
