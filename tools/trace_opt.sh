#!/bin/bash
# kernel trace (start / end timestamps) of optimize_feature at (C 640, 64 x 64) under a given FRESCO_OPT_SPLIT: shows whether the
# two CFG-half pipelines actually overlap.  usage: bash tools/trace_opt.sh <tag> <split>
TAG=${1:-t}; SPLIT=${2:-1}
REPO=$PWD
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tools")
import bench_opt, fresco_amd
from fresco_amd import ops
g = torch.Generator().manual_seed(0)
N, R, dev = 8, 512, "cuda"
C, h = int(sys.argv[2]), int(sys.argv[3])
flows, occs, sal = bench_opt._inputs(N, R, dev, g)
x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
for _ in range(2):
    fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=6)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
FRESCO_OPT_SPLIT=$SPLIT timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python $OUT/run.py $REPO ${3:-640} ${4:-64} > $OUT/log.txt 2>&1
cd $REPO
f=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "opt_" in r["Kernel_Name"] or "gram16" in r["Kernel_Name"] or "sv16" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-48:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fresco::", "")[:18]
    print("%-18s q%-3s %9.1f -> %9.1f  (%7.1f us) grid %s" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                               (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "?")))
PY
