#!/bin/bash
# rocprofv3 kernel stats + PMC passes (HBM bytes, L2 hit rate, matrix-pipe busy) of one native GMFlow forward (8 x 512^2).
# usage: bash tools/pmc_gmflow.sh <tag>
TAG=${1:-g}
REPO=$PWD
OUT=$PWD/gpurun_out/pmcg_$TAG
mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests/golden")
import closed_form as cf
import fresco_amd.gmflow as G
N, R, dev = 8, 512, "cuda"
m = G.GMFlow().eval()
sd = m.state_dict()
m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
m = m.to(dev)
imgs = cf.gmflow_frames(N, R, R).to(dev)
nxt = list(range(1, N)) + [0]
kw = dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow=True)
with torch.no_grad():
    for _ in range(3):
        m(imgs, imgs[nxt], **kw)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $OUT/run.py $REPO > $OUT/stats.log 2>&1
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python $OUT/run.py $REPO > $OUT/pass$i.log 2>&1
done
cd $REPO
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -14 $f | cut -c1-160
find $OUT -name "*kernel_trace.csv" -delete
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/pmcg_" + os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in sorted(glob.glob(base + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if "fn_" not in k and "attn_f32" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(base + "/summary.csv", "w") as f:
    f.write("kernel,counter,launches,per_launch\n")
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            line = "%s,%s,%d,%.5g" % (k, c, n[(k, c)], v / max(n[(k, c)], 1))
            f.write(line + "\n"); print(line)
PY
