# A/B of the small-plane Gram forms (round 6): usage (repo root on the GPU box): bash tools/ab_small.sh
export BENCH_OPT_LAYERS=0,1
run() { echo "== $1"; shift; env "$@" python tools/bench_opt.py 20 --no-baselines 2>&1 | grep -E "^layer" | cut -c1-420; }
run "round-5 forms (one-launch gram16s at 8x8, gram16s at 16x16)" FRESCO_GRAM_SPLIT_WG=0 FRESCO_GRAM_TILED_MIN_HW=512
run "8x8: split-K slices on their own CUs + ordered sum" FRESCO_GRAM_SPLIT_WG=1 FRESCO_GRAM_TILED_MIN_HW=512
run "16x16: DMA-staged 128x128 tiles (gram16z)" FRESCO_GRAM_SPLIT_WG=1 FRESCO_GRAM_TILED_MIN_HW=256
run "16x16: gram16z, one stream" FRESCO_GRAM_SPLIT_WG=1 FRESCO_GRAM_TILED_MIN_HW=256 FRESCO_OPT_SPLIT=0
python - <<'PY'
import os, sys, torch
sys.path.insert(0, "tests")
import synth
import fresco_amd.ops as ops
from fresco_amd.warp import _prep_flow_occ
from oracle import fresco_oracle as O
dev = "cuda"
for C, h in ((1280, 8), (1280, 16)):
    g = synth.gen(5 + h)
    N, R = 8, 512
    x = torch.randn(2 * N, C, h, h, generator=g).to(dev)
    flows, occs = synth.make_flows(N, R, g)
    fd, od = [f.to(dev) for f in flows], [o.to(dev) for o in occs]
    td = O.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
    prep = _prep_flow_occ(h, fd, od, with_dilate=False)
    outs = {}
    for name, env in (("r5", dict(FRESCO_GRAM_SPLIT_WG="0", FRESCO_GRAM_TILED_MIN_HW="512")),
                      ("new", dict(FRESCO_GRAM_SPLIT_WG="1", FRESCO_GRAM_TILED_MIN_HW="256"))):
        os.environ.update(env)
        cs = x.clone()
        ops.opt_run(cs, prep, td, 100.0, 20, 2)
        outs[name] = cs
    d = float((outs["r5"] - outs["new"]).abs().max())
    print("C=%d %dx%d: 20 iterations, round-5 forms vs new forms: equal=%s max|d|=%.3e" % (C, h, h, torch.equal(outs["r5"], outs["new"]), d))
PY
