"""optimize_feature at the four decoder-layer shapes of a 768 x 768 batch (config 5's frames: 12 x 12 and 24 x 24 planes take
the generic / plain-layout kernels, 48 x 48 and 96 x 96 the tiled ones): one Adam iteration against the oracle evaluated by
torch on the GPU, and timing of the pipeline's 20.  usage: python tools/check_opt_768.py [frames]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth
import fresco_amd
from oracle import fresco_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
for C, h in ((1280, 12), (1280, 24), (1280, 48), (640, 96)):
    case = synth.make_opt_case(N, C, h, 768, seed=7)
    x = case["x"].to(dev)
    fd, od, td = [f.to(dev) for f in case["flows"]], [o.to(dev) for o in case["occs"]], case["target"].to(dev)
    out = fresco_amd.optimize_feature(x, fd, od, [td], iters=1)
    ref = O.optimize_feature(x, fd, od, [td], iters=1)
    df = (out - ref).abs()
    frac = float((df > 1e-3).double().mean())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o20 = fresco_amd.optimize_feature(x, fd, od, [td], iters=20)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    print("C=%d %dx%d: 1 iteration vs oracle: median %.1e, off by > 1e-3: %.1e of the elements | 20 iterations %.2f ms, finite %s"
          % (C, h, h, float(df.median()), frac, ms, bool(torch.isfinite(o20).all())), flush=True)
    assert frac < 0.02 and torch.isfinite(o20).all()
