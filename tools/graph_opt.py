"""Does a hipGraph form of the 20-iteration Adam loop buy anything (VERDICT r04 missing #6)?
optimize_feature at the two small decoder layers -- (1280, 8 x 8): 80 launches of 15 - 26 us; (1280, 16 x 16) -- timed eagerly
and as the replay of ONE captured graph of the whole call (every launch of the library goes to the capturing stream; the
16 x 16 layer's second pipeline stream is forked / joined by events, which capture follows).
usage: python tools/graph_opt.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_opt
import fresco_amd
from fresco_amd import ops


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


dev = "cuda"
g = torch.Generator().manual_seed(0)
flows, occs, sal = bench_opt._inputs(8, 512, dev, g)
for C, h in ((1280, 8), (1280, 16)):
    x = torch.randn(16, C, h, h, generator=g).half().to(dev)
    tgt = ops.gram_target(torch.randn(16, C, h, h, generator=g).to(dev))
    run = lambda: fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=20)
    eager = timeit(run)
    out_e = run()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    status = "ok"
    try:
        with torch.cuda.stream(s):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_g = run()
        replay = timeit(graph.replay)
        graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out_g, out_e))
    except Exception as e:  # noqa: BLE001
        status, replay, same = "capture failed: %s" % (str(e).splitlines()[0][:160] if str(e) else type(e).__name__), float("nan"), None
    print("C=%d %dx%d: eager %.3f ms per call, graph replay %.3f ms (%s; result identical to eager: %s)" % (C, h, h, eager, replay, status, same))
