"""Root-cause hunt for the 5 - 25 ms stalls of tools/bench_opt.py: K calls of
optimize_feature + warp_tensor at one decoder-layer shape, each bracketed by a host clock AND a pair of HIP events on the
launch stream, with the collector's pauses and the caching allocator's counters sampled around every call:

  * wall >> events  -> the HOST stalled (GC pause, allocator hipMalloc / hipFree, a blocking runtime call): the GPU was idle
  * events ~ wall   -> the stall is on the GPU side (a kernel or a gap between kernels): look at the kernel trace of the
                       same run (rocprofv3 --kernel-trace: tools/trace_gaps.py prints the largest gaps)

usage: python tools/stall_hunt.py [calls=200] [layer=3] [split=default]      (layer: 0..3 = (1280,8) (1280,16) (1280,32) (640,64))
Prints one JSON line: per-call statistics, the slow calls with what was observed around them."""
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_opt  # noqa: E402
import fresco_amd  # noqa: E402
from fresco_amd import ops  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    layer = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if len(sys.argv) > 3 and sys.argv[3] != "default":
        os.environ["FRESCO_OPT_SPLIT"] = sys.argv[3]
    dev = "cuda"
    N, R = 8, 512
    C, h = bench_opt.LAYERS[layer]
    g = torch.Generator().manual_seed(0)
    flows, occs, sal = bench_opt._inputs(N, R, dev, g)
    x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
    tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
    gc_events = []

    def on_gc(phase, info):
        gc_events.append((time.perf_counter(), phase, info.get("generation")))

    gc.callbacks.append(on_gc)
    for _ in range(2):
        out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=20)
        fresco_amd.warp_tensor(out, flows, occs, sal, 2)
    torch.cuda.synchronize()
    rows = []
    for i in range(K):
        st0 = torch.cuda.memory_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=20)
        fresco_amd.warp_tensor(out, flows, occs, sal, 2)
        t_issued = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st1 = torch.cuda.memory_stats()
        rows.append(dict(i=i, wall_ms=1e3 * (t1 - t0), issue_ms=1e3 * (t_issued - t0), gpu_ms=e0.elapsed_time(e1),
                         t0=t0, t1=t1,
                         malloc=st1.get("num_device_alloc", 0) - st0.get("num_device_alloc", 0),
                         free=st1.get("num_device_free", 0) - st0.get("num_device_free", 0),
                         retries=st1.get("num_alloc_retries", 0) - st0.get("num_alloc_retries", 0)))
    gc.callbacks.remove(on_gc)
    wall = sorted(r["wall_ms"] for r in rows)
    med = wall[len(wall) // 2]
    slow = [r for r in rows if r["wall_ms"] > med + 2.0]
    for r in slow:
        r["gc_inside"] = [(round(1e3 * (t - r["t0"]), 2), ph, gen) for (t, ph, gen) in gc_events if r["t0"] <= t <= r["t1"]]
        r["verdict"] = ("host stall: the GPU finished in %.2f ms, the host needed %.2f ms to issue" % (r["gpu_ms"], r["issue_ms"])
                        if r["gpu_ms"] < med + 1.0 else
                        ("GPU-side: the events span %.2f ms (host issue took %.2f ms%s)"
                         % (r["gpu_ms"], r["issue_ms"], ": the GPU was waiting for the host" if r["issue_ms"] > med else "")))
    for r in rows:
        r.pop("t0"), r.pop("t1")
    res = dict(layer=dict(C=C, h=h), calls=K, split=os.environ.get("FRESCO_OPT_SPLIT", "default"),
               wall_ms=dict(median=round(med, 3), mean=round(sum(wall) / K, 3), min=round(wall[0], 3), max=round(wall[-1], 3),
                            p99=round(wall[int(0.99 * (K - 1))], 3)),
               gpu_ms_median=round(sorted(r["gpu_ms"] for r in rows)[K // 2], 3),
               issue_ms_median=round(sorted(r["issue_ms"] for r in rows)[K // 2], 3),
               slow_calls=[{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in slow],
               gc_collections=len([e for e in gc_events if e[1] == "stop"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
