"""Time fresco_temporal_attn alone (the library's HIP events around each launch) at the bench's layer shapes.
   python tools/bench_temporal.py [N] [res]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ctypes  # noqa: E402
import fresco_amd.ops as ops  # noqa: E402
from fresco_amd import _lib  # noqa: E402
import synth  # noqa: E402
from oracle import fresco_oracle as O  # noqa: E402  (only builds the trajectory maps of the synthetic flows)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = "cuda"
g = synth.gen(1)
flows, occs = synth.make_flows(N, R, g)
imgs = torch.rand(N, 3, R, R, generator=g)
for layer, C, down in (("L3", 320, 8), ("L2", 640, 16)):
    HW = (R // down) ** 2
    fwd_map, _, tmask = O.mapping_ind(flows[1], occs[1], imgs, scale=float(down))
    q, k, v = (torch.randn(2 * N, HW, C, generator=g).half().to(dev) for _ in range(3))
    fm, tm = fwd_map.to(dev), tmask.to(dev)
    scale = 0.2 / math.sqrt(C // 8)
    for _ in range(5):
        out = ops.temporal_attention(q, k, v, fm, tm, 8, scale, 2)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.fresco_prof_enable(64)
    for _ in range(20):
        out = ops.temporal_attention(q, k, v, fm, tm, 8, scale, 2)
    torch.cuda.synchronize()
    lib.fresco_prof_disable()
    tags, dims, ms = (ctypes.c_int * 64)(), (ctypes.c_int * 256)(), (ctypes.c_float * 64)()
    n = lib.fresco_prof_read(64, tags, dims, ms)
    t = sorted(ms[i] for i in range(n))
    best = t[len(t) // 2] * 1e3  # median, us (HIP events on the launch stream)
    byts = 4 * 2 * N * HW * C * 2
    print("temporal %s N=%d HW=%d C=%d: %.1f us/launch (median of 20, events around the kernel)  %.2f TB/s algorithmic"
          % (layer, N, HW, C, best, byts / best / 1e6))
