"""Kernel times at the shapes ONE rank sees in a frame-sharded 8-GPU run of config 2 (one frame of each CFG half per rank):
the flash launches (queries of 2 batch rows against the full cross-frame key set), the key pack, the projections and the
packed temporal kernel -- the inputs of the scaling estimate in DESIGN.md section 6.  One GPU, HIP events.
usage: python tools/bench_shard_shapes.py [world]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fresco_amd import ops, _lib

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
g = torch.Generator().manual_seed(0)
N, chunk, H = 8, 2, 8
n_loc = N // world
B_loc = chunk * n_loc


def prof(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    cap = 8 * reps + 16
    lib.fresco_prof_enable(cap)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    tags = (ctypes.c_int * cap)(); dims = (ctypes.c_int * (4 * cap))(); ms = (ctypes.c_float * cap)()
    n = lib.fresco_prof_read(cap, tags, dims, ms)
    lib.fresco_prof_disable()
    agg = {}
    for i in range(n):
        agg.setdefault(tags[i], []).append(ms[i] * 1e3)
    return {k: sum(v) / len(v) for k, v in agg.items()}


for name, HW, C, D in (("up_blocks.3", 4096, 320, 40), ("up_blocks.2", 1024, 640, 80)):
    q = (0.3 * torch.randn(B_loc, HW, C, generator=g)).half().cuda()
    mask = torch.rand(N, HW, generator=g) < 0.004
    mask[0] = True
    M = int(mask.sum())
    # exchange-buffer addressing: rows t*chunk + c of a (rows, chunk, 2C) buffer (fresco_amd/dist.py)
    R = HW + world * 16
    buf = (0.3 * torch.randn(R, chunk, 2 * C, generator=g)).half().cuda()
    table = (torch.randperm(R, generator=g)[:M].sort().values * chunk).to(torch.int32).cuda()
    flat = buf.view(-1, 2 * C)
    t = prof(lambda: ops.attention(q, flat[:, :C], flat[:, C:], H, 1 / math.sqrt(D), kv_rows=table, n_groups=chunk, M=M, group_rows=1))
    x = torch.randn(B_loc * HW, C, generator=g).half().cuda()
    Ws = [(torch.randn(C, C, generator=g) / C ** 0.5).half().cuda() for _ in range(3)]
    bias = torch.randn(C, generator=g).half().cuda()
    tq = prof(lambda: ops.linear(x, Ws))
    to = prof(lambda: ops.linear(x, [Ws[0]], [bias])) if C == 320 else {10: float("nan")}
    print("%s  rank shard of %d: flash %.1f us, kv_pack %.1f us (M = %d), q,k,v projection (%d rows) %.1f us, out projection %.1f us"
          % (name, world, t[1], t[2], M, B_loc * HW, tq[10], to[10]))
