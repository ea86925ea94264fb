"""Kernel-only timing of the cross-frame / spatial attention launches (HIP events from fresco_prof_*):
cfg2 up_blocks.3 (HW=4096, D=40) and up_blocks.2 (HW=1024, D=80), small-M and block-occlusion large-M masks.
usage: python tools/bench_flash.py [reps] [gain] [case]    (gain scales q and k: 1.0 = N(0,1) per channel, logit bound ~19 log2
units, the exact-scale regime; 0.3 = small logits, the regime of the headline bench)"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fresco_amd import ops, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
only = sys.argv[3] if len(sys.argv) > 3 else None   # e.g. "spatial40": one (label, D) case
lib = _lib.load()
g = torch.Generator().manual_seed(0)
N, chunk, H = 8, 2, 8
B = chunk * N
for (HW, C, D) in ((4096, 320, 40), (1024, 640, 80)):
    q = (gain * torch.randn(B, HW, C, generator=g)).half().cuda()
    k = (gain * torch.randn(B, HW, C, generator=g)).half().cuda()
    v = torch.randn(B, HW, C, generator=g).half().cuda()
    for label, p in (("small-M", 0.004), ("large-M", 0.5), ("spatial", None)):
        if only and only != "%s%d" % (label, D):
            continue
        if p is None:
            kw = dict(n_groups=B, M=HW, group_rows=HW)
            scale = 0.2 / math.sqrt(D)
        else:
            mask = torch.rand(N, HW, generator=g) < p
            mask[0] = True
            rows = mask.reshape(-1).nonzero().squeeze(1).to(torch.int32).cuda()
            kw = dict(kv_rows=rows, n_groups=chunk, M=rows.numel(), group_rows=N * HW)
            scale = 1 / math.sqrt(D)
        for _ in range(3):
            ops.attention(q, k, v, H, scale, **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_enable(4 * reps + 8)
        for _ in range(reps):
            ops.attention(q, k, v, H, scale, **kw)
        torch.cuda.synchronize()
        n = 4 * reps + 8
        tags = (ctypes.c_int * n)(); dims = (ctypes.c_int * (4 * n))(); ms = (ctypes.c_float * n)()
        got = lib.fresco_prof_read(n, tags, dims, ms)
        lib.fresco_prof_disable()
        fl = [ms[i] for i in range(got) if tags[i] == 1]
        pk = [ms[i] for i in range(got) if tags[i] == 2]
        M = kw["M"]
        flop = 4.0 * B * HW * M * C
        t = sum(fl) / len(fl)
        print("%-8s HW=%d D=%d M=%5d: flash %7.1f us (min %7.1f)  pack %5.1f us  %6.0f TFLOP/s algorithmic = %.3f of 2.5 PF"
              % (label, HW, D, M, t * 1e3, min(fl) * 1e3, sum(pk) / len(pk) * 1e3, flop / t / 1e9, flop / t / 1e9 / 2500))
