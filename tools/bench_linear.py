"""fresco_linear vs torch.nn.functional.linear at the cfg2 projection shapes (us per call, CUDA events).
FRESCO_LINEAR=resident|tiled forces one of the two kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fresco_amd import ops

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

g = torch.Generator().manual_seed(0)
for (M, C) in ((16 * 4096, 320), (16 * 1024, 640), (8474, 320), (2112, 640)):
    x = torch.randn(M, C, generator=g).half().cuda()
    Ws = [(torch.randn(C, C, generator=g) / C ** 0.5).half().cuda() for _ in range(3)]
    b = torch.randn(C, generator=g).half().cuda()
    lin3 = t(lambda: [torch.nn.functional.linear(x, w) for w in Ws])
    fus3 = t(lambda: ops.linear(x, Ws))
    fus2 = t(lambda: ops.linear(x, Ws[:2]))
    lin1 = t(lambda: torch.nn.functional.linear(x, Ws[0], b))
    fus1 = t(lambda: ops.linear(x, [Ws[0]], [b]))
    print("M=%d C=%d: q,k,v torch %.1f us  fused %.1f us | two projections %.1f us | out-proj (bias) torch %.1f us  ours %.1f us"
          % (M, C, lin3, fus3, fus2, lin1, fus1))
