"""cfg3 probe: optimize_feature (20 Adam iterations) + feature-space warp_tensor at the four decoder
resolutions of an 8-frame 512^2 batch; per-kernel HIP-event timings via the library's opt-in profiler."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fresco_amd
from fresco_amd import _lib, ops

NAMES = {4: "temporal_sign", 5: "temporal_grad", 6: "colnorm", 7: "gram", 8: "sv", 9: "adam_update"}
dev = "cuda"
N, R, iters = 8, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0)
base = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1)
bwd = (base + 0.3 * torch.randn(N, 2, R, R, generator=g)).to(dev)
flows = [-bwd, bwd]
occs = [(torch.rand(N, R, R, generator=g) < 0.1).float().to(dev) for _ in range(2)]
sal = torch.rand(N, 1, R // 2, R // 2, generator=g).to(dev)
lib = _lib.load()
total = 0.0
for C, h in ((1280, 8), (1280, 16), (1280, 32), (640, 64)):
    x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
    tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
    for rep in range(2):
        torch.cuda.synchronize()
        if rep == 1:
            lib.fresco_prof_enable(4096)
        t0 = time.perf_counter()
        out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=iters)
        torch.cuda.synchronize()
        t_opt = time.perf_counter() - t0
        t0 = time.perf_counter()
        w = fresco_amd.warp_tensor(out, flows, occs, sal, 2)
        torch.cuda.synchronize()
        t_warp = time.perf_counter() - t0
    lib.fresco_prof_disable()
    cap = 4096
    tags = (ctypes.c_int * cap)(); dims = (ctypes.c_int * (4 * cap))(); ms = (ctypes.c_float * cap)()
    n = lib.fresco_prof_read(cap, tags, dims, ms)
    agg = {}
    for i in range(n):
        agg.setdefault(NAMES.get(tags[i], tags[i]), []).append(ms[i])
    hw = h * h
    gflop = 2.0 * 2 * N * hw * hw * C / 1e9  # one GEMM (gram or sv)
    line = ", ".join("%s %.1f us" % (k, 1e3 * sum(v) / len(v)) for k, v in agg.items())
    tf = {k: gflop / (sum(v) / len(v)) for k, v in agg.items() if k in ("gram", "sv")}
    print("layer C=%d h=%d: optimize_feature(%d it) %.2f ms, warp_tensor %.3f ms | per launch: %s | fp32 MFMA TFLOP/s: %s"
          % (C, h, iters, 1e3 * t_opt, 1e3 * t_warp, line, {k: round(v, 1) for k, v in tf.items()}))
    total += t_opt + t_warp
    assert torch.isfinite(out.float()).all()
print("cfg3 extra per denoising step (4 layers): %.1f ms" % (1e3 * total))
