"""cfg3 probe: optimize_feature (20 Adam iterations) + feature-space warp_tensor at the four decoder
resolutions of an 8-frame 512^2 batch; per-kernel HIP-event timings via the library's opt-in profiler.
`measure()` is also called by bench.py (auxiliary `cfg3` entry of its JSON line)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {4: "temporal_sign", 5: "temporal_grad", 6: "colnorm", 7: "gram", 8: "sv", 9: "adam_update"}
LAYERS = ((1280, 8), (1280, 16), (1280, 32), (640, 64))


def measure(iters=20, N=8, R=512, dev="cuda", verbose=False):
    import fresco_amd
    from fresco_amd import _lib, ops
    g = torch.Generator().manual_seed(0)
    base = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1)
    bwd = (base + 0.3 * torch.randn(N, 2, R, R, generator=g)).to(dev)
    flows = [-bwd, bwd]
    occs = [(torch.rand(N, R, R, generator=g) < 0.1).float().to(dev) for _ in range(2)]
    sal = torch.rand(N, 1, R // 2, R // 2, generator=g).to(dev)
    lib = _lib.load()
    total = 0.0
    per_layer = []
    for C, h in LAYERS:
        x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
        tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
        best = None
        for rep in range(3):  # the first pass pays for workspace / allocator growth; keep the fastest
            torch.cuda.synchronize()
            if rep == 2 and verbose:
                lib.fresco_prof_enable(4096)
            t0 = time.perf_counter()
            out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=iters)
            torch.cuda.synchronize()
            t_opt = time.perf_counter() - t0
            t0 = time.perf_counter()
            fresco_amd.warp_tensor(out, flows, occs, sal, 2)
            torch.cuda.synchronize()
            t_warp = time.perf_counter() - t0
            if best is None or t_opt + t_warp < best:
                best = t_opt + t_warp
        assert torch.isfinite(out.float()).all()
        per_layer.append(round(1e3 * best, 3))
        total += best
        if verbose:
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
            print("layer C=%d h=%d: optimize_feature(%d it) %.2f ms, warp_tensor %.3f ms | per launch: %s | algorithmic TFLOP/s: %s"
                  % (C, h, iters, 1e3 * t_opt, 1e3 * t_warp, line, {k: round(v, 1) for k, v in tf.items()}))
        del x, tgt, out
    return dict(ms_per_step=round(1e3 * total, 2), per_layer_ms=per_layer,
                workload="cfg3 extra work per denoising step: optimize_feature (%d Adam iterations, intra_weight 100) + "
                         "feature-space warp_tensor at the inputs of the four up-blocks, %d frames %dx%d" % (iters, N, R, R))


if __name__ == "__main__":
    r = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 20, verbose=True)
    print("cfg3 extra per denoising step (4 layers): %.1f ms" % r["ms_per_step"])
