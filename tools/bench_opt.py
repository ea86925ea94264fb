"""cfg3 probe: optimize_feature (20 Adam iterations) + feature-space warp_tensor at the four decoder
resolutions of an 8-frame 512^2 batch; per-kernel HIP-event timings via the library's opt-in profiler.
`measure()` is also called by bench.py (auxiliary `cfg3` entry of its JSON line)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {4: "prep", 6: "colnorm", 7: "gram", 8: "sv", 9: "adam_update"}  # 4: residual signs + normalisation + fp16 splits (one launch); 6 only on the generic path
LAYERS = ((1280, 8), (1280, 16), (1280, 32), (640, 64))
if os.environ.get("BENCH_OPT_LAYERS"):  # experiments: a subset, e.g. "3" or "2,3"
    LAYERS = tuple(LAYERS[int(i)] for i in os.environ["BENCH_OPT_LAYERS"].split(","))


PEAK_F16_DENSE = 2.5e15     # MI355X_MICROARCH.md: dense fp16 MFMA (the Gram / S V products run as split-fp16 MFMAs)
PEAK_HBM = 8.0e12           # spec; ~6.3e12 achievable (same guide)


def _inputs(N, R, dev, g):
    base = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1)
    bwd = base + 0.3 * torch.randn(N, 2, R, R, generator=g)
    flows = [(-bwd).to(dev), bwd.to(dev)]
    occs = [(torch.rand(N, R, R, generator=g) < 0.1).float().to(dev) for _ in range(2)]
    sal = torch.rand(N, 1, R // 2, R // 2, generator=g).to(dev)
    return flows, occs, sal


def _pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a config-3 kernel from the newest committed counter passes over (640, 64 x 64)
    (profiles/r*_pmc_opt_C640_h64.csv, written by tools/pmc_opt.sh: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950, + WRITE_SIZE, both KiB per launch, separate --pmc passes).  None without a profile."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_opt_C640_h64.csv")))
    if not files:
        return None
    for line in open(files[-1]):
        f = line.strip().split(",")
        if len(f) == 5 and kernel_substr in f[0]:
            try:
                return dict(bytes=int((2.0 * float(f[2]) + float(f[3])) * 1024), source=os.path.basename(files[-1]))
            except ValueError:
                pass
    return None


def _dominant_roofline(kern):
    """roofline of cfg3's dominant kernel: the Gram product at the largest layer (C = 640, 64 x 64).  `achieved` counts
    the ALGORITHMIC flop of one G = V V^T (2 B hw^2 C, what the reference's bmm computes) against the dense fp16 MFMA
    peak the kernel's instructions run at; `frac_executed` counts what it executes (upper triangle, three split-fp16
    products per fp32-accurate product)."""
    k = kern.get("C640_h64", {}).get("gram_roofline")
    if not k:
        return None
    pmc = _pmc_traffic("gram16y_kernel") or {}
    return dict(bound="mfma", kernel="gram16y_kernel at (C 640, 64 x 64)", achieved=k["algorithmic_tflops"],
                peak=PEAK_F16_DENSE / 1e12, unit="TFLOP/s", frac=k["frac_algorithmic"], frac_executed=k["frac_executed"],
                traffic=pmc.get("bytes"), traffic_source=pmc.get("source"),
                # targets (fp32, upper triangle) + sign bytes (direct + mirror position) + the (hi, lo) pixel-major operands once
                algorithmic_bytes_per_launch=int(16 * 4096 * 4096 * 4 // 2 + 16 * 4096 * 4096 + 16 * 640 * 4096 * 4),
                note="the Gram and S V products are fp32-accurate products built from 3 / 2 fp16 MFMAs on hi / lo halves "
                     "(exact to ~2^-22); per-kernel fractions of every launch, MFMA- and HBM-bound alike, are in "
                     "kernel_avg_us.*_roofline")


def measure(iters=20, N=8, R=512, dev="cuda", verbose=False, baselines=True, reps=5):
    """cfg3's extra work per denoising step: optimize_feature + feature-space warp_tensor at the inputs of the four
    up-blocks.  Ours: 1 warm-up + `reps` timed runs per layer; `per_layer_ms` / `ms_per_step` are the MEAN over ALL of
    them (nothing is dropped: round 4 dropped the slowest run because about one run in a hundred carries a 5 - 25 ms stall
    that no kernel accounts for -- tools/stall_hunt.py looks for its cause; until it is explained the line carries mean,
    median, min and max).  Per-kernel HIP-event times from an instrumented run.  Baselines ("time 2 Adam iterations x 10"): the reference's autograd + Adam op sequence
    on the same GPU (oracle/torch_opt_path.py, pinned against the reference goldens) and the analytic CPU port."""
    import fresco_amd
    from fresco_amd import _lib, ops
    g = torch.Generator().manual_seed(0)
    flows, occs, sal = _inputs(N, R, dev, g)
    lib = _lib.load()
    total = 0.0
    per_layer, kern, torch_ms, cpu_ms = [], {}, [], []
    per_layer_stats, total_median = [], 0.0
    miss_extra = []
    for C, h in LAYERS:
        hw = h * h
        x = torch.randn(2 * N, C, h, h, generator=g).half().to(dev)
        tgt = ops.gram_target(torch.randn(2 * N, C, h, h, generator=g).to(dev))
        times = []
        for rep in range(reps + 1):  # the first pass pays for workspace / allocator growth: warm-up, not counted
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=iters)
            fresco_amd.warp_tensor(out, flows, occs, sal, 2)
            torch.cuda.synchronize()
            if rep > 0:
                times.append(time.perf_counter() - t0)
        assert os.environ.get("BENCH_OPT_NOCHECK") or torch.isfinite(out.float()).all()  # (NOCHECK: timing ablations)
        kept = times
        mean = sum(kept) / len(kept)
        med = sorted(kept)[len(kept) // 2]
        per_layer.append(round(1e3 * mean, 3))
        per_layer_stats.append(dict(mean=round(1e3 * mean, 3), median=round(1e3 * med, 3), min=round(1e3 * min(kept), 3),
                                    max=round(1e3 * max(kept), 3), runs=[round(1e3 * t, 3) for t in kept]))
        total += mean
        total_median += med
        # The timed runs above hit fresco_amd.warp's cache of the tensors derived from flows / occlusions / saliency (resized
        # flows, pooled + dilated occlusions, warped saliency): constants of a batch of frames, which the pipeline computes
        # once per layer and batch and reuses over its 10 - 15 optimised steps.  One more call with the cache emptied (warm
        # allocator) gives the cost of that miss; the line carries it separately and amortised over 10 steps.
        from fresco_amd import warp as _warp
        _warp.invalidate_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=iters)
        fresco_amd.warp_tensor(out, flows, occs, sal, 2)
        torch.cuda.synchronize()
        miss_extra.append(max(0.0, 1e3 * ((time.perf_counter() - t0) - med)))
        # instrumented run: per-kernel means
        lib.fresco_prof_enable(4096)
        fresco_amd.optimize_feature(x, flows, occs, [tgt], iters=iters)
        torch.cuda.synchronize()
        lib.fresco_prof_disable()
        cap = 4096
        tags = (ctypes.c_int * cap)(); dims = (ctypes.c_int * (4 * cap))(); ms = (ctypes.c_float * cap)()
        n = lib.fresco_prof_read(cap, tags, dims, ms)
        agg = {}
        for i in range(n):
            agg.setdefault(NAMES.get(tags[i], str(tags[i])), []).append(ms[i])
        one = 2.0 * 2 * N * hw * hw * C  # algorithmic flop of one Gram / one S V product
        kl = {k: round(1e3 * sum(v) / len(v), 1) for k, v in agg.items()}
        for k in ("gram", "sv"):
            if k in agg:
                t = sum(agg[k]) / len(agg[k]) * 1e-3
                # fp16-split forms: gram = 3 products (hi.hi + hi.lo + lo.hi) over the tiles it computes (256 x 128 tiles with
                # tj >= 2 ti: the upper triangle + one 128 x 128 block per 256 rows; 128 x 128 tiles with tj >= ti for launches that
                # take gram16z; ALL tiles for planes <= 256 pixels),
                # sv = 2 products (S exact)
                if k == "gram":
                    n128, n256 = hw // 128, hw // 256
                    tiles_y = n256 * n128 - n256 * (n256 - 1)
                    if hw < 512:
                        share = 1.0
                    elif tiles_y * 2 * N < 1024 and os.environ.get("FRESCO_GRAM_Z", "1") != "0":
                        share = (n128 * (n128 + 1) / 2.0) / (n128 * n128)  # gram16z: 128 x 128 tiles (ti, tj >= ti), all of each
                    else:
                        share = tiles_y * 2.0 / (n128 * n128)              # gram16y: 256 x 128 tiles with tj >= 2 ti
                    execd = one * 3.0 * share
                else:
                    execd = one * 2.0
                kl[k + "_roofline"] = dict(bound="mfma", algorithmic_tflops=round(one / t / 1e12, 1),
                                           frac_algorithmic=round(one / t / PEAK_F16_DENSE, 3),
                                           executed_fp16_tflops=round(execd / t / 1e12, 1),
                                           frac_executed=round(execd / t / PEAK_F16_DENSE, 3))
        # the HBM-bound passes: algorithmic bytes per launch (fp32 tensors of B*C*hw elements; S V's dV^T counted with its reader)
        el = 4.0 * 2 * N * C * hw
        # prep: x once, signs (2 x 1 byte) and four fp16 copies out; adam: x, m, v, dV in + signs, x, m, v out
        for k, nbytes in (("prep", el + el / 2 + 2 * el), ("adam_update", 4 * el + el / 2 + 3 * el)):
            if k in agg:
                t = sum(agg[k]) / len(agg[k]) * 1e-3
                kl[k + "_roofline"] = dict(bound="hbm", algorithmic_gb=round(nbytes / 1e9, 3),
                                           achieved_tbs=round(nbytes / t / 1e12, 2), frac=round(nbytes / t / PEAK_HBM, 3))
        kern["C%d_h%d" % (C, h)] = kl
        if verbose:
            print("layer C=%d h=%d: %.2f ms (mean of %d) | per launch us: %s" % (C, h, 1e3 * mean, len(kept), kl))
        if baselines:
            from oracle import torch_opt_path as TO
            from oracle import fresco_oracle as O
            xf = x.float()
            with torch.no_grad():
                TO.optimize_feature(xf, flows, occs, [tgt], iters=1)  # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                TO.optimize_feature(xf, flows, occs, [tgt], iters=2)
                torch.cuda.synchronize()
                torch_ms.append(1e3 * (time.perf_counter() - t0) * iters / 2)
            xc, fc, oc, tc = xf.cpu(), [f.cpu() for f in flows], [o.cpu() for o in occs], tgt.cpu()
            t0 = time.perf_counter()
            O.optimize_feature(xc, fc, oc, [tc], iters=2)
            cpu_ms.append(1e3 * (time.perf_counter() - t0) * iters / 2)
            del xc, tc
        del x, tgt, out
    res = dict(ms_per_step=round(1e3 * total, 2), per_layer_ms=per_layer, ms_per_step_median=round(1e3 * total_median, 2),
               per_layer_stats=per_layer_stats,
               timing="1 warm-up + %d timed runs per layer; ms_per_step / per_layer_ms = mean of ALL %d (nothing dropped); median, "
                      "min, max and every run in per_layer_stats.  The timed runs find the flow / occlusion / saliency-derived "
                      "tensors in fresco_amd.warp's per-batch cache (as steps 2.. of a keyframe batch do); the cost of a miss is "
                      "derived_tensor_miss_ms" % (reps, reps),
               derived_tensor_miss_ms=dict(per_layer=[round(v, 3) for v in miss_extra], total=round(sum(miss_extra), 3),
                                           ms_per_step_with_miss_amortised_over_10_steps=round(1e3 * total + sum(miss_extra) / 10.0, 2),
                                           note="extra time of one call per layer with the cache emptied (resize / max-pool / "
                                                "dilate / saliency warps); paid once per batch of frames and layer; rounds 1-4 "
                                                "paid it on every call"),
               workload="cfg3 extra work per denoising step: optimize_feature (%d Adam iterations, intra_weight 100) + "
                        "feature-space warp_tensor at the inputs of the four up-blocks, %d frames %dx%d" % (iters, N, R, R),
               roofline=_dominant_roofline(kern),
               kernel_avg_us=kern)
    if baselines:
        res["torch_gpu_baseline"] = dict(ms_per_step=round(sum(torch_ms), 1), per_layer_ms=[round(v, 1) for v in torch_ms], kind="port",
                                         sample="oracle/torch_opt_path.optimize_feature (the reference's autograd + torch.optim.Adam op "
                                                "sequence, fp32, same GPU): 2 Adam iterations timed per layer, x%d" % (iters // 2))
        res["cpu_baseline"] = dict(ms_per_step=round(sum(cpu_ms), 1), per_layer_ms=[round(v, 1) for v in cpu_ms], cores=torch.get_num_threads(),
                                   kind="port", sample="oracle.optimize_feature (analytic gradients, fp32, torch CPU): 2 Adam iterations "
                                                       "timed per layer, x%d" % (iters // 2))
        res["speedup_vs_torch_gpu"] = round(sum(torch_ms) / (1e3 * total), 2)
    return res


if __name__ == "__main__":
    import json
    r = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 20, verbose=True, baselines="--no-baselines" not in sys.argv)
    print(json.dumps(r))
    print("cfg3 extra per denoising step (4 layers): %.1f ms" % r["ms_per_step"])
