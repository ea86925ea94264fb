"""f3 timing: GMFlow (fresco_amd.gmflow, stand-in weights) + the rest of get_flow_and_interframe_paras for an
8-frame 512x512 batch on one MI355X; per-launch times of fresco_attn_f32 via the library's profiler.
`measure()` is also called by bench.py (auxiliary `f3_gmflow` entry of its JSON line)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

PEAK_F16_DENSE = 2.5e15  # MI355X_MICROARCH.md: dense fp16 MFMA (fresco_attn_f32 runs split-fp16 products on it)
PEAK_F32_MFMA = 157.3e12  # the fp32 matrix peak: what an fp32-MFMA attention is bounded by


def measure(N=8, R=512, dev="cuda", reps=3, verbose=False):
    import closed_form as cf
    import fresco_amd
    import fresco_amd.gmflow as G
    from fresco_amd import _lib
    m = G.GMFlow().eval()
    sd = m.state_dict()
    m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
    m = m.to(dev)
    frames = [f.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy() for f in cf.gmflow_frames(N, R, R)]
    imgs = torch.stack([torch.from_numpy(f).permute(2, 0, 1).float() for f in frames]).to(dev)
    nxt = list(range(1, N)) + [0]
    kw = dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow=True)
    lib = _lib.load()
    t_net = t_all = 1e9
    with torch.no_grad():
        for rep in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m(imgs, imgs[nxt], **kw)
            torch.cuda.synchronize(); t_net = min(t_net, time.perf_counter() - t0)
            t0 = time.perf_counter()
            fresco_amd.get_flow_and_interframe_paras(m, frames)
            torch.cuda.synchronize(); t_all = min(t_all, time.perf_counter() - t0)
        lib.fresco_prof_enable(4096)
        m(imgs, imgs[nxt], **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_disable()
    cap = 4096
    tags = (ctypes.c_int * cap)(); dims = (ctypes.c_int * (4 * cap))(); ms = (ctypes.c_float * cap)()
    n = lib.fresco_prof_read(cap, tags, dims, ms)
    agg = {}
    for i in range(n):
        if tags[i] == 11:
            agg.setdefault(tuple(dims[4 * i: 4 * i + 4]), []).append(ms[i])
    # the dense layers (csrc/flownet.hip, tag 12: dims = {M, N, K, kernel height}): per-shape means and the total
    fn = {}
    for i in range(n):
        if tags[i] == 12:
            fn.setdefault(tuple(dims[4 * i: 4 * i + 4]), []).append(ms[i])
    fn_tot = sum(sum(v) for v in fn.values())
    fn_flop = sum(2.0 * k[0] * k[1] * k[2] * len(v) for k, v in fn.items())
    fn_launches = {"M%d_N%d_K%d_kh%d" % k: dict(launches=len(v), avg_us=round(1e3 * sum(v) / len(v), 1),
                                                algorithmic_tflops=round(2.0 * k[0] * k[1] * k[2] / (sum(v) / len(v) * 1e-3) / 1e12, 1))
                   for k, v in sorted(fn.items())}
    tot, launches, best = 0.0, {}, None
    for k, v in sorted(agg.items()):
        B, Lq, Lk, D = k
        flop = 4.0 * B * Lq * Lk * D  # algorithmic: QK^T + PV at head dim D (the value dim of the matching launch is 2)
        t = sum(v) / len(v) * 1e-3
        # executed on the fp16 pipe: 6 MFMAs per QK product (33-bit logits), 3 per PV product
        ex = 2.0 * B * Lq * Lk * D * 6 + 2.0 * B * Lq * Lk * D * 3
        launches["B%d_L%d_D%d" % (B, Lq, D)] = dict(launches=len(v), avg_us=round(1e6 * t, 1),
                                                     algorithmic_tflops=round(flop / t / 1e12, 1),
                                                     speedup_vs_fp32_mfma_bound=round(flop / t / PEAK_F32_MFMA, 3),  # (NOT a roofline fraction: the kernel runs on the fp16 pipe)
                                                     executed_fp16_tflops=round(ex / t / 1e12, 1),
                                                     frac_executed_of_fp16_peak=round(ex / t / PEAK_F16_DENSE, 3))
        tot += sum(v)
        if verbose:
            print("  attn_f32 (B=%d, Lq=%d, Lk=%d, D=%d): %d launches, %.1f us each, %.1f TFLOP/s (QK + PV at D)"
                  % (B, Lq, Lk, D, len(v), 1e6 * t, flop / t / 1e12))
        if Lq == 1024 and B >= 64:
            best = launches["B%d_L%d_D%d" % (B, Lq, D)]
    res = dict(workload="f3: GMFlow forward (stand-in weights) for %d frame pairs x 2 directions at %dx%d + occlusions, masks, "
                        "mappings (get_flow_and_interframe_paras), once per batch of frames" % (N, R, R),
               gmflow_forward_ms=round(1e3 * t_net, 2), get_flow_and_interframe_paras_ms=round(1e3 * t_all, 2),
               attention_ms_of_forward=round(tot, 2), timing="fastest of %d" % reps, attn_f32_launches=launches,
               dense_layers=dict(kernel="fn_gemm_kernel / fn_conv7_rgb_kernel (csrc/flownet.hip): convolutions as implicit GEMM + "
                                        "linear layers, fp32-accurate split-fp16 products (3 MFMAs per product)",
                                 ms_of_forward=round(fn_tot, 2), launches=sum(len(v) for v in fn.values()),
                                 algorithmic_tflops=round(fn_flop / max(fn_tot, 1e-9) / 1e9, 1),
                                 executed_fp16_tflops=round(3 * fn_flop / max(fn_tot, 1e-9) / 1e9, 1),
                                 frac_executed_of_fp16_peak=round(3 * fn_flop / max(fn_tot, 1e-9) / 1e9 / (PEAK_F16_DENSE / 1e12), 3),
                                 per_shape=fn_launches))
    # the same forward with PyTorch's own convolutions / norms / linears around the same attention kernel (rounds 1-4)
    os.environ["FRESCO_GMFLOW_LIBRARY_OPS"] = "1"
    t_lib = 1e9
    with torch.no_grad():
        for rep in range(0 if os.environ.get("BENCH_GMFLOW_NO_LIBRARY_LEG") else reps):  # (skipped under a kernel profiler)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m(imgs, imgs[nxt], **kw)
            torch.cuda.synchronize(); t_lib = min(t_lib, time.perf_counter() - t0)
    del os.environ["FRESCO_GMFLOW_LIBRARY_OPS"]
    res["gmflow_forward_library_ops_ms"] = round(1e3 * t_lib, 2) if t_lib < 1e8 else None
    if best:
        res["roofline"] = dict(bound="mfma", kernel="kv_split_kernel + attn_f32p_kernel<128,128> (swin window attention, B 64, L 1024, D 128)",
                               achieved=best["executed_fp16_tflops"], peak=PEAK_F16_DENSE / 1e12, unit="TFLOP/s",
                               frac=best["frac_executed_of_fp16_peak"],
                               algorithmic_tflops=best["algorithmic_tflops"],
                               note="fp32 attention as split-fp16 products (6 + 3 MFMAs per product pair): `achieved` counts "
                                    "executed fp16 flop; the algorithmic rate is %.2f x the 157 TFLOP/s an fp32-MFMA kernel "
                                    "is bounded by (a speed-up over that bound, not a fraction of this kernel's roof)" % best["speedup_vs_fp32_mfma_bound"])
    return res


if __name__ == "__main__":
    import json
    r = measure(verbose=True)
    print("GMFlow forward: %.1f ms;  get_flow_and_interframe_paras (network + occlusions + masks + mappings): %.1f ms;  "
          "attention %.2f ms of the forward" % (r["gmflow_forward_ms"], r["get_flow_and_interframe_paras_ms"],
                                                 r["attention_ms_of_forward"]))
    print(json.dumps(r))
