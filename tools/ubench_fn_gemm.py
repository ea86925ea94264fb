"""timing of fresco_fn_gemm at the flow network's linear shapes (what bounds the short-K products?): python tools/ubench_fn_gemm.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fresco_amd.ops as ops

def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

dev = "cuda"
g = torch.Generator().manual_seed(0)
for M in (65536, 16384, 262144):
    for K, N in ((128, 128), (128, 64), (128, 384), (256, 1024), (1024, 128), (256, 128)):
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        _, xs = ops.fn_prep(x)
        _, ws = ops.fn_prep(w, scale=ops.FN_W_SCALE)
        tab = torch.randperm(M, generator=g).to(torch.int32).to(dev)
        out = torch.empty(M, N, device=dev)
        r = dict(plain=t(lambda: ops.fn_gemm(xs, ws, N, K, out_f32=out)),
                 a_rows=t(lambda: ops.fn_gemm(xs, ws, N, K, a_rows=tab, out_f32=out)),
                 out_rows=t(lambda: ops.fn_gemm(xs, ws, N, K, out_rows=tab, out_f32=out)))
        if N % 8 == 0:
            r["planes_only"] = t(lambda: ops.fn_gemm(xs, ws, N, K, want_f32=False, want_split=True))
        flop = 2.0 * M * N * K
        print("M=%d K=%d N=%d: %s | plain: %.0f TFLOP/s alg, %.2f TB/s (A planes + fp32 out)" % (
            M, K, N, {k: round(v, 1) for k, v in r.items()}, flop / r["plain"] / 1e6, (M * K * 4 + M * N * 4) / r["plain"] / 1e6))
        del x, xs, out
