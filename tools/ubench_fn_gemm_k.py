"""fresco_fn_gemm: fixed cost (prologue + epilogue) vs cost per 32-wide K chunk, at N = 1024 / 128 / 64: python tools/ubench_fn_gemm_k.py"""
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
M = 65536
for N in (1024, 128, 64):
    rows = []
    for K in (32, 64, 128, 256, 512):
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        _, xs = ops.fn_prep(x)
        _, ws = ops.fn_prep(w, scale=ops.FN_W_SCALE)
        out = torch.empty(M, N, device=dev)
        r = dict(K=K, f32=round(t(lambda: ops.fn_gemm(xs, ws, N, K, out_f32=out)), 1),
                 planes=round(t(lambda: ops.fn_gemm(xs, ws, N, K, want_f32=False, want_split=True)), 1),
                 gelu_planes=round(t(lambda: ops.fn_gemm(xs, ws, N, K, act=2, want_f32=False, want_split=True)), 1),
                 both=round(t(lambda: ops.fn_gemm(xs, ws, N, K, out_f32=out, want_split=True)), 1))
        rows.append(r)
    wgs = (M // 256) * ((N + 127) // 128 if N > 64 else 1)
    print("M=%d N=%d (%d workgroups = %.1f rounds of 256):" % (M, N, wgs, wgs / 256))
    for r in rows: print("   ", r)
