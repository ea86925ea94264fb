"""fresco_attn_f32 alone at one shape (timing + a target for rocprofv3 --pmc): python tools/run_attn32_only.py reps B L D Dv"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fresco_amd.ops as ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, L, D, Dv = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (64, 1024, 128, 128)
g = torch.Generator().manual_seed(0)
q = torch.randn(B, L, D, generator=g).cuda(); k = torch.randn(B, L, D, generator=g).cuda(); v = torch.randn(B, L, Dv, generator=g).cuda()
sc = D ** -0.5
for _ in range(2): ops.attention_f32(q, k, v, sc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.attention_f32(q, k, v, sc)
e1.record(); torch.cuda.synchronize()
print("attention_f32 B=%d L=%d D=%d Dv=%d: %.1f us per call (range pass + kv_split + attention + guarded exact kernel)" % (B, L, D, Dv, 1e3 * e0.elapsed_time(e1) / reps))
