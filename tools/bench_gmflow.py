"""f3 timing: GMFlow (fresco_amd.gmflow, stand-in weights) + the rest of get_flow_and_interframe_paras for an
8-frame 512x512 batch on one MI355X; per-launch times of fresco_attn_f32 via the library's profiler."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import closed_form as cf
import fresco_amd
import fresco_amd.gmflow as G
from fresco_amd import _lib

N, R = 8, 512
m = G.GMFlow().eval()
sd = m.state_dict()
m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
m = m.cuda()
frames = [f.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy() for f in cf.gmflow_frames(N, R, R)]
imgs = torch.stack([torch.from_numpy(f).permute(2, 0, 1).float() for f in frames]).cuda()
nxt = list(range(1, N)) + [0]
kw = dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow=True)
lib = _lib.load()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    flow = m(imgs, imgs[nxt], **kw)["flow_preds"][-1]
    torch.cuda.synchronize(); t_net = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = fresco_amd.get_flow_and_interframe_paras(m, frames)
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("GMFlow forward, %d pairs x 2 directions at %dx%d: %.1f ms;  get_flow_and_interframe_paras (network + occlusions + masks + "
      "mappings): %.1f ms" % (N, R, R, 1e3 * t_net, 1e3 * t_all))
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
tot = 0.0
for k, v in sorted(agg.items()):
    B, Lq, Lk, D = k
    flop = 4.0 * B * Lq * Lk * D
    print("  attn_f32 (B=%d, Lq=%d, Lk=%d, D=%d): %d launches, %.1f us each, %.1f TFLOP/s (QK + PV at D)" % (B, Lq, Lk, D, len(v), 1e3 * sum(v) / len(v), flop / (sum(v) / len(v)) / 1e9))
    tot += sum(v)
print("  attention total %.2f ms of the forward" % tot)
