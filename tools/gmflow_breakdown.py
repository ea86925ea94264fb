"""Where the time of get_flow_and_interframe_paras goes (8 frames x 512^2, stand-in weights): host-side frame conversion,
the flow network (native dense layers / library ops), occlusions + masks, the two trajectory-map scales.
usage: python tools/gmflow_breakdown.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import closed_form as cf
import fresco_amd
import fresco_amd.gmflow as G
from fresco_amd import ops, paras
from fresco_amd.mapping import cross_frame_masks, get_mapping_ind


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best, r


N, R, dev = 8, 512, "cuda"
m = G.GMFlow().eval()
sd = m.state_dict()
m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
m = m.to(dev)
frames = [f.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy() for f in cf.gmflow_frames(N, R, R)]
kw = dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow=True)
nxt = list(range(1, N)) + [0]
with torch.no_grad():
    ms, images = t(lambda: torch.stack([torch.from_numpy(i).permute(2, 0, 1).float() for i in frames], 0).to(dev))
    print("frames -> device tensor: %.2f ms" % ms)
    ms, out = t(lambda: m(images, images[nxt], **kw)["flow_preds"][-1])
    print("flow network (native dense layers): %.2f ms" % ms)
    os.environ["FRESCO_GMFLOW_LIBRARY_OPS"] = "1"
    ms, _ = t(lambda: m(images, images[nxt], **kw)["flow_preds"][-1])
    print("flow network (library ops): %.2f ms" % ms)
    del os.environ["FRESCO_GMFLOW_LIBRARY_OPS"]
    fwd, bwd = (x.float().contiguous() for x in out.chunk(2))
    ms, (fo, bo) = t(lambda: ops.flow_occlusion(fwd, bwd, images, 0.01, 0.5, 255 * 0.25))
    print("occlusions: %.2f ms" % ms)
    ms, _ = t(lambda: cross_frame_masks(bo, (8.0, 16.0, 32.0)))
    print("cross-frame masks: %.2f ms" % ms)
    it = images / 255.0 * 2.0 - 1.0
    for s in (8.0, 16.0):
        ms, _ = t(lambda: get_mapping_ind(bwd, bo, it, scale=s))
        print("get_mapping_ind scale %g: %.2f ms" % (s, ms))
    ms, _ = t(lambda: fresco_amd.get_flow_and_interframe_paras(m, frames))
    print("get_flow_and_interframe_paras: %.2f ms" % ms)
