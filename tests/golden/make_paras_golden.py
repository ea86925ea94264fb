"""Generate tests/golden/paras_golden.npz by running the UNMODIFIED reference
`get_flow_and_interframe_paras` (src/diffusion_hacked.py:904-957) on CPU with a stand-in flow model that
returns closed-form flows (GMFlow weights are absent; everything AFTER the flow network is the
reference's own code: fb-consistency check, colour-difference occlusion refinement, cross-frame masks,
FLATTEN pixel mappings).  Build container only:  python tests/golden/make_paras_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as cf  # noqa: E402
import _ref_harness  # noqa: E402


case_inputs = cf.video_case


class FixedFlow:
    def __init__(self, fwd, bwd):
        self.fwd, self.bwd = fwd, bwd
        self.calls = []

    def __call__(self, a, b, **kw):
        self.calls.append((tuple(a.shape), tuple(b.shape), dict(kw)))
        return {"flow_preds": [torch.cat([self.fwd, self.bwd], 0)]}


def main():
    torch.set_num_threads(8)
    if "torchvision" not in sys.modules:
        _ref_harness._stub("torchvision")
    dh, fu, geo, ut = _ref_harness.load_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self  # utils.py:9, diffusion_hacked.py:911 hard-code .cuda()
    out = {}
    for tag, (N, H, W) in {"a": (4, 64, 64), "b": (3, 96, 160)}.items():
        frames, fwd, bwd = case_inputs(N, H, W)
        fm = FixedFlow(fwd, bwd)
        flows, occs, attn_mask, paras = dh.get_flow_and_interframe_paras(fm, frames)
        assert fm.calls[0][2] == dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1],
                                      pred_bidir_flow=True)
        out[tag + "_fwd_occ"] = occs[0].to(torch.uint8).numpy()
        out[tag + "_bwd_occ"] = occs[1].to(torch.uint8).numpy()
        fo, bo = geo.forward_backward_consistency_check(fwd, bwd)
        out[tag + "_fb_fwd_occ"] = fo.to(torch.uint8).numpy()
        out[tag + "_fb_bwd_occ"] = bo.to(torch.uint8).numpy()
        for i, m in enumerate(attn_mask):
            out["%s_attn_mask%d" % (tag, i)] = m.to(torch.uint8).numpy()
        for i in range(2):
            out["%s_fwd_map%d" % (tag, i)] = paras["fwd_mappings"][i].numpy()
            out["%s_bwd_map%d" % (tag, i)] = paras["bwd_mappings"][i].numpy()
            out["%s_imask%d" % (tag, i)] = paras["interattn_masks"][i].to(torch.uint8).numpy()
        print(tag, "occ sums", int(occs[0].sum()), int(occs[1].sum()), "fb-only", int(fo.sum()), int(bo.sum()),
              "mask rows", [int(m.sum()) for m in attn_mask])
    np.savez_compressed(os.path.join(HERE, "paras_golden.npz"), **out)
    print("wrote paras_golden.npz", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
