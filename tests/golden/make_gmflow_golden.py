"""Generate tests/golden/gmflow_golden.npz: the UNMODIFIED reference GMFlow (src/ebsynth/deps/gmflow/gmflow) in
FRESCO's configuration (run_fresco.py:38-45), on CPU in fp32, with closed-form stand-in weights
(closed_form.gmflow_param: the published checkpoint is absent) on closed-form frames, called exactly as
get_flow_and_interframe_paras calls it.  Build container only:  python tests/golden/make_gmflow_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as cf  # noqa: E402

CASES = {"a": (2, 96, 128), "b": (3, 64, 96), "c": (2, 256, 256)}
# case c (round 6): every map of the encoder is whole 16 x 16 patches and 64-pixel rows -- the sizes at which the GPU path takes the
# window-in-LDS convolutions, the fused InstanceNorm sums of the stem and 256-token windows; stored at every 4th pixel
STRIDE = {"c": 4}


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/src/ebsynth/deps/gmflow")
    from gmflow.gmflow import GMFlow
    torch.set_num_threads(8)
    m = GMFlow(feature_channels=128, num_scales=1, upsample_factor=8, num_head=1, attention_type="swin",
               ffn_dim_expansion=4, num_transformer_layers=6).eval()
    sd = m.state_dict()
    m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
    out = {"param_names": np.array(sorted(sd.keys()))}
    for tag, (N, H, W) in CASES.items():
        imgs = cf.gmflow_frames(N, H, W)
        nxt = list(range(1, N)) + [0]
        with torch.no_grad():
            flow = m(imgs, imgs[nxt], attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1],
                     pred_bidir_flow=True)["flow_preds"][-1]
        st = STRIDE.get(tag, 1)
        out["flow_" + tag] = flow[:, :, ::st, ::st].contiguous().numpy()
        print(tag, tuple(flow.shape), "mean |flow| %.3f max %.3f" % (float(flow.abs().mean()), float(flow.abs().max())))
    path = os.path.join(HERE, "gmflow_golden.npz")
    if os.path.exists(path):  # earlier cases stay byte for byte what they were (report how a re-run compares)
        old = dict(np.load(path))
        for k, v in old.items():
            if k in out and k.startswith("flow_"):
                print("re-run vs stored", k, "max |d| = %.3g" % float(np.abs(out[k] - v).max()))
            out[k] = v
    np.savez_compressed(path, **out)
    print("wrote gmflow_golden.npz")


if __name__ == "__main__":
    main()
