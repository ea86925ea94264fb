"""Processor goldens at the head dims the SD-1.5 decoder uses (D = 40 in up_blocks.3, D = 80 in up_blocks.2): outputs
of the UNMODIFIED reference processor (/root/reference, fp32, CPU) for the closed-form KAT-7 inputs widened to C = 80
(2 heads of 40 / 1 head of 80), in the three attention modes of the schedule.

Run in the build container only:  python tests/golden/make_proc_headdim_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as cf  # noqa: E402
import _ref_harness  # noqa: E402
from make_golden import FakeAttn  # noqa: E402

C, HW, B = 80, 64, 8
MODES = ("full", "cf_temporal", "cf")


def main():
    torch.set_num_threads(8)
    dh, fu, geo, ut = _ref_harness.load_reference()
    d = cf.base_case()
    out = {}
    with torch.no_grad():
        W = cf.attn_weights(C)
        hs = cf.attn_hidden(B, HW, C, 0.0)
        ref = cf.attn_hidden(B, HW, C, 0.4)
        fm, bm, tm = fu.get_mapping_ind(d["bwd"], d["bo"], d["imgs"], scale=8.0)
        o = F.interpolate(d["bo"][:-1].unsqueeze(1), scale_factor=1.0 / 8.0, mode="bilinear")
        cfm = torch.cat((o[0:1].reshape(1, -1) > -1, o.reshape(o.shape[0], -1) > 0.5), dim=0)
        paras = {"fwd_mappings": [fm], "bwd_mappings": [bm], "interattn_masks": [tm]}
        for heads in (2, 1):
            attn = FakeAttn(C, heads, W)
            for mode in MODES:
                ctl = dh.AttentionControl()
                proc = dh.FRESCOAttnProcessor2_0(2, ctl)
                if mode == "full":
                    ctl.enable_store()
                    proc(attn, ref)
                    ctl.disable_store()
                    ctl.enable_controller(interattn_paras=paras, attn_mask=[cfm])
                elif mode == "cf_temporal":
                    ctl.enable_interattn(paras)
                    ctl.enable_cfattn([cfm])
                else:
                    ctl.enable_cfattn([cfm])
                out["proc_d%d_%s" % (C // heads, mode)] = proc(attn, hs)
    np.savez_compressed(os.path.join(HERE, "proc_headdim_golden.npz"),
                        **{k: v.detach().cpu().numpy() for k, v in out.items()})
    for k, v in out.items():
        print(k, tuple(v.shape), "sum %.6f abs %.6f" % cf.checksum(v))


if __name__ == "__main__":
    main()
