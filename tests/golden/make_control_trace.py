"""Record the state trace of the UNMODIFIED reference AttentionControl (src/diffusion_hacked.py:23-137)
under a scripted sequence of calls -> tests/golden/control_trace.json.  Build container only."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness  # noqa: E402

SCRIPT = [
    ("enable_intraattn",), ("call", None), ("enable_store",), ("call", 1), ("call", 2), ("call", 3),
    ("disable_store",), ("enable_intraattn",), ("call", None), ("call", None), ("call", None), ("call", None),
    ("enable_store",), ("call", 7), ("enable_cfattn", None), ("enable_cfattn", "mask"),
    ("enable_cfattn", None), ("disable_cfattn",), ("enable_cfattn", "empty"), ("enable_interattn", None),
    ("enable_interattn", "paras"), ("disable_interattn",), ("enable_interattn", None),
    ("disable_controller",), ("enable_controller", "paras2", "mask2"), ("call", None), ("call", 9),
    ("clear_store",), ("call", 4), ("enable_intraattn",), ("enable_controller", None, None),
    ("disable_intraattn",), ("enable_store",), ("call", 5), ("enable_intraattn",), ("call", None), ("call", None),
]
OBJ = {"mask": [torch.ones(2, 4, dtype=torch.bool)], "mask2": [torch.zeros(2, 4, dtype=torch.bool)],
       "paras": {"fwd_mappings": [1]}, "paras2": {"fwd_mappings": [2]}, "empty": []}


def run(ctrl_cls):
    c = ctrl_cls()
    trace = []
    for step in SCRIPT:
        ret = None
        if step[0] == "call":
            arg = None if step[1] is None else torch.full((1,), float(step[1]))
            out = c(arg)
            ret = None if out is None else float(out[0])
        else:
            args = [OBJ.get(a, a) if isinstance(a, str) else a for a in step[1:]]
            getattr(c, step[0])(*args)
        trace.append(dict(ret=ret, store=c.store, index=c.index, intra=c.use_intraattn, cf=c.use_cfattn,
                          inter=c.use_interattn, n=len(c.stored_attn["decoder_attn"]),
                          mask=None if c.attn_mask is None else int(c.attn_mask[0].sum()) if len(c.attn_mask) else -1,
                          paras=None if c.interattn_paras is None else c.interattn_paras["fwd_mappings"][0]))
    return trace


if __name__ == "__main__":
    dh, _, _, _ = _ref_harness.load_reference()
    with open(os.path.join(HERE, "control_trace.json"), "w") as f:
        json.dump(run(dh.AttentionControl), f, indent=0)
    print("wrote control_trace.json")
