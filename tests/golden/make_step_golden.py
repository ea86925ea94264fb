"""Golden vectors for the DDPM step (src/pipe_FRESCO.py:14-77) from the UNMODIFIED reference on CPU ->
tests/golden/step_golden.npz.  Build container only.  The scheduler is a stand-in exposing exactly what
step() reads (previous_timestep, alphas_cumprod, one) with SD-1.5's scaled-linear beta schedule."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as cf  # noqa: E402
import _ref_harness  # noqa: E402


class Sched:
    def __init__(self):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.one = torch.tensor(1.0)

    def previous_timestep(self, t):
        return t - 50


class P:
    scheduler = Sched()


def inputs():
    x = cf.feat(8, 4, 16, 16, 0.1) * 1.3
    eps = cf.feat(8, 4, 16, 16, 0.8)
    return x, eps


if __name__ == "__main__":
    _ref_harness.load_reference()
    os.chdir(_ref_harness.REF_ROOT)
    import types
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))  # only used by the debug visualiser
    import src.pipe_FRESCO as pf
    out = {}
    x, eps = inputs()
    for t in (701, 1):
        for rep in (False, True):
            g = torch.Generator().manual_seed(7)
            noise = torch.randn(eps.shape, generator=torch.Generator().manual_seed(7))
            prev, x0 = pf.step(P(), eps, t, x, g, repeat_noise=rep)
            tag = "t%d_%s" % (t, "rep" if rep else "ind")
            out[tag + "_prev"] = prev.numpy()
            out[tag + "_x0"] = x0.numpy()
            out[tag + "_noise"] = noise.numpy()
    np.savez_compressed(os.path.join(HERE, "step_golden.npz"), **out)
    print("wrote step_golden.npz", sorted(out))
