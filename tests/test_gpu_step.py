"""f2: DDPM step + classifier-free guidance kernels against the reference's own step() (golden, CPU fp32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sg():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "step_golden.npz")))


@pytest.mark.parametrize("t", [701, 1])
@pytest.mark.parametrize("rep", [False, True])
def test_step_matches_reference(sg, t, rep, monkeypatch):
    import sys
    import fresco_amd
    S = sys.modules["fresco_amd.step"]  # `fresco_amd.step` itself is the function
    import make_step_golden as msg
    tag = "t%d_%s" % (t, "rep" if rep else "ind")
    x, eps = msg.inputs()
    noise = torch.from_numpy(sg[tag + "_noise"]).to(DEV)
    monkeypatch.setattr(S.torch, "randn", lambda *a, **k: noise.to(k.get("dtype", torch.float32)))
    prev, x0 = fresco_amd.step(msg.P(), eps.to(DEV), t, x.to(DEV), None, repeat_noise=rep)
    assert float((x0.cpu() - torch.from_numpy(sg[tag + "_x0"])).abs().max()) < 1e-5
    assert float((prev.cpu() - torch.from_numpy(sg[tag + "_prev"])).abs().max()) < 1e-5
    # fp16 latents (what the pipeline runs): same formulas, fp16 storage
    prev16, x016 = fresco_amd.step(msg.P(), eps.to(DEV).half(), t, x.to(DEV).half(), None, repeat_noise=rep)
    assert prev16.dtype == torch.float16
    assert float((x016.float().cpu() - torch.from_numpy(sg[tag + "_x0"])).abs().max()) < 2e-2
    assert float((prev16.float().cpu() - torch.from_numpy(sg[tag + "_prev"])).abs().max()) < 1e-2


def test_cfg_fused_x0():
    import fresco_amd
    g = torch.Generator().manual_seed(3)
    xt, eu, ec = (torch.randn(4, 4, 8, 8, generator=g) for _ in range(3))
    x0, eps = fresco_amd.predict_x0(xt.to(DEV), eu.to(DEV), ec.to(DEV), guidance_scale=7.5, alpha_prod_t=0.37)
    eref = eu + 7.5 * (ec - eu)
    assert float((eps.cpu() - eref).abs().max()) < 1e-5
    assert float((x0.cpu() - (xt - (1 - 0.37) ** 0.5 * eref) / 0.37 ** 0.5).abs().max()) < 1e-4
