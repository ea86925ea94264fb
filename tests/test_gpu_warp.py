"""GPU parity of the warp family against the reference goldens and the oracle."""
import pytest
import torch

import closed_form as cf
import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(a)


def maxdiff(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


@pytest.fixture(scope="module")
def d():
    return cf.base_case()


def test_flow_warp_kat1(d, golden):
    import fresco_amd
    w = fresco_amd.flow_warp(d["x"].to(DEV), d["bwd"].to(DEV))
    assert maxdiff(w, T(golden["flow_warp_x_bwd"])) < 2e-5
    s, sa = cf.checksum(w.cpu())
    assert abs(s - (-17.644410)) < 2e-3 and abs(sa - 29293.912217) < 0.5


def test_flow_warp_out_of_range_and_broadcast():
    import fresco_amd.ops as ops
    g = synth.gen(3)
    x = torch.randn(6, 5, 9, 13, generator=g)
    flow = 6 * torch.randn(3, 2, 9, 13, generator=g)  # many samples fall outside -> zeros padding
    flow[0, 0, 0, 0] = 1e9
    flow[1, 1, 2, 3] = -1e9
    out = ops.flow_warp(x.to(DEV), flow.to(DEV))
    ref = O.flow_warp(x, flow.repeat(2, 1, 1, 1))
    assert maxdiff(out, ref) < 1e-5


def test_resize_maxpool_dilate(d, golden):
    import fresco_amd
    import fresco_amd.ops as ops
    assert maxdiff(fresco_amd.Dilate(13)(d["bo"].unsqueeze(1).to(DEV)), T(golden["dilate13_bo"])) == 0
    assert maxdiff(fresco_amd.Dilate(7)(d["fo"].unsqueeze(1).to(DEV)), T(golden["dilate7_fo"])) == 0
    fl = ops.resize_bilinear(d["bwd"].to(DEV), 8 / 64, mul=8 / 64)
    assert maxdiff(fl, T(golden["prep_f32_bwd_flow"])) < 1e-6
    oc = ops.max_pool(d["bo"].unsqueeze(1).to(DEV), 8)
    assert maxdiff(oc, T(golden["prep_f32_bwd_occ"])) == 0
    # odd sizes / non-integer scale against the oracle
    g = synth.gen(4)
    x = torch.randn(2, 3, 37, 53, generator=g)
    for s in (0.5, 0.3, 1.0, 2.0):
        assert maxdiff(ops.resize_bilinear(x.to(DEV), s), O.resize_bilinear(x, s)) < 1e-5
    assert maxdiff(ops.max_pool(x.to(DEV), 3), O.max_pool(x, 3)) == 0


def test_adain(golden):
    import fresco_amd
    c_ = cf.feat(8, 16, 8, 8, 0.0) * 1.7 + 0.3
    s_ = cf.feat(8, 16, 8, 8, 0.9) * 0.6 - 0.2
    out = fresco_amd.adaptive_instance_normalization(c_.to(DEV), s_.to(DEV))
    assert maxdiff(out, T(golden["adain"])) < 5e-6
    out16 = fresco_amd.adaptive_instance_normalization(c_.to(DEV).half(), s_.to(DEV).half())
    assert out16.dtype == torch.float16
    ref16 = O.adain(c_.half().float(), s_.half().float())
    assert maxdiff(out16, ref16) < 4e-3


def test_calc_mean_std():
    """src/utils.py:58-67 on the reduction of the AdaIN kernel: mean and sqrt(unbiased variance + eps) per (sample, channel)"""
    import fresco_amd
    x = cf.feat(8, 16, 8, 8, 0.4) * 1.3 - 0.1
    for xx, tol in ((x, 2e-6), (x.half(), 2e-3)):
        m, s = fresco_amd.calc_mean_std(xx.to(DEV), eps=1e-5)
        assert m.shape == s.shape == (8, 16, 1, 1) and m.dtype == xx.dtype
        xf = xx.float().reshape(8, 16, -1)
        assert maxdiff(m.float(), xf.mean(2).view(8, 16, 1, 1)) < tol
        assert maxdiff(s.float(), (xf.var(2) + 1e-5).sqrt().view(8, 16, 1, 1)) < tol
    m, s = fresco_amd.calc_mean_std(x.to(DEV), 1)  # the reference's positional slip: eps = chunk = 1 (utils.py:73)
    assert maxdiff(s, (x.reshape(8, 16, -1).var(2) + 1.0).sqrt().view(8, 16, 1, 1)) < 2e-6


def test_warp_tensor_kat4_kat5(d, golden):
    import fresco_amd
    fl = [d["fwd"].to(DEV), d["bwd"].to(DEV)]
    oc = [d["fo"].to(DEV), d["bo"].to(DEV)]
    wt = fresco_amd.warp_tensor(d["lat"].to(DEV), fl, oc, d["sal"].to(DEV), 2)
    assert maxdiff(wt, T(golden["warp_tensor_lat"])) < 2e-5
    wi = fresco_amd.warp_tensor(d["x"].to(DEV), fl, oc, d["sal"].to(DEV), 1)  # scale==1 -> Dilate-13 path
    assert maxdiff(wi, T(golden["warp_tensor_img"])) < 2e-5
    s, sa = cf.checksum(wi.cpu())
    assert abs(s - (-122.289417)) < 5e-3 and abs(sa - 31129.745621) < 0.6


def test_warp_tensor_16_and_fp16(golden):
    import fresco_amd
    d8 = cf.base_case(N=8)
    lat16 = cf.feat(16, 12, 16, 16, 0.7)
    fl = [d8["fwd"].to(DEV), d8["bwd"].to(DEV)]
    oc = [d8["fo"].to(DEV), d8["bo"].to(DEV)]
    w = fresco_amd.warp_tensor(lat16.to(DEV), fl, oc, d8["sal"].to(DEV), 2)
    assert maxdiff(w, T(golden["warp_tensor_lat16"])) < 2e-5
    wh = fresco_amd.warp_tensor(lat16.to(DEV).half(), fl, oc, d8["sal"].to(DEV), 2)
    assert wh.dtype == torch.float16
    ref = O.warp_tensor(lat16.half(), [d8["fwd"], d8["bwd"]], [d8["fo"], d8["bo"]], d8["sal"], 2)
    assert maxdiff(wh, ref) < 1e-3


def test_warp_tensor_cfg3_layer_shapes():
    """feature-space warp at the four decoder resolutions of 8 x 512^2 (reduced channels), vs oracle."""
    import fresco_amd
    case = synth.make_opt_case(8, 24, 8, 512, seed=9)
    g = synth.gen(10)
    for h in (8, 16, 32, 64):
        x = torch.randn(16, 24, h, h, generator=g)
        out = fresco_amd.warp_tensor(x.to(DEV), [f.to(DEV) for f in case["flows"]],
                                     [o.to(DEV) for o in case["occs"]], case["sal"].to(DEV), 2)
        ref = O.warp_tensor(x, case["flows"], case["occs"], case["sal"], 2)
        assert maxdiff(out, ref) < 5e-5, h
