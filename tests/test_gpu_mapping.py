"""GPU parity of the FLATTEN correspondence kernels (integer path): bit-exact against the reference's
golden outputs (Appendix-B KAT 3) and against the oracle on random cases incl. forced colour ties."""
import pytest
import torch

import closed_form as cf
import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(a)


@pytest.mark.parametrize("scale", [4.0, 8.0])
def test_mapping_ind_kat3_bit_exact(golden, scale):
    import fresco_amd
    d = cf.base_case()
    fm, bm, tm = fresco_amd.get_mapping_ind(d["bwd"].to(DEV), d["bo"].to(DEV), d["imgs"].to(DEV), scale=scale)
    tag = "map_s%d" % int(scale)
    assert torch.equal(fm.cpu(), T(golden[tag + "_fwd"]))
    assert torch.equal(bm.cpu(), T(golden[tag + "_bwd"]))
    assert torch.equal(tm.cpu().to(torch.uint8), T(golden[tag + "_mask"]))
    if scale == 4.0:
        p1 = torch.arange(1, fm.shape[2] + 1)
        assert [int((fm[f, 0].cpu() * p1).sum()) for f in range(4)] == [5592320, 5591833, 5591161, 5590439]


@pytest.mark.parametrize("seed,N,R,scale,ties", [(0, 4, 64, 8.0, False), (1, 8, 128, 8.0, False),
                                                  (2, 5, 96, 4.0, True), (3, 3, 160, 16.0, True),
                                                  (4, 8, 512, 16.0, False), (5, 8, 512, 8.0, False)])
def test_mapping_ind_random_vs_oracle(seed, N, R, scale, ties):
    import fresco_amd
    g = synth.gen(100 + seed)
    flows, occs = synth.make_flows(N, R, g, "blocks" if seed % 2 else "bernoulli")
    flows[1] = flows[1] * (1.0 + seed)  # larger motions: more collisions and out-of-range targets
    imgs = torch.rand(N, 3, R, R, generator=g)
    if ties:  # few distinct colours -> many exactly equal errors: the earliest source must win
        imgs = (imgs * 3).floor() / 3
    fm, bm, tm = fresco_amd.get_mapping_ind(flows[1].to(DEV), occs[1].to(DEV), imgs.to(DEV), scale=scale)
    fo, bo, to_ = O.mapping_ind(flows[1], occs[1], imgs, scale=scale)
    assert torch.equal(fm.cpu(), fo) and torch.equal(bm.cpu(), bo) and torch.equal(tm.cpu(), to_)
    # every row is a permutation and bwd is its inverse
    hw = fm.shape[2]
    ar = torch.arange(hw)
    for f in range(N):
        assert torch.equal(torch.sort(fm[f, 0].cpu())[0], ar)
        assert torch.equal(fm[f, 0].cpu()[bm[f, 0].cpu()], ar)


def test_single_mapping_and_cross_frame_masks(golden):
    import fresco_amd
    d = cf.base_case()
    m, unl = fresco_amd.get_single_mapping_ind(d["bwd"][0:1].to(DEV), d["bo"][0:1].to(DEV), d["imgs"][0:2].to(DEV), 4.0)
    mo, uo = O.single_mapping_ind(d["bwd"][0:1], d["bo"][0:1], d["imgs"][0:2], 4.0)
    assert torch.equal(m.cpu(), mo) and torch.equal(unl.cpu(), uo)
    ms = fresco_amd.cross_frame_masks(d["bo"].to(DEV), scales=(2.0, 4.0, 8.0))
    for mk, s in zip(ms, (2, 4, 8)):
        assert torch.equal(mk.cpu().to(torch.uint8), T(golden["cfmask_s%d" % s]))
