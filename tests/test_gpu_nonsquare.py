"""Non-square frames (the reference resizes to a 512-px short side and keeps the aspect ratio, run_fresco.py:170: real clips
are 512 x 896 and the like): every entry point of the hot path at the layer shapes of a 4-frame 256 x 448 batch -- token
grids 32 x 56 / 16 x 28, feature planes 4 x 7 ... 32 x 56 (hw = 28, 112: no multiple of 64 -> the generic kernels; 448 = 7 x 64
and 1792 = 28 x 64: the four-launch pipeline on planes that are no power of two) -- against the oracle.  References:
src/diffusion_hacked.py:169-387, 416-488; src/flow_utils.py:18-138."""
import copy

import pytest
import torch

import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, H, W = 4, 256, 448


def _flows(g):
    base = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1)
    bwd = base + 0.3 * torch.randn(N, 2, H, W, generator=g)
    fo = (torch.rand(N, H, W, generator=g) < 0.1).float()
    bo = (torch.rand(N, H, W, generator=g) < 0.1).float()
    return [-bwd, bwd], [fo, bo]


@pytest.mark.parametrize("layer", ["L3", "L2"])
@pytest.mark.parametrize("mode", ["full", "cf_temporal", "cf"])
def test_processor_nonsquare(layer, mode):
    import fresco_amd
    g = synth.gen(77)
    C, down = (640, 16) if layer == "L2" else (320, 8)
    HW = (H // down) * (W // down)
    attn = synth.FakeAttn(C, 8)
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(p.half().float())
    hidden = torch.randn(2 * N, HW, C, generator=g).half()
    ref = (hidden.float() + 0.1 * torch.randn(2 * N, HW, C, generator=g)).half()
    flows, occs = _flows(g)
    imgs = torch.rand(N, 3, H, W, generator=g)
    fwd_map, bwd_map, tmask = O.mapping_ind(flows[1], occs[1], imgs, scale=float(down))
    cf_mask = O.cross_frame_masks(occs[1], scales=(float(down),))[0]
    assert tuple(fwd_map.shape) == (N, 1, HW) and tuple(cf_mask.shape) == (N, HW)
    case = dict(attn=attn, hidden=hidden, ref=ref, fwd_map=fwd_map, bwd_map=bwd_map, tmask=tmask, cf_mask=cf_mask, N=N, HW=HW,
                C=C, heads=8)
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, synth.controller_for(case, mode, DEV))
    with torch.no_grad():
        out = proc(copy.deepcopy(attn).to(DEV).half(), hidden.to(DEV))
    want = synth.oracle_attention(case, mode, round_dtype=None)
    err = (out.float().cpu() - want).abs()
    assert bool((err <= 1e-3 + 1e-3 * want.abs()).all()), float(err.max())
    print("non-square %s %-11s (HW = %d = %d x %d): max |HIP - fp32 oracle| = %.2e" % (layer, mode, HW, H // down, W // down,
                                                                                       float(err.max())))


@pytest.mark.parametrize("C,div", [(1280, 64), (1280, 32), (1280, 16), (640, 8)])
def test_optimize_feature_and_warp_nonsquare(C, div):
    import fresco_amd
    import fresco_amd.ops as ops
    from fresco_amd.warp import _prep_flow_occ
    g = synth.gen(80 + div)
    h, w = H // div, W // div
    flows, occs = _flows(g)
    sal = torch.rand(N, 1, H // 2, W // 2, generator=g)
    x = torch.randn(2 * N, C, h, w, generator=g)
    target = O.gram_target(torch.randn(2 * N, C, h, w, generator=g))
    fd, od = [f.to(DEV) for f in flows], [o.to(DEV) for o in occs]
    # one closure vs the fp64 oracle
    prep = _prep_flow_occ(h, fd, od, with_dilate=False)
    loss, grad = ops.opt_loss_grad(x.to(DEV), prep, target.to(DEV), 100.0, 2)
    prep64 = O.opt_prepare(h, flows, occs, 2, torch.float64)
    lo, go = O.opt_loss_and_grad(x.double(), prep64, target.double(), 100.0)
    assert abs(float(loss.sum()) - float(lo)) < 1e-5 * float(lo)
    err = (grad.cpu().double() - go).abs()
    scale = float(go.abs().max())
    assert float((err > 1e-3 * scale).double().mean()) <= 2e-4 * max(h * w / 256.0, 1.0) + 4.0 / go.numel()
    # three Adam iterations + AdaIN through the public entry, then the feature-space warp
    out = fresco_amd.optimize_feature(x.half().to(DEV), fd, od, [target.to(DEV)], iters=3)
    assert out.dtype == torch.float16 and bool(torch.isfinite(out).all())
    wt = fresco_amd.warp_tensor(x.to(DEV), fd, od, sal.to(DEV), 2)
    e = float((wt.cpu() - O.warp_tensor(x, flows, occs, sal, 2)).abs().max())
    assert e < 5e-5, e
    print("non-square opt / warp C=%d %dx%d (hw = %d): gradient outliers %.1e, warp max err %.1e"
          % (C, h, w, h * w, float((err > 1e-3 * scale).double().mean()), e))


def test_interframe_paras_and_flow_network_nonsquare():
    """occlusions + masks + trajectory maps bit-exact vs the oracle on 256 x 448 frames; the flow network (native dense
    layers: 32 x 56 token grid, row blocks that are not whole 16 x 16 patches at the coarsest level) runs and is finite"""
    import closed_form as cf
    import fresco_amd
    import fresco_amd.gmflow as G
    from fresco_amd import paras
    g = synth.gen(91)
    flows, _ = _flows(g)
    images = (torch.rand(N, 3, H, W, generator=g) * 255).round()
    got = paras.interframe_paras_from_flows(images.to(DEV), flows[0].to(DEV), flows[1].to(DEV))
    want = O.interframe_paras(images, flows[0], flows[1])
    for a, b in zip(got[1], want[1]):
        assert int((a.cpu() != b).sum()) <= 4                      # (last-bit ties of a float threshold)
    same_occ = all(torch.equal(a.cpu(), b) for a, b in zip(got[1], want[1]))
    if same_occ:
        for a, b in zip(got[2], want[2]):
            assert torch.equal(a.cpu(), b)
        for k in ("fwd_mappings", "bwd_mappings", "interattn_masks"):
            for a, b in zip(got[3][k], want[3][k]):
                assert torch.equal(a.cpu(), b), k
    m = G.GMFlow().eval()
    sd = m.state_dict()
    m.load_state_dict({k: cf.gmflow_param(k, tuple(v.shape)) for k, v in sd.items()})
    m = m.to(DEV)
    frames = [f.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy() for f in cf.gmflow_frames(N, H, W)]
    fl, oc, masks, pr = fresco_amd.get_flow_and_interframe_paras(m, frames)
    assert tuple(fl[0].shape) == (N, 2, H, W) and bool(torch.isfinite(fl[0]).all()) and bool(torch.isfinite(fl[1]).all())
    assert [tuple(t.shape) for t in pr["fwd_mappings"]] == [(N, 1, (H // 8) * (W // 8)), (N, 1, (H // 16) * (W // 16))]
