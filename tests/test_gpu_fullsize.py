"""Parity at the REAL configurations of BASELINE.json (not the reduced shapes of the other GPU tests), against the
un-rounded fp32 / fp64 CPU oracle -- never against another GPU implementation:

  * the whole FRESCOAttnProcessor2_0 call at config 2 (8 frames x 512^2), up_blocks.3 (HW 4096, C 320, D 40) and
    up_blocks.2 (HW 1024, C 640, D 80), in the three attention modes of the denoising schedule
    (full = spatial + cross-frame + temporal, cf_temporal, cf), every output element, tolerance = the contract of
    BASELINE.json's north star: 1e-3 absolute on O(1) outputs (reference: src/diffusion_hacked.py:169-387);
  * the same with the block-occlusion masks of SURVEY 8d (M ~ 4.5 HW cross-frame keys);
  * the temporal-guided kernel at N = 8, HW = 4096 and at config 5's N = 32, HW = 9216 with the trajectory maps and
    masks the oracle's get_mapping_ind restatement produces from synthetic flows (src/flow_utils.py:56-138);
  * one closure (loss + gradient) of optimize_feature at the shipping layer shapes (C = 640, 64 x 64) and
    (C = 1280, 32 x 32) against the fp64 oracle (src/diffusion_hacked.py:455-485).
Every call goes through the C ABI of libfresco_hip.so.  The CPU oracle needs tens of seconds per case on the GPU
box's host cores."""
import copy
import math

import pytest
import torch

import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL, RTOL = 1e-3, 1e-3  # north star: max per-element deviation < 1e-3 on O(1) outputs


def _check(out, ref, atol=ATOL, rtol=RTOL, what=""):
    out = out.float().cpu()
    err = (out - ref).abs()
    bound = atol + rtol * ref.abs()
    assert bool((err <= bound).all()), "%s: max err %.3e (ref max %.3e)" % (what, float(err.max()), float(ref.abs().max()))
    return float(err.max())


_cases = {}


def _case(layer, occ_mode):
    key = (layer, occ_mode)
    if key not in _cases:
        _cases[key] = synth.make_attention_case(8, 512, layer, seed=5, occ_mode=occ_mode)
    return _cases[key]


def _run_processor(case, mode):
    import fresco_amd
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, synth.controller_for(case, mode, DEV))
    attn = copy.deepcopy(case["attn"]).to(DEV).half()
    with torch.no_grad():
        return proc(attn, case["hidden"].to(DEV).half())


@pytest.mark.parametrize("layer", ["L3", "L2"])
@pytest.mark.parametrize("mode", ["full", "cf_temporal", "cf"])
def test_processor_cfg2_every_element(layer, mode):
    case = _case(layer, "bernoulli")
    assert case["HW"] == (4096 if layer == "L3" else 1024) and case["hidden"].shape[0] == 16
    out = _run_processor(case, mode)
    ref = synth.oracle_attention(case, mode, round_dtype=None)
    e = _check(out, ref, what="cfg2 %s %s" % (layer, mode))
    print("cfg2 %s %-11s: max |HIP - fp32 oracle| = %.2e over %d elements (|ref| max %.2f)"
          % (layer, mode, e, ref.numel(), float(ref.abs().max())))


@pytest.mark.parametrize("mode", ["cf", "cf_temporal"])
def test_processor_cfg2_block_occlusions(mode):
    """SURVEY 8d's second mask setting: Bernoulli(0.5) blocks of 32 x 32 px -> M ~ (1 + 0.5 (N-1)) HW keys, many
    broken trajectories; the regime where the packed key images stop fitting an XCD's L2."""
    case = _case("L3", "blocks")
    M = int(case["cf_mask"].sum())
    assert M > 3 * case["HW"]
    out = _run_processor(case, mode)
    ref = synth.oracle_attention(case, mode, round_dtype=None)
    e = _check(out, ref, what="cfg2 blocks %s" % mode)
    print("cfg2 L3 blocks %-11s (M = %d): max err %.2e" % (mode, M, e))


def test_oracle_on_the_gpu_is_the_same_oracle():
    """The configurations below evaluate the oracle's tensor code with torch on the GPU (fp32; the CPU needs minutes for
    32 frames x 768^2).  Same code, same arithmetic type: at config 2's up_blocks.2 the two devices agree to fp32
    round-off, far below the 1e-3 bar they are used for."""
    case = _case("L2", "bernoulli")
    a = synth.oracle_attention(case, "full", round_dtype=None)
    b = synth.oracle_attention(case, "full", round_dtype=None, device=DEV)
    d = float((a - b).abs().max())
    print("oracle cpu vs cuda (cfg2 L2 full): max |d| = %.2e" % d)
    assert d < 2e-5


@pytest.mark.parametrize("N,R,layer", [(4, 256, "L3"), (4, 256, "L2"), (16, 512, "L3"), (16, 512, "L2")])
@pytest.mark.parametrize("mode", ["full", "cf_temporal", "cf"])
def test_processor_cfg1_cfg4_every_element(N, R, layer, mode):
    """Config 1's shapes (4 keyframes at 256^2: HW 1024 / 256) and config 4's (16 frames at 512^2: B = 32), every
    element of the processor output vs the fp32 oracle (reference: src/diffusion_hacked.py:169-387)."""
    case = synth.make_attention_case(N, R, layer, seed=7 + N, occ_mode="bernoulli")
    out = _run_processor(case, mode)
    ref = synth.oracle_attention(case, mode, round_dtype=None, device=DEV if N > 4 else None)
    e = _check(out, ref, what="N=%d R=%d %s %s" % (N, R, layer, mode))
    print("N=%d %dx%d %s %-11s: max |HIP - fp32 oracle| = %.2e over %d elements"
          % (N, R, R, layer, mode, e, ref.numel()))


@pytest.mark.parametrize("mode", ["cf_temporal", "cf"])
def test_processor_cfg5_one_cfg_half(mode):
    """Config 5's up_blocks.3 call (32 frames x 768^2: HW = 9216 per frame, ~10 300 cross-frame keys), ONE CFG half
    (unet_chunk_size 1, B = 32): the shape the frame-parallel claim rests on."""
    import fresco_amd
    case = synth.make_attention_case(32, 768, "L3", seed=9, occ_mode="bernoulli")
    half = dict(case)
    half["hidden"] = case["hidden"][:32].contiguous()
    M = int(case["cf_mask"].sum())
    proc = fresco_amd.FRESCOAttnProcessor2_0(1, synth.controller_for(half, mode, DEV))
    attn = copy.deepcopy(case["attn"]).to(DEV).half()
    with torch.no_grad():
        out = proc(attn, half["hidden"].to(DEV).half())
    ref = synth.oracle_attention(half, mode, round_dtype=None, device=DEV, chunk=1)
    e = _check(out, ref, what="cfg5 L3 %s" % mode)
    print("cfg5 L3 %-11s (HW 9216, M = %d, B 32): max |HIP - fp32 oracle| = %.2e" % (mode, M, e))


@pytest.mark.parametrize("vscale,atol", [(1.0, 2e-3), (0.4, 1e-3)])
@pytest.mark.parametrize("N,R,layer", [(8, 512, "L3"), (32, 768, "L3"), (32, 768, "L2")])
def test_temporal_kernel_at_real_sizes(N, R, layer, vscale, atol):
    """fresco_temporal_attn alone, fp16 inputs, vs the fp32 per-pixel restatement (oracle.temporal_attention).
    The contract is BASELINE.json's 1e-3: it holds on the pipeline's activations -- q, k, v are projections of
    LayerNorm outputs, |v| <~ 2, where half an fp16 ulp of the OUTPUT is <= 4.9e-4 (vscale 0.4: |v| reaches ~2) --
    and cannot hold for N(0,1) v (|v| reaches 4-5, half an output ulp there is 1.95e-3: the fp16 output grid itself):
    that case keeps the 2e-3 bar and documents why."""
    import fresco_amd.ops as ops
    g = synth.gen(40 + N)
    C, down = (640, 16) if layer == "L2" else (320, 8)
    H = 8
    side = R // down
    HW = side * side
    flows, occs = synth.make_flows(N, R, g)
    imgs = torch.rand(N, 3, R, R, generator=g)
    fwd_map, _, tmask = O.mapping_ind(flows[1], occs[1], imgs, scale=float(down))
    assert tuple(fwd_map.shape) == (N, 1, HW) and tuple(tmask.shape) == (HW, 1, N, N)
    q = (vscale * torch.randn(2 * N, HW, C, generator=g)).half()
    k = (vscale * torch.randn(2 * N, HW, C, generator=g)).half()
    v = (vscale * torch.randn(2 * N, HW, C, generator=g)).half()
    scale = 0.2 / math.sqrt(C // H)
    out = ops.temporal_attention(q.to(DEV), k.to(DEV), v.to(DEV), fwd_map.to(DEV), tmask.to(DEV), H, scale, 2)
    ref = O.temporal_attention(q.float(), k.float(), v.float(), fwd_map[:, 0], tmask[:, 0], H, scale, 2)
    e = _check(out, ref, atol=atol, rtol=1e-3, what="temporal N=%d HW=%d" % (N, HW))
    print("temporal N=%d HW=%d C=%d |v| max %.1f: max err %.2e (bar %.0e), %.1f %% of the frame pairs masked"
          % (N, HW, C, float(v.abs().max()), e, atol, 100 * (1 - float(tmask.float().mean()))))


@pytest.mark.parametrize("C,h", [(640, 64), (1280, 32), (1280, 16), (1280, 8)])
def test_opt_closure_at_shipping_shapes(C, h):
    """One evaluation of optimize_feature's closure (temporal L1 + Gram L1 and their analytic gradient) at ALL FOUR
    shapes the pipeline runs it on at 8 x 512^2: up_blocks.3's input (C=640, 64 x 64: the 32 x 32 grid of 128-wide
    Gram tiles, upper triangle + mirrored writes), up_blocks.2's (C=1280, 32 x 32), up_blocks.1's (1280, 16 x 16: the
    8-wave DMA-staged Gram kernel on a 2 x 2 tile grid) and up_blocks.0's (1280, 8 x 8: a plane smaller than one tile,
    the 4-wave register-staged Gram kernel with K chunk 64), vs the fp64 oracle."""
    import fresco_amd.ops as ops
    from fresco_amd.warp import _prep_flow_occ
    N, R = 8, 512
    case = synth.make_opt_case(N, C, h, R, seed=31)
    x = case["x"]
    prep64 = O.opt_prepare(h, case["flows"], case["occs"], 2, torch.float64)
    loss_ref, grad_ref = O.opt_loss_and_grad(x.double(), prep64, case["target"].double(), 100.0, chunk=2)
    prep = _prep_flow_occ(h, [f.to(DEV) for f in case["flows"]], [o.to(DEV) for o in case["occs"]], with_dilate=False)
    loss, grad = ops.opt_loss_grad(x.to(DEV), prep, case["target"].to(DEV), 100.0, 2)
    lt, ls = float(loss[0]), float(loss[1])
    tot_ref = float(loss_ref)
    assert abs((lt + ls) - tot_ref) <= 1e-5 * abs(tot_ref), (lt, ls, tot_ref)
    g = grad.double().cpu()
    err = (g - grad_ref).abs()
    # The gradient of a row is a sum of hw sign() terms: a Gram entry within fp32 rounding of its target may flip its
    # sign, which moves the C gradient entries of that row by one term (2 w V_jc / (B hw^2), ~1 % of the row's scale).
    # The chance that a row holds such a near-tie grows with its hw entries, so the reduced-size budget of
    # test_gpu_opt.py (2e-4 of the elements off by more than 1e-3 of the gradient's scale at hw = 256) scales with
    # hw / 256; and a flipped term can never move an element by more than a few per cent of the scale.
    scale = float(grad_ref.abs().max())
    frac_bad = float((err > 1e-3 * scale).double().mean())
    assert frac_bad <= 2e-4 * (h * h / 256.0), frac_bad
    assert float(err.max()) <= 5e-2 * scale, (float(err.max()), scale)
    print("opt closure C=%d %dx%d: loss rel err %.1e, gradient outliers %.2e, worst %.1e of scale"
          % (C, h, h, abs(lt + ls - tot_ref) / abs(tot_ref), frac_bad, float(err.max()) / scale))


def test_opt_20_iterations_final_loss_at_a_shipping_shape():
    """SURVEY section 7's criterion at a shipping shape: after the pipeline's 20 Adam iterations at up_blocks.1's input
    (C = 1280, 16 x 16, 8 frames, CFG batch 16) the loss reached must agree with the oracle's within 1 % (element-wise
    agreement is not attainable: L1 losses + Adam are chaotic, tests/test_gpu_opt.py), and two runs must be bit-identical."""
    import fresco_amd.ops as ops
    from fresco_amd.warp import _prep_flow_occ
    N, R, C, h = 8, 512, 1280, 16
    case = synth.make_opt_case(N, C, h, R, seed=41)
    x, tgt = case["x"], case["target"]
    prep = _prep_flow_occ(h, [f.to(DEV) for f in case["flows"]], [o.to(DEV) for o in case["occs"]], with_dilate=False)
    cs = x.to(DEV).clone()
    ops.opt_run(cs, prep, tgt.to(DEV), 100.0, 20, 2)
    cs2 = x.to(DEV).clone()
    ops.opt_run(cs2, prep, tgt.to(DEV), 100.0, 20, 2)
    assert torch.equal(cs, cs2)
    ref = O.optimize_feature(x, case["flows"], case["occs"], [tgt], iters=20, return_raw=True)
    prep32 = O.opt_prepare(h, case["flows"], case["occs"], 2, torch.float32)
    l_ours, _ = O.opt_loss_and_grad(cs.cpu(), prep32, tgt, 100.0)
    l_ref, _ = O.opt_loss_and_grad(ref, prep32, tgt, 100.0)
    l_0, _ = O.opt_loss_and_grad(x, prep32, tgt, 100.0)
    print("opt 20 iterations C=%d %dx%d: loss %.6f -> ours %.6f, oracle %.6f (rel diff %.2e)"
          % (C, h, h, float(l_0), float(l_ours), float(l_ref), abs(float(l_ours) - float(l_ref)) / float(l_ref)))
    assert float(l_ours) < float(l_0)
    assert abs(float(l_ours) - float(l_ref)) < 0.01 * float(l_ref)


@pytest.mark.parametrize("C,h", [(640, 64), (1280, 32)])
def test_opt_20_iterations_final_loss_at_the_big_shapes(C, h, monkeypatch):
    """The same criterion at the two shapes that cost 89 % of the feature optimisation's time: up_blocks.3's input
    (C = 640, 64 x 64) and up_blocks.2's (C = 1280, 32 x 32), 8 frames, CFG batch 16, 20 Adam iterations
    (src/diffusion_hacked.py:432-485).  The oracle's loop (analytic gradients, fp32) is evaluated by torch on the GPU
    -- the CPU needs ~4 minutes for it -- and both end points are scored by the oracle's fp64 loss.  These shapes run
    the two-stream form (one pipeline per CFG half): two runs must still be bit-identical."""
    import fresco_amd.ops as ops
    from fresco_amd.warp import _prep_flow_occ
    N, R = 8, 512
    case = synth.make_opt_case(N, C, h, R, seed=43)
    x, tgt = case["x"], case["target"]
    fd, od, td = [f.to(DEV) for f in case["flows"]], [o.to(DEV) for o in case["occs"]], tgt.to(DEV)
    prep = _prep_flow_occ(h, fd, od, with_dilate=False)
    cs = x.to(DEV).clone()
    ops.opt_run(cs, prep, td, 100.0, 20, 2)
    cs2 = x.to(DEV).clone()
    ops.opt_run(cs2, prep, td, 100.0, 20, 2)
    assert torch.equal(cs, cs2)
    # ... and the launch form must not matter: one stream (FRESCO_OPT_SPLIT=0), two pipelines started half an iteration
    # apart (2), whole S V tiles only (FRESCO_OPT_SVTAIL=0: these shapes run a half-tile tail round in one form or the other)
    # nor the Gram tile shape (FRESCO_GRAM_Z=0: 256 x 128 tiles also at 32 x 32, where the default is gram16z's 128 x 128)
    for var, val in (("FRESCO_OPT_SPLIT", "0"), ("FRESCO_OPT_SPLIT", "2"), ("FRESCO_OPT_SVTAIL", "0"), ("FRESCO_GRAM_Z", "0")):
        monkeypatch.setenv(var, val)
        cs3 = x.to(DEV).clone()
        ops.opt_run(cs3, prep, td, 100.0, 20, 2)
        monkeypatch.delenv(var)
        assert torch.equal(cs, cs3), (var, val, int((cs != cs3).sum()))
    ref = O.optimize_feature(x.to(DEV), fd, od, [td], iters=20, return_raw=True)
    prep64 = O.opt_prepare(h, fd, od, 2, torch.float64)
    l_ours = float(O.opt_loss_and_grad(cs.double(), prep64, td.double(), 100.0)[0])
    l_ref = float(O.opt_loss_and_grad(ref.double(), prep64, td.double(), 100.0)[0])
    l_0 = float(O.opt_loss_and_grad(x.to(DEV).double(), prep64, td.double(), 100.0)[0])
    print("opt 20 iterations C=%d %dx%d: loss %.6f -> ours %.6f, oracle %.6f (rel diff %.2e)"
          % (C, h, h, l_0, l_ours, l_ref, abs(l_ours - l_ref) / l_ref))
    assert l_ours < l_0
    assert abs(l_ours - l_ref) < 0.01 * l_ref


@pytest.mark.parametrize("C,h", [(640, 64), (1280, 32)])
def test_opt_gradient_deviation_vs_the_oracles_one_ulp_envelope(C, h):
    """SURVEY section 7, hard part 1: the gradient is a sum of sign() terms, so ANY fp32 evaluation differs from the
    exact one wherever a residual sits within rounding of zero.  The yardstick is the oracle itself: (a) its fp32
    gradient vs its fp64 gradient, (b) its fp32 gradient at x vs at x moved by ONE ulp in every element.  The HIP
    gradient's deviation from the fp64 oracle (elements off by more than 1e-3 of the gradient's scale) must stay within
    twice the larger of the two (plus four sign flips) -- i.e. it is rounding noise of the same size, not an error of the
    kernels."""
    import fresco_amd.ops as ops
    from fresco_amd.warp import _prep_flow_occ
    N, R = 8, 512
    case = synth.make_opt_case(N, C, h, R, seed=31)
    x = case["x"].to(DEV)
    fd, od, td = [f.to(DEV) for f in case["flows"]], [o.to(DEV) for o in case["occs"]], case["target"].to(DEV)
    prep64 = O.opt_prepare(h, fd, od, 2, torch.float64)
    prep32 = O.opt_prepare(h, fd, od, 2, torch.float32)
    g64 = O.opt_loss_and_grad(x.double(), prep64, td.double(), 100.0)[1]
    g32 = O.opt_loss_and_grad(x, prep32, td, 100.0)[1].double()
    x1 = torch.nextafter(x, torch.full_like(x, float("inf")))
    g32p = O.opt_loss_and_grad(x1, prep32, td, 100.0)[1].double()
    prep = _prep_flow_occ(h, fd, od, with_dilate=False)
    ghip = ops.opt_loss_grad(x, prep, td, 100.0, 2)[1].double()
    scale = float(g64.abs().max())
    frac = lambda a, b: float(((a - b).abs() > 1e-3 * scale).double().mean())
    f_f32, f_env, f_hip = frac(g32, g64), frac(g32p, g32), frac(ghip, g64)
    # ONE flipped sign of the Gram term moves the 2 C gradient entries of its two pixels: count the pixels that hold the
    # HIP outliers, and allow -- on top of twice the oracle's own noise -- what four such flips cost (at (1280, 32^2) the
    # oracle happens to have none for this seed; one near-tie entry out of 16.7 M is not an error of the kernels)
    bad = ((ghip - g64).abs() > 1e-3 * scale)
    pixels = int(bad.any(dim=1).sum())
    flips4 = 4 * 2.0 * C / ghip.numel()
    print("opt gradient C=%d %dx%d, elements off by > 1e-3 of scale: oracle fp32 vs fp64 %.2e | oracle fp32, input moved by "
          "1 ulp %.2e | HIP vs fp64 oracle %.2e (in %d pixels)" % (C, h, h, f_f32, f_env, f_hip, pixels))
    assert f_hip <= 2.0 * max(f_f32, f_env) + flips4 + 1e-5

