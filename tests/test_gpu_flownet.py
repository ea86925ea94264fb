"""GPU parity of the flow network's dense-layer kernels (csrc/flownet.hip; SURVEY 8 row f3: GMFlow's CNN encoder, transformer
projections / FFN / LayerNorms, upsampler head -- gmflow/backbone.py:7-117, transformer.py:111-237, gmflow.py:44-90) against
torch in fp64 on the same fp32 inputs.  The products run on the fp16 matrix pipe from (hi, lo) operand planes: the bar is
fp32-class accuracy, a few 1e-6 of sum |a| |w| (an fp32 GEMM's own error grows like sqrt(K) 6e-8 of the same sum)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bar(a_abs_w_abs):
    return 4e-6 * float(a_abs_w_abs.max()) + 1e-6


@pytest.mark.parametrize("M,K,N,act,bias", [(1000, 128, 128, 0, False), (777, 256, 1024, 2, False), (515, 1024, 128, 0, False),
                                             (300, 256, 576, 0, True), (129, 128, 64, 1, True), (4096, 128, 96, 0, True)])
def test_fn_gemm_linear(M, K, N, act, bias):
    import fresco_amd.ops as ops
    g = synth.gen(M + K + N)
    x = torch.randn(M, K, generator=g) * 1.7
    x[::7] *= 1e-3                                   # small activations: their lo pieces would flush without the pre-scale
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) if bias else None
    _, xs = ops.fn_prep(x.to(DEV))
    _, ws = ops.fn_prep(w.to(DEV), scale=ops.FN_W_SCALE)
    out, outs = ops.fn_gemm(xs, ws, N, K, bias=None if b is None else b.to(DEV), act=act, want_split=True)
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    if act == 1:
        ref = ref.clamp_min(0)
    if act == 2:
        ref = F.gelu(ref)
    bar = _bar(x.abs().double() @ w.abs().double().t())
    err = float((out.cpu().double() - ref).abs().max())
    assert err < bar, (err, bar)
    # the epilogue's own (hi, lo) planes reproduce the fp32 result to 2^-22
    rec = (outs[0].float() + outs[1].float()) / ops.FN_A_SCALE
    assert float((rec - out).abs().max()) <= 2 ** -21 * float(out.abs().max()) + 1e-7


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,bias", [(64, 64, 3, 1, 1, 20, 28, False), (64, 96, 3, 2, 1, 20, 28, False),
                                                            (96, 128, 3, 2, 1, 16, 24, False), (64, 96, 1, 2, 0, 20, 28, True),
                                                            (128, 128, 3, 1, 1, 9, 13, False), (130, 256, 3, 1, 1, 8, 12, True),
                                                            (64, 64, 3, 1, 1, 32, 32, False), (64, 96, 3, 2, 1, 32, 64, False),
                                                            (96, 128, 1, 2, 0, 32, 64, True),
                                                            (128, 128, 3, 1, 1, 16, 32, False), (96, 96, 3, 1, 1, 32, 16, True),
                                                            (130, 256, 3, 1, 1, 16, 16, True), (64, 64, 3, 1, 1, 48, 16, True)])
def test_fn_gemm_implicit_conv(cin, cout, k, stride, pad, H, W, bias):
    """nn.Conv2d as the implicit GEMM over NHWC rows: every encoder / upsampler shape class (3 x 3 stride 1 / 2, the 1 x 1
    stride-2 shortcut with bias, a map whose width is no multiple of anything, and the 130-channel input padded to 160);
    the 3 x 3 / stride-1 cases on maps of whole 16 x 16 patches take the window-in-LDS form of the kernel (1 - 5 channel
    chunks, one and several patches per image, every border)"""
    import fresco_amd.ops as ops
    g = synth.gen(cin + cout + k + H)
    n = 3
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) if bias else None
    cp = (cin + 31) // 32 * 32
    xr = x.permute(0, 2, 3, 1).reshape(n * H * W, cin)
    if cin % 4:
        xr = F.pad(xr, (0, 4 - cin % 4))
    _, xs = ops.fn_prep(xr.contiguous().to(DEV), ld=cp)
    wr = F.pad(w.permute(0, 2, 3, 1), (0, cp - cin)).reshape(cout, -1).contiguous()
    _, ws = ops.fn_prep(wr.to(DEV), scale=ops.FN_W_SCALE)
    out, _, (mean, rstd) = ops.fn_gemm(xs, ws, cout, k * k * cp, bias=None if b is None else b.to(DEV),
                                       conv=(n, H, W, k, k, stride, pad), instance_norm_eps=1e-5)
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad)
    OH, OW = ref.shape[2], ref.shape[3]
    # InstanceNorm2d statistics of the result: out of the epilogue's partial sums when OH * OW % 128 == 0 (the last three
    # cases, which also take the 8 x 16 patch order of the row blocks), else from a second pass (fresco_fn_colstats)
    rf = ref.reshape(n, cout, -1)
    assert float((mean.cpu().double() - rf.mean(2)).abs().max()) < 2e-6
    assert float((rstd.cpu().double() * (rf.var(2, unbiased=False) + 1e-5).sqrt() - 1).abs().max()) < 1e-5
    got = out.view(n, OH, OW, cout).permute(0, 3, 1, 2).cpu().double()
    bar = _bar(F.conv2d(x.abs().double(), w.abs().double(), None, stride=stride, padding=pad))
    err = float((got - ref).abs().max())
    assert err < bar, (err, bar)


def test_fn_instance_norm_prep_and_layernorm():
    import fresco_amd.ops as ops
    g = synth.gen(5)
    n, rows, C = 3, 700, 96
    x = torch.randn(n * rows, C, generator=g) * 3.0 + 1.5
    res = torch.randn(n * rows, C, generator=g)
    mean, rstd = ops.fn_colstats(x.to(DEV), n)
    xd = x.double().view(n, rows, C)
    m_ref, v_ref = xd.mean(1), xd.var(1, unbiased=False)
    assert float((mean.cpu().double() - m_ref).abs().max()) < 1e-6
    assert float((rstd.cpu().double() - 1 / (v_ref + 1e-5).sqrt()).abs().max()) < 1e-6
    y, ys = ops.fn_prep(x.to(DEV), mean, rstd, residual=res.to(DEV), rows_per_img=rows, relu_a=True, relu_b=True, want_f32=True,
                        ld=128)
    ref = (((xd - m_ref[:, None]) / (v_ref[:, None] + 1e-5).sqrt()).clamp_min(0).view(-1, C) + res.double()).clamp_min(0)
    assert float((y.cpu().double() - ref).abs().max()) < 5e-6
    rec = (ys[0].float() + ys[1].float()) / ops.FN_A_SCALE
    assert tuple(rec.shape) == (n * rows, 128) and float(rec[:, C:].abs().max()) == 0.0
    assert float((rec[:, :C].cpu().double() - ref).abs().max()) < 5e-6
    # LayerNorm(128) + residual
    t = torch.randn(1000, 128, generator=g) * 2.0
    r2 = torch.randn(1000, 128, generator=g)
    ln = torch.nn.LayerNorm(128)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(128, generator=g))
        ln.bias.copy_(torch.randn(128, generator=g))
    y2, _ = ops.fn_layernorm(t.to(DEV), ln.weight.to(DEV), ln.bias.to(DEV), residual=r2.to(DEV))
    ref2 = F.layer_norm(t.double(), (128,), ln.weight.double(), ln.bias.double(), 1e-5) + r2.double()
    assert float((y2.cpu().double() - ref2).abs().max()) < 5e-6


@pytest.mark.parametrize("n,H,W", [(2, 40, 72), (1, 37, 51), (2, 64, 256), (1, 6, 1030), (3, 128, 128)])
def test_fn_conv7_rgb_stem(n, H, W):
    """the 7 x 7 / stride-2 stem (backbone.py:69) as split-fp16 MFMA products over the k' = 32 ky + 3 kx + ci layout: odd
    maps, maps narrower / wider than one 256-column tile, every border; InstanceNorm statistics out of the epilogue
    ((2, 64, 256): OW = 128, (3, 128, 128): OW = 64 -- two waves of a tile lie outside the map) and from the second pass"""
    import fresco_amd.ops as ops
    g = synth.gen(7 + H)
    x = torch.randn(n, 3, H, W, generator=g) * 1.3
    x[:, :, ::5] *= 1e-3
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    ws = ops.fn_conv7_weight(w.to(DEV))
    out, (mean, rstd) = ops.fn_conv7_rgb(x.permute(0, 2, 3, 1).contiguous().to(DEV), ws, instance_norm_eps=1e-5)
    ref = F.conv2d(x.double(), w.double(), None, stride=2, padding=3)
    got = out.permute(0, 3, 1, 2).cpu().double()
    assert tuple(got.shape) == tuple(ref.shape)
    bar = _bar(F.conv2d(x.abs().double(), w.abs().double(), None, stride=2, padding=3))
    err = float((got - ref).abs().max())
    assert err < bar, (err, bar)
    rf = ref.reshape(n, 64, -1)
    assert float((mean.cpu().double() - rf.mean(2)).abs().max()) < 2e-6
    assert float((rstd.cpu().double() * (rf.var(2, unbiased=False) + 1e-5).sqrt() - 1).abs().max()) < 1e-5
    # without the statistics: the same map, bit for bit
    assert torch.equal(ops.fn_conv7_rgb(x.permute(0, 2, 3, 1).contiguous().to(DEV), ws), out)
    # an input outside the operand planes' range trips the guard
    with ops.fn_range_guard(out.device) as g2:
        ops.fn_conv7_rgb((x * 2000).permute(0, 2, 3, 1).contiguous().to(DEV), ws)
    assert g2.tripped()
    with ops.fn_range_guard(out.device) as g3:
        ops.fn_conv7_rgb(x.permute(0, 2, 3, 1).contiguous().to(DEV), ws)
    assert not g3.tripped()


def test_fn_convex_upsample():
    """gmflow.py:75-90 restated with torch ops (softmax over the 9 taps, F.unfold of 8 * flow) vs the fused kernel"""
    import fresco_amd.ops as ops
    g = synth.gen(21)
    B, h, w = 3, 6, 10
    logits_nchw = torch.randn(B, 576, h, w, generator=g) * 2.0
    flow = torch.randn(B, 2, h, w, generator=g) * 3.0
    mask = logits_nchw.view(B, 1, 9, 8, 8, h, w).softmax(2)
    nb = F.unfold(8 * flow, (3, 3), padding=1).view(B, 2, 9, 1, 1, h, w)
    ref = (mask * nb).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * h, 8 * w)
    out = ops.fn_convex_upsample(logits_nchw.permute(0, 2, 3, 1).reshape(B * h * w, 576).contiguous().to(DEV),
                                 flow.flatten(2).transpose(1, 2).contiguous().to(DEV), B, h, w)
    assert float((out.cpu() - ref).abs().max()) < 2e-5


def test_fn_gemm_row_tables():
    """a_rows / out_rows: problem row m reads input row a_rows[m], its results land in output row out_rows[m] (the token
    gather / scatter of the window attention folded into the projections)"""
    import fresco_amd.ops as ops
    g = synth.gen(33)
    M, K, N = 777, 128, 128
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    perm = torch.randperm(M, generator=g)
    _, xs = ops.fn_prep(x.to(DEV))
    _, ws = ops.fn_prep(w.to(DEV), scale=ops.FN_W_SCALE)
    ref = (x.double() @ w.double().t())
    bar = _bar(x.abs().double() @ w.abs().double().t())
    t = perm.to(torch.int32).to(DEV)
    a, _ = ops.fn_gemm(xs, ws, N, K, a_rows=t)                      # a[m] = (x W^T)[perm[m]]
    assert float((a.cpu().double() - ref[perm]).abs().max()) < bar
    out = torch.zeros(M, N, device=DEV)
    ops.fn_gemm(xs, ws, N, K, out_rows=t, out_f32=out)                # out[perm[m]] = (x W^T)[m]
    back = torch.empty_like(ref)
    back[perm] = ref
    assert float((out.cpu().double() - back).abs().max()) < bar
    _, sp = ops.fn_gemm(xs, ws, N, K, out_rows=t, want_f32=False, want_split=True)
    rec = (sp[0].float() + sp[1].float()) / ops.FN_A_SCALE
    assert float((rec.cpu().double() - back).abs().max()) < bar + 2 ** -20 * float(back.abs().max())


def test_fn_range_flag_is_raised_by_every_producer():
    """the (hi, lo) planes hold x * 2^6 (weights: w * 2^10) and saturate at 65000 -- finite, wrong.  Every
    producer (fn_prep, fn_layernorm, the GEMM epilogue) must say so through the range word of the enclosing guard, and must
    NOT raise it for in-range operands."""
    import fresco_amd.ops as ops
    g = synth.gen(77)
    x = (torch.randn(256, 128, generator=g) * 3.0).to(DEV)
    w = (torch.randn(128, 128, generator=g) * 0.05).to(DEV)
    gam, bet = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    with ops.fn_range_guard(x.device) as ok:
        _, xs = ops.fn_prep(x)
        _, ws = ops.fn_prep(w, scale=ops.FN_W_SCALE)
        ops.fn_gemm(xs, ws, 128, 128, want_split=True)
        ops.fn_layernorm(x, gam, bet, want_split=True)
    assert not ok.tripped()
    big = x.clone()
    big[17, 5] = 1016.0                                   # 1016 * 64 = 65024 > 65000
    with ops.fn_range_guard(x.device) as g1:
        ops.fn_prep(big)
    assert g1.tripped()
    with ops.fn_range_guard(x.device) as g1b:             # (no planes requested: nothing is split, nothing can saturate)
        ops.fn_prep(big, want_f32=True, want_split=False)
    assert not g1b.tripped()
    wbig = w.clone()
    wbig[3, 3] = -64.0                                    # 64 * 1024 = 65536
    with ops.fn_range_guard(x.device) as g2:
        ops.fn_prep(wbig, scale=ops.FN_W_SCALE)
    assert g2.tripped()
    with ops.fn_range_guard(x.device) as g3:              # in-range operands, out-of-range RESULT planes (30 * 128 * 0.5 = 1920)
        _, xs3 = ops.fn_prep(torch.full((256, 128), 30.0, device=DEV))
        _, ws3 = ops.fn_prep(torch.full((128, 128), 0.5, device=DEV), scale=ops.FN_W_SCALE)
        assert not g3.tripped()
        out, _ = ops.fn_gemm(xs3, ws3, 128, 128, want_split=True)
    assert g3.tripped() and abs(float(out[0, 0]) - 1920.0) < 1e-2   # (the fp32 result itself is exact: only the planes clamp)
    with ops.fn_range_guard(x.device) as g4:
        ops.fn_gemm(xs3, ws3, 128, 128, want_split=False)            # fp32 output only: no planes, no flag
    assert not g4.tripped()
    with ops.fn_range_guard(x.device) as g5:
        ops.fn_layernorm(x, gam * 500.0, bet, want_split=True)       # LayerNorm output ~ N(0, 1) * 500 reaches 1015
    assert g5.tripped()
    with ops.fn_range_guard(x.device) as g6:                         # a NaN is out of range too
        bad = x.clone()
        bad[0, 0] = float("nan")
        ops.fn_prep(bad)
    assert g6.tripped()


def test_fn_side_operands_are_checked():
    """bias / gamma / beta / mean / rstd / residual are read as raw fp32 words by the kernels -- a .half() or
    strided tensor is converted on the way in, a table of the wrong dtype is refused"""
    import fresco_amd.ops as ops
    g = synth.gen(5)
    x = torch.randn(64, 128, generator=g).to(DEV)
    w = (torch.randn(128, 128, generator=g) * 0.05).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    _, xs = ops.fn_prep(x)
    _, ws = ops.fn_prep(w, scale=ops.FN_W_SCALE)
    ref, _ = ops.fn_gemm(xs, ws, 128, 128, bias=b)
    half_bias, _ = ops.fn_gemm(xs, ws, 128, 128, bias=b.half())
    assert float((half_bias - ref).abs().max()) < 2e-3              # (fp16 rounding of the bias, not garbage)
    strided = torch.stack((b, b), 1)[:, 0]
    got, _ = ops.fn_gemm(xs, ws, 128, 128, bias=strided)
    assert torch.equal(got, ref)
    res = torch.randn(64, 256, generator=g).to(DEV)[:, ::2]          # non-contiguous residual
    y1, _ = ops.fn_layernorm(x, torch.ones(128, device=DEV).half(), torch.zeros(128, device=DEV), residual=res)
    y2, _ = ops.fn_layernorm(x, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), residual=res.contiguous())
    assert torch.equal(y1, y2)
    with pytest.raises(ValueError):
        ops.fn_gemm(xs, ws, 128, 128, a_rows=torch.arange(64, device=DEV))           # int64 table
    with pytest.raises(ValueError):
        ops.fn_gemm(xs, ws, 128, 128, bias=b.cpu())


def test_fn_gemm_stacked_projections_in_one_product():
    """Round 6: q | k | v of one input as ONE product (weights stacked, out_blocks = 3): every projection lands in a matrix of its
    own, bit-identical to three separate products (a column's accumulation does not depend on the tiling of N); with a
    row table, as the flow network's window attention uses it."""
    import fresco_amd.ops as ops
    g = synth.gen(31)
    M, C = 3000, 128
    x = torch.randn(M, C, generator=g).to(DEV)
    ws = [(torch.randn(C, C, generator=g) * 0.05).to(DEV) for _ in range(3)]
    table = torch.randperm(M, generator=g).to(torch.int32).to(DEV)
    _, xs = ops.fn_prep(x)
    planes = [ops.fn_prep(w, scale=ops.FN_W_SCALE)[1] for w in ws]
    stacked = (torch.cat([p[0] for p in planes], 0).contiguous(), torch.cat([p[1] for p in planes], 0).contiguous())
    for nb in (3, 2):
        st = (stacked[0][: nb * C].contiguous(), stacked[1][: nb * C].contiguous())
        fused, _ = ops.fn_gemm(xs, st, nb * C, C, a_rows=table, out_blocks=nb)
        assert tuple(fused.shape) == (nb, M, C)
        for i in range(nb):
            single, _ = ops.fn_gemm(xs, planes[i], C, C, a_rows=table)
            assert torch.equal(fused[i], single), i
    ref = x.double()[table.long()] @ ws[1].double().t()
    assert float((fused[1].double() - ref).abs().max()) < _bar(x.abs().double()[table.long()] @ ws[1].abs().double().t())
    with pytest.raises(ValueError):
        ops.fn_gemm(xs, stacked, 3 * C, C, out_blocks=3, want_split=True)


_FORMS_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import synth
import fresco_amd.ops as ops
g = synth.gen(99)
dev = "cuda"
res = {}
# a 3 x 3 / stride-1 convolution on whole 16 x 16 patches (the window-in-LDS form when FRESCO_FN_CONV_PATCH != 0) ...
n, cin, cout, H, W = 3, 96, 128, 64, 48
x = torch.randn(n * H * W, cin, generator=g)
w = torch.randn(cout, 9 * cin, generator=g) * 0.03
_, xs = ops.fn_prep(x.to(dev))
_, ws = ops.fn_prep(w.to(dev), scale=ops.FN_W_SCALE)
out, sp, (mean, rstd) = ops.fn_gemm(xs, ws, cout, 9 * cin, conv=(n, H, W, 3, 3, 1, 1), want_split=True, instance_norm_eps=1e-5)
res["conv"], res["conv_hi"], res["conv_lo"], res["mean"], res["rstd"] = out.cpu(), sp[0].cpu(), sp[1].cpu(), mean.cpu(), rstd.cpu()
# ... and a linear product with five column blocks (the XCD-aware workgroup order when FRESCO_FN_XCD_MAP != 0), 19 row blocks
M, K, N = 19 * 256 - 77, 256, 576
a = torch.randn(M, K, generator=g)
b = torch.randn(N, K, generator=g) * 0.05
_, as_ = ops.fn_prep(a.to(dev))
_, bs = ops.fn_prep(b.to(dev), scale=ops.FN_W_SCALE)
lin, lsp = ops.fn_gemm(as_, bs, N, K, act=2, want_split=True)
res["lin"], res["lin_hi"], res["lin_lo"] = lin.cpu(), lsp[0].cpu(), lsp[1].cpu()
torch.save(res, sys.argv[2])
"""


def test_fn_gemm_forms_agree(tmp_path):
    """the build's two A/B switches, each in a process of its own (they are read once per process): the XCD-aware workgroup
    order is a pure re-ordering of workgroups -- identical bits; the window-in-LDS convolution contracts in another order
    (channel chunks outer, taps inner) -- fp32-class agreement with the im2col form, identical InstanceNorm statistics to
    1e-6, and (hi, lo) planes that reproduce each form's own fp32 output"""
    import subprocess
    import sys
    import fresco_amd.ops as ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("default", {}), ("im2col", {"FRESCO_FN_CONV_PATCH": "0"}), ("plain_order", {"FRESCO_FN_XCD_MAP": "0"})):
        path = str(tmp_path / (tag + ".pt"))
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", _FORMS_SCRIPT, root, path], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = torch.load(path)
    d, p, q = outs["default"], outs["plain_order"], outs["im2col"]
    for k in d:  # workgroup order: nothing changes
        assert torch.equal(d[k], p[k]), k
    for k in ("lin", "lin_hi", "lin_lo"):  # the convolution switch does not touch linear products
        assert torch.equal(d[k], q[k]), k
    scale = float(d["conv"].abs().max())
    assert float((d["conv"] - q["conv"]).abs().max()) < 2e-6 * scale + 1e-7, "window form vs im2col form"
    assert not torch.equal(d["conv"], q["conv"])  # (the two forms really are two: another summation order)
    assert float((d["mean"] - q["mean"]).abs().max()) < 1e-6 and float((d["rstd"] / q["rstd"] - 1).abs().max()) < 1e-5
    for o in (d, q):
        rec = (o["conv_hi"].float() + o["conv_lo"].float()) / ops.FN_A_SCALE
        assert float((rec - o["conv"]).abs().max()) <= 2 ** -21 * scale + 1e-7
