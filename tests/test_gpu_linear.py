"""GPU parity of the fused projection kernel (fresco_linear) against fp32 matmul of the same fp16 operands."""
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(x, W, b, nw):
    y = x.float().reshape(-1, x.shape[-1]) @ W.float().t()
    if b is not None:
        y = y + b.float()
    return [t.reshape(x.shape[:-1] + (W.shape[0] // nw,)) for t in y.chunk(nw, dim=1)]


@pytest.mark.parametrize("K,N,nw,M,bias", [(320, 320, 3, 1000, False), (640, 640, 3, 300, False),
                                            (320, 320, 1, 4096, True), (640, 640, 2, 129, True),
                                            (320, 64, 1, 5, False), (640, 128, 3, 128, True), (320, 960, 1, 777, True)])
def test_linear_matches_fp32_matmul(K, N, nw, M, bias):
    import fresco_amd.ops as ops
    g = synth.gen(K + N + nw + M)
    x = torch.randn(M, K, generator=g).half()
    W = (torch.randn(nw * N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(nw * N, generator=g).half() if bias else None
    Ws = [w.contiguous().to(DEV) for w in W.chunk(nw, 0)]
    bs = None if b is None else [t.contiguous().to(DEV) for t in b.chunk(nw, 0)]
    outs = ops.linear(x.to(DEV), Ws, bs)
    assert len(outs) == nw
    for o, r in zip(outs, _ref(x, W, b, nw)):
        assert o.dtype == torch.float16 and tuple(o.shape) == (M, N)
        err = (o.float().cpu() - r).abs()
        assert float(err.max()) <= 2e-3 + 2e-3 * float(r.abs().max()), float(err.max())
        # fp32 accumulation + one rounding: almost every element is the correctly rounded fp16 of the exact sum
        assert float((o.cpu() != r.half()).double().mean()) < 0.02


def test_linear_strided_views_and_batch_dims():
    """x as a (B, HW, C) tensor, outputs written into the halves of a fused K|V buffer"""
    import fresco_amd.ops as ops
    g = synth.gen(3)
    B, HW, C = 4, 96, 320
    x = torch.randn(B, HW, C, generator=g).half().to(DEV)
    W = (torch.randn(2 * C, C, generator=g) / C ** 0.5).half().to(DEV)
    kv = torch.zeros(2, B, HW, C, dtype=torch.float16, device=DEV)
    outs = ops.linear(x, [W[:C].contiguous(), W[C:].contiguous()], None, outs=[kv[0], kv[1]])
    assert outs[0].data_ptr() == kv[0].data_ptr()
    ref = _ref(x.cpu(), W.cpu(), None, 2)
    assert float((kv[0].float().cpu() - ref[0]).abs().max()) < 3e-3
    assert float((kv[1].float().cpu() - ref[1]).abs().max()) < 3e-3
    # column-sliced input (row stride 2C)
    wide = torch.randn(B, HW, 2 * C, generator=g).half().to(DEV)
    o = ops.linear(wide[..., C:], [W[:C].contiguous()])[0]
    r = _ref(wide[..., C:].cpu(), W[:C].cpu(), None, 1)[0]
    assert float((o.float().cpu() - r).abs().max()) < 3e-3


def test_linear_rejects_unsupported():
    import fresco_amd
    import fresco_amd.ops as ops
    x = torch.zeros(8, 256, dtype=torch.float16, device=DEV)
    W = torch.zeros(256, 256, dtype=torch.float16, device=DEV)
    assert not ops.linear_supported(256, 256, torch.float16)
    with pytest.raises(fresco_amd.FrescoHipError):
        ops.linear(x, W)
    with pytest.raises(ValueError):
        ops.linear(x.float(), W)


class _Scaled(torch.nn.Linear):
    """a wrapped projection (what a LoRA / quantisation layer looks like to the processor): NOT a plain Linear"""

    def forward(self, x):
        return super().forward(x) * 0.5


@torch.no_grad()
def test_processor_fuses_only_plain_linears_and_tracks_weight_updates():
    import copy
    import fresco_amd
    from oracle import fresco_oracle as O
    g = synth.gen(21)
    C, H, B, HW = 320, 8, 4, 64
    attn = synth.FakeAttn(C, H).half().to(DEV)
    x = torch.randn(B, HW, C, generator=g).half().to(DEV)
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, fresco_amd.AttentionControl())
    seen = []
    real = fresco_amd.ops.linear

    def spy(x_, weights, *a, **k):
        seen.append(len(weights))
        return real(x_, weights, *a, **k)

    fresco_amd.ops.linear = spy
    try:
        y = proc(attn, x)
        assert seen == [3, 1]  # q,k,v in one launch, to_out (C = 320) in another
        W = [w.detach().float().cpu() for w in attn.weights()]
        ref = O.fresco_attention(x.float().cpu(), W[0], W[1], W[2], W[3], attn.to_out[0].bias.detach().float().cpu(), H,
                                 round_dtype=torch.float16)
        assert float((y.float().cpu() - ref).abs().max()) < 3e-3
        # in-place weight updates, also the kind that leaves `_version` and the address alone (EMA copy_to / LoRA
        # merges write through `.data`): the kernel reads the live weights, there is no stacked copy to go stale
        with torch.no_grad():
            attn.to_k.weight.mul_(-1.0)
        attn.to_v.weight.data.mul_(0.5)
        y2 = proc(attn, x)
        W[1] = -W[1]
        W[2] = 0.5 * W[2]
        ref2 = O.fresco_attention(x.float().cpu(), W[0], W[1], W[2], W[3], attn.to_out[0].bias.detach().float().cpu(), H,
                                  round_dtype=torch.float16)
        assert float((y2.float().cpu() - ref2).abs().max()) < 3e-3
        # a wrapped module keeps its own forward: no fused q,k,v launch, and its scaling is honoured
        seen.clear()
        attn2 = copy.deepcopy(attn)
        wrapped = _Scaled(C, C, bias=False).half().to(DEV)
        with torch.no_grad():
            wrapped.weight.copy_(attn2.to_q.weight)
        attn2.to_q = wrapped
        y3 = proc(attn2, x)
        assert 3 not in seen
        ref3 = O.fresco_attention(x.float().cpu(), 0.5 * W[0], W[1], W[2], W[3],
                                  attn.to_out[0].bias.detach().float().cpu(), H, round_dtype=torch.float16)
        assert float((y3.float().cpu() - ref3).abs().max()) < 3e-3
        # switch off: module calls only
        seen.clear()
        proc.fuse_projections = False
        y4 = proc(attn, x)
        assert seen == [] and float((y4.float() - y2.float()).abs().max()) < 2e-3
    finally:
        fresco_amd.ops.linear = real


def test_linear_gathered_rows():
    """fresco_linear_rows: output row m = projection of input row x_rows[m] (the K / V projection of the selected
    cross-frame tokens); out-of-range tables are refused."""
    import fresco_amd.ops as ops
    g = torch.Generator().manual_seed(3)
    Mx, C = 5000, 320
    x = torch.randn(Mx, C, generator=g).half().to(DEV)
    Ws = [(torch.randn(C, C, generator=g) / C ** 0.5).half().to(DEV) for _ in range(2)]
    rows = torch.randperm(Mx, generator=g)[:777].to(torch.int32).to(DEV)
    k, v = ops.linear(x, Ws, x_rows=rows)
    assert tuple(k.shape) == (777, C)
    xs = x.index_select(0, rows.long())
    for out, W in zip((k, v), Ws):
        ref = (xs.float() @ W.float().t())
        assert float((out.float() - ref).abs().max()) < 2e-3 + 2e-3 * float(ref.abs().max())
    same_k, same_v = ops.linear(xs, Ws)
    assert torch.equal(k, same_k) and torch.equal(v, same_v)  # bit-identical to gathering first
    bad = rows.clone()
    bad[3] = Mx
    with pytest.raises(ValueError):
        ops.linear(x, Ws, x_rows=bad)
