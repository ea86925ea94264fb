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
                                            (320, 64, 1, 5, False), (640, 96, 3, 128, True)])
def test_linear_matches_fp32_matmul(K, N, nw, M, bias):
    import fresco_amd.ops as ops
    g = synth.gen(K + N + nw + M)
    x = torch.randn(M, K, generator=g).half()
    W = (torch.randn(nw * N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(nw * N, generator=g).half() if bias else None
    outs = ops.linear(x.to(DEV), W.to(DEV), None if b is None else b.to(DEV), nw)
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
    outs = ops.linear(x, W, None, 2, outs=[kv[0], kv[1]])
    assert outs[0].data_ptr() == kv[0].data_ptr()
    ref = _ref(x.cpu(), W.cpu(), None, 2)
    assert float((kv[0].float().cpu() - ref[0]).abs().max()) < 3e-3
    assert float((kv[1].float().cpu() - ref[1]).abs().max()) < 3e-3
    # column-sliced input (row stride 2C)
    wide = torch.randn(B, HW, 2 * C, generator=g).half().to(DEV)
    o = ops.linear(wide[..., C:], W[:C], None, 1)[0]
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
