"""GPU parity of the fp32 attention kernel (fresco_attn_f32: fp32 operands split into fp16 pieces on the fp16 matrix pipe,
33-bit logits, 22-bit P V products) against an fp64 softmax(q k^T) v."""
import math

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(q, k, v, scale):
    s = (q.double() @ k.double().transpose(1, 2)) * scale
    return torch.softmax(s, -1) @ v.double()


@pytest.mark.parametrize("D,Dv,B,Lq,Lk", [(128, 128, 3, 200, 333), (128, 2, 2, 130, 257), (64, 64, 2, 64, 64),
                                           (32, 5, 1, 33, 31), (128, 128, 8, 1024, 1024), (128, 2, 2, 4096, 4096)])
def test_attention_f32(D, Dv, B, Lq, Lk):
    import fresco_amd.ops as ops
    g = synth.gen(D + Dv + Lq)
    q = torch.randn(B, Lq, D, generator=g) * 1.5
    k = torch.randn(B, Lk, D, generator=g) * 1.5
    v = torch.randn(B, Lk, Dv, generator=g)
    if Dv == 2:  # a pixel grid, as in the global matching step
        v = torch.stack((torch.arange(Lk) % 64, torch.arange(Lk) // 64), -1).float().expand(B, Lk, 2).contiguous()
    scale = 1.0 / math.sqrt(D)
    out = ops.attention_f32(q.to(DEV), k.to(DEV), v.to(DEV), scale)
    ref = _ref(q, k, v, scale)
    assert out.dtype == torch.float32 and tuple(out.shape) == (B, Lq, Dv)
    err = (out.cpu().double() - ref).abs().max()
    assert float(err) < 2e-5 * max(1.0, float(ref.abs().max())), float(err)


def test_attention_f32_peaked_logits_and_validation():
    import fresco_amd
    import fresco_amd.ops as ops
    g = synth.gen(9)
    q = torch.randn(1, 96, 128, generator=g) * 6.0   # logits of several tens: near one-hot rows
    k = torch.randn(1, 500, 128, generator=g) * 6.0
    v = torch.randn(1, 500, 128, generator=g)
    out = ops.attention_f32(q.to(DEV), k.to(DEV), v.to(DEV), 1.0 / math.sqrt(128))
    assert float((out.cpu().double() - _ref(q, k, v, 1.0 / math.sqrt(128))).abs().max()) < 5e-5
    with pytest.raises(fresco_amd.FrescoHipError):
        ops.attention_f32(q.to(DEV)[..., :48], k.to(DEV)[..., :48], v.to(DEV), 1.0)   # D = 48 unsupported
    with pytest.raises(ValueError):
        ops.attention_f32(q.to(DEV), k.to(DEV)[:, :10], v.to(DEV), 1.0)


def test_attention_f32_workspace_form_is_bit_identical():
    """fresco_attn_f32_ws (K / V split once per launch, operand images by LDS-DMA) runs the arithmetic of fresco_attn_f32 on the
    same numbers: the results must be equal bit for bit (ragged last key tile, Lq not a multiple of 128, Dv 128 / 2)."""
    import fresco_amd.ops as ops
    from fresco_amd import _lib
    lib = _lib.load()
    g = synth.gen(77)
    for (B, Lq, Lk, D, Dv) in ((3, 600, 333, 128, 128), (2, 1024, 1024, 128, 2), (2, 512, 96, 64, 40), (1, 515, 700, 32, 5)):
        q = (torch.randn(B, Lq, D, generator=g) * 1.5).to(DEV)
        k = (torch.randn(B, Lk, D, generator=g) * 1.5).to(DEV)
        v = torch.randn(B, Lk, Dv, generator=g).to(DEV)
        a = ops.attention_f32(q, k, v, 1.0 / math.sqrt(D))          # Lq >= 256: the workspace form
        b = torch.empty_like(a)
        rc = lib.fresco_attn_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), b.data_ptr(), B, Lq, Lk, D, Dv,
                                 1.0 / math.sqrt(D), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b), (B, Lq, Lk, D, Dv, float((a - b).abs().max()))
        assert float((a.cpu().double() - _ref(q.cpu(), k.cpu(), v.cpu(), 1.0 / math.sqrt(D))).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))



@pytest.mark.parametrize("what", ["k", "v", "q"])
@pytest.mark.parametrize("Lq", [96, 700])
def test_attention_f32_operands_beyond_fp16_range(what, Lq):
    """the split-fp16 kernels scale operands by 2^6 before the split, so |k|, |v|, |q scale log2 e| >= ~1000
    overflow the hi piece to inf (NaN out of the MFMA).  The guarded entry's range pass must hand such a launch to the
    exact-fp32 MFMA kernel: finite, fp32-accurate results for pixel-grid-like values of 5000 and features of 2000, in the
    per-workgroup form (Lq < 256) and the workspace form; the same operands scaled into range take the split-fp16
    kernels (bit-identical to fresco_attn_f32, test above)."""
    import fresco_amd.ops as ops
    g = synth.gen(123 + Lq)
    B, Lk, D, Dv = 2, 300, 128, 2 if what == "v" else 64
    q = torch.randn(B, Lq, D, generator=g)
    k = torch.randn(B, Lk, D, generator=g)
    v = torch.randn(B, Lk, Dv, generator=g)
    scale = 1.0 / math.sqrt(D)
    if what == "v":
        v = v * 2000.0 + 5000.0
    elif what == "k":   # one huge key coordinate, met by a tiny query coordinate: logits stay moderate
        k[:, :, 0] = 2000.0
        q[:, :, 0] *= 1e-3
    else:
        q[:, :, 1] = 9000.0
        k[:, :, 1] *= 1e-3
    out = ops.attention_f32(q.to(DEV), k.to(DEV), v.to(DEV), scale)
    ref = _ref(q, k, v, scale)
    assert bool(torch.isfinite(out).all())
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err
