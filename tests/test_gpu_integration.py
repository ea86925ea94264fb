"""End-to-end on the GPU over a stand-in decoder: the forward hook (optimize_feature + warp_tensor at the
input of every up-block) and the shared FRESCO processor (cross-frame + temporal attention in the last two
blocks) running together in fp16, against the same composition of oracle functions on the CPU.
iters = 0 keeps the comparison deterministic (optimize_feature then reduces to its AdaIN epilogue); the
chaotic Adam loop is covered by test_gpu_opt.py."""
import copy

import pytest
import torch

import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, R = 4, 128
CH = [32, 32, 64, 64]      # channels entering up-block i
SIDE = [4, 8, 16, 16]      # feature side entering up-block i (R/32 .. R/8)


class Block(torch.nn.Module):
    """up-block stand-in: [self-attention through the processor] -> 1x1 conv -> optional 2x upsample"""

    def __init__(self, cin, cout, attn, up, proc_ref):
        super().__init__()
        self.attn = attn
        self.proj = torch.nn.Conv2d(cin, cout, 1)
        self.up = up
        self.proc_ref = proc_ref

    def forward(self, hidden_states, temb=None):
        x = hidden_states
        if self.attn is not None:
            b, c, h, w = x.shape
            t = self.proc_ref[0](self.attn, x.view(b, c, h * w).transpose(1, 2).contiguous())
            x = x + t.transpose(1, 2).reshape(b, c, h, w)
        x = self.proj(x)
        if self.up:
            x = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
        return x


class UNet(torch.nn.Module):
    def __init__(self, proc_ref):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        outs = CH[1:] + [8]
        ups = [True, True, False, False]
        blocks = []
        for i in range(4):
            attn = synth.FakeAttn(CH[i], 8) if i >= 2 else None
            blocks.append(Block(CH[i], outs[i], attn, ups[i], proc_ref))
        self.up_blocks = torch.nn.ModuleList(blocks)

    def forward(self, sample, timestep, return_dict=True):
        for b in self.up_blocks:
            sample = b(hidden_states=sample, temb=None)
        return (sample,) if not return_dict else {"sample": sample}


class Pipe:
    pass


def test_hook_and_processor_on_stand_in_decoder():
    import fresco_amd

    g = synth.gen(31)
    flows, occs = synth.make_flows(N, R, g)
    sal = torch.rand(N, 1, R // 2, R // 2, generator=g)
    imgs = torch.rand(N, 3, R, R, generator=g)
    x0 = torch.randn(2 * N, CH[0], SIDE[0], SIDE[0], generator=g).half()
    fmap, bmap, tmask = O.mapping_ind(flows[1], occs[1], imgs, scale=8.0)     # 16 x 16 tokens
    cfm = O.cross_frame_masks(occs[1], scales=(8.0,))[0]

    proc_ref = [None]
    unet = UNet(proc_ref)
    with torch.no_grad():
        for p in unet.parameters():
            p.copy_(p.half().float())
    unet_gpu = copy.deepcopy(unet).to(DEV).half()
    ctrl = fresco_amd.AttentionControl()
    ctrl.enable_interattn(dict(fwd_mappings=[fmap.to(DEV)], bwd_mappings=[bmap.to(DEV)],
                               interattn_masks=[tmask.to(DEV)]))
    ctrl.enable_cfattn([cfm.to(DEV)])
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, ctrl)
    for b in unet_gpu.up_blocks:
        b.proc_ref = [proc]
    pipe = Pipe()
    pipe.unet = unet_gpu
    fresco_amd.apply_FRESCO_opt(pipe, steps=torch.tensor([700, 650]), layers=[0, 1, 2, 3],
                                flows=[f.to(DEV) for f in flows], occs=[o.to(DEV) for o in occs],
                                correlation_matrix=[], iters=0, saliency=sal.to(DEV))
    with torch.no_grad():
        out = pipe.unet(x0.to(DEV), torch.tensor(650, device=DEV), return_dict=False)
        out_off = pipe.unet(x0.to(DEV), torch.tensor(300, device=DEV), return_dict=False)
    assert len(out) == 5 and [tuple(t.shape[1:]) for t in out[1:]] == [(CH[i], SIDE[i], SIDE[i]) for i in range(4)]

    # the same composition with the oracle on the CPU (fp32, storage rounded to fp16 where the GPU path stores fp16)
    def r16(t):
        return t.half().float()

    def oracle_forward(x, active):
        for i, blk in enumerate(unet.up_blocks):
            if active:
                x = r16(O.adain(r16(x), x))                       # optimize_feature with iters = 0
                x = r16(O.warp_tensor(x, flows, occs, sal, 2))
            if blk.attn is not None:
                b, c, h, w = x.shape
                W = [p.detach() for p in blk.attn.weights()]
                t = O.fresco_attention(x.view(b, c, h * w).transpose(1, 2), W[0], W[1], W[2], W[3],
                                       blk.attn.to_out[0].bias.detach(), 8, use_cf=True, cf_mask=cfm,
                                       fwd_map=fmap[:, 0], tmask=tmask[:, 0], round_dtype=torch.float16)
                x = r16(x + t.transpose(1, 2).reshape(b, c, h, w))
            x = r16(blk.proj(x))
            if blk.up:
                x = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
        return x

    with torch.no_grad():
        ref_on = oracle_forward(x0.float(), True)
        ref_off = oracle_forward(x0.float(), False)
    for got, ref, what in ((out[0], ref_on, "hooks on"), (out_off[0], ref_off, "hooks off")):
        err = (got.float().cpu() - ref).abs()
        assert float(err.max()) < 2e-2 and float(err.mean()) < 2e-3, (what, float(err.max()), float(err.mean()))
    assert float((out[0].float() - out_off[0].float()).abs().max()) > 1e-2  # the hook did change the result
