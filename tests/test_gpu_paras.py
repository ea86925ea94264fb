"""GPU parity of the per-batch parameter producers (SURVEY.md 8f-3 minus the flow network, 8f-4):
forward_backward_consistency_check, get_flow_and_interframe_paras (stand-in flow model returning fixed
flows) and get_intraframe_paras (stand-in pipe) against the reference-generated goldens and the oracle.
Occlusions / masks / mappings are booleans and integers: compared exactly on the closed-form cases; on
random data a float threshold can legitimately flip on a last-bit tie, so the count of differing pixels
is bounded instead (and the downstream integers are compared from identical occlusions)."""
import pytest
import torch

import closed_form as cf
import synth
from oracle import fresco_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(a)


class FixedFlow:
    """what GMFlow returns for pred_bidir_flow=True: flow_preds[-1] = cat(fwd, bwd) over the batch"""

    def __init__(self, fwd, bwd):
        self.fwd, self.bwd, self.calls = fwd, bwd, []

    def __call__(self, a, b, **kw):
        self.calls.append((a, b, kw))
        return {"flow_preds": [torch.cat([self.fwd, self.bwd], 0).to(a.device)]}


@pytest.mark.parametrize("tag,shape", [("a", (4, 64, 64)), ("b", (3, 96, 160))])
def test_fb_consistency_check_golden(paras_golden, tag, shape):
    import fresco_amd
    _, fwd, bwd = cf.video_case(*shape)
    fo, bo = fresco_amd.forward_backward_consistency_check(fwd.to(DEV), bwd.to(DEV))
    assert fo.dtype == torch.float32 and tuple(fo.shape) == (shape[0], shape[1], shape[2])
    assert torch.equal(fo.cpu().to(torch.uint8), T(paras_golden[tag + "_fb_fwd_occ"]))
    assert torch.equal(bo.cpu().to(torch.uint8), T(paras_golden[tag + "_fb_bwd_occ"]))


@pytest.mark.parametrize("tag,shape", [("a", (4, 64, 64)), ("b", (3, 96, 160))])
def test_get_flow_and_interframe_paras_golden(paras_golden, tag, shape):
    import fresco_amd
    g = paras_golden
    frames, fwd, bwd = cf.video_case(*shape)
    fm = FixedFlow(fwd, bwd)
    flows, occs, attn_mask, paras = fresco_amd.get_flow_and_interframe_paras(fm, frames)
    a, b, kw = fm.calls[0]
    assert kw == dict(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow=True)
    assert a.is_cuda and float(a.max()) > 1.5 and torch.equal(b, a[list(range(1, shape[0])) + [0]])
    assert torch.equal(flows[0].cpu(), fwd) and torch.equal(flows[1].cpu(), bwd)
    assert torch.equal(occs[0].cpu().to(torch.uint8), T(g[tag + "_fwd_occ"]))
    assert torch.equal(occs[1].cpu().to(torch.uint8), T(g[tag + "_bwd_occ"]))
    assert len(attn_mask) == 3
    for i, m in enumerate(attn_mask):
        assert m.dtype == torch.bool and torch.equal(m.cpu().to(torch.uint8), T(g["%s_attn_mask%d" % (tag, i)]))
    for i in range(2):
        assert torch.equal(paras["fwd_mappings"][i].cpu(), T(g["%s_fwd_map%d" % (tag, i)]))
        assert torch.equal(paras["bwd_mappings"][i].cpu(), T(g["%s_bwd_map%d" % (tag, i)]))
        assert torch.equal(paras["interattn_masks"][i].cpu().to(torch.uint8), T(g["%s_imask%d" % (tag, i)]))


def test_flow_occlusion_random_512_vs_oracle():
    import fresco_amd.ops as ops
    g = synth.gen(77)
    N, R = 8, 512
    flows, _ = synth.make_flows(N, R, g)
    fwd, bwd = flows
    # make the round trip close almost everywhere, with patches where it does not
    bwd = bwd + 0.4 * (torch.rand(N, 1, R, R, generator=g) < 0.05).float() * torch.randn(N, 2, R, R, generator=g)
    images = torch.rand(N, 3, R, R, generator=g) * 255.0
    images = torch.nn.functional.avg_pool2d(images, 9, 1, 4)  # smooth: colour test near its threshold
    images = (images - images.mean()) * 6.0 + 128.0
    fo, bo = ops.flow_occlusion(fwd.to(DEV), bwd.to(DEV), images.to(DEV))
    ro, rb = O.flow_occlusions(images, fwd, bwd)
    for a, b in ((fo, ro), (bo, rb)):
        frac = float(b.mean())
        assert 0.02 < frac < 0.98, frac
        assert int((a.cpu() != b).sum()) <= 4  # last-bit ties of a float threshold
    f2, b2 = ops.flow_occlusion(fwd.to(DEV), bwd.to(DEV), None)
    r2, rb2 = O.fb_consistency_check(fwd, bwd)
    assert int((f2.cpu() != r2).sum()) <= 4 and int((b2.cpu() != rb2).sum()) <= 4


def test_flow_occlusion_validation():
    import fresco_amd
    import fresco_amd.ops as ops
    f = torch.zeros(2, 2, 8, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.flow_occlusion(f, torch.zeros(2, 2, 8, 9, device=DEV))
    with pytest.raises(ValueError):
        ops.flow_occlusion(f, f, torch.zeros(3, 3, 8, 8, device=DEV))
    with pytest.raises(fresco_amd.FrescoHipError):
        ops.flow_occlusion(f.cpu(), f.cpu())
    # zero flow, identical frames: nothing occluded
    fo, bo = ops.flow_occlusion(f, f, torch.full((2, 3, 8, 8), 7.0, device=DEV))
    assert float(fo.sum()) == 0 and float(bo.sum()) == 0


class _Dist:
    def __init__(self, x):
        self.x = x

    def sample(self):
        return self.x


class _Enc:
    def __init__(self, x):
        self.latent_dist = _Dist(x)


class _Cfg:
    pass


class _VAE:
    def __init__(self):
        self.config = _Cfg()
        self.config.scaling_factor = 0.5

    def encode(self, imgs):
        return _Enc(torch.nn.functional.avg_pool2d(imgs, 8)[:, [0, 1, 2, 0]])


class _Sched:
    timesteps = torch.tensor([981, 500, 21])

    def add_noise(self, x0, noise, t):
        self.seen_t = int(t)
        return 0.9 * x0 + 0.1 * noise


class _Block(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = torch.nn.Conv2d(cin, cout, 1)

    def forward(self, hidden_states, temb=None):
        return self.proj(hidden_states)


class _UNet(torch.nn.Module):
    def __init__(self, ctrl_ref):
        super().__init__()
        self.config = _Cfg()
        self.config.in_channels = 4
        self.up_blocks = torch.nn.ModuleList([_Block(4, 8), _Block(8, 16), _Block(16, 8), _Block(8, 4)])
        self.ctrl_ref = ctrl_ref

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=True):
        for b in self.up_blocks:
            sample = b(hidden_states=sample, temb=None)
            c = self.ctrl_ref[0]
            if c.store:  # what the processor does at diffusion_hacked.py:206-207
                c(sample.flatten(2).transpose(1, 2).detach().clone())
        return (sample,) if not return_dict else {"sample": sample}


class _Pipe:
    _execution_device = torch.device(DEV)

    def prepare_latents(self, B, C, H, W, dtype, device, generator, latents=None):
        self.prep = (B, C, H, W)
        return torch.randn(B, C, H // 8, W // 8, generator=generator, device=device, dtype=dtype)


class _Proc:
    pass


def test_get_intraframe_paras_on_stand_in_pipe():
    import fresco_amd
    torch.manual_seed(0)
    ctrl = fresco_amd.AttentionControl()
    proc = _Proc()
    proc.controller = ctrl
    pipe = _Pipe()
    pipe.scheduler, pipe.vae = _Sched(), _VAE()
    pipe.unet = _UNet([ctrl]).to(DEV).half()
    imgs = torch.rand(3, 3, 64, 96, generator=synth.gen(5)).to(DEV) * 2 - 1
    emb = torch.zeros(6, 77, 16, device=DEV, dtype=torch.float16)
    ctrl.enable_cfattn(None)  # must be switched off by the call
    corr = fresco_amd.get_intraframe_paras(pipe, imgs, proc, emb, do_classifier_free_guidance=True, seed=3)
    assert pipe.scheduler.seen_t == 21 and pipe.prep == (3, 4, 64, 96)
    assert not ctrl.store and not ctrl.use_cfattn and len(ctrl.stored_attn["decoder_attn"]) == 4
    # features entering each up-block, recomputed with plain torch on the same pipe
    gen = torch.Generator(device=DEV).manual_seed(3)
    lat = torch.randn(3, 4, 8, 12, generator=gen, device=DEV, dtype=torch.float16)
    x0 = 0.5 * pipe.vae.encode(imgs.half()).latent_dist.sample()
    x = torch.cat([0.9 * x0 + 0.1 * lat] * 2)
    feats = []
    with torch.no_grad():
        for b in pipe.unet.up_blocks:
            feats.append(x)
            x = b.proj(x)
    assert len(corr) == 4
    for c, f in zip(corr, feats):
        assert c.dtype == torch.float32 and tuple(c.shape) == (6, 96, 96)
        ref = O.gram_target(f.float().cpu())
        assert float((c.cpu() - ref).abs().max()) < 2e-6
        # the reference's own arithmetic (normalise + bmm in the UNet dtype, cast) is within fp16 rounding
        v = f.flatten(2).transpose(1, 2)
        v = v / ((v ** 2).sum(dim=2, keepdims=True) ** 0.5)
        assert float((c - torch.bmm(v, v.transpose(-1, -2)).float()).abs().max()) < 2e-3
    # the forward keeps returning the decoder features afterwards, as disable_FRESCO_opt leaves it
    out = pipe.unet(torch.cat([lat] * 2), torch.tensor(21, device=DEV), encoder_hidden_states=emb, return_dict=False)
    assert len(out) == 5
