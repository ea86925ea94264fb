"""Synthetic, seeded inputs shared by the parity tests, smoke() and bench.py.

Everything is generated on the CPU with a seeded generator and then moved, so the CPU oracle and the
GPU path see identical bits.  The integer parameters (trajectory maps, masks) come from the oracle's
restatement of the reference's own prep code, so both sides consume identical integers.
"""
import math

import torch


class FakeAttn(torch.nn.Module):
    """Stand-in for diffusers' `Attention` of an SD-1.5 `attn1` block: exactly the attributes the
    processor touches: no norms, no residual, rescale 1, bias-free q/k/v."""

    def __init__(self, C, heads, weights=None, out_bias=None):
        super().__init__()
        self.heads = heads
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(C, C, bias=False)
        self.to_v = torch.nn.Linear(C, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        if weights is not None:
            with torch.no_grad():
                self.to_q.weight.copy_(weights[0])
                self.to_k.weight.copy_(weights[1])
                self.to_v.weight.copy_(weights[2])
                self.to_out[0].weight.copy_(weights[3])
                if out_bias is None:
                    self.to_out[0].bias.zero_()
                else:
                    self.to_out[0].bias.copy_(out_bias)

    def weights(self):
        return [self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight]


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def make_flows(N, R, g, occ_mode="bernoulli"):
    """Smooth synthetic flow: constant (3,-2) px + 0.3 px noise; fwd = -bwd; N entries (incl. the
    wrap-around pair).  Occlusions: Bernoulli(0.1) per pixel, or 32x32-px blocks with p = 0.5."""
    base = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1)
    bwd = base + 0.3 * torch.randn(N, 2, R, R, generator=g)
    fwd = -bwd
    if occ_mode == "bernoulli":
        bo = (torch.rand(N, R, R, generator=g) < 0.1).float()
        fo = (torch.rand(N, R, R, generator=g) < 0.1).float()
    else:
        nb = max(R // 32, 1)
        bo = (torch.rand(N, nb, nb, generator=g) < 0.5).float().repeat_interleave(R // nb, 1).repeat_interleave(R // nb, 2)
        fo = (torch.rand(N, nb, nb, generator=g) < 0.5).float().repeat_interleave(R // nb, 1).repeat_interleave(R // nb, 2)
    return [fwd, bwd], [fo, bo]


def make_attention_case(N, R, layer, seed=0, occ_mode="bernoulli", dtype=torch.float16):
    """One FRESCO self-attention layer call.  layer 'L2' = up_blocks.2 (C=640, D=80, HW=(R/16)^2),
    'L3' = up_blocks.3 (C=320, D=40, HW=(R/8)^2).  Returns a dict of CPU tensors."""
    from oracle import fresco_oracle as O

    g = gen(seed)
    C, down = (640, 16) if layer == "L2" else (320, 8)
    heads = 8
    side = R // down
    HW = side * side
    B = 2 * N
    attn = FakeAttn(C, heads)
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(p.to(dtype).float())  # weights representable in the compute dtype
    hidden = torch.randn(B, HW, C, generator=g).to(dtype)
    ref = (hidden.float() + 0.1 * torch.randn(B, HW, C, generator=g)).to(dtype)
    flows, occs = make_flows(N, R, g, occ_mode)
    imgs = torch.rand(N, 3, R, R, generator=g)
    fwd_map, bwd_map, tmask = O.mapping_ind(flows[1], occs[1], imgs, scale=float(down))
    cf_mask = O.cross_frame_masks(occs[1], scales=(float(down),))[0]
    return dict(attn=attn, hidden=hidden, ref=ref, flows=flows, occs=occs, fwd_map=fwd_map,
                bwd_map=bwd_map, tmask=tmask, cf_mask=cf_mask, N=N, HW=HW, C=C, heads=heads)


def oracle_attention(case, mode, round_dtype=torch.float16, device=None, chunk=2):
    """fp32 oracle output of one layer call for mode in {'plain','full','cf_temporal','cf','temporal'}.
    device: where torch evaluates the oracle's (device-agnostic) tensor code -- None = CPU; the biggest configurations
    pass "cuda" (the CPU needs minutes there); tests/test_gpu_fullsize.py checks that the device does not matter."""
    from oracle import fresco_oracle as O

    dev = device or "cpu"
    a = case["attn"]
    W = [w.detach().float().to(dev) for w in a.weights()]
    bo = a.to_out[0].bias.detach().float().to(dev)
    kw = {}
    if mode in ("full", "cf_temporal", "cf"):
        kw.update(use_cf=True, cf_mask=case["cf_mask"].to(dev))
    if mode in ("full", "cf_temporal", "temporal"):
        kw.update(fwd_map=case["fwd_map"][:, 0].to(dev), tmask=case["tmask"][:, 0].to(dev))
    if mode == "full":
        kw.update(ref=case["ref"].float().to(dev))
    out = O.fresco_attention(case["hidden"].float().to(dev), W[0], W[1], W[2], W[3], bo, case["heads"],
                             round_dtype=round_dtype, chunk=chunk, **kw)
    return out.cpu()


def controller_for(case, mode, device, dtype=None):
    """A fresco_amd AttentionControl configured like the pipeline would for `mode`."""
    import fresco_amd

    c = fresco_amd.AttentionControl()
    if mode == "full":
        c.stored_attn["decoder_attn"] = [case["ref"].to(device).to(dtype or case.get("dtype", torch.float16))]
        c.enable_intraattn()
    if mode in ("full", "cf_temporal", "temporal"):
        c.enable_interattn(dict(fwd_mappings=[case["fwd_map"].to(device)],
                                bwd_mappings=[case["bwd_map"].to(device)],
                                interattn_masks=[case["tmask"].to(device)]))
    if mode in ("full", "cf_temporal", "cf"):
        c.enable_cfattn([case["cf_mask"].to(device)])
    return c


def make_opt_case(N, C, h, R, seed=0):
    """Features + flows + Gram target for optimize_feature at feature side h, flow side R."""
    from oracle import fresco_oracle as O

    g = gen(seed)
    B = 2 * N
    x = torch.randn(B, C, h, h, generator=g)
    flows, occs = make_flows(N, R, g)
    tgt_feat = torch.randn(B, C, h, h, generator=g)
    target = O.gram_target(tgt_feat)
    sal = torch.rand(N, 1, R // 2, R // 2, generator=g)
    return dict(x=x, flows=flows, occs=occs, target=target, sal=sal, N=N)
