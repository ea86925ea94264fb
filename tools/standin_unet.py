"""A stand-in for the SD-1.5 UNet + ControlNet the reference drives: the module tree, tensor
shapes and attribute names of diffusers 0.19.3's `UNet2DConditionModel` / `ControlNetModel`, plain torch modules,
random weights.  NOT part of the product and not a re-implementation of diffusers: it exists so that
tools/bench_full_step.py can time one whole denoising step (everything that is not FRESCO's hot path is PyTorch's own
conv / GEMM / SDPA code here, as it would be with the real model).

Every attention is an `Attention` module with diffusers' attribute names (to_q / to_k / to_v / to_out, heads,
spatial_norm, group_norm, norm_cross, residual_connection, rescale_output_factor) and a replaceable `.processor`
callable `(attn, hidden_states, encoder_hidden_states=None, ...)` -- the plugin surface FRESCO uses."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SDPAProcessor:
    """what diffusers' AttnProcessor2_0 does: projections + torch SDPA"""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        B, L, _ = hidden_states.shape
        h = attn.heads
        q = attn.to_q(hidden_states).view(B, L, h, -1).transpose(1, 2)
        k = attn.to_k(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        v = attn.to_v(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, -1)
        return attn.to_out[1](attn.to_out[0](o))


class Attention(nn.Module):
    def __init__(self, dim, heads, cross_dim=None):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(cross_dim or dim, dim, bias=False)
        self.to_v = nn.Linear(cross_dim or dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.spatial_norm = self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = SDPAProcessor()

    def forward(self, hidden_states, encoder_hidden_states=None):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.ff_in = nn.Linear(dim, 8 * dim)  # GEGLU
        self.ff_out = nn.Linear(4 * dim, dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        a, g = self.ff_in(self.norm3(x)).chunk(2, dim=-1)
        return x + self.ff_out(a * F.gelu(g))


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, cross_dim):
        super().__init__()
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        y = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            y = blk(y, ctx)
        return x + self.proj_out(y.reshape(B, H, W, C).permute(0, 3, 1, 2))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch=1280):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(32, cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2, self.conv2 = nn.GroupNorm(32, cout), nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x))) + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class DownBlock(nn.Module):
    def __init__(self, cin, cout, cross, down, heads=8, cross_dim=768):
        super().__init__()
        self.has_cross_attention = cross
        self.resnets = nn.ModuleList([ResnetBlock2D(cin, cout), ResnetBlock2D(cout, cout)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim) for _ in range(2)]) if cross else None
        self.downsamplers = nn.ModuleList([nn.Conv2d(cout, cout, 3, stride=2, padding=1)]) if down else None

    def forward(self, hidden_states, temb, encoder_hidden_states=None):
        outs = ()
        for i, r in enumerate(self.resnets):
            hidden_states = r(hidden_states, temb)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class MidBlock(nn.Module):
    def __init__(self, ch, heads=8, cross_dim=768):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch), ResnetBlock2D(ch, ch)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, cross_dim)])

    def forward(self, hidden_states, temb, encoder_hidden_states=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states, encoder_hidden_states)
        return self.resnets[1](hidden_states, temb)


class UpBlock(nn.Module):
    def __init__(self, cprev, cout, skips, cross, up, heads=8, cross_dim=768):
        super().__init__()
        self.has_cross_attention = cross
        cins = [cprev] + [cout] * 2
        self.resnets = nn.ModuleList([ResnetBlock2D(cins[i] + skips[i], cout) for i in range(3)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim) for _ in range(3)]) if cross else None
        self.upsamplers = nn.ModuleList([nn.Conv2d(cout, cout, 3, padding=1)]) if up else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states=None):
        for i, r in enumerate(self.resnets):
            hidden_states = r(torch.cat([hidden_states, res_hidden_states_tuple[-1 - i]], dim=1), temb)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](F.interpolate(hidden_states, scale_factor=2.0, mode="nearest"))
        return hidden_states


def _time_embedding(t, dim=320):
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, device=t.device, dtype=torch.float32) / half)
    a = t.float()[:, None] * f[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class _Encoder(nn.Module):
    """conv_in + time embedding + the four down blocks + mid block (shared by the UNet and the ControlNet)"""

    def __init__(self, cin=4):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, 320, 3, padding=1)
        self.time_embedding = nn.Sequential(nn.Linear(320, 1280), nn.SiLU(), nn.Linear(1280, 1280))
        self.down_blocks = nn.ModuleList([DownBlock(320, 320, True, True), DownBlock(320, 640, True, True),
                                          DownBlock(640, 1280, True, True), DownBlock(1280, 1280, False, False)])
        self.mid_block = MidBlock(1280)

    def encode(self, sample, timestep, ctx, extra=None):
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], device=sample.device)
        temb = self.time_embedding(_time_embedding(t.reshape(-1).expand(sample.shape[0])).to(sample.dtype))
        h = self.conv_in(sample)
        if extra is not None:
            h = h + extra
        res = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, temb, ctx)
            res += outs
        return self.mid_block(h, temb, ctx), res, temb


class ControlNet(_Encoder):
    def __init__(self):
        super().__init__()
        chs = [16, 32, 96, 256]
        convs = [nn.Conv2d(3, 16, 3, padding=1)]
        for a, b in zip(chs[:-1], chs[1:]):
            convs += [nn.Conv2d(a, a, 3, padding=1), nn.Conv2d(a, b, 3, padding=1, stride=2)]
        self.cond_convs = nn.ModuleList(convs)
        self.cond_out = nn.Conv2d(256, 320, 3, padding=1)
        res_ch = [320] * 4 + [640] * 3 + [1280] * 5
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(c, c, 1) for c in res_ch])
        self.controlnet_mid_block = nn.Conv2d(1280, 1280, 1)

    def forward(self, sample, timestep, ctx, cond, scale=1.0):
        c = cond
        for conv in self.cond_convs:
            c = F.silu(conv(c))
        mid, res, _ = self.encode(sample, timestep, ctx, extra=self.cond_out(c))
        return [z(r) * scale for z, r in zip(self.controlnet_down_blocks, res)], self.controlnet_mid_block(mid) * scale


class UNet(_Encoder):
    def __init__(self):
        super().__init__()
        self.up_blocks = nn.ModuleList([UpBlock(1280, 1280, [1280, 1280, 1280], False, True),
                                        UpBlock(1280, 1280, [1280, 1280, 640], True, True),
                                        UpBlock(1280, 640, [640, 640, 320], True, True),
                                        UpBlock(640, 320, [320, 320, 320], True, False)])
        self.conv_norm_out, self.conv_act = nn.GroupNorm(32, 320), nn.SiLU()
        self.conv_out = nn.Conv2d(320, 4, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True):
        h, res, temb = self.encode(sample, timestep, encoder_hidden_states)
        if down_block_additional_residuals is not None:
            res = tuple(r + a for r, a in zip(res, down_block_additional_residuals))
            h = h + mid_block_additional_residual
        for blk in self.up_blocks:
            skips, res = res[-3:], res[:-3]
            h = blk(h, skips, temb, encoder_hidden_states)
        out = self.conv_out(self.conv_act(self.conv_norm_out(h)))
        return (out,)

    def fresco_self_attentions(self):
        """the six decoder self-attentions FRESCO replaces (keys up_blocks.2.* / up_blocks.3.*, attn1), in call order"""
        return [t.transformer_blocks[0].attn1 for i in (2, 3) for t in self.up_blocks[i].attentions]


def reinit_unit_gain(model, seed=0, branch=0.3):
    """Variance-preserving re-initialisation (Next #4): torch's default init (uniform, var = 1 / (3 fan_in))
    plus un-damped residual branches gives this stand-in decoder an input -> output gain of several tens, which SD-1.5's
    trained weights do not have.  Here every Linear / Conv2d weight is N(0, 1 / fan_in) (unit gain on unit-variance input),
    biases are zero, and the LAST layer of every residual branch (ResnetBlock2D.conv2, Attention.to_out[0],
    BasicTransformerBlock.ff_out, Transformer2DModel.proj_out) is scaled by `branch`, so that a block maps unit-variance
    features to variance 1 + branch^2 and a perturbation passes with gain ~ 1.  Deterministic for a seed; weights are drawn
    on the CPU generator and copied, so CPU / GPU construction give the same network."""
    g = torch.Generator().manual_seed(seed)
    last = set()
    for m in model.modules():
        if isinstance(m, ResnetBlock2D):
            last.add(m.conv2)
        elif isinstance(m, Attention):
            last.add(m.to_out[0])
        elif isinstance(m, BasicTransformerBlock):
            last.add(m.ff_out)
        elif isinstance(m, Transformer2DModel):
            last.add(m.proj_out)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                w = m.weight
                fan_in = w[0].numel()
                val = torch.randn(w.shape, generator=g) * (1.0 / math.sqrt(fan_in)) * (branch if m in last else 1.0)
                w.copy_(val.to(w.dtype))
                if m.bias is not None:
                    m.bias.zero_()
    return model
