"""GMFlow inference for FRESCO's flow / occlusion producer, in the one configuration the
reference instantiates (run_fresco.py:38-45: feature_channels 128, one scale, upsample 8, one head, swin
attention with 2 x 2 splits, 6 transformer blocks, global matching, global propagation, bidirectional).

Reference: src/ebsynth/deps/gmflow/gmflow/{gmflow,backbone,transformer,matching,position,utils}.py.  The module
tree carries the reference's parameter names, so `load_state_dict` takes the published checkpoint
(gmflow_sintel-0c07dcb3.pth, `checkpoint['model']`) unchanged; everything else is written for this package:

  * every attention -- window self / cross attention, shifted windows, the global correlation softmax, the
    flow propagation -- is `fresco_attn_f32` (fp32 MFMA, csrc/attn32.hip): the flows feed integer decisions,
    fp16 attention operands move them by 0.1 - 2 px;
  * shifted-window attention needs no mask: the reference's additive -100 mask only separates the (at most
    four) regions a rolled window is stitched from, so tokens are grouped by (window, region) once per
    resolution and every group is an ordinary attention problem; equally sized groups share a launch;
  * no roll / split / merge copies, no L x L score or mask tensors;
  * round 5: on the GPU the dense layers run on this package's kernels too (csrc/flownet.hip: implicit-GEMM convolutions,
    linear layers, InstanceNorm / LayerNorm, all fp32-accurate products on the fp16 matrix pipe from (hi, lo) operand
    planes), in NHWC -- which is the transformer's token layout, so nothing is transposed between the encoder and the
    attention layers.  The nn.Module tree stays (it carries the checkpoint's parameter names); its own forward methods --
    PyTorch's convolutions / norms / linears -- are what runs for tensors that are not on the GPU (the CPU architecture
    test with a stubbed attention) and with FRESCO_GMFLOW_LIBRARY_OPS=1 (A/B measurements).

`GMFlow.forward(img0, img1, attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1],
pred_bidir_flow=True)` returns {'flow_preds': [flow]} like the reference, flow (2B, 2, H, W) in pixels.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------------
# backbone (gmflow/backbone.py): 1/8-resolution features
# ------------------------------------------------------------------------------------------------
class ResidualBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.norm1 = nn.InstanceNorm2d(cout)
        self.norm2 = nn.InstanceNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.norm3 = nn.InstanceNorm2d(cout)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class CNNEncoder(nn.Module):
    def __init__(self, output_dim=128):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.norm1 = nn.InstanceNorm2d(64)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, 1), ResidualBlock(64, 64, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, 2), ResidualBlock(96, 96, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, 2), ResidualBlock(128, 128, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)

    def forward(self, x):
        x = F.relu(self.norm1(self.conv1(x)))
        return self.conv2(self.layer3(self.layer2(self.layer1(x))))


# ------------------------------------------------------------------------------------------------
# token groups of the (shifted) window attention
# ------------------------------------------------------------------------------------------------
def window_groups(h, w, splits, shifted, device):
    """Partition of the h*w tokens into the sets that attend to each other (transformer.py:20-44, 48-108).
    Plain windows: splits^2 windows.  Shifted windows: the feature map is rolled by half a window and the
    reference's mask confines attention to tokens of the same pre-roll region, i.e. to the groups
    (window, region) computed here in UN-rolled coordinates.  Returns a list of (idx (G, n) int64) tensors, one
    per group size."""
    wh, ww = h // splits, w // splits
    sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
    y = torch.arange(h).view(h, 1).expand(h, w)
    x = torch.arange(w).view(1, w).expand(h, w)
    yr, xr = (y - sh) % h, (x - sw) % w  # position after torch.roll(shifts=(-sh, -sw))

    def slice_id(c, size, win, shift):  # the reference's three slices: [0, size-win), [size-win, size-shift), rest
        if shift == 0:
            return torch.zeros_like(c)
        return (c >= size - win).long() + (c >= size - shift).long()

    gid = (((yr // wh) * splits + (xr // ww)) * 3 + slice_id(yr, h, wh, sh)) * 3 + slice_id(xr, w, ww, sw)
    gid = gid.reshape(-1)
    by_size = {}
    for g in torch.unique(gid).tolist():
        idx = (gid == g).nonzero().squeeze(1)
        by_size.setdefault(idx.numel(), []).append(idx)
    return [torch.stack(v, 0).to(device) for _, v in sorted(by_size.items())]


def grouped_attention(q, k, v, groups, scale):
    """q, k, v (B, L, C) fp32; attention inside every token group; groups from window_groups()."""
    B, L, C = q.shape
    out = torch.empty_like(q)
    for idx in groups:  # (G, n): G groups of n tokens -> one launch over B*G problems
        G, n = idx.shape
        flat = idx.reshape(-1)
        qs = q[:, flat].reshape(B * G, n, C)
        ks = k[:, flat].reshape(B * G, n, C)
        vs = v[:, flat].reshape(B * G, n, C)
        out[:, flat] = ops.attention_f32(qs, ks, vs, scale).reshape(B, G * n, C)
    return out


# ------------------------------------------------------------------------------------------------
# transformer (gmflow/transformer.py:111-293)
# ------------------------------------------------------------------------------------------------
class TransformerLayer(nn.Module):
    def __init__(self, d_model, no_ffn, ffn_dim_expansion=4):
        super().__init__()
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.norm1 = nn.LayerNorm(d_model)
        self.no_ffn = no_ffn
        if not no_ffn:
            self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model * ffn_dim_expansion, bias=False), nn.GELU(),
                                     nn.Linear(2 * d_model * ffn_dim_expansion, d_model, bias=False))
            self.norm2 = nn.LayerNorm(d_model)

    def forward(self, source, target, groups):
        C = source.shape[-1]
        msg = grouped_attention(self.q_proj(source), self.k_proj(target), self.v_proj(target), groups,
                                1.0 / math.sqrt(C))
        msg = self.norm1(self.merge(msg))
        if not self.no_ffn:
            msg = self.norm2(self.mlp(torch.cat((source, msg), -1)))
        return source + msg


class TransformerBlock(nn.Module):
    def __init__(self, d_model, ffn_dim_expansion=4):
        super().__init__()
        self.self_attn = TransformerLayer(d_model, True, ffn_dim_expansion)
        self.cross_attn_ffn = TransformerLayer(d_model, False, ffn_dim_expansion)

    def forward(self, source, target, groups):
        source = self.self_attn(source, source, groups)
        return self.cross_attn_ffn(source, target, groups)


class FeatureTransformer(nn.Module):
    def __init__(self, num_layers=6, d_model=128, ffn_dim_expansion=4):
        super().__init__()
        self.layers = nn.ModuleList([TransformerBlock(d_model, ffn_dim_expansion) for _ in range(num_layers)])
        self._groups = {}

    def _get_groups(self, h, w, splits, shifted, device):
        key = (h, w, splits, shifted, str(device))
        if key not in self._groups:
            self._groups[key] = window_groups(h, w, splits, shifted, device)
        return self._groups[key]

    def forward(self, feature0, feature1, attn_num_splits):
        b, c, h, w = feature0.shape
        f0 = feature0.flatten(2).transpose(1, 2)
        f1 = feature1.flatten(2).transpose(1, 2)
        # both directions in one batch: (f0 | f1) attends to (f1 | f0)
        x = torch.cat((f0, f1), 0)
        for i, layer in enumerate(self.layers):
            shifted = attn_num_splits > 1 and i % 2 == 1
            groups = self._get_groups(h, w, attn_num_splits, shifted, x.device)
            y = torch.cat((x[b:], x[:b]), 0)
            x = layer(x, y, groups)
        f0, f1 = x[:b], x[b:]
        return (f0.transpose(1, 2).reshape(b, c, h, w).contiguous(),
                f1.transpose(1, 2).reshape(b, c, h, w).contiguous())


class FeatureFlowAttention(nn.Module):
    """flow propagation: softmax(q k^T / sqrt(C)) flow, with the reference's quirk k = k_proj(q_proj(x))
    (transformer.py:345-357) kept -- the checkpoint was trained with it"""

    def __init__(self, in_channels):
        super().__init__()
        self.q_proj = nn.Linear(in_channels, in_channels)
        self.k_proj = nn.Linear(in_channels, in_channels)

    def forward(self, feature0, flow):
        b, c, h, w = feature0.shape
        query = self.q_proj(feature0.flatten(2).transpose(1, 2))
        key = self.k_proj(query)
        value = flow.flatten(2).transpose(1, 2)
        out = ops.attention_f32(query, key, value, 1.0 / math.sqrt(c))
        return out.transpose(1, 2).reshape(b, flow.shape[1], h, w)


# ------------------------------------------------------------------------------------------------
# the dense layers on csrc/flownet.hip (GPU tensors): NHWC rows, (hi, lo) operand planes
# ------------------------------------------------------------------------------------------------
class _Weights:
    """(hi, lo) fp16 planes of a layer's weight as the (N, K) matrix fresco_fn_gemm reads, made once per parameter version.
    Range: the planes hold w * 2^10, so |w| >= 63.5 saturates; the split pass flags that on the device (ops.fn_range_guard)
    and the verdict is kept WITH the cached planes -- `out_of_range` turns True whenever such planes are handed out, and the
    forward that used them is recomputed with library ops (GMFlow.forward).  Staleness: the key is the parameter's version
    counter and address; an update that bypasses the counter (`p.data.copy_()`) needs `invalidate()`."""

    def __init__(self):
        self.cache = {}
        self.out_of_range = False

    def invalidate(self):
        self.cache.clear()
        self.out_of_range = False

    @staticmethod
    def _stamp(p):
        try:
            return (p._version, p.data_ptr())
        except RuntimeError:  # inference tensors track no version: never served from the cache
            return None

    def get(self, p, kind, pad_cin=None):
        key = (id(p), kind)
        hit = self.cache.get(key)
        stamp = self._stamp(p)
        if hit is not None and stamp is not None and hit[0] == stamp:
            self.out_of_range |= hit[2]
            return hit[1]
        w = p.detach().float()
        if kind == "conv":  # (cout, cin, kh, kw) -> (cout, kh, kw, cin [padded]) -> (cout, K)
            w = w.permute(0, 2, 3, 1)
            if pad_cin is not None and pad_cin > w.shape[-1]:
                w = F.pad(w, (0, pad_cin - w.shape[-1]))
            w = w.reshape(w.shape[0], -1)
        with ops.fn_range_guard(w.device) as g:
            if kind == "stem":  # (64, 3, 7, 7) -> the (64, 224) layout of fresco_fn_conv7_rgb
                val = ops.fn_conv7_weight(w)
            else:
                _, val = ops.fn_prep(w.contiguous(), scale=ops.FN_W_SCALE)
        bad = g.tripped()  # (one host sync per parameter version)
        self.out_of_range |= bad
        self.cache[key] = (stamp, val, bad)
        return val

    def get_stacked(self, params):
        """planes of the row-stacked weights of several nn.Linear layers of one input (q | k | v as ONE product): the
        single layers' planes concatenated, cached per version of every member"""
        key = tuple(id(p) for p in params) + ("stack",)
        stamps = tuple(self._stamp(p) for p in params)
        hit = self.cache.get(key)
        if hit is not None and None not in stamps and hit[0] == stamps:
            self.out_of_range |= hit[2]
            return hit[1]
        parts = [self.get(p, "lin") for p in params]
        bad = any(self.cache[(id(p), "lin")][2] for p in params)
        val = (torch.cat([h for h, _ in parts], 0).contiguous(), torch.cat([l for _, l in parts], 0).contiguous())
        self.out_of_range |= bad
        self.cache[key] = (stamps, val, bad)
        return val


def _conv(wts, x_split, conv, n, H, W, stride, act=0, want_f32=True, want_split=False, pad_cin=None, norm=None):
    """Conv2d on the NHWC tensor behind the planes x_split -> rows (n * OH * OW, cout); norm: the InstanceNorm2d module that
    follows (its statistics come out of the convolution's epilogue)"""
    kh = conv.kernel_size[0]
    w = wts.get(conv.weight, "conv", pad_cin)
    K = w[0].shape[1]
    return ops.fn_gemm(x_split, w, conv.out_channels, K, bias=conv.bias, act=act,
                       conv=(n, H, W, kh, kh, stride, conv.padding[0]), want_f32=want_f32, want_split=want_split,
                       instance_norm_eps=None if norm is None else norm.eps)


def _res_block(wts, blk, x, xs, n, H, W):
    """ResidualBlock.forward on NHWC rows: x fp32 (n H W, cin), xs its planes -> (out fp32, planes, OH, OW)"""
    s = blk.conv1.stride[0]
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    c1, _, (m1, r1) = _conv(wts, xs, blk.conv1, n, H, W, s, norm=blk.norm1)
    _, a1 = ops.fn_prep(c1, m1, r1, rows_per_img=OH * OW, relu_a=True)
    c2, _, (m2, r2) = _conv(wts, a1, blk.conv2, n, OH, OW, 1, norm=blk.norm2)
    if blk.downsample is not None:
        c3, _, (m3, r3) = _conv(wts, xs, blk.downsample[0], n, H, W, s, norm=blk.downsample[1])
        x, _ = ops.fn_prep(c3, m3, r3, rows_per_img=OH * OW, want_f32=True, want_split=False)
    out, outs = ops.fn_prep(c2, m2, r2, residual=x, rows_per_img=OH * OW, relu_a=True, relu_b=True, want_f32=True)
    return out, outs, OH, OW


def _backbone_native(bb, wts, x_nchw):
    """CNNEncoder.forward -> tokens (n, h * w, C) fp32 (= NHWC), h, w"""
    n, _, H, W = x_nchw.shape
    c, (m, r) = ops.fn_conv7_rgb(x_nchw.permute(0, 2, 3, 1).contiguous(), wts.get(bb.conv1.weight, "stem"),
                                 instance_norm_eps=bb.norm1.eps)
    H, W = c.shape[1], c.shape[2]
    c = c.view(n * H * W, 64)
    x, xs = ops.fn_prep(c, m, r, rows_per_img=H * W, relu_a=True, want_f32=True)
    for layer in (bb.layer1, bb.layer2, bb.layer3):
        for blk in layer:
            x, xs, H, W = _res_block(wts, blk, x, xs, n, H, W)
    w = wts.get(bb.conv2.weight, "conv")
    feats, _ = ops.fn_gemm(xs, w, bb.conv2.out_channels, w[0].shape[1], bias=bb.conv2.bias)
    return feats.view(n, H * W, -1), H, W


def _group_rows(groups, B, L, device):
    """Row table of the grouped attention for (B, L, C) tokens: position g of the GROUP-MAJOR order -> token row b L + l.
    Size class k (G groups of n tokens) occupies the contiguous positions [off, off + B G n) as (b, group, token): exactly
    the (B G, n, C) problem list fresco_attn_f32 takes.  Returns (table int32 (B L), [(off, G, n), ...])."""
    parts, spans, off = [], [], 0
    for idx in groups:
        G, n = idx.shape
        parts.append((torch.arange(B, device=device).view(B, 1) * L + idx.reshape(1, -1).to(device)).reshape(-1))
        spans.append((off, G, n))
        off += B * G * n
    return torch.cat(parts).to(torch.int32).contiguous(), spans


def _cat_planes(M, C, device):
    """(hi, lo) operand planes as the LEFT halves of two (M, 2C) buffers: a layer's output planes are born where the next
    layer's FFN wants them -- mlp(cat(source, norm1(merge))) fills the right halves and reads the buffers whole (round 6:
    the two copies per layer that built the concatenation are gone)"""
    ch = torch.empty(M, 2 * C, dtype=torch.float16, device=device)
    cl = torch.empty(M, 2 * C, dtype=torch.float16, device=device)
    return ch, cl


def _cat_base(planes, C):
    """the (M, 2C) buffers whose left halves `planes` are (see _cat_planes), or None"""
    bases = []
    for t in planes:
        b = t._base
        if b is None or b.dim() != 2 or tuple(b.shape) != (t.shape[0], 2 * C) or not b.is_contiguous() \
                or b.data_ptr() != t.data_ptr() or t.stride(0) != 2 * C:
            return None
        bases.append(b)
    return tuple(bases)


def _layer_native(layer, wts, src, src_s, tgt_s, grows, tgt_table=None):
    """TransformerLayer.forward on token rows: src (B, L, C) fp32, src_s / tgt_s planes (B L, C) -> (out fp32, planes).
    The window grouping costs nothing: the q / k / v projections READ their rows through the group-major table (their
    outputs are the attention problems, contiguous per size class) and the merge projection WRITES its rows back through
    it -- no gather / scatter passes over the tokens (PyTorch index kernels: 1.5 ms of the forward before)."""
    B, L, C = src.shape
    table, spans = grows
    # tgt_table (cross-attention): the k / v projections read the OTHER image's rows of tgt_s through it -- the swap of the two
    # image groups (transformer.py:279-288) is a row table, not two concatenation kernels per block (round 6)
    ktable = table if tgt_table is None else tgt_table
    lin = lambda p: wts.get(p.weight, "lin")
    # q | k | v of one source (self-attention) as ONE product with three output matrices, k | v of the other image's tokens
    # (cross-attention) as one with two: the operand rows are read once instead of three / two times (round 6)
    fuse = os.environ.get("FRESCO_GMFLOW_FUSE_QKV", "1") != "0"  # (A/B switch)
    if fuse and src_s is tgt_s and tgt_table is None:
        qkv, _ = ops.fn_gemm(src_s, wts.get_stacked((layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight)), 3 * C, C,
                             a_rows=table, out_blocks=3)
        q, k, v = qkv[0], qkv[1], qkv[2]
    elif fuse:
        q, _ = ops.fn_gemm(src_s, lin(layer.q_proj), C, C, a_rows=table)
        kv, _ = ops.fn_gemm(tgt_s, wts.get_stacked((layer.k_proj.weight, layer.v_proj.weight)), 2 * C, C, a_rows=ktable,
                            out_blocks=2)
        k, v = kv[0], kv[1]
    else:
        q, _ = ops.fn_gemm(src_s, lin(layer.q_proj), C, C, a_rows=table)
        k, _ = ops.fn_gemm(tgt_s, lin(layer.k_proj), C, C, a_rows=ktable)
        v, _ = ops.fn_gemm(tgt_s, lin(layer.v_proj), C, C, a_rows=ktable)
    scale = 1.0 / math.sqrt(C)
    outs_ = []
    for off, G, n in spans:
        rows = slice(off, off + B * G * n)
        outs_.append(ops.attention_f32(q[rows].view(B * G, n, C), k[rows].view(B * G, n, C), v[rows].view(B * G, n, C),
                                       scale).view(-1, C))
    msg = outs_[0] if len(outs_) == 1 else torch.cat(outs_, 0)
    _, ms = ops.fn_prep(msg)
    mg = torch.empty(B * L, C, dtype=torch.float32, device=src.device)
    ops.fn_gemm(ms, lin(layer.merge), C, C, out_rows=table, out_f32=mg)
    src2 = src.view(B * L, C)
    dev = src.device
    if layer.no_ffn:
        oh, ol = _cat_planes(B * L, C, dev)
        outs = (oh[:, :C], ol[:, :C])
        out, _ = ops.fn_layernorm(mg, layer.norm1.weight, layer.norm1.bias, residual=src2, eps=layer.norm1.eps,
                                  out_split=outs + (2 * C,))
        return out.view(B, L, C), outs
    # mlp(cat(source, norm1(merge))): the LayerNorm writes its planes into the right half of the concatenated operand, whose
    # left half the source's planes already are when the layer before made them (_cat_planes)
    base = _cat_base(src_s, C)
    if base is not None:
        ch, cl = base
    else:
        ch, cl = _cat_planes(B * L, C, dev)
        ch[:, :C].copy_(src_s[0])
        cl[:, :C].copy_(src_s[1])
    ops.fn_layernorm(mg, layer.norm1.weight, layer.norm1.bias, eps=layer.norm1.eps, want_f32=False,
                     out_split=(ch[:, C:], cl[:, C:], 2 * C))
    w1, w2 = lin(layer.mlp[0]), lin(layer.mlp[2])
    hid = layer.mlp[0].out_features
    _, hs = ops.fn_gemm((ch, cl), w1, hid, 2 * C, act=2, want_f32=False, want_split=True)
    o2, _ = ops.fn_gemm(hs, w2, C, hid)
    oh, ol = _cat_planes(B * L, C, dev)
    outs = (oh[:, :C], ol[:, :C])
    out, _ = ops.fn_layernorm(o2, layer.norm2.weight, layer.norm2.bias, residual=src2, eps=layer.norm2.eps,
                              out_split=outs + (2 * C,))
    return out.view(B, L, C), outs


def _transformer_native(tr, wts, tok, b, h, w, splits):
    """FeatureTransformer.forward on tokens (2b, L, C) (both directions in one batch) -> tokens"""
    x = tok
    _, xs = ops.fn_prep(x.reshape(-1, x.shape[-1]))
    L, C = x.shape[1], x.shape[2]
    B = x.shape[0]
    for i, blk in enumerate(tr.layers):
        shifted = splits > 1 and i % 2 == 1
        key = ("rows", h, w, splits, shifted, B, str(x.device))
        if key not in tr._groups:
            grows = _group_rows(tr._get_groups(h, w, splits, shifted, x.device), B, L, x.device)
            # the same positions in the OTHER image group: row r of swap(t) = cat(t[b L:], t[:b L]) is row (r + b L) mod 2 b L of t
            swapped = ((grows[0].long() + b * L) % (2 * b * L)).to(torch.int32).contiguous()
            tr._groups[key] = (grows, swapped)
        grows, swapped = tr._groups[key]
        xs_before = xs  # the other image's tokens BEFORE this block (transformer.py:279-288), read through `swapped`
        x, xs = _layer_native(blk.self_attn, wts, x, xs, xs, grows)
        x, xs = _layer_native(blk.cross_attn_ffn, wts, x, xs, xs_before, grows, tgt_table=swapped)
    return x, xs


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def sine_position(c, h, w, device, temperature=10000.0):
    """DETR-style normalised sine embedding (gmflow/position.py), (1, c, h, w)"""
    npf = c // 2
    ys = (torch.arange(1, h + 1, dtype=torch.float32, device=device) / (h + 1e-6) * (2 * math.pi)).view(h, 1, 1)
    xs = (torch.arange(1, w + 1, dtype=torch.float32, device=device) / (w + 1e-6) * (2 * math.pi)).view(1, w, 1)
    i = torch.arange(npf, dtype=torch.float32, device=device)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
    py = (ys / dim_t).expand(h, w, npf)
    px = (xs / dim_t).expand(h, w, npf)

    def interleave(p):
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), -1).flatten(-2)

    return torch.cat((interleave(py), interleave(px)), -1).permute(2, 0, 1).unsqueeze(0)


def pixel_grid(h, w, device):
    y, x = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack((x, y), 0).float()  # (2, h, w): channel 0 = x


class GMFlow(nn.Module):
    def __init__(self, num_scales=1, upsample_factor=8, feature_channels=128, attention_type="swin",
                 num_transformer_layers=6, ffn_dim_expansion=4, num_head=1, **kwargs):
        super().__init__()
        if num_scales != 1 or attention_type != "swin" or num_head != 1:
            raise NotImplementedError("fresco_amd.gmflow covers FRESCO's GMFlow configuration: one scale, swin "
                                      "attention, one head (run_fresco.py:38-45)")
        self.feature_channels = feature_channels
        self.upsample_factor = upsample_factor
        self.backbone = CNNEncoder(feature_channels)
        self.transformer = FeatureTransformer(num_transformer_layers, feature_channels, ffn_dim_expansion)
        self.feature_flow_attn = FeatureFlowAttention(feature_channels)
        self.upsampler = nn.Sequential(nn.Conv2d(2 + feature_channels, 256, 3, 1, 1), nn.ReLU(inplace=True),
                                       nn.Conv2d(256, upsample_factor ** 2 * 9, 1, 1, 0))
        self._wts = _Weights()

    def _forward_native(self, x, splits, pred_bidir_flow):
        """the forward on csrc/flownet.hip + attn32.hip: NHWC / token layout end to end (x: normalised (2b, 3, H, W))"""
        wts, K = self._wts, self.upsample_factor
        b = x.shape[0] // 2
        tok, h, w = _backbone_native(self.backbone, wts, x)
        c = tok.shape[-1]
        if h % splits or w % splits:
            raise ValueError("feature map %dx%d is not divisible into %d x %d windows" % (h, w, splits, splits))
        pos = sine_position(c, h // splits, w // splits, x.device).repeat(1, 1, splits, splits)
        tok = tok + pos.flatten(2).transpose(1, 2)
        tok, toks = _transformer_native(self.transformer, wts, tok.contiguous(), b, h, w, splits)
        L = h * w
        t0, t1 = tok[:b], tok[b:]
        grid = pixel_grid(h, w, x.device)
        gtok = grid.flatten(1).t().unsqueeze(0)
        if pred_bidir_flow:
            qs, ks, B2, fs = tok, torch.cat((t1, t0), 0), 2 * b, toks
        else:
            qs, ks, B2, fs = t0, t1, b, (toks[0][:b * L], toks[1][:b * L])
        corr = ops.attention_f32(qs, ks, gtok.expand(B2, -1, -1).contiguous(), 1.0 / math.sqrt(c))   # (B2, L, 2)
        flow_tok = corr - gtok                                                                          # matching.py:30-34
        # flow propagation (transformer.py:353-372; k = k_proj(q_proj(x)) kept)
        fa = self.feature_flow_attn
        query, qsp = ops.fn_gemm(fs, wts.get(fa.q_proj.weight, "lin"), c, c, bias=fa.q_proj.bias, want_split=True)
        key, _ = ops.fn_gemm(qsp, wts.get(fa.k_proj.weight, "lin"), c, c, bias=fa.k_proj.bias)
        flow_tok = ops.attention_f32(query.view(B2, L, c), key.view(B2, L, c), flow_tok.contiguous(), 1.0 / math.sqrt(c))
        # convex upsampling (gmflow.py:75-90): mask head on NHWC rows [flow | feature | zero padding to 160 channels]
        cat = torch.zeros(B2 * L, c + 4, dtype=torch.float32, device=x.device)
        cat[:, :2] = flow_tok.reshape(B2 * L, 2)
        cat[:, 2:2 + c] = qs.reshape(B2 * L, c)
        _, cs = ops.fn_prep(cat, ld=160)
        up0, up2 = self.upsampler[0], self.upsampler[2]
        _, hs = _conv(wts, cs, up0, B2, h, w, 1, act=1, want_f32=False, want_split=True, pad_cin=160)
        w2 = wts.get(up2.weight, "conv")
        mk, _ = ops.fn_gemm(hs, w2, up2.out_channels, w2[0].shape[1], bias=up2.bias)
        if K != 8:
            raise NotImplementedError("fresco_amd.gmflow: upsample_factor 8 only (FRESCO's configuration)")
        return ops.fn_convex_upsample(mk, flow_tok, B2, h, w)

    def upsample_flow(self, flow, feature):
        """convex upsampling (gmflow.py:75-90): every fine pixel is a softmax-weighted mix of its coarse 3x3"""
        K = self.upsample_factor
        b, _, h, w = flow.shape
        mask = self.upsampler(torch.cat((flow, feature), 1)).view(b, 1, 9, K, K, h, w).softmax(2)
        nb = F.unfold(K * flow, (3, 3), padding=1).view(b, 2, 9, 1, 1, h, w)
        up = (mask * nb).sum(2)  # (b, 2, K, K, h, w)
        return up.permute(0, 1, 4, 2, 5, 3).reshape(b, 2, K * h, K * w)

    @torch.no_grad()
    def forward(self, img0, img1, attn_splits_list=None, corr_radius_list=None, prop_radius_list=None,
                pred_bidir_flow=False, **kwargs):
        if list(attn_splits_list) != [attn_splits_list[0]] or corr_radius_list[0] != -1 or prop_radius_list[0] != -1:
            raise NotImplementedError("global matching / global propagation at one scale only")
        splits = attn_splits_list[0]
        dev = img0.device
        mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
        x = torch.cat((img0, img1), 0).float()
        if x.is_cuda and os.environ.get("FRESCO_GMFLOW_LIBRARY_OPS", "0") != "1":
            # the dense layers' operands live as fp16 planes of x * 2^6 (weights: w * 2^10): an activation beyond +-1015 or a
            # weight beyond +-63 would saturate there -- finite, wrong, and these flows feed occlusion thresholds and pixel
            # correspondences.  Every producer kernel flags it on the device; one word is read back per forward and the
            # forward is recomputed with library ops (exact fp32 range) when it is set.
            self._wts.out_of_range = False
            with ops.fn_range_guard(x.device) as guard:
                flow = self._forward_native((x / 255.0 - mean) / std, splits, pred_bidir_flow)
            if not (guard.tripped() or self._wts.out_of_range):
                return {"flow_preds": [flow]}
            import warnings
            warnings.warn("fresco_amd.GMFlow: an activation or weight left the range of the split-fp16 dense layers "
                          "(|activation| < 1015, |weight| < 63); this forward is recomputed with library ops",
                          RuntimeWarning, stacklevel=2)
        feats = self.backbone((x / 255.0 - mean) / std)
        f0, f1 = feats.chunk(2, 0)
        b, c, h, w = f0.shape
        if h % splits or w % splits:
            raise ValueError("feature map %dx%d is not divisible into %d x %d windows" % (h, w, splits, splits))
        # position added per window (utils.py:69-86): the same embedding tiled over the windows
        pos = sine_position(c, h // splits, w // splits, dev).repeat(1, 1, splits, splits)
        f0, f1 = self.transformer(f0 + pos, f1 + pos, splits)
        # global matching (matching.py:7-36): expected coordinate under softmax(f0 f1^T / sqrt(c)) minus own
        grid = pixel_grid(h, w, dev)
        gtok = grid.flatten(1).t().unsqueeze(0)  # (1, hw, 2)
        t0 = f0.flatten(2).transpose(1, 2)
        t1 = f1.flatten(2).transpose(1, 2)
        if pred_bidir_flow:
            qs, ks = torch.cat((t0, t1), 0), torch.cat((t1, t0), 0)
        else:
            qs, ks = t0, t1
        corr = ops.attention_f32(qs, ks, gtok.expand(qs.shape[0], -1, -1).contiguous(), 1.0 / math.sqrt(c))
        flow = corr.transpose(1, 2).reshape(-1, 2, h, w) - grid.unsqueeze(0)
        feature0 = torch.cat((f0, f1), 0) if pred_bidir_flow else f0
        flow = self.feature_flow_attn(feature0, flow)
        return {"flow_preds": [self.upsample_flow(flow, feature0)]}
