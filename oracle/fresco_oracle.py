"""CPU restatement of FRESCO's hot path -- TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP kernels in ``fresco_amd/csrc``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package ``fresco_amd`` never does (it fails loudly when the HIP library is missing).

It restates, with explicit index arithmetic on plain torch CPU tensors (matmul / softmax /
index gathers; no autograd, no SDPA, no grid_sample, no einops), the algorithm of these reference
functions (paths relative to /root/reference):

* ``flow_warp`` / ``bilinear_sample`` / ``coords_grid``   src/ebsynth/deps/gmflow/gmflow/geometry.py:5-72
* ``warp_tensor``                                         src/flow_utils.py:18-53
* ``Dilate``, ``calc_mean_std``, ``adaptive_instance_normalization``   src/utils.py:58-93
* ``optimize_feature``                                    src/diffusion_hacked.py:416-488
* ``FRESCOAttnProcessor2_0.__call__``                     src/diffusion_hacked.py:169-387
* ``get_single_mapping_ind`` / ``get_mapping_ind``        src/flow_utils.py:56-138  (integer path)
* cross-frame key mask construction                       src/diffusion_hacked.py:935-938

Parity pinning: the reference ships no tests or golden vectors, so the oracle
is pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (closed-form inputs, no RNG) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every one of them, plus the
known-answer checksums of SURVEY.md Appendix B.

Floating-point type: every function computes in the dtype of its inputs (use float64 inputs for a
high-precision reference, float32 to mirror the reference's CPU path and for the timed CPU
baseline).
"""
import math

import torch

# --------------------------------------------------------------------------------------------
# geometry: bilinear flow warp (geometry.py:41-72)
# --------------------------------------------------------------------------------------------


def _pixel_grid(h, w, dtype, device):
    ys = torch.arange(h, dtype=dtype, device=device).view(h, 1).expand(h, w)
    xs = torch.arange(w, dtype=dtype, device=device).view(1, w).expand(h, w)
    return xs, ys


def sample_coords(flow):
    """Absolute sampling coordinates (ix, iy) the reference ends up using for ``flow_warp``.

    geometry.py:65-72 adds the pixel grid (x in channel 0, y in channel 1), geometry.py:50-51
    normalises ``2*x/(w-1)-1`` and ``grid_sample(align_corners=True)`` maps back with
    ``(g+1)/2*(size-1)``; the round trip is kept so that rounding matches to the last few ulps.
    """
    b, two, h, w = flow.shape
    assert two == 2
    xs, ys = _pixel_grid(h, w, flow.dtype, flow.device)
    x = xs + flow[:, 0]
    y = ys + flow[:, 1]
    gx = 2 * x / (w - 1) - 1
    gy = 2 * y / (h - 1) - 1
    ix = (gx + 1) / 2 * (w - 1)
    iy = (gy + 1) / 2 * (h - 1)
    return ix, iy


def bilinear_taps(flow):
    """4 taps of the zero-padded bilinear sampler: flat indices (b,4,h*w) int64 (clamped) and
    weights (b,4,h*w) already zeroed for out-of-range taps."""
    b, _, h, w = flow.shape
    ix, iy = sample_coords(flow)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    taps_i, taps_w = [], []
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx = x0 + dx
            yy = y0 + dy
            ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
            wgt = torch.where(ok, wx * wy, torch.zeros_like(wx))
            idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).to(torch.int64)
            taps_i.append(idx.reshape(b, h * w))
            taps_w.append(wgt.reshape(b, h * w))
    return torch.stack(taps_i, 1), torch.stack(taps_w, 1)


def flow_warp(feature, flow):
    """out[b,c,y,x] = bilinear(feature[b,c], (x+flow[b,0,y,x], y+flow[b,1,y,x])), zeros outside."""
    b, c, h, w = feature.shape
    assert flow.shape == (b, 2, h, w)
    idx, wgt = bilinear_taps(flow.to(feature.dtype))
    src = feature.reshape(b, c, h * w)
    out = torch.zeros_like(src)
    for t in range(4):
        out = out + torch.gather(src, 2, idx[:, t : t + 1].expand(b, c, h * w)) * wgt[:, t : t + 1]
    return out.reshape(b, c, h, w)


def flow_warp_adjoint(grad_out, flow):
    """Adjoint of ``flow_warp`` w.r.t. the feature: scatter-add of the 4 tap weights."""
    b, c, h, w = grad_out.shape
    idx, wgt = bilinear_taps(flow.to(grad_out.dtype))
    g = grad_out.reshape(b, c, h * w)
    out = torch.zeros_like(g)
    for t in range(4):
        out.scatter_add_(2, idx[:, t : t + 1].expand(b, c, h * w), g * wgt[:, t : t + 1])
    return out.reshape(b, c, h, w)


# --------------------------------------------------------------------------------------------
# resolution helpers used by warp_tensor / optimize_feature
# --------------------------------------------------------------------------------------------


def resize_bilinear(x, scale):
    """F.interpolate(x, scale_factor=scale, mode='bilinear') (align_corners=False, no antialias,
    coordinate scale = 1/scale_factor, output size floor(in*scale)) -- flow_utils.py:26,30,35 and
    diffusion_hacked.py:439,441."""
    b, c, h, w = x.shape
    ho, wo = int(math.floor(h * scale)), int(math.floor(w * scale))
    inv = 1.0 / scale

    def axis(n_out, n_in):
        src = (torch.arange(n_out, dtype=x.dtype, device=x.device) + 0.5) * inv - 0.5
        src = src.clamp(min=0)
        i0 = src.floor().to(torch.int64).clamp(max=n_in - 1)
        i1 = torch.where(i0 < n_in - 1, i0 + 1, i0)
        l1 = (src - i0.to(x.dtype)).clamp(0, 1)
        return i0, i1, 1 - l1, l1

    y0, y1, wy0, wy1 = axis(ho, h)
    x0, x1, wx0, wx1 = axis(wo, w)
    rows0 = x[:, :, y0, :]
    rows1 = x[:, :, y1, :]
    top = rows0[:, :, :, x0] * wx0 + rows0[:, :, :, x1] * wx1
    bot = rows1[:, :, :, x0] * wx0 + rows1[:, :, :, x1] * wx1
    return top * wy0.view(-1, 1) + bot * wy1.view(-1, 1)


def max_pool(x, k):
    """F.max_pool2d(x, kernel_size=k) (stride k, floor) -- diffusion_hacked.py:440,442."""
    b, c, h, w = x.shape
    ho, wo = h // k, w // k
    return x[:, :, : ho * k, : wo * k].reshape(b, c, ho, k, wo, k).amax(dim=(3, 5))


def dilate(x, k):
    """utils.py:81-93: replicate pad (k-1)//2, k x k all-ones box sum, clamp to [0,1]."""
    r = (k - 1) // 2
    b, c, h, w = x.shape
    yi = (torch.arange(-r, h + r)).clamp(0, h - 1)
    xi = (torch.arange(-r, w + r)).clamp(0, w - 1)
    p = x[:, :, yi][:, :, :, xi]
    cs = torch.cumsum(torch.cumsum(p, 2), 3)
    cs = torch.nn.functional.pad(cs, (1, 0, 1, 0))
    s = cs[:, :, k:, k:] - cs[:, :, :-k, k:] - cs[:, :, k:, :-k] + cs[:, :, :-k, :-k]
    return s.clamp(0, 1)


# --------------------------------------------------------------------------------------------
# warp_tensor (flow_utils.py:18-53)
# --------------------------------------------------------------------------------------------


def warp_prepare(h, flows, occs, saliency, dtype):
    """Resize flow / occlusion / saliency to feature height ``h`` (flow_utils.py:24-35)."""
    H = flows[0].shape[2]
    scale = h * 1.0 / H
    kernel = int(1 / scale)
    bwd_flow = resize_bilinear(flows[1].to(dtype) * scale, scale)
    fwd_flow = resize_bilinear(flows[0].to(dtype) * scale, scale)
    bwd_occ = max_pool(occs[1].to(dtype).unsqueeze(1), kernel)
    fwd_occ = max_pool(occs[0].to(dtype).unsqueeze(1), kernel)
    if scale == 1:
        bwd_occ = dilate(bwd_occ, 13)
        fwd_occ = dilate(fwd_occ, 13)
    sal = None
    if saliency is not None:
        scale2 = h * 1.0 / saliency.shape[2]
        sal = resize_bilinear(saliency.to(dtype), scale2)
    return fwd_flow, bwd_flow, fwd_occ, bwd_occ, sal


def warp_tensor(sample, flows, occs, saliency, unet_chunk_size, compute_dtype=torch.float32):
    """flow_utils.py:18-53.  The frame chain is sequential: frame i+1 is blended with the warp
    of the ALREADY UPDATED frame i; the last step warps frame 0 into frame N-1 with the
    wrap-around forward flow."""
    dt = compute_dtype
    fwd_flow, bwd_flow, fwd_occ, bwd_occ, sal = warp_prepare(sample.shape[2], flows, occs, saliency, dt)
    lat = sample.to(dt).clone()
    n = sample.shape[0] // unet_chunk_size
    warp_sal = flow_warp(sal, bwd_flow)
    warp_sal_last = flow_warp(sal[0:1], fwd_flow[n - 1 : n])
    for j in range(unet_chunk_size):
        base = n * j
        for ii in range(n - 1):
            i = base + ii
            warped = flow_warp(lat[i : i + 1], bwd_flow[ii : ii + 1])
            m = (1 - bwd_occ[ii : ii + 1]) * sal[ii + 1 : ii + 2] * warp_sal[ii : ii + 1]
            lat[i + 1 : i + 2] = lat[i + 1 : i + 2] * (1 - m) + warped * m
        ii = n - 1
        warped = flow_warp(lat[base : base + 1], fwd_flow[ii : ii + 1])
        m = (1 - fwd_occ[ii : ii + 1]) * sal[ii : ii + 1] * warp_sal_last
        lat[base + ii : base + ii + 1] = lat[base + ii : base + ii + 1] * (1 - m) + warped * m
    return lat.to(sample.dtype)


# --------------------------------------------------------------------------------------------
# AdaIN (utils.py:58-78)
# --------------------------------------------------------------------------------------------


def adain(content, style, content_eps=1e-5, style_eps=1.0):
    """utils.py:70-78.  NOTE the reference passes ``chunk`` (=1) positionally into ``eps`` for the
    style statistics (utils.py:73 vs :58), so the style std is sqrt(var + 1.0); kept on purpose."""
    b, c = content.shape[:2]
    xc = content.reshape(b, c, -1)
    xs = style.reshape(b, c, -1)
    mu_c = xc.mean(2, keepdim=True)
    mu_s = xs.mean(2, keepdim=True)
    sd_c = (xc.var(2, keepdim=True, unbiased=True) + content_eps).sqrt()
    sd_s = (xs.var(2, keepdim=True, unbiased=True) + style_eps).sqrt()
    return (((xc - mu_c) / sd_c) * sd_s + mu_s).reshape(content.shape)


# --------------------------------------------------------------------------------------------
# optimize_feature (diffusion_hacked.py:416-488), analytic gradients instead of autograd
# --------------------------------------------------------------------------------------------


def opt_prepare(h, flows, occs, chunk, dtype):
    """diffusion_hacked.py:437-442 (no Dilate here, unlike warp_tensor)."""
    H = flows[0].shape[2]
    scale = h * 1.0 / H
    kernel = int(1 / scale)
    bwd_flow = resize_bilinear(flows[1].to(dtype) * scale, scale).repeat(chunk, 1, 1, 1)
    fwd_flow = resize_bilinear(flows[0].to(dtype) * scale, scale).repeat(chunk, 1, 1, 1)
    bwd_occ = max_pool(occs[1].to(dtype).unsqueeze(1), kernel).repeat(chunk, 1, 1, 1)
    fwd_occ = max_pool(occs[0].to(dtype).unsqueeze(1), kernel).repeat(chunk, 1, 1, 1)
    return fwd_flow, bwd_flow, fwd_occ, bwd_occ


def _next_frame(x, n, chunk):
    """x[(b f)] -> x[(b, (f+1) mod n)]  (reshuffle_list, diffusion_hacked.py:444,463)."""
    shp = x.shape
    return torch.roll(x.reshape(chunk, n, *shp[1:]), shifts=-1, dims=1).reshape(shp)


def _prev_frame(x, n, chunk):
    shp = x.shape
    return torch.roll(x.reshape(chunk, n, *shp[1:]), shifts=1, dims=1).reshape(shp)


def opt_loss_and_grad(cs, prep, target, intra_weight, chunk=2, temporal=True):
    """One evaluation of the closure at diffusion_hacked.py:455-484: returns (loss, dL/dcs).

    cs: (B,C,h,w).  prep = (fwd_flow, bwd_flow, fwd_occ, bwd_occ) at feature resolution or None.
    target: (B,hw,hw) Gram target or None.  Gradients per SURVEY.md Appendix A.5.
    """
    B, C, h, w = cs.shape
    n = B // chunk
    loss = cs.new_zeros(())
    grad = torch.zeros_like(cs)
    if temporal and prep is not None:
        fwd_flow, bwd_flow, fwd_occ, bwd_occ = prep
        c1 = cs
        c2 = _next_frame(cs, n, chunk)
        mb = 1 - bwd_occ
        mf = 1 - fwd_occ
        r1 = (c2 - flow_warp(c1, bwd_flow)) * mb
        r2 = (c1 - flow_warp(c2, fwd_flow)) * mf
        cnt = B * C * h * w
        loss = loss + (r1.abs() + r2.abs()).sum() / cnt * 2
        s1 = torch.sign(r1) * mb * (2.0 / cnt)
        s2 = torch.sign(r2) * mf * (2.0 / cnt)
        g1 = s2 - flow_warp_adjoint(s1, bwd_flow)  # d/dc1
        g2 = s1 - flow_warp_adjoint(s2, fwd_flow)  # d/dc2, c2[f] = cs[f+1]
        grad = grad + g1 + _prev_frame(g2, n, chunk)
    if target is not None and intra_weight > 0:
        X = cs.reshape(B, C, h * w).transpose(1, 2)  # (B,hw,C)
        nrm = (X * X).sum(2, keepdim=True).sqrt()
        V = X / nrm
        G = V @ V.transpose(1, 2)
        diff = G - target.to(cs.dtype)
        hw = h * w
        loss = loss + diff.abs().sum() / (B * hw * hw) * intra_weight
        S = torch.sign(diff) * (intra_weight / (B * hw * hw))
        dV = (S + S.transpose(1, 2)) @ V
        dX = (dV - V * (V * dV).sum(2, keepdim=True)) / nrm
        grad = grad + dX.transpose(1, 2).reshape(B, C, h, w)
    return loss, grad


def adam_step(p, g, m, v, t, lr=0.2, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (diffusion_hacked.py:433); t is the 1-based step count."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p.addcdiv_(m, denom, value=-lr / bc1)


def find_target(correlation_matrix, hw):
    for t in correlation_matrix:
        if t.shape[1] == hw:
            return t
    return None


def optimize_feature(sample, flows, occs, correlation_matrix=(), intra_weight=1e2, iters=20,
                     unet_chunk_size=2, optimize_temporal=True, compute_dtype=torch.float32,
                     return_raw=False):
    """diffusion_hacked.py:416-488."""
    no_temporal = flows is None or occs is None or (not optimize_temporal)
    if no_temporal and (intra_weight == 0 or len(correlation_matrix) == 0):
        return sample
    dt = compute_dtype
    B, C, h, w = sample.shape
    cs = sample.to(dt).clone()
    prep = None
    if flows is not None and occs is not None:
        prep = opt_prepare(h, flows, occs, unet_chunk_size, dt)
    target = find_target(correlation_matrix, h * w)
    m = torch.zeros_like(cs)
    v = torch.zeros_like(cs)
    for it in range(iters):
        _, g = opt_loss_and_grad(cs, prep, target, intra_weight, unet_chunk_size, optimize_temporal)
        adam_step(cs, g, m, v, it + 1)
    if return_raw:
        return cs
    # reference casts cs to the sample dtype BEFORE AdaIN (diffusion_hacked.py:488)
    return adain(cs.to(sample.dtype), sample)


def gram_target(feat):
    """diffusion_hacked.py:889-895: cosine Gram of (B,C,h,w) features, fp32 (B,hw,hw)."""
    B, C, h, w = feat.shape
    X = feat.reshape(B, C, h * w).transpose(1, 2)
    V = X / (X * X).sum(2, keepdim=True).sqrt()
    return (V @ V.transpose(1, 2)).to(torch.float32)


# --------------------------------------------------------------------------------------------
# FRESCO attention (diffusion_hacked.py:169-387)
# --------------------------------------------------------------------------------------------


def _heads(x, heads):
    """(B,L,C) -> (B,heads,L,D)"""
    B, L, C = x.shape
    return x.reshape(B, L, heads, C // heads).transpose(1, 2)


def _merge(x):
    B, H, L, D = x.shape
    return x.transpose(1, 2).reshape(B, L, H * D)


# bench.py's cpu_baseline leg sets this: dense attention then goes through torch's fused CPU SDPA --
# the very op the reference calls (diffusion_hacked.py:281,303,357) -- instead of the explicit
# matmul/softmax restatement, so the timed CPU baseline is not handicapped by the restatement.
USE_TORCH_SDPA = False


def dense_attention(q, k, v, scale, diag_bias=0.0, q_block=512):
    """softmax(q k^T * scale + diag_bias*I) v over the last two dims; q (..,Lq,D), k,v (..,Lk,D).
    Query rows are processed in blocks of `q_block` only to bound the size of the logits tensor
    (rows are independent, so the result does not depend on the blocking)."""
    if USE_TORCH_SDPA and diag_bias == 0.0:
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=scale)
    Lq, Lk = q.shape[-2], k.shape[-2]
    kt = k.transpose(-1, -2)
    outs = []
    for r0 in range(0, Lq, q_block):
        r1 = min(r0 + q_block, Lq)
        logits = (q[..., r0:r1, :] @ kt) * scale
        if diag_bias != 0.0 and r0 < Lk:
            i = torch.arange(r0, min(r1, Lk))
            logits[..., i - r0, i] += diag_bias
        outs.append(torch.softmax(logits, dim=-1) @ v)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=-2)


def compact_cross_frame(t, mask, n, chunk):
    """diffusion_hacked.py:234-247: (chunk*n, HW, C) -> (chunk, M, C); rows where mask (n,HW) is
    True, row-major over (frame, pixel).  mask None -> frame 0 only."""
    Bt, HW, C = t.shape
    t4 = t.reshape(chunk, n, HW, C)
    if mask is None:
        return t4[:, 0]
    sel = mask.reshape(-1).nonzero().squeeze(1)
    return t4.reshape(chunk, n * HW, C)[:, sel]


def temporal_attention(q_raw, k_raw, v, fwd_map, tmask, heads, scale, chunk=2):
    """diffusion_hacked.py:309-367 restated per aligned pixel.

    q_raw,k_raw,v: (chunk*n, HW, C); fwd_map (n,HW) int64; tmask (HW,n,n) bool (True = attend).
    Returns (chunk*n, HW, C)."""
    Bt, HW, C = q_raw.shape
    n = Bt // chunk
    D = C // heads
    idx = fwd_map.reshape(1, n, HW, 1).expand(chunk, n, HW, C)
    qa = torch.gather(q_raw.reshape(chunk, n, HW, C), 2, idx)
    ka = torch.gather(k_raw.reshape(chunk, n, HW, C), 2, idx)
    va = torch.gather(v.reshape(chunk, n, HW, C), 2, idx)

    def per_pixel(x):  # (chunk,n,HW,C) -> (chunk,HW,heads,n,D)
        return x.reshape(chunk, n, HW, heads, D).permute(0, 2, 3, 1, 4)

    qa, ka, va = per_pixel(qa), per_pixel(ka), per_pixel(va)
    logits = (qa @ ka.transpose(-1, -2)) * scale  # (chunk,HW,heads,n,n)
    allow = tmask.reshape(1, HW, 1, n, n)
    logits = logits.masked_fill(~allow, float("-inf"))
    oa = torch.softmax(logits, dim=-1) @ va  # (chunk,HW,heads,n,D)
    oa = oa.permute(0, 3, 1, 2, 4).reshape(chunk, n, HW, C)
    out = torch.empty_like(oa)
    out.scatter_(2, idx, oa)  # out[b,f,fwd_map[f,p]] = oa[b,f,p]; fwd_map[f] is a permutation
    return out.reshape(Bt, HW, C)


def fresco_attention(x, Wq, Wk, Wv, Wo, bo, heads, *, ref=None, intra_scale=0.2, intra_bias=0.0,
                     use_cf=False, cf_mask=None, fwd_map=None, tmask=None, inter_scale=0.2,
                     chunk=2, round_dtype=None, return_parts=False):
    """Self-attention branch of FRESCOAttnProcessor2_0.__call__ for SD-1.5 ``attn1`` modules
    (no norms, no residual, rescale 1; diffusion_hacked.py:201-385).

    x: (B,HW,C).  ref: stored hidden states for the spatial-guided pass or None.
    use_cf/cf_mask: efficient cross-frame K/V (mask (n,HW) bool or None = frame 0).
    fwd_map/tmask: temporal-guided pass parameters or None.
    round_dtype: if set (e.g. torch.float16) intermediates that the GPU path materialises in that
    dtype (projections, attention outputs) are rounded through it, to mirror storage rounding.
    """
    def rnd(t):
        return t.to(round_dtype).to(t.dtype) if round_dtype is not None else t

    B, HW, C = x.shape
    D = C // heads
    n = B // chunk
    q = rnd(x @ Wq.t())
    k = rnd(x @ Wk.t())
    v = rnd(x @ Wv.t())
    q_raw, k_raw = q, k
    if use_cf:
        kc = compact_cross_frame(k, cf_mask, n, chunk)  # (chunk,M,C)
        vc = compact_cross_frame(v, cf_mask, n, chunk)
        k_att = kc.unsqueeze(1).expand(chunk, n, *kc.shape[1:]).reshape(B, -1, C)
        v_att = vc.unsqueeze(1).expand(chunk, n, *vc.shape[1:]).reshape(B, -1, C)
    else:
        k_att, v_att = k, v
    q_att = q
    parts = {}
    if ref is not None:
        q_ = rnd(ref @ Wq.t())
        k_ = rnd(ref @ Wk.t())
        q_att = _merge(dense_attention(_heads(q_, heads), _heads(k_, heads), _heads(q, heads),
                                       intra_scale / math.sqrt(D), intra_bias))
        q_att = rnd(q_att)
        parts["intra"] = q_att
    hs = _merge(dense_attention(_heads(q_att, heads), _heads(k_att, heads), _heads(v_att, heads),
                                1.0 / math.sqrt(D)))
    hs = rnd(hs)
    parts["cross"] = hs
    if fwd_map is not None:
        hs = rnd(temporal_attention(q_raw, k_raw, hs, fwd_map, tmask, heads,
                                    inter_scale / math.sqrt(D), chunk))
        parts["temporal"] = hs
    out = hs @ Wo.t()
    if bo is not None:
        out = out + bo
    out = rnd(out)
    if return_parts:
        return out, parts
    return out


def cross_frame_masks(bwd_occs, scales=(8.0, 16.0, 32.0)):
    """diffusion_hacked.py:935-938: row 0 all True, rows 1..N-1 = resized bwd_occs[:-1] > 0.5."""
    out = []
    for s in scales:
        o = resize_bilinear(bwd_occs[:-1].unsqueeze(1).to(torch.float32), 1.0 / s)
        flat = o.reshape(o.shape[0], -1)
        out.append(torch.cat((torch.ones_like(flat[0:1], dtype=torch.bool), flat > 0.5), 0))
    return out


# --------------------------------------------------------------------------------------------
# FLATTEN pixel correspondences (flow_utils.py:56-138) -- integer outputs, must match exactly
# --------------------------------------------------------------------------------------------


def single_mapping_ind(bwd_flow, bwd_occ, imgs, scale=1.0):
    """flow_utils.py:56-103 without the per-pixel Python loop.

    The loop keeps, for every target pixel f1, the source f0 with the smallest colour MSE
    (ties: the earliest f0), among sources that are in range and not occluded; losers are marked
    unused; unlinked targets are filled with the unused sources in ascending order.
    """
    flows = resize_bilinear(bwd_flow, 1.0 / scale)[0][[1, 0]] / scale  # (y,x) order, :72
    _, H, W = flows.shape
    occ = resize_bilinear(bwd_occ[None], 1.0 / scale)
    free = torch.logical_not(occ > 0.5)[0, 0]
    frames = resize_bilinear(imgs, 1.0 / scale).reshape(2, 3, -1)
    gy = torch.arange(H).view(H, 1).expand(H, W)
    gx = torch.arange(W).view(1, W).expand(H, W)
    wy = torch.round(gy + flows[0])
    wx = torch.round(gx + flows[1])
    valid = ((wy >= 0) & (wy < H) & (wx >= 0) & (wx < W) & free).reshape(-1)
    tgt = (wy.reshape(-1) * W + wx.reshape(-1)).to(torch.long)
    hw = H * W
    src = torch.arange(hw)
    vs = src[valid]
    vt = tgt[valid]
    err = ((frames[1][:, vs] - frames[0][:, vt]) ** 2).mean(0)
    # lexicographic winner per target: (target, err, source).  The loop replaces the incumbent only
    # on strictly smaller error, so the earliest source wins ties.
    order = torch.argsort(vs, stable=True)
    order = order[torch.argsort(err[order], stable=True)]
    order = order[torch.argsort(vt[order], stable=True)]
    st = vt[order]
    first = torch.ones_like(st, dtype=torch.bool)
    first[1:] = st[1:] != st[:-1]
    win_t = st[first]
    win_s = vs[order][first]
    mapping = torch.full((hw,), -1, dtype=torch.long)
    mapping[win_t] = win_s
    used = torch.zeros(hw, dtype=torch.bool)
    used[win_s] = True
    unlinked = mapping == -1
    mapping[unlinked] = src[~used]
    return mapping, unlinked


def mapping_ind(bwd_flows, bwd_occs, imgs, scale=1.0):
    """flow_utils.py:106-138."""
    N = imgs.shape[0]
    H, W = int(imgs.shape[2] // scale), int(imgs.shape[3] // scale)
    tmask = torch.ones(H * W, N, N, dtype=torch.bool)
    fwd = [torch.arange(H * W)]
    bwd = [torch.arange(H * W)]
    for i in range(N - 1):
        one = torch.ones(N, N, dtype=torch.bool)
        one[: i + 1, i + 1 :] = False
        one[i + 1 :, : i + 1] = False
        m, unlinked = single_mapping_ind(bwd_flows[i : i + 1], bwd_occs[i : i + 1], imgs[i : i + 2], scale)
        broken = unlinked[fwd[-1]]
        tmask[broken] = torch.logical_and(tmask[broken], one)
        fwd.append(m[fwd[-1]])
        bwd.append(torch.sort(fwd[-1])[1])
    return torch.stack(fwd, 0).unsqueeze(1), torch.stack(bwd, 0).unsqueeze(1), tmask.unsqueeze(1)


# --------------------------------------------------------------------------------------------
# flows -> occlusions -> attention parameters (the part of get_flow_and_interframe_paras after the
# flow network; diffusion_hacked.py:914-957, gmflow/geometry.py:75-96)
# --------------------------------------------------------------------------------------------


def fb_consistency_check(fwd_flow, bwd_flow, alpha=0.01, beta=0.5):
    """geometry.py:75-96: a pixel is occluded when the round trip fwd + bwd(warped) does not close to
    within alpha*(|fwd| + |bwd|) + beta.  Returns (fwd_occ, bwd_occ), float {0,1}, (B,H,W)."""
    mag = fwd_flow.square().sum(1).sqrt() + bwd_flow.square().sum(1).sqrt()
    diff_f = (fwd_flow + flow_warp(bwd_flow, fwd_flow)).square().sum(1).sqrt()
    diff_b = (bwd_flow + flow_warp(fwd_flow, bwd_flow)).square().sum(1).sqrt()
    thr = alpha * mag + beta
    return (diff_f > thr).to(fwd_flow.dtype), (diff_b > thr).to(fwd_flow.dtype)


def flow_occlusions(images, fwd_flows, bwd_flows, color_thr=255 * 0.25):
    """diffusion_hacked.py:919-926.  images (N,3,H,W) in 0..255; flows of the pairs (i, i+1 mod N).
    fb-consistency occlusion OR mean absolute colour difference between a frame and its neighbour
    warped onto it above `color_thr`."""
    N = images.shape[0]
    nxt = list(range(1, N)) + [0]
    fwd_occ, bwd_occ = fb_consistency_check(fwd_flows, bwd_flows)
    w1 = flow_warp(images, bwd_flows)
    bwd_occ = torch.clamp(bwd_occ + ((images[nxt] - w1).abs().mean(1) > color_thr).to(bwd_occ.dtype), 0, 1)
    w2 = flow_warp(images[nxt], fwd_flows)
    fwd_occ = torch.clamp(fwd_occ + ((images - w2).abs().mean(1) > color_thr).to(fwd_occ.dtype), 0, 1)
    return fwd_occ, bwd_occ


def interframe_paras(images, fwd_flows, bwd_flows):
    """Everything get_flow_and_interframe_paras derives from the predicted flows (914-953).
    images (N,3,H,W) float in 0..255.  Returns ([fwd,bwd] flows, [fwd,bwd] occs, attn_mask (3 scales),
    dict(fwd_mappings, bwd_mappings, interattn_masks) at scales 8 and 16)."""
    fwd_occ, bwd_occ = flow_occlusions(images, fwd_flows, bwd_flows)
    imgs_torch = images / 255.0 * 2.0 - 1.0  # utils.py:9
    masks = cross_frame_masks(bwd_occ)
    paras = dict(fwd_mappings=[], bwd_mappings=[], interattn_masks=[])
    for scale in (8.0, 16.0):
        f, b, m = mapping_ind(bwd_flows, bwd_occ, imgs_torch, scale)
        paras["fwd_mappings"].append(f)
        paras["bwd_mappings"].append(b)
        paras["interattn_masks"].append(m)
    return [fwd_flows, bwd_flows], [fwd_occ, bwd_occ], masks, paras
