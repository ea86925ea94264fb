"""TEST / BENCH INFRASTRUCTURE -- not part of the product.

Op-for-op restatement of how the reference's FRESCOAttnProcessor2_0.__call__ drives PyTorch for one
SD-1.5 ``attn1`` call (src/diffusion_hacked.py:201-385), for timing "the reference PyTorch path" on
the SAME GPU as our kernels (bench.py `torch_gpu_baseline`).  fresco_oracle.fresco_attention computes
the same numbers with fewer, leaner ops (no clones, no mask tensors, its own temporal softmax); this
file keeps every tensor op the reference issues, in its order, because that sequence IS the baseline:

  * q/k/v projections (201, 214-215), `query.clone(), key.clone()` while the temporal pass is on (219)
  * cross-frame K/V: boolean-mask row select of the (frame, pixel) grid, `repeat` to every frame,
    materialised (234-247)
  * spatial-guided pass: two more projections, `key_ * scale` materialised, a dense
    `torch.eye(HW, HW) * bias` float mask handed to SDPA as attn_mask (even when bias == 0), then
    `torch.cuda.empty_cache()` (258-288)
  * cross-frame SDPA (303-305)
  * temporal pass: `empty_cache()`, three `(b f) d c -> f (b c) d` relayouts + `torch.gather` along
    pixels with the expanded int64 map, three relayouts to (b*HW, f, C), head split, `key * scale`,
    SDPA over 2*HW problems of length N with the bool mask repeated per CFG half, relayout, gather
    with the inverse map, relayout (311-367)
  * head merge, output projection, `/ rescale_output_factor` (371-385)

tests/test_oracle_golden.py checks it against the reference-generated goldens and the oracle.
"""
import math

import torch
import torch.nn.functional as F


def _empty_cache(t):
    if t.is_cuda:
        torch.cuda.empty_cache()


def processor_call(x, Wq, Wk, Wv, Wo, bo, heads, *, ref=None, intra_scale=0.2, intra_bias=0.0,
                   use_cf=False, cf_mask=None, fwd_map=None, bwd_map=None, tmask=None,
                   inter_scale=0.2, chunk=2):
    """x (B,HW,C); ref: stored hidden states or None; cf_mask (n,HW) bool or None; fwd_map/bwd_map
    (n,HW) int64 (bwd_map = inverse permutation, derived when omitted); tmask (HW,n,n) bool."""
    B, HW, C = x.shape
    D = C // heads
    n = B // chunk
    query = F.linear(x, Wq)                                        # :201
    key = F.linear(x, Wk)                                          # :214
    value = F.linear(x, Wv)                                        # :215
    temporal = fwd_map is not None
    if temporal:
        query_raw, key_raw = query.clone(), key.clone()            # :219
    if use_cf:                                                     # :225-247
        def select(t):
            t = t.reshape(chunk, n, HW, C)
            if cf_mask is None:
                t = t[:, [0] * n]
            else:
                t = t[:, cf_mask]                                  # (chunk, M, C)
                t = t.unsqueeze(1).repeat(1, n, 1, 1)              # einops repeat materialises
            return t.reshape(chunk * n, -1, C).detach()
        key, value = select(key), select(value)
    query = query.view(B, -1, heads, D).transpose(1, 2)            # :250-254
    key = key.view(B, -1, heads, D).transpose(1, 2)
    value = value.view(B, -1, heads, D).transpose(1, 2)
    if ref is not None:                                            # :256-288
        query_ = F.linear(ref, Wq).view(B, -1, heads, D).transpose(1, 2)
        key_ = F.linear(ref, Wk).view(B, -1, heads, D).transpose(1, 2)
        eye = torch.eye(query_.size(-2), key_.size(-2), dtype=query.dtype, device=query.device) * intra_bias
        query = F.scaled_dot_product_attention(query_, key_ * intra_scale, query, attn_mask=eye).detach()
        del query_, key_
        _empty_cache(x)
    hidden = F.scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False)
    if temporal:                                                   # :309-367
        del query, key, value
        _empty_cache(x)
        if bwd_map is None:
            bwd_map = torch.argsort(fwd_map, dim=1)
        fmap = fwd_map.view(n, 1, HW)
        bmap = bwd_map.view(n, 1, HW)

        def align(t):  # (b f) d c -> f (b c) d, gather along pixels, -> (b d) f c
            t = t.reshape(chunk, n, HW, C).permute(1, 0, 3, 2).reshape(n, chunk * C, HW)
            t = torch.gather(t, 2, fmap.expand(-1, t.shape[1], -1))
            return t.reshape(n, chunk, C, HW).permute(1, 3, 0, 2).reshape(chunk * HW, n, C)

        key = align(key_raw)
        query = align(query_raw)
        # (b f) h d c -> f (b h c) d
        value = hidden.reshape(chunk, n, heads, HW, D).permute(1, 0, 2, 4, 3).reshape(n, chunk * C, HW)
        value = torch.gather(value, 2, fmap.expand(-1, value.shape[1], -1))
        value = value.reshape(n, chunk, C, HW).permute(1, 3, 0, 2).reshape(chunk * HW, n, C)
        query = query.view(-1, n, heads, D).transpose(1, 2).detach()
        key = key.view(-1, n, heads, D).transpose(1, 2).detach()
        value = value.view(-1, n, heads, D).transpose(1, 2).detach()
        mask = tmask.view(HW, 1, n, n).repeat(chunk, 1, 1, 1)
        hidden_ = F.scaled_dot_product_attention(query, key * inter_scale, value, attn_mask=mask)
        # (b d) h f c -> f (b h c) d
        hidden_ = hidden_.reshape(chunk, HW, heads, n, D).permute(3, 0, 2, 4, 1).reshape(n, chunk * C, HW)
        hidden_ = torch.gather(hidden_, 2, bmap.expand(-1, hidden_.shape[1], -1)).detach()
        # f (b h c) d -> (b f) h d c
        hidden = hidden_.reshape(n, chunk, heads, D, HW).permute(1, 0, 2, 4, 3).reshape(B, heads, HW, D)
    hidden = hidden.transpose(1, 2).reshape(B, -1, C)              # :371
    hidden = hidden.to(x.dtype)
    out = F.linear(hidden, Wo, bo)                                 # :375
    return out / 1.0                                               # :385
