"""flow_warp / warp_tensor / AdaIN / Dilate with the reference's signatures, on the HIP kernels.

Reference: src/ebsynth/deps/gmflow/gmflow/geometry.py:41-72 (flow_warp), src/flow_utils.py:18-53
(warp_tensor), src/utils.py:58-93 (calc_mean_std, adaptive_instance_normalization, Dilate).
"""
import torch

from . import ops


def flow_warp(feature, flow, mask=False, padding_mode="zeros"):
    """geometry.py:65-72.  Only the configuration FRESCO uses: zeros padding, no validity mask."""
    if mask or padding_mode != "zeros":
        raise NotImplementedError("fresco_amd.flow_warp: mask=False, padding_mode='zeros' only")
    assert flow.size(1) == 2
    return ops.flow_warp(feature, flow).to(feature.dtype)


# flows / occs / saliency are constants of a batch of frames, but the hook calls optimize_feature + warp_tensor at 4
# layers x 15 steps with them: the derived tensors (resized flows, pooled occlusions, resized + warped saliency) are kept
# per (tensor identities, feature height).  An entry holds weak references to its source tensors and their versions: a
# new tensor at a recycled address, or an in-place update, misses.
_derived = {}


def _version_of(t):
    """autograd version counter, or None when the tensor does not track one (torch.inference_mode tensors raise)"""
    try:
        return t._version
    except RuntimeError:
        return None


def _cached(kind, srcs, extra, build):
    import weakref
    vers = [_version_of(t) for t in srcs]
    if any(v is None for v in vers):
        return build()  # inference tensors: an in-place update would be invisible -> never cached
    key = (kind, extra) + tuple((t.data_ptr(), tuple(t.shape), v, str(t.device)) for t, v in zip(srcs, vers))
    hit = _derived.get(key)
    if hit is not None and all(r() is t for r, t in zip(hit[0], srcs)):
        return hit[1]
    # a miss: entries whose source tensors have died release their GPU tensors now, not when 33 have piled up
    for k in [k for k, (refs, _) in _derived.items() if any(r() is None for r in refs)]:
        del _derived[k]
    if len(_derived) > 32:
        _derived.clear()
    val = build()
    _derived[key] = ([weakref.ref(t) for t in srcs], val)
    return val


def invalidate_cache():
    """drop every cached derived tensor (resized flows, pooled occlusions, warped saliency).  Needed only after a
    `t.data.copy_()`-style update of a flow / occlusion / saliency tensor, which does not bump its version counter."""
    _derived.clear()


def _prep_flow_occ(h, flows, occs, with_dilate):
    """flow_utils.py:24-31 / diffusion_hacked.py:437-442: flows, occs at feature height h (cached per batch of frames:
    the results are read-only for every consumer)."""
    return _cached("flow_occ", (flows[0], flows[1], occs[0], occs[1]), (int(h), bool(with_dilate)),
                   lambda: _prep_flow_occ_build(h, flows, occs, with_dilate))


def _prep_flow_occ_build(h, flows, occs, with_dilate):
    H = flows[0].shape[2]
    scale = h * 1.0 / H
    kernel = int(1 / scale)
    out = []
    for i in (0, 1):
        fl = ops.resize_bilinear(flows[i], scale, mul=scale)
        oc = ops.max_pool(occs[i].unsqueeze(1), kernel)
        if with_dilate and scale == 1:
            oc = ops.dilate(oc, 13)
        out.append((fl, oc))
    (fwd_flow, fwd_occ), (bwd_flow, bwd_occ) = out
    if fwd_flow.shape[2] != h:
        raise ValueError("flow of height %d does not resize to feature height %d" % (H, h))
    return fwd_flow, bwd_flow, fwd_occ, bwd_occ


@torch.no_grad()
def warp_tensor(sample, flows, occs, saliency, unet_chunk_size, shard=None):
    """flow_utils.py:18-53: warp frame i into frame i+1 along the chain and blend by
    (1-occ) * saliency * warped saliency; the last step warps frame 0 into frame N-1.

    shard (extension): a fresco_amd.dist.FrameShard.  The chain is a scan over frames, so it does not
    shard: the ranks all-gather their frames, every rank runs the whole chain (cheap, HBM-bound) and
    keeps its own frames ("replicas only" for this op)."""
    if shard is not None:
        allf, _ = shard.all_gather(sample.contiguous())  # (world, chunk*n_loc, C, h, w)
        full = allf.view(shard.world, shard.chunk, shard.n_loc, *sample.shape[1:]).transpose(0, 1)
        full = full.reshape(shard.chunk * shard.N, *sample.shape[1:])
        out = warp_tensor(full, flows, occs, saliency, unet_chunk_size)
        return out.index_select(0, shard.local_batch_index().to(out.device)).contiguous()
    h = sample.shape[2]
    fwd_flow, bwd_flow, fwd_occ, bwd_occ = _prep_flow_occ(h, flows, occs, with_dilate=True)
    n = sample.shape[0] // unet_chunk_size

    def sal_terms():
        scale2 = h * 1.0 / saliency.shape[2]
        sal = ops.resize_bilinear(saliency, scale2)
        return sal, ops.flow_warp(sal, bwd_flow), ops.flow_warp(sal[0:1], fwd_flow[n - 1:n])

    sal, warp_sal, warp_sal_last = _cached("saliency", (saliency, flows[0], flows[1], occs[0], occs[1]), (int(h), int(n)),
                                           sal_terms)
    lat = sample.contiguous().to(torch.float32, copy=True)  # (one pass: the cast IS the private copy)
    ops.warp_fuse_chain(lat, bwd_flow, fwd_flow, bwd_occ, fwd_occ, sal, warp_sal, warp_sal_last,
                        unet_chunk_size)
    return lat.to(sample.dtype)


def calc_mean_std(feat, eps=1e-5, chunk=1):
    """utils.py:58-67: per-(sample, channel) mean and sqrt(unbiased variance + eps) over the plane, both (N, C, 1, 1).
    Runs on adain_kernel's reduction: fresco_chan_mean_std.  chunk = 1 only (all the pipeline ever passes: AdaIN's
    `chunk` lands in `eps`, utils.py:73): the reference's chunk = 2 branch concatenates the two CFG halves along the
    width, takes joint statistics over N // 2 rows and then fails at its own `.view(N, C, 1, 1)` (utils.py:61-65)."""
    size = feat.size()
    assert len(size) == 4
    if chunk != 1:
        raise NotImplementedError("fresco_amd.calc_mean_std: chunk = 1 only (the reference's chunk = 2 branch raises at its "
                                  "own view, src/utils.py:61-65)")
    mean, std = ops.chan_mean_std(feat, float(eps))
    return mean.view(size[0], size[1], 1, 1).to(feat.dtype), std.view(size[0], size[1], 1, 1).to(feat.dtype)


def adaptive_instance_normalization(content_feat, style_feat, chunk=1):
    """utils.py:70-78.  The reference passes `chunk` positionally into calc_mean_std's `eps` for the
    style statistics (utils.py:73 vs :58), i.e. style std = sqrt(var + chunk); reproduced here."""
    assert content_feat.size()[:2] == style_feat.size()[:2]
    return ops.adain(content_feat, style_feat, eps_content=1e-5, eps_style=float(chunk))


class Dilate:
    """utils.py:81-93: replicate-pad + k x k box sum + clamp to [0, 1]."""

    def __init__(self, kernel_size=7, channels=1, device="cpu"):
        if channels != 1:
            raise NotImplementedError("fresco_amd.Dilate: channels=1 only (all the reference uses)")
        self.kernel_size = kernel_size
        self.channels = channels
        self.mean = (kernel_size - 1) // 2

    def __call__(self, x):
        return ops.dilate(x, self.kernel_size).to(x.dtype)
