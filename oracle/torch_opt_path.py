"""TEST / BENCH INFRASTRUCTURE -- not part of the product.

How the reference's `optimize_feature` (src/diffusion_hacked.py:416-488) drives PyTorch: an fp32 `nn.Parameter`
copy of the features, `torch.optim.Adam(lr=0.2)` stepping a closure that builds the temporal L1 term from two
`F.grid_sample` warps (gmflow/geometry.py:41-72: `coords_grid` rebuilt on the host and copied over on every call)
and the spatial term from a normalised `torch.bmm` Gram matrix against the stored target, `loss.backward()`
through autograd, and AdaIN with the reference's eps quirk (src/utils.py:58-78) at the end.  fresco_oracle.py
computes the same numbers with analytic gradients; this file keeps the reference's op sequence, because that
sequence on the SAME GPU is the baseline bench.py's `cfg3.torch_gpu_baseline` times.

tests/test_oracle_golden.py checks it against the reference-generated golden (Appendix-B KAT 6).
"""
import torch
import torch.nn.functional as F


def _coords_grid(b, h, w):
    # geometry.py:5-21: built on the host each time (the caller moves it to the flow's device)
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], dim=0).float()[None].repeat(b, 1, 1, 1)


def flow_warp(feature, flow):
    # geometry.py:41-72
    b, c, h, w = feature.size()
    grid = _coords_grid(b, h, w).to(flow.device) + flow
    xg = 2 * grid[:, 0] / (w - 1) - 1
    yg = 2 * grid[:, 1] / (h - 1) - 1
    return F.grid_sample(feature, torch.stack([xg, yg], dim=-1), mode="bilinear", padding_mode="zeros",
                         align_corners=True)


def _mean_std(feat, eps=1e-5):
    # utils.py:58-67
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps
    return feat.reshape(n, c, -1).mean(dim=2).view(n, c, 1, 1), var.sqrt().view(n, c, 1, 1)


def adain(content, style):
    # utils.py:70-78: the style statistics are computed with eps = 1 (the positional-argument slip at :73)
    s_mean, s_std = _mean_std(style, 1)
    c_mean, c_std = _mean_std(content)
    return (content - c_mean) / c_std * s_std + s_mean


def optimize_feature(sample, flows, occs, correlation_matrix=(), intra_weight=1e2, iters=20, unet_chunk_size=2,
                     optimize_temporal=True):
    if (flows is None or occs is None or (not optimize_temporal)) and (intra_weight == 0 or len(correlation_matrix) == 0):
        return sample
    if sample.is_cuda:
        torch.cuda.empty_cache()
    n = sample.shape[0] // unet_chunk_size
    B, C, h, w = sample.shape
    cs = torch.nn.Parameter(sample.to(torch.float32).reshape(unet_chunk_size, n, C, h, w).detach().clone())
    optimizer = torch.optim.Adam([cs], lr=0.2)
    if flows is not None and occs is not None:
        scale = h * 1.0 / flows[0].shape[2]
        kernel = int(1 / scale)
        rep = (unet_chunk_size, 1, 1, 1)
        bwd_flow = F.interpolate(flows[1] * scale, scale_factor=scale, mode="bilinear").repeat(*rep)
        bwd_occ = F.max_pool2d(occs[1].unsqueeze(1), kernel_size=kernel).repeat(*rep)
        fwd_flow = F.interpolate(flows[0] * scale, scale_factor=scale, mode="bilinear").repeat(*rep)
        fwd_occ = F.max_pool2d(occs[0].unsqueeze(1), kernel_size=kernel).repeat(*rep)
        nxt = list(range(1, n)) + [0]
    target = None
    for t in correlation_matrix:
        if h * w == t.shape[1]:
            target = t
            break
    done = [0]
    while done[0] < iters:
        def closure():
            optimizer.zero_grad()
            loss = 0
            if optimize_temporal and flows is not None and occs is not None:
                c1 = cs.reshape(B, C, h, w)
                c2 = cs[:, nxt].reshape(B, C, h, w)
                w1 = flow_warp(c1, bwd_flow)
                w2 = flow_warp(c2, fwd_flow)
                loss = (abs((c2 - w1) * (1 - bwd_occ)) + abs((c1 - w2) * (1 - fwd_occ))).mean() * 2
            if target is not None and intra_weight > 0:
                vec = cs.reshape(B, C, h * w).transpose(1, 2)
                vec = vec / ((vec ** 2).sum(dim=2, keepdims=True) ** 0.5)
                gram = torch.bmm(vec, vec.transpose(-1, -2))
                loss = F.l1_loss(gram, target) * intra_weight + loss
            loss.backward()
            done[0] += 1
            return loss
        optimizer.step(closure)
    if sample.is_cuda:
        torch.cuda.empty_cache()
    return adain(cs.data.to(sample.dtype).reshape(B, C, h, w), sample)
