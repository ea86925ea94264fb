"""optimize_feature with the reference's signature (src/diffusion_hacked.py:416-488), running the
whole Adam loop (analytic gradients, no autograd graph) in libfresco_hip.so."""
import torch

from . import ops
from .warp import _prep_flow_occ, adaptive_instance_normalization


@torch.no_grad()
def optimize_feature(sample, flows, occs, correlation_matrix=[], intra_weight=1e2, iters=20,
                     unet_chunk_size=2, optimize_temporal=True, _workspace=None, shard=None):
    """FRESCO-guided latent feature optimisation.

    sample (2N,C,h,w); flows = [fwd, bwd] (N,2,H,W); occs = [fwd, bwd] (N,H,W);
    correlation_matrix: list of (2N,hw,hw) fp32 Gram targets (matched by hw).
    Returns AdaIN(optimised features cast to sample.dtype, sample)  (diffusion_hacked.py:488).
    shard (extension, not in the reference): a fresco_amd.dist.FrameShard for frame-parallel runs.
    """
    no_temporal = flows is None or occs is None or (not optimize_temporal)
    if no_temporal and (intra_weight == 0 or len(correlation_matrix) == 0):
        return sample
    h, w = sample.shape[2], sample.shape[3]
    prep = None
    if not no_temporal:
        # diffusion_hacked.py:437-442 (no Dilate here, unlike warp_tensor)
        prep = _prep_flow_occ(h, flows, occs, with_dilate=False)
    target = None
    for tmp in correlation_matrix:
        if h * w == tmp.shape[1]:
            target = tmp
            break
    if prep is None and (target is None or not intra_weight > 0):
        # the reference reaches `loss = 0; loss.backward()` here and dies with AttributeError
        raise ValueError("optimize_feature: no loss term is active (no flows and no Gram target of "
                         "%d x %d tokens)" % (h * w, h * w))
    cs = sample.contiguous().to(torch.float32, copy=True)  # (one pass: the cast IS the private copy)
    if shard is not None:
        # frame-parallel form (fresco_amd/dist.py): `sample` and the Gram target hold this rank's frames,
        # flows / occs describe all N pairs; the pairs touching the local frames are selected here
        if prep is not None:
            idx = torch.tensor(shard.pair_index(), device=cs.device)
            prep = tuple(t.index_select(0, idx) for t in prep)
        # (an object with halo_start / halo_finish gets the overlapped neighbour exchange; anything else that offers
        # exchange_halos the blocking form)
        exch = shard if hasattr(shard, "halo_start") else shard.exchange_halos
        ops.opt_run_sharded(cs, prep, target, float(intra_weight), int(iters), unet_chunk_size, shard.N,
                            exch, workspace=_workspace)
    else:
        ops.opt_run(cs, prep, target, float(intra_weight), int(iters), unet_chunk_size, workspace=_workspace)
    return adaptive_instance_normalization(cs.to(sample.dtype), sample)
