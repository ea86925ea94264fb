"""Per-batch FRESCO parameters on the GPU: the FLATTEN pixel correspondences of
src/flow_utils.py:56-138 and the cross-frame key masks of src/diffusion_hacked.py:935-938, with the
reference's signatures and integer-exact results -- without the per-pixel Python loop."""
import torch

from . import _lib, ops


def _resized(bwd_flows, bwd_occs, imgs, scale):
    s = 1.0 / scale
    flow = ops.resize_bilinear(bwd_flows, s)                       # (P,2,H,W); the kernel divides by scale
    occ = ops.resize_bilinear(bwd_occs.unsqueeze(1), s)[:, 0].contiguous()   # (P,H,W)
    frames = ops.resize_bilinear(imgs, s)                          # (N,3,H,W)
    return flow, occ, frames


def get_mapping_ind(bwd_flows, bwd_occs, imgs, scale=1.0):
    """flow_utils.py:106-138.  bwd_flows (N-1,2,H,W), bwd_occs (N-1,H,W), imgs (N,3,H,W) ->
    fwd_mappings (N,1,HW) int64, bwd_mappings (N,1,HW) int64, mask (HW,1,N,N) bool."""
    N = imgs.shape[0]
    P = N - 1
    flow, occ, frames = _resized(bwd_flows[:P], bwd_occs[:P], imgs, scale)
    H, W = frames.shape[2], frames.shape[3]
    hw = H * W
    dev = imgs.device
    fwd = torch.empty(N, hw, dtype=torch.int64, device=dev)
    bwd = torch.empty(N, hw, dtype=torch.int64, device=dev)
    mask = torch.empty(hw, N, N, dtype=torch.uint8, device=dev)
    lib = _lib.load()
    nbytes = lib.fresco_mapping_workspace_bytes(N, H, W)
    ws = ops._default_ws.get(nbytes, dev)
    rc = lib.fresco_mapping_ind(flow.data_ptr(), occ.data_ptr(), frames.data_ptr(), fwd.data_ptr(), bwd.data_ptr(),
                                mask.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, float(scale), ops._stream())
    _lib.check(rc, "fresco_mapping_ind(N=%d,H=%d,W=%d)" % (N, H, W))
    return fwd.unsqueeze(1), bwd.unsqueeze(1), mask.bool().unsqueeze(1)


def get_single_mapping_ind(bwd_flow, bwd_occ, imgs, scale=1.0):
    """flow_utils.py:56-103 for one frame pair: (mapping_ind (HW,) int64, unlinkedmask (HW,) bool)."""
    fwd, _, mask = get_mapping_ind(bwd_flow, bwd_occ, imgs, scale)
    return fwd[1, 0], torch.logical_not(mask[:, 0, 0, 1])


def cross_frame_masks(bwd_occs, scales=(8.0, 16.0, 32.0)):
    """diffusion_hacked.py:935-938: per scale a (N, HW) bool mask, row 0 all True, rows 1.. = resized
    bwd_occs[:-1] > 0.5 (the keys of frame j that frame 0 cannot explain)."""
    out = []
    for s in scales:
        o = ops.resize_bilinear(bwd_occs[:-1].unsqueeze(1), 1.0 / s)
        flat = o.reshape(o.shape[0], -1)
        out.append(torch.cat((torch.ones_like(flat[0:1], dtype=torch.bool), flat > 0.5), 0))
    return out
