"""Frame-parallel FRESCO attention over the GPUs of one node (new design -- the
reference is single-GPU).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank r owns the n_loc =
N / world consecutive frames [r*n_loc, (r+1)*n_loc) of BOTH CFG halves; its local batch axis is
(c, f_loc).  Everything per-frame (projections, spatial-guided pass, the queries of every pass) stays
local.  Two exchange steps per layer, both plain all-gathers into rank-major buffers:

  1. K|V, fused into one (2, B_loc, HW, C) message  ->  (world, 2, B_loc, HW, C).
     The cross-frame kernel reads its compacted key rows straight out of that buffer: the flat
     (frame, pixel) indices of controller.attn_mask are remapped once per batch to buffer rows
     (`remap_kv_rows`), so no re-layout pass runs after the collective.  The gather is launched right
     after the K / V projections and overlaps the Q projection and the spatial-guided pass.
  2. the cross-frame output (the temporal pass's V)  ->  (world, B_loc, HW, C), only while the
     temporal-guided pass is active.  The temporal kernel gathers K from buffer 1 and V from buffer 2
     along the trajectories and writes only the local frames' rows (fresco_temporal_attn_sharded).

Payload per rank and layer at 8 x 512^2 on 8 GPUs: K|V 10.5 MB (up_blocks.3) / 5.2 MB (up_blocks.2);
xGMI is point-to-point (7 links x ~153 GB/s), so a direct all-gather moves each shard over its own
link: ~70 us -- comparable to the sharded kernels, hence the overlap.

optimize_feature: Gram loss, normalisation, Adam and AdaIN are per frame (local); the temporal L1 term
couples frame f with f+-1, so every Adam iteration starts with a halo exchange of the ranks' boundary
frames (`exchange_halos`: one all-gather of 2 frames per rank, 42 MB at the largest layer) and each
rank evaluates the n_loc+1 frame pairs touching its frames (`fresco_opt_sharded_step`).  warp_tensor's
frame chain is a scan over frames and is not sharded (replicas).

The index arithmetic lives in plain functions so that it is testable on CPU (gloo, world_size 2).
"""
import torch
import torch.distributed as dist


def local_batch_index(N, chunk, rank, world):
    """Global (c*N + f) batch indices of this rank's rows, in local (c, f_loc) order."""
    n_loc = N // world
    f0 = rank * n_loc
    return torch.tensor([c * N + f0 + fl for c in range(chunk) for fl in range(n_loc)], dtype=torch.long)


def remap_kv_rows(rows, N, HW, chunk, world):
    """Flat indices into one CFG half's (N*HW) token grid -> rows of the fused K|V gather buffer
    (world, 2, chunk*n_loc, HW) for group 0; group g adds g*group_rows with group_rows = n_loc*HW,
    and V is the same row + chunk*n_loc*HW (passed as a pointer offset)."""
    n_loc = N // world
    B_loc = chunk * n_loc
    rows = rows.to(torch.int64)
    f = rows // HW
    pix = rows - f * HW
    r = f // n_loc
    fl = f - r * n_loc
    out = r * (2 * B_loc * HW) + fl * HW + pix
    return out.to(torch.int32), n_loc * HW


def gathered_batch(g_frame, c, n_loc, rank_stride):
    """Batch index (units of HW rows) of frame g, CFG half c inside a rank-major gather buffer."""
    return (g_frame // n_loc) * rank_stride + c * n_loc + g_frame % n_loc


class FrameShard:
    def __init__(self, N, chunk, rank, world, group=None):
        if N % world != 0:
            raise ValueError("frames (%d) must divide evenly over %d ranks" % (N, world))
        self.N, self.chunk, self.rank, self.world, self.group = N, chunk, rank, world, group
        self.n_loc = N // world
        self.f0 = rank * self.n_loc
        self.B_loc = chunk * self.n_loc
        self._rows_cache = {}

    def local_batch_index(self):
        return local_batch_index(self.N, self.chunk, self.rank, self.world)

    def all_gather(self, x, async_op=False):
        """x (any shape, contiguous) -> (world, *x.shape); returns (buffer, work-or-None)."""
        x = x.contiguous()
        if x.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional testing of the multi-process path without RCCL (e.g. several ranks on one GPU):
            # stage through host memory
            xc = x.cpu()
            outc = xc.new_empty((self.world * xc.shape[0],) + tuple(xc.shape[1:]))
            dist.all_gather_into_tensor(outc, xc, group=self.group)
            return outc.to(x.device).view((self.world,) + tuple(x.shape)), None
        out = x.new_empty((self.world * x.shape[0],) + tuple(x.shape[1:]))  # rank-major concatenation
        work = dist.all_gather_into_tensor(out, x, group=self.group, async_op=async_op)
        return out.view((self.world,) + tuple(x.shape)), work

    def pair_index(self):
        """Global indices of the n_loc+1 frame pairs (f, f+1 mod N) this rank evaluates in the temporal
        term of optimize_feature: pairs f0-1 .. f0+n_loc-1 (ring order)."""
        return [(self.f0 - 1 + j) % self.N for j in range(self.n_loc + 1)]

    def exchange_halos(self, cs):
        """cs: local (chunk*n_loc, C, h, w).  Returns (halo_l, halo_r) = the current frame before / after
        the owned range, (chunk, C, h, w) each, via one all-gather of every rank's first and last frame."""
        x = cs.view(self.chunk, self.n_loc, *cs.shape[1:])
        edges = torch.stack((x[:, 0], x[:, self.n_loc - 1]))  # (2, chunk, C, h, w): first, last
        allb, _ = self.all_gather(edges)
        left = (self.rank - 1) % self.world
        right = (self.rank + 1) % self.world
        return allb[left, 1], allb[right, 0]

    def kv_rows(self, rows, HW, key, device):
        """Remapped int32 key rows (on `device`) + group_rows for the fused K|V gather buffer;
        rows = flat (frame, pixel) indices of the cross-frame mask, or None for "frame 0 only"."""
        hit = self._rows_cache.get(key)
        if hit is None:
            if len(self._rows_cache) > 16:
                self._rows_cache.clear()
            if rows is None:
                rows = torch.arange(HW)
            remapped, group_rows = remap_kv_rows(rows.cpu(), self.N, HW, self.chunk, self.world)
            hit = (remapped.to(device), group_rows)
            self._rows_cache[key] = hit
        return hit
