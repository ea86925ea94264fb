"""Frame-parallel FRESCO attention over the GPUs of one node (new design -- the
reference is single-GPU).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank r owns the n_loc =
N / world consecutive frames [r*n_loc, (r+1)*n_loc) of BOTH CFG halves; its local batch axis is
(c, f_loc).  Everything per-frame (projections, spatial-guided pass, the queries of the cross-frame pass)
stays local.  Exchange steps per layer (every byte crosses the fabric once, nothing is replicated that is
not read):

  1. cross-frame keys (`exchange_cf`): the efficient cross-frame pass reads frame 0 (all tokens) and the
     occluded tokens of the other frames (controller.attn_mask, src/diffusion_hacked.py:935-938).  Frame 0's fused
     K|V rows go from their owner to every rank; the other frames' selected rows are compacted per rank, padded
     to the largest rank's count (the mask is replicated, so every rank knows the counts) and sent to every rank --
     as one broadcast + one all-gather (default), or as ONE grouped launch of point-to-point transfers
     (`p2p_exchange = True`; opt-in until it has been validated on RCCL).
     Both land in one (HW + world*Rmax, chunk, 2C) buffer -- the CFG halves side by side in a row, so that frame 0's
     rows of BOTH halves are one contiguous block and every rank's selected rows another: one transfer each per
     peer, all of them in one launch.  The kernel addresses the buffer through the remapped row table of
     `cf_plan` (row t of half c = flat row t*chunk + c) -- no re-layout pass after the collectives.  At 8 x 512^2 on
     8 GPUs a rank receives 10.5 MB + 0.1 MB per up_blocks.3 call instead of the 73 MB of an all-gather of every
     frame's K|V.  The exchange is launched right after the K / V projection and overlaps the spatial pass.
  2. temporal-guided pass (`temporal`): sharded by TRAJECTORY, not by frame.  fresco_temporal_pack gathers the
     local frames' q | k | v rows along the trajectories into per-destination ranges, an ALL-TO-ALL delivers
     to rank r all N frames of its HW/world trajectories, the packed kernel runs on them, and the way back
     mirrors it (all-to-all, fresco_temporal_unpack).  Fabric bytes per rank: 4 x (local rows) x (world-1)/world
     -- a quarter of what all-gathering K and V of every frame costs at 8 ranks -- and the kernel's HBM
     traffic is 1/world of the single-GPU pass.

optimize_feature: Gram loss, normalisation, Adam and AdaIN are per frame (local); the temporal L1 term
couples frame f with f+-1, so every Adam iteration needs the neighbours' boundary frames.  NEIGHBOUR-ONLY exchange
(`halo_start` / `halo_finish`, round 6; rounds 1-5 all-gathered every rank's first and last frame: world x 2 slabs
received to use 2): a rank sends its last frame to the right neighbour and its first frame to the left one -- two
(chunk, C, h, w) slabs out, two in, whatever the world size; one slab each way when n_loc = 1 on two ranks (first frame
= last frame, left neighbour = right neighbour).  The exchange is started asynchronously right after Adam(it - 1), runs
under the launches of step it that read no halo frame (normalise, signs of the interior pairs, Gram, S V:
`fresco_opt_sharded_step_part` part 1) and is waited for only in front of the two boundary pairs' signs + Adam (part 2).
Each rank evaluates the n_loc+1 frame pairs touching its frames.  warp_tensor's frame chain is a scan over frames and is
not sharded (replicas).

The index arithmetic lives in plain functions so that it is testable on CPU (gloo, world_size 2).
"""
import weakref

import torch
import torch.distributed as dist


def local_batch_index(N, chunk, rank, world):
    """Global (c*N + f) batch indices of this rank's rows, in local (c, f_loc) order."""
    n_loc = N // world
    f0 = rank * n_loc
    return torch.tensor([c * N + f0 + fl for c in range(chunk) for fl in range(n_loc)], dtype=torch.long)


def cf_plan(mask, N, HW, world, rank):
    """Index plan of the sparse cross-frame exchange for one feature scale.  mask: (N, HW) bool on any device
    (row 0 all True: the reference builds it so, src/diffusion_hacked.py:937-938), or None for "every frame uses
    frame 0" (former_frame_index, :227).
    Returns a dict:
      table      int32 (M,): row of the m-th key (row-major over (frame, pixel), the reference's order, :239) inside
                 one CFG half's slab of the exchange buffer: frame 0's pixel p -> p; a selected token of frame f >= 1
                 owned by rank r, the i-th such token of that rank -> HW + r*Rmax + i
      group_rows HW + world*Rmax: rows per CFG half's slab
      Rmax       largest per-rank count of selected tokens in frames >= 1
      local_sel  int64 (Rmax,): this rank's selected rows, as indices into its (n_loc*HW) local token grid (padded
                 with 0: the pad rows travel but no table entry points at them)"""
    n_loc = N // world
    if mask is None:
        sel = torch.zeros(N, HW, dtype=torch.bool)
        sel[0] = True
    else:
        sel = mask.detach().to("cpu", torch.bool).reshape(N, HW).clone()
        if not bool(sel[0].all()):
            raise ValueError("fresco_amd: the cross-frame mask must select every token of frame 0")
    flat = sel.reshape(-1).nonzero().squeeze(1)  # (M,) row-major (frame, pixel)
    f = flat // HW
    pix = flat - f * HW
    owner = f // n_loc
    rest = f > 0
    counts = [int(((owner == r) & rest).sum()) for r in range(world)]
    Rmax = max(counts) if counts else 0
    table = pix.clone()
    for r in range(world):
        m = (owner == r) & rest
        table[m] = HW + r * Rmax + torch.arange(int(m.sum()))
    mine = (owner == rank) & rest
    local_sel = torch.zeros(Rmax, dtype=torch.int64)
    local_sel[: int(mine.sum())] = (f[mine] - rank * n_loc) * HW + pix[mine]
    return dict(table=table.to(torch.int32), group_rows=HW + world * Rmax, Rmax=Rmax, local_sel=local_sel,
                M=int(flat.numel()))


class FrameShard:
    def __init__(self, N, chunk, rank, world, group=None):
        if N % world != 0:
            raise ValueError("frames (%d) must divide evenly over %d ranks" % (N, world))
        self.N, self.chunk, self.rank, self.world, self.group = N, chunk, rank, world, group
        self.n_loc = N // world
        self.f0 = rank * self.n_loc
        self.B_loc = chunk * self.n_loc
        self._rows_cache = {}
        # cross-frame exchange form (exchange_cf).  False (default): one broadcast + one all-gather -- the two stock
        # collectives.  True: ONE grouped launch of point-to-point transfers (dist.batch_isend_irecv).  The grouped form
        # has only ever run on gloo / emulated ranks (the build boxes have one GPU), so it stays opt-in until an RCCL
        # run has shown parity and timing: `FRESCO_BENCH_P2P=1 bench.py --gpus N` checks and times BOTH forms side by side.
        self.p2p_exchange = False

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

    # ---- halo frames of optimize_feature's temporal term: neighbour-only, asynchronous ----------------------------
    def neighbour_exchange(self, to_left, to_right, from_left, from_right):
        """Point-to-point ring step: `to_left` goes to rank-1, `to_right` to rank+1 (mod world); `from_left` /
        `from_right` (preallocated, same shape) receive what the left / right neighbour sent here.  With world == 2 both
        neighbours are the same peer: the two messages are told apart by their tags (gloo) and by the order in which
        both sides post them (RCCL: `right-going` first on every rank).  `to_left is None` (n_loc == 1 on two ranks: one
        frame serves both sides) sends and receives ONE message; `from_right` then aliases `from_left`.
        Returns the list of outstanding works (empty for host-staged test backends, which complete inside)."""
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        single = to_left is None
        if self._host_staged(to_right):
            # functional testing without RCCL (several ranks on one GPU, gloo): through host memory, blocking
            tr = to_right.cpu()
            fl = torch.empty_like(tr)
            ops = [dist.P2POp(dist.isend, tr, self._global_rank(right), self.group, 1),
                   dist.P2POp(dist.irecv, fl, self._global_rank(left), self.group, 1)]
            if not single:
                tl = to_left.cpu()
                fr = torch.empty_like(tl)
                ops += [dist.P2POp(dist.isend, tl, self._global_rank(left), self.group, 2),
                        dist.P2POp(dist.irecv, fr, self._global_rank(right), self.group, 2)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            from_left.copy_(fl)
            if not single:
                from_right.copy_(fr)
            return []
        ops = [dist.P2POp(dist.isend, to_right, self._global_rank(right), self.group, 1),
               dist.P2POp(dist.irecv, from_left, self._global_rank(left), self.group, 1)]
        if not single:
            ops += [dist.P2POp(dist.isend, to_left, self._global_rank(left), self.group, 2),
                    dist.P2POp(dist.irecv, from_right, self._global_rank(right), self.group, 2)]
        return [w for w in dist.batch_isend_irecv(ops) if w is not None]

    def halo_start(self, cs):
        """cs: local (chunk*n_loc, C, h, w), as the last Adam launch left it (current stream).  Starts the exchange of
        the boundary frames with the two ring neighbours and returns a handle for `halo_finish`; nothing here waits for
        a peer.  The handle owns packed copies of the frames sent, so `cs` may be read (not written) meanwhile."""
        x = cs.view(self.chunk, self.n_loc, *cs.shape[1:])
        # PRIVATE copies (clone, not .contiguous(): with n_loc == 1 the slice already is contiguous and would alias cs):
        # the transfer may still be reading them when the caller's next launch writes cs
        first = x[:, 0].clone(memory_format=torch.contiguous_format)   # (chunk, C, h, w) -> the left neighbour's halo_r
        if self.world == 1:                                # ring of one rank: its own last / first frame
            return dict(halo_l=x[:, self.n_loc - 1].clone(memory_format=torch.contiguous_format), halo_r=first, works=[],
                        keep=())
        single = self.n_loc == 1 and self.world == 2       # one frame, one peer: a single message serves both sides
        # -> the right neighbour's halo_l
        last = first if self.n_loc == 1 else x[:, self.n_loc - 1].clone(memory_format=torch.contiguous_format)
        halo_l = torch.empty_like(first)
        halo_r = halo_l if single else torch.empty_like(first)
        works = self.neighbour_exchange(None if single else first, last, halo_l, halo_r)
        nb = first.numel() * first.element_size()
        self.halo_bytes_received = getattr(self, "halo_bytes_received", 0) + (nb if single else 2 * nb)
        self.halo_exchanges = getattr(self, "halo_exchanges", 0) + 1
        return dict(halo_l=halo_l, halo_r=halo_r, works=works, keep=(first, last))

    def halo_finish(self, handle):
        """-> (halo_l, halo_r) = the current frame before / after the owned range, (chunk, C, h, w) each; the current
        stream waits for the transfers (no host wait under RCCL)."""
        for w in handle["works"]:
            w.wait()
        return handle["halo_l"], handle["halo_r"]

    def exchange_halos(self, cs):
        """blocking form (start + finish); kept for callers that do not overlap the exchange"""
        return self.halo_finish(self.halo_start(cs))

    def _host_staged(self, x):
        # functional testing of the multi-process path without RCCL (several ranks on one GPU): gloo, via host memory
        return x.is_cuda and dist.get_backend(self.group) == "gloo"

    def broadcast(self, x, src, async_op=False):
        """in place; returns work-or-None"""
        if self._host_staged(x):
            xc = x.cpu()
            dist.broadcast(xc, src=src, group=self.group)
            x.copy_(xc)
            return None
        return dist.broadcast(x, src=src, group=self.group, async_op=async_op)

    def all_gather_into(self, out, x, async_op=False):
        """out (world*rows, ...) contiguous view <- every rank's x (rows, ...); returns work-or-None"""
        if self._host_staged(x):
            oc = out.cpu()
            dist.all_gather_into_tensor(oc, x.contiguous().cpu(), group=self.group)
            out.copy_(oc)
            return None
        return dist.all_gather_into_tensor(out, x.contiguous(), group=self.group, async_op=async_op)

    def all_to_all(self, x):
        """x (world, ...) contiguous: slab d goes to rank d; returns (world, ...) with slab s = what rank s sent here"""
        x = x.contiguous()
        if self._host_staged(x):
            xc = x.cpu()
            oc = torch.empty_like(xc)
            dist.all_to_all_single(oc, xc, group=self.group)
            return oc.to(x.device)
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x, group=self.group)
        return out

    def cf_plan(self, mask, HW, device):
        """cached `cf_plan` of a mask tensor (or None) for this rank, index tensors on `device`.  Adds the flat
        addressing of the exchange buffer: `kv_table` = table * chunk with `kv_group_rows` = 1 (row t of CFG half c
        is flat row t*chunk + c of the (rows, chunk, 2C) buffer).  The cache entry holds a weak reference to the mask:
        the pipeline builds a new mask per keyframe batch and the allocator may hand back an old address."""
        device = torch.device(device)
        key = ("none", HW, str(device)) if mask is None else (mask.data_ptr(), tuple(mask.shape), mask._version, str(device))
        hit = self._rows_cache.get(key)
        if hit is None or (mask is not None and hit["mask_ref"]() is not mask):
            if len(self._rows_cache) > 16:
                self._rows_cache.clear()
            hit = cf_plan(mask, self.N, HW, self.world, self.rank)
            hit["table"] = hit["table"].to(device)
            hit["kv_table"] = (hit["table"] * self.chunk).to(torch.int32)
            hit["kv_group_rows"] = 1
            hit["local_sel"] = hit["local_sel"].to(device)
            hit["mask_ref"] = weakref.ref(mask) if mask is not None else None
            self._rows_cache[key] = hit
        return hit

    def _global_rank(self, group_rank):
        """rank inside `self.group` -> global rank (what dist.broadcast's `src` means)"""
        if self.group is None or not dist.is_initialized():
            return group_rank
        return dist.get_global_rank(self.group, group_rank)

    def exchange_cf(self, kv_loc, plan):
        """kv_loc: this rank's fused K|V rows (chunk*n_loc, HW, 2C).  Fills a (HW + world*Rmax, chunk, 2C) buffer with
        frame 0's rows of both CFG halves (from their owner, rank 0) and every rank's selected rows of its other frames;
        returns (buffer, [works]) -- wait on the works before the kernel reads the buffer through plan["kv_table"] /
        plan["kv_group_rows"].
        Default: one broadcast + one all-gather.  `p2p_exchange = True` (round 4, opt-in): ONE grouped launch of
        point-to-point transfers (dist.batch_isend_irecv: the owner of frame 0 sends its block to every peer, every
        rank sends its selected rows to every peer): xGMI is point-to-point -- the owner drives its 7 links in
        parallel either way -- and a layer call pays ONE collective launch (12 -> 6 per step at 8 x 512^2); a
        host-staged test backend always takes the two-collective form."""
        Bl, HW, C2 = kv_loc.shape
        Rmax = plan["Rmax"]
        buf = torch.empty(HW + self.world * Rmax, self.chunk, C2, dtype=kv_loc.dtype, device=kv_loc.device)
        x = kv_loc.view(self.chunk, self.n_loc, HW, C2)
        if self.rank == 0:  # frame 0 lives on the group's rank 0
            buf[:HW].copy_(x[:, 0].transpose(0, 1))
        mine = None
        if Rmax > 0:
            mine = x.reshape(self.chunk, self.n_loc * HW, C2).index_select(1, plan["local_sel"]).transpose(0, 1).contiguous()
        if self.p2p_exchange and self.world > 1 and not self._host_staged(kv_loc):
            ops = []
            peers = [p for p in range(self.world) if p != self.rank]
            if self.rank == 0:
                ops += [dist.P2POp(dist.isend, buf[:HW], self._global_rank(p), self.group) for p in peers]
            else:
                ops.append(dist.P2POp(dist.irecv, buf[:HW], self._global_rank(0), self.group))
            if Rmax > 0:
                buf[HW + self.rank * Rmax: HW + (self.rank + 1) * Rmax].copy_(mine)
                for p in peers:
                    ops.append(dist.P2POp(dist.isend, mine, self._global_rank(p), self.group))
                    ops.append(dist.P2POp(dist.irecv, buf[HW + p * Rmax: HW + (p + 1) * Rmax], self._global_rank(p), self.group))
            works = dist.batch_isend_irecv(ops) if ops else []
            return buf, [w for w in works if w is not None]
        works = [self.broadcast(buf[:HW], src=self._global_rank(0), async_op=True)]
        if Rmax > 0:
            works.append(self.all_gather_into(buf[HW:], mine, async_op=True))
        return buf, [w for w in works if w is not None]

    def cf_collectives(self, plan, device_tensor=None):
        """collective launches one cross-frame exchange issues in the branch `exchange_cf` actually takes (bench.py reports
        the count per step): 1 for the grouped point-to-point form, else broadcast + (all-gather when any rank has rows)"""
        staged = device_tensor is not None and self._host_staged(device_tensor)
        if self.p2p_exchange and self.world > 1 and not staged:
            return 1
        return 1 + (1 if plan["Rmax"] > 0 else 0)

    def temporal(self, q, k, v, fwd_map, mask, heads, scale):
        """trajectory-sharded temporal-guided pass: q, k, v local (chunk*n_loc, HW, C); returns the local rows"""
        from . import ops
        Bl, HW, C = q.shape
        if HW % self.world != 0:
            raise ValueError("tokens (%d) must divide evenly over %d ranks" % (HW, self.world))
        Pw = HW // self.world
        owner = fwd_map
        fwd_map, mask = ops._prep_maps(fwd_map, mask, self.N, HW)
        ops._check_permutations(owner, fwd_map, HW)
        send = ops.temporal_pack(q, k, v, fwd_map, self.chunk, self.n_loc, self.f0, self.world)
        recv = self.all_to_all(send)  # (src, fl, c, pl, 3C) = (frame, c, pl, 3C)
        outp = ops.temporal_attention_packed(recv.view(self.N, self.chunk, Pw, 3 * C),
                                             mask[self.rank * Pw:(self.rank + 1) * Pw], heads, scale, self.chunk)
        back = self.all_to_all(outp.view(self.world, self.n_loc, self.chunk, Pw, C))
        return ops.temporal_unpack(back, fwd_map, self.chunk, self.n_loc, self.f0, self.world)
