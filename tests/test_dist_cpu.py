"""Frame-sharding host logic on CPU: world_size-2 gloo run of the collectives + index plans that feed the
sharded kernels -- sparse cross-frame exchange (broadcast + all-gather), trajectory all-to-all, halos
(the kernels themselves are covered on the GPU by test_gpu_sharded.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, chunk, HW, C, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fresco_amd.dist import FrameShard

        g = torch.Generator().manual_seed(0)
        K = torch.randn(chunk * N, HW, C, generator=g)  # the global tensors every rank can rebuild
        V = torch.randn(chunk * N, HW, C, generator=g)
        Q = torch.randn(chunk * N, HW, C, generator=g)
        mask = torch.rand(N, HW, generator=g) < 0.4
        mask[0] = True
        sh = FrameShard(N, chunk, rank, world)
        sel = sh.local_batch_index()
        kv_loc = torch.cat((K[sel], V[sel]), dim=-1).contiguous()  # fused K|V rows, the exchange layout
        ok = True
        # (1) sparse cross-frame exchange, in both forms: ONE grouped launch of point-to-point transfers (opt-in) and ONE
        # broadcast of frame 0 (both CFG halves) + ONE all-gather of the other frames' selected rows; C-ABI addressing g*kv_group_rows + kv_table[m] into the flat (rows*chunk, 2C) exchange
        # buffer must give the reference's key rows
        rows = mask.reshape(-1).nonzero().squeeze(1)
        for m_, p2p in ((mask, True), (None, True), (mask, False), (None, False)):
            sh.p2p_exchange = p2p
            plan = sh.cf_plan(m_, HW, "cpu")
            buf, works = sh.exchange_cf(kv_loc, plan)
            for w in works:
                w.wait()
            flat = buf.view(-1, 2 * C)
            want_rows = rows if m_ is not None else torch.arange(HW)
            assert plan["M"] == want_rows.numel()
            for grp in range(chunk):
                got = flat[grp * plan["kv_group_rows"] + plan["kv_table"].long()]
                ok &= torch.equal(got[:, :C], K.view(chunk, N * HW, C)[grp, want_rows])
                ok &= torch.equal(got[:, C:], V.view(chunk, N * HW, C)[grp, want_rows])
            # what crosses the fabric: frame 0 once + the padded selected rows, not every frame
            n_rest = int(mask[1:].sum()) if m_ is not None else 0
            assert buf.shape[0] <= HW + world * max(n_rest, 1) and buf.shape[0] < N * HW
            assert sh.cf_collectives(plan, kv_loc) == (1 if p2p else (2 if plan["Rmax"] > 0 else 1))
        assert FrameShard(N, chunk, rank, world).p2p_exchange is False  # the grouped form is opt-in until validated on RCCL
        sh.p2p_exchange = True
        # a NEW mask object at a recycled address must not hit the cached plan of the old one
        m1 = mask.clone()
        p1 = sh.cf_plan(m1, HW, "cpu")
        addr = m1.data_ptr()
        m1.copy_(~mask | (torch.arange(N).view(N, 1) == 0))   # same storage, new contents: _version differs -> rebuilt
        p2 = sh.cf_plan(m1, HW, "cpu")
        ok &= p2 is not p1 and p2["M"] == int(m1.sum())
        fake = dict(p2)
        fake["mask_ref"] = lambda: None                        # entry whose mask object died: same key, other tensor
        sh._rows_cache[(addr, tuple(m1.shape), m1._version, "cpu")] = fake
        ok &= sh.cf_plan(m1, HW, "cpu") is not fake
        # (2) the all-to-all that turns frame shards into trajectory shards and back (the pack / unpack kernels are
        # emulated with index arithmetic here; the kernels themselves are checked on the GPU)
        fwd = torch.stack([torch.randperm(HW, generator=g) for _ in range(N)])
        Pw = HW // world
        n_loc = sh.n_loc
        send = torch.empty(world, n_loc, chunk, Pw, 3 * C)
        for d in range(world):
            for fl in range(n_loc):
                r = fwd[sh.f0 + fl, d * Pw:(d + 1) * Pw]
                for c in range(chunk):
                    b = c * N + sh.f0 + fl
                    send[d, fl, c] = torch.cat((Q[b, r], K[b, r], V[b, r]), dim=-1)
        recv = sh.all_to_all(send).view(N, chunk, Pw, 3 * C)
        for gf in range(N):
            r = fwd[gf, rank * Pw:(rank + 1) * Pw]
            for c in range(chunk):
                ok &= torch.equal(recv[gf, c], torch.cat((Q[c * N + gf, r], K[c * N + gf, r], V[c * N + gf, r]), -1))
        back = sh.all_to_all(recv[..., :C].contiguous().view(world, n_loc, chunk, Pw, C))  # "result" = the q rows
        out = torch.empty(chunk * n_loc, HW, C)
        for d in range(world):
            for fl in range(n_loc):
                r = fwd[sh.f0 + fl, d * Pw:(d + 1) * Pw]
                for c in range(chunk):
                    out[c * n_loc + fl, r] = back[d, fl, c]
        ok &= torch.equal(out, Q[sel])
        # (3) optimize_feature halos: frame before / after the owned range (ring), and the pair list
        X = torch.randn(chunk * N, 3, 4, 5, generator=g)
        hl, hr = sh.exchange_halos(X[sel].contiguous())
        Xg = X.view(chunk, N, 3, 4, 5)
        ok &= torch.equal(hl, Xg[:, (sh.f0 - 1) % N]) and torch.equal(hr, Xg[:, (sh.f0 + sh.n_loc) % N])
        ok &= sh.pair_index() == [(sh.f0 - 1 + j) % N for j in range(sh.n_loc + 1)]
        # (4) warp_tensor's re-assembly of the global (c, f) batch order from the rank-major gather
        allf, _ = sh.all_gather(X[sel].contiguous())
        full = allf.view(world, chunk, sh.n_loc, 3, 4, 5).transpose(0, 1).reshape(chunk * N, 3, 4, 5)
        ok &= torch.equal(full, X)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _halo_worker(rank, world, port, N, chunk, ret):
    """Neighbour-only halo exchange of optimize_feature's temporal term, in the loop form the sharded optimiser uses:
    start (asynchronous) -> local work -> finish -> boundary work, over several iterations of a stand-in update in which
    every frame mixes with its ring neighbours (what the temporal L1 term does to the dependencies)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fresco_amd.dist import FrameShard

        g = torch.Generator().manual_seed(5)
        C, h, w = 3, 4, 5
        X = torch.randn(chunk * N, C, h, w, generator=g)
        sh = FrameShard(N, chunk, rank, world)
        sel = sh.local_batch_index()
        slab = chunk * C * h * w * 4
        ok = True
        # single-process run of the stand-in loop (every rank can afford it)
        ref = X.clone().view(chunk, N, C, h, w)
        for it in range(3):
            ref = ref + 0.25 * torch.roll(ref, 1, 1) - 0.5 * torch.roll(ref, -1, 1)
        cs = X[sel].contiguous()
        n_loc = sh.n_loc
        for it in range(3):
            before = getattr(sh, "halo_bytes_received", 0)
            hnd = sh.halo_start(cs)
            local = cs.clone()                                   # (stands for the launches that read no halo frame)
            hl, hr = sh.halo_finish(hnd)
            got = sh.halo_bytes_received - before
            single = n_loc == 1 and world == 2
            # what crossed the fabric into this rank: two slabs (one when a single frame serves both sides), not 2 * world
            ok &= got == (slab if single else 2 * slab)
            Xg = None
            x = local.view(chunk, n_loc, C, h, w)
            prev = torch.cat((hl.unsqueeze(1), x[:, :-1]), 1)    # frame f-1 of every local frame
            nxt = torch.cat((x[:, 1:], hr.unsqueeze(1)), 1)      # frame f+1
            cs = (x + 0.25 * prev - 0.5 * nxt).reshape(chunk * n_loc, C, h, w).contiguous()
        ok &= torch.equal(cs, ref.reshape(chunk * N, C, h, w)[sel])  # exactly the single-process result
        ok &= sh.halo_exchanges == 3
        # the blocking form returns the same frames
        hl, hr = sh.exchange_halos(X[sel].contiguous())
        Xg = X.view(chunk, N, C, h, w)
        ok &= torch.equal(hl, Xg[:, (sh.f0 - 1) % N]) and torch.equal(hr, Xg[:, (sh.f0 + sh.n_loc) % N])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 2), (2, 4), (4, 4), (4, 8)])
def test_halo_neighbour_exchange_gloo(world, N):
    """2 slabs received per Adam iteration whatever the world size (1 when n_loc = 1 on two ranks), results equal to the
    single-process loop exactly"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_halo_worker, args=(world, _free_port(), N, 2, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_halo_ring_of_one_rank():
    from fresco_amd.dist import FrameShard

    sh = FrameShard(4, 2, 0, 1)
    X = torch.randn(8, 3, 4, 5)
    hl, hr = sh.exchange_halos(X)
    Xg = X.view(2, 4, 3, 4, 5)
    assert torch.equal(hl, Xg[:, 3]) and torch.equal(hr, Xg[:, 0]) and getattr(sh, "halo_bytes_received", 0) == 0


@pytest.mark.parametrize("N,chunk", [(4, 2), (6, 2)])
def test_frame_shard_gather_and_remap_gloo(N, chunk):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), N, chunk, 24, 16, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_local_batch_index_partition():
    from fresco_amd.dist import local_batch_index

    N, chunk, world = 8, 2, 4
    allidx = torch.cat([local_batch_index(N, chunk, r, world) for r in range(world)])
    assert sorted(allidx.tolist()) == list(range(chunk * N))
    assert local_batch_index(N, chunk, 1, world).tolist() == [2, 3, 10, 11]


def test_cf_plan_tables():
    from fresco_amd.dist import cf_plan

    N, HW, world = 4, 6, 2
    mask = torch.zeros(N, HW, dtype=torch.bool)
    mask[0] = True
    mask[1, [1, 4]] = True   # rank 0 (frames 0, 1)
    mask[2, [0]] = True      # rank 1 (frames 2, 3)
    mask[3, [2, 3, 5]] = True
    p0, p1 = cf_plan(mask, N, HW, world, 0), cf_plan(mask, N, HW, world, 1)
    assert p0["Rmax"] == p1["Rmax"] == 4 and p0["group_rows"] == HW + 2 * 4 and p0["M"] == HW + 6
    assert p0["table"].tolist() == list(range(HW)) + [HW + 0, HW + 1] + [HW + 4 + i for i in range(4)]
    assert torch.equal(p0["table"], p1["table"])
    assert p0["local_sel"].tolist() == [HW + 1, HW + 4, 0, 0]          # frame 1 = local frame 1 of rank 0
    assert p1["local_sel"].tolist() == [0, HW + 2, HW + 3, HW + 5]     # frames 2, 3 = local frames 0, 1 of rank 1
    none = cf_plan(None, N, HW, world, 1)
    assert none["Rmax"] == 0 and none["table"].tolist() == list(range(HW)) and none["group_rows"] == HW
    with pytest.raises(ValueError):
        cf_plan(~mask, N, HW, world, 0)


@pytest.mark.parametrize("world,N", [(2, 8), (4, 8), (8, 8), (8, 32), (4, 16)])
@pytest.mark.parametrize("density", [0.0, 0.05, 0.5, 1.0])
def test_cf_plan_emulated_exchange(world, N, density):
    """The sparse cross-frame exchange at the world sizes the 8-GPU node runs (no process group: the broadcast and the
    all-gather are emulated by writing every rank's contribution where the collectives put it).  Addressing the buffer
    through the plan's table must reproduce the reference's key rows -- the True entries of the mask in row-major
    (frame, pixel) order, src/diffusion_hacked.py:239 -- on every rank, including ranks that contribute no row."""
    from fresco_amd.dist import cf_plan

    HW, C, chunk = 40, 3, 2
    g = torch.Generator().manual_seed(world * 100 + N + int(density * 10))
    mask = torch.rand(N, HW, generator=g) < density
    mask[0] = True
    if density == 0.05:
        mask[N // 2:] = False  # the upper ranks own no selected row at all
        mask[0] = True
    KV = torch.randn(chunk, N, HW, C, generator=g)
    n_loc = N // world
    plans = [cf_plan(mask, N, HW, world, r) for r in range(world)]
    Rmax, rows = plans[0]["Rmax"], plans[0]["group_rows"]
    want = mask.reshape(-1).nonzero().squeeze(1)
    assert all(p["Rmax"] == Rmax and p["group_rows"] == rows and torch.equal(p["table"], plans[0]["table"]) for p in plans)
    assert plans[0]["M"] == want.numel() and rows == HW + world * Rmax
    buf = torch.full((chunk, rows, C), float("nan"))
    buf[:, :HW] = KV[:, 0]                                   # broadcast of frame 0 (owner: rank 0)
    for r, p in enumerate(plans):                            # all-gather: rank r's padded rows land at HW + r*Rmax
        loc = KV[:, r * n_loc:(r + 1) * n_loc].reshape(chunk, n_loc * HW, C)
        assert p["local_sel"].numel() == Rmax and (Rmax == 0 or int(p["local_sel"].max()) < n_loc * HW)
        buf[:, HW + r * Rmax:HW + (r + 1) * Rmax] = loc[:, p["local_sel"]]
    got = buf[:, plans[0]["table"].long()]
    assert torch.equal(got, KV.reshape(chunk, N * HW, C)[:, want])
    assert plans[0]["table"].unique().numel() == want.numel()  # every key row has its own slot
