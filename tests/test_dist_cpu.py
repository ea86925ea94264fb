"""Frame-sharding host logic on CPU: world_size-2 gloo run of the all-gather + index remapping that
feed the sharded kernels (the kernels themselves are covered on the GPU by test_gpu_sharded.py)."""
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
        from fresco_amd.dist import FrameShard, gathered_batch

        g = torch.Generator().manual_seed(0)
        K = torch.randn(chunk * N, HW, C, generator=g)  # the global tensors every rank can rebuild
        V = torch.randn(chunk * N, HW, C, generator=g)
        mask = torch.rand(N, HW, generator=g) < 0.4
        mask[0] = True
        sh = FrameShard(N, chunk, rank, world)
        sel = sh.local_batch_index()
        k_loc, v_loc = K[sel].contiguous(), V[sel].contiguous()
        kv, work = sh.all_gather(torch.stack((k_loc, v_loc)), async_op=True)
        work.wait()
        flat = kv.view(-1, C)
        B_loc = chunk * sh.n_loc
        # (1) cross-frame key rows: C-ABI addressing g*group_rows + kv_rows[m] into the gather buffer
        rows = mask.reshape(-1).nonzero().squeeze(1)
        remapped, group_rows = sh.kv_rows(rows, HW, "m", "cpu")
        ok = True
        for grp in range(chunk):
            got_k = flat[grp * group_rows + remapped.long()]
            got_v = flat[B_loc * HW:][grp * group_rows + remapped.long()]
            want_k = K.view(chunk, N * HW, C)[grp, rows]
            want_v = V.view(chunk, N * HW, C)[grp, rows]
            ok &= torch.equal(got_k, want_k) and torch.equal(got_v, want_v)
        # frame-0-only fallback
        r0, gr0 = sh.kv_rows(None, HW, "f0", "cpu")
        for grp in range(chunk):
            ok &= torch.equal(flat[grp * gr0 + r0.long()], K.view(chunk, N, HW, C)[grp, 0])
        # (2) temporal pass addressing: frame g of half c inside the fused and the plain buffer
        hs_all, _ = sh.all_gather(v_loc)
        for c in range(chunk):
            for gf in range(N):
                kb = gathered_batch(gf, c, sh.n_loc, 2 * B_loc)
                vb = gathered_batch(gf, c, sh.n_loc, B_loc)
                ok &= torch.equal(kv.view(-1, HW, C)[kb], K[c * N + gf])
                ok &= torch.equal(hs_all.view(-1, HW, C)[vb], V[c * N + gf])
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
