"""Frame-sharded attention path on ONE GPU: every rank of a world of 2 / 4 is emulated by a thread
whose FrameShard collectives are barrier-synchronised in-process exchanges, so the sharded kernels
(remapped key rows out of the sparse cross-frame exchange buffer, fresco_temporal_pack / _attn_packed /
_unpack around the trajectory all-to-all) and the processor's sharded branch are exercised exactly as under RCCL.  Result must equal the single-GPU
processor's rows for the same global batch."""
import copy
import os
import threading

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


class ThreadShard:
    """Drop-in for fresco_amd.dist.FrameShard whose collectives are barrier-synchronised in-process exchanges."""

    def __init__(self, base, slots, barrier):
        self.__dict__.update(base.__dict__)
        self._base, self._slots, self._barrier = base, slots, barrier
        self.p2p_exchange = False  # (threads have no process group: the broadcast + all-gather form, emulated below)

    def _exchange(self, x):
        self._slots[self.rank] = x
        self._barrier.wait()
        got = [self._slots[r] for r in range(self.world)]
        self._barrier.wait()  # everyone has read the slots before they are reused
        return got

    def cf_plan(self, *a):
        return self._base.cf_plan(*a)

    def _global_rank(self, r):
        return r

    def all_gather(self, x, async_op=False):
        return torch.stack([t.contiguous() for t in self._exchange(x.contiguous())], 0), None

    def all_gather_into(self, out, x, async_op=False):
        out.copy_(torch.cat([t.contiguous() for t in self._exchange(x.contiguous())], 0).view_as(out))
        return None

    def broadcast(self, x, src, async_op=False):
        got = self._exchange(x.clone())
        x.copy_(got[src])
        return None

    def all_to_all(self, x):
        got = self._exchange(x.contiguous())
        return torch.stack([got[s][self.rank] for s in range(self.world)], 0).contiguous()

    def exchange_cf(self, kv_loc, plan):
        from fresco_amd.dist import FrameShard
        return FrameShard.exchange_cf(self, kv_loc, plan)

    # neighbour-only halo exchange of optimize_feature (dist.FrameShard.neighbour_exchange, emulated)
    def neighbour_exchange(self, to_left, to_right, from_left, from_right):
        got = self._exchange((to_left, to_right))
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        from_left.copy_(got[left][1])           # what the left neighbour sent to ITS right
        if to_left is not None:
            from_right.copy_(got[right][0])     # what the right neighbour sent to its left
        return []

    def halo_start(self, cs):
        from fresco_amd.dist import FrameShard
        return FrameShard.halo_start(self, cs)

    def halo_finish(self, handle):
        from fresco_amd.dist import FrameShard
        return FrameShard.halo_finish(self, handle)

    def exchange_halos(self, cs):
        from fresco_amd.dist import FrameShard
        return FrameShard.exchange_halos(self, cs)

    def temporal(self, *a):
        from fresco_amd.dist import FrameShard
        return FrameShard.temporal(self, *a)


# (world, N, which frames keep their occlusion tokens in the cross-frame mask): 8 ranks x 1 frame is the node the
# driver runs (rank 0 owns nothing but frame 0); "f1" leaves the last rank without a single selected token (its slab of
# the all-gather is padding only); "f0" selects frame 0 alone (no all-gather at all, one broadcast)
@pytest.mark.parametrize("world,N,keep", [(2, 4, "all"), (4, 4, "all"), (8, 8, "all"), (2, 4, "f1"), (4, 8, "f0")])
@pytest.mark.parametrize("mode", ["cf", "cf_temporal", "full", "temporal"])
def test_sharded_processor_equals_single_gpu(world, N, keep, mode):
    import fresco_amd
    from fresco_amd.dist import FrameShard

    if keep != "all" and mode == "temporal":
        pytest.skip("the mask variants only touch the cross-frame pass")
    case = synth.make_attention_case(N, 128, "L3", seed=4)
    if keep == "f1":
        case["cf_mask"][2:] = False
    elif keep == "f0":
        case["cf_mask"][1:] = False
    attn = copy.deepcopy(case["attn"]).to(DEV).half()
    hidden = case["hidden"].to(DEV)
    with torch.no_grad():
        ref_out = fresco_amd.FRESCOAttnProcessor2_0(2, synth.controller_for(case, mode, DEV))(attn, hidden)
    torch.cuda.synchronize()

    slots = [None] * world
    barrier = threading.Barrier(world)
    outs, errs = [None] * world, []

    def rank_fn(r):
        try:
            base = FrameShard(N, 2, r, world)
            sel = base.local_batch_index().to(DEV)
            ctrl = synth.controller_for(case, mode, DEV)
            if mode == "full":  # the stored reference features are sharded like the hidden states
                ctrl.stored_attn["decoder_attn"] = [case["ref"].to(DEV).half().index_select(0, sel)]
            proc = fresco_amd.FRESCOAttnProcessor2_0(2, ctrl)
            proc.shard = ThreadShard(base, slots, barrier)
            with torch.no_grad():
                outs[r] = (sel, proc(attn, hidden.index_select(0, sel).contiguous()))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    torch.cuda.synchronize()
    for sel, o in outs:
        d = float((o.float() - ref_out.index_select(0, sel).float()).abs().max())
        # same kernels, same key order: only the projection GEMMs see a different batch size
        assert d < 5e-4, (world, mode, d)


@pytest.mark.parametrize("overlapped", [True, False])
@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_optimize_feature_equals_single_gpu(world, overlapped):
    """frame-sharded optimize_feature (neighbour-only halo exchange every Adam step) == the single-GPU loop.  Same
    kernels, same summation orders -> the optimised features agree EXACTLY, in the overlapped two-part form (exchange
    under the launches that read no halo frame) and in the blocking undivided form; a rank receives two slabs per
    iteration whatever the world size."""
    import fresco_amd
    from fresco_amd.dist import FrameShard

    N, chunk = 4, 2
    case = synth.make_opt_case(N, 24, 16, 64, seed=11)
    x = case["x"].to(DEV)
    flows = [f.to(DEV) for f in case["flows"]]
    occs = [o.to(DEV) for o in case["occs"]]
    target = case["target"].to(DEV)
    ref = fresco_amd.optimize_feature(x, flows, occs, [target], iters=4)
    torch.cuda.synchronize()

    slots = [None] * world
    barrier = threading.Barrier(world)
    outs, errs = [None] * world, []

    def rank_fn(r):
        try:
            base = FrameShard(N, chunk, r, world)
            sh = ThreadShard(base, slots, barrier)
            sh.pair_index = base.pair_index
            if not overlapped:  # an exchange object without halo_start: the blocking, undivided step
                class Blocking:
                    N = sh.N
                    pair_index = staticmethod(base.pair_index)
                    exchange_halos = staticmethod(sh.exchange_halos)
                shard_arg = Blocking()
            else:
                shard_arg = sh
            sel = base.local_batch_index().to(DEV)
            # every emulated rank needs its own scratch (real ranks are separate processes)
            outs[r] = (sel, fresco_amd.optimize_feature(x.index_select(0, sel).contiguous(), flows, occs,
                                                        [target.index_select(0, sel).contiguous()], iters=4,
                                                        shard=shard_arg, _workspace=fresco_amd.ops.Workspace()),
                       getattr(sh, "halo_bytes_received", 0), getattr(sh, "halo_exchanges", 0))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    torch.cuda.synchronize()
    slab = chunk * x.shape[1] * x.shape[2] * x.shape[3] * 4
    for sel, o, nbytes, nex in outs:
        assert torch.equal(o, ref.index_select(0, sel)), (world, float((o - ref.index_select(0, sel)).abs().max()))
        assert nex == (0 if world == 1 else 4) and nbytes == (0 if world == 1 else 4 * 2 * slab), (world, nex, nbytes, slab)


def test_sharded_warp_tensor_replicas():
    import fresco_amd
    from fresco_amd.dist import FrameShard

    N, chunk, world = 4, 2, 2
    case = synth.make_opt_case(N, 12, 16, 64, seed=12)
    x = case["x"].to(DEV)
    flows = [f.to(DEV) for f in case["flows"]]
    occs = [o.to(DEV) for o in case["occs"]]
    sal = case["sal"].to(DEV)
    ref = fresco_amd.warp_tensor(x, flows, occs, sal, chunk)
    slots, barrier = [None] * world, threading.Barrier(world)
    outs, errs = [None] * world, []

    def rank_fn(r):
        try:
            base = FrameShard(N, chunk, r, world)
            sh = ThreadShard(base, slots, barrier)
            sh.local_batch_index = base.local_batch_index
            sel = base.local_batch_index().to(DEV)
            outs[r] = (sel, fresco_amd.warp_tensor(x.index_select(0, sel).contiguous(), flows, occs, sal, chunk,
                                                   shard=sh))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for sel, o in outs:
        assert torch.equal(o, ref.index_select(0, sel))


def _opt_worker(rank, world, port, N, ret, backend="gloo"):
    """one REAL process per rank (gloo, both on cuda:0: collectives staged through the host): the frame-sharded
    optimize_feature with the overlapped neighbour exchange, against the single-process loop computed by rank 0"""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fresco_amd
        from fresco_amd.dist import FrameShard
        case = synth.make_opt_case(N, 24, 16, 64, seed=11)
        x = case["x"].to(DEV)
        flows = [f.to(DEV) for f in case["flows"]]
        occs = [o.to(DEV) for o in case["occs"]]
        target = case["target"].to(DEV)
        sh = FrameShard(N, 2, rank, world)
        sel = sh.local_batch_index().to(DEV)
        out = fresco_amd.optimize_feature(x.index_select(0, sel).contiguous(), flows, occs,
                                          [target.index_select(0, sel).contiguous()], iters=4, shard=sh)
        torch.cuda.synchronize()
        ref = fresco_amd.optimize_feature(x, flows, occs, [target], iters=4)
        slab = 2 * x.shape[1] * x.shape[2] * x.shape[3] * 4
        single = sh.n_loc == 1 and world == 2
        ok = torch.equal(out, ref.index_select(0, sel))
        ok = ok and sh.halo_exchanges == 4 and sh.halo_bytes_received == 4 * (1 if single else 2) * slab
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [4, 2])
def test_sharded_optimize_feature_two_processes_gloo(N):
    """the same check with REAL processes and a real process group (gloo: the collectives go through host memory), so that
    dist.batch_isend_irecv, the tags of the two messages to one peer and the start / finish protocol run as they will under
    RCCL; N = 2: one frame per rank (one message serves both halos)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_opt_worker, args=(2, port, N, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.skipif(os.environ.get("FRESCO_TEST_RCCL") != "1" or torch.cuda.device_count() < 2,
                    reason="opt-in (FRESCO_TEST_RCCL=1) and needs two GPUs: the build and test boxes have one")
@pytest.mark.parametrize("N", [4, 2])
def test_sharded_optimize_feature_two_processes_rccl(N):
    """the same on RCCL, one GPU per rank: opt-in, for whoever has a multi-GPU node (no builder box has had one)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_opt_worker, args=(2, port, N, ret, "nccl"), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
