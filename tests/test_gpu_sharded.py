"""Frame-sharded attention path on ONE GPU: every rank of a world of 2 / 4 is emulated by a thread
whose FrameShard.all_gather is a barrier-synchronised in-process gather, so the sharded kernels
(remapped key rows out of the fused K|V gather buffer, fresco_temporal_attn_sharded) and the
processor's sharded branch are exercised exactly as under RCCL.  Result must equal the single-GPU
processor's rows for the same global batch."""
import copy
import threading

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


class ThreadShard:
    """Drop-in for fresco_amd.dist.FrameShard with an in-process all_gather."""

    def __init__(self, base, slots, barrier):
        self.__dict__.update(base.__dict__)
        self._base, self._slots, self._barrier = base, slots, barrier

    def kv_rows(self, *a):
        return self._base.kv_rows(*a)

    def all_gather(self, x, async_op=False):
        self._slots[self.rank] = x.contiguous()
        self._barrier.wait()
        out = torch.stack([self._slots[r] for r in range(self.world)], 0)
        self._barrier.wait()  # everyone has read the slots before they are reused
        return out, None


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ["cf", "cf_temporal", "full", "temporal"])
def test_sharded_processor_equals_single_gpu(world, mode):
    import fresco_amd
    from fresco_amd.dist import FrameShard

    N = 4
    case = synth.make_attention_case(N, 128, "L3", seed=4)
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
