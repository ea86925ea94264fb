"""CPU-side checks: the C-ABI library loads and exports every declared symbol, host-only queries,
loud failure without a GPU, the controller state machine against the reference's recorded trace, and
the forward-hook logic on a stand-in UNet (no kernels are launched here)."""
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    from fresco_amd import _lib
    header = open(os.path.join(ROOT, "include", "fresco_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fresco_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.fresco_version().decode().startswith("fresco_hip")


def test_workspace_queries_are_host_only():
    from fresco_amd import _lib
    lib = _lib.load()
    # cfg2 up_blocks.3 cross-frame: 2 groups x 8 heads x (67 tiles + 1: the V^T half of a pack lags its K half by one
    # tile) packs of 64 keys x (48 + 64) halfs (K | V^T fragment images: head dim padded to 48 for the QK
    # contraction, to 64 rows for the PV product), plus one fp32 max|k|^2 per tile; each block rounded up to 256 B
    r256 = lambda n: (n + 255) // 256 * 256
    assert lib.fresco_attn_workspace_bytes(2, 8, 4237, 40) == r256(2 * 8 * 68 * 64 * 112 * 2) + r256(2 * 8 * 67 * 4)
    assert lib.fresco_attn_workspace_bytes(0, 8, 10, 40) == 0
    full = lib.fresco_opt_workspace_bytes(2, 8, 640, 64, 64, 1, 1)
    assert full > lib.fresco_opt_workspace_bytes(2, 8, 640, 64, 64, 1, 0) > lib.fresco_opt_workspace_bytes(2, 8, 640, 64, 64, 0, 0)
    assert full >= 16 * 4096 * 4096 + 5 * 16 * 640 * 4096 * 4


def test_operators_refuse_cpu_tensors():
    import fresco_amd
    x = torch.zeros(2, 3, 4, 4)
    with pytest.raises(fresco_amd.FrescoHipError):
        fresco_amd.flow_warp(x, torch.zeros(2, 2, 4, 4))
    with pytest.raises(fresco_amd.FrescoHipError):
        fresco_amd.adaptive_instance_normalization(x, x)
    with pytest.raises(fresco_amd.FrescoHipError):
        fresco_amd.optimize_feature(x, [torch.zeros(1, 2, 8, 8)] * 2, [torch.zeros(1, 8, 8)] * 2, [], iters=1)
    # the early-out contract needs no GPU (diffusion_hacked.py:423-424)
    assert fresco_amd.optimize_feature(x, None, None, [], iters=1) is x


def test_missing_library_fails_loudly(monkeypatch):
    from fresco_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfresco_hip.so")
    with pytest.raises(_lib.FrescoHipError):
        _lib.load()


def test_attention_control_matches_reference_trace():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import fresco_amd
    import make_control_trace as mct
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "control_trace.json")))
    got = mct.run(fresco_amd.AttentionControl)
    assert got == want


class _Blk(torch.nn.Module):
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, hidden_states, temb=None):
        return hidden_states + self.k


class _UNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.up_blocks = torch.nn.ModuleList([_Blk(1.0), _Blk(10.0), _Blk(100.0)])

    def forward(self, sample, timestep, encoder_hidden_states=None, return_dict=True):
        for b in self.up_blocks:
            sample = b(hidden_states=sample, temb=None)
        return (sample,) if not return_dict else {"sample": sample}


class _Pipe:
    def __init__(self):
        self.unet = _UNet()


def test_forward_hook_semantics(monkeypatch):
    import fresco_amd
    from fresco_amd import hook
    calls = []

    def fake_opt(sample, flows, occs, corr, intra_weight, iters, optimize_temporal=True):
        calls.append(("opt", float(sample[0]), intra_weight, iters))
        return sample * 2

    def fake_warp(sample, flows, occs, saliency, chunk):
        calls.append(("warp", float(sample[0]), chunk))
        return sample + 0.5

    monkeypatch.setattr(hook._opt, "optimize_feature", fake_opt)
    monkeypatch.setattr(hook._warp, "warp_tensor", fake_warp)
    pipe = _Pipe()
    x = torch.zeros(1)
    steps = torch.tensor([900, 800])
    fresco_amd.apply_FRESCO_opt(pipe, steps=steps, layers=[0, 2], flows="f", occs="o", correlation_matrix=["c"],
                                intra_weight=7.0, iters=3, saliency="s")
    out = pipe.unet(x, torch.tensor(800), return_dict=False)
    # layer 0: (0*2+0.5)+1 = 1.5 ; layer 1 untouched: 11.5 ; layer 2: (11.5*2+0.5)+100 = 123.5
    assert float(out[0]) == 123.5
    assert [float(t) for t in out[1:]] == [0.0, 11.5]  # up_samples = inputs of layers 0 and 2 BEFORE optimisation
    assert calls == [("opt", 0.0, 7.0, 3), ("warp", 0.0, 2), ("opt", 11.5, 7.0, 3), ("warp", 23.0, 2)]
    calls.clear()
    out = pipe.unet(x, torch.tensor(700), return_dict=False)  # timestep not in steps
    assert float(out[0]) == 111.0 and len(out) == 3 and not calls
    assert isinstance(pipe.unet(x, 800), dict)  # return_dict=True: output untouched
    # no saliency -> no warp; re-applying replaces (does not stack) the hooks
    fresco_amd.apply_FRESCO_opt(pipe, steps=[800], layers=[1], flows="f", occs="o", correlation_matrix=["c"])
    calls.clear()
    out = pipe.unet(x, 800, return_dict=False)
    assert float(out[0]) == 112.0 and [c[0] for c in calls] == ["opt"] and len(out) == 2
    fresco_amd.disable_FRESCO_opt(pipe)
    calls.clear()
    out = pipe.unet(x, 800, return_dict=False)
    assert float(out[0]) == 111.0 and not calls and len(out) == 1 + 3  # default layers [0,1,2,3] -> 3 blocks here


def test_capi_argument_validation_without_gpu():
    """Argument checks run before any launch, so the error codes can be exercised on the CPU with
    dummy non-null pointers (nothing is dereferenced)."""
    from fresco_amd import _lib
    lib = _lib.load()
    p = 4096  # fake, aligned, never touched
    EINVAL, EUNSUP, EWS = -1, -2, -3
    ws = lib.fresco_attn_workspace_bytes(2, 8, 100, 40)
    ok_args = dict(B=4, H=8, Lq=64, D=40, G=2, M=100, rows=200)

    def attn(q=p, k=p, v=p, out=p, w=p, wsb=ws, B=4, H=8, Lq=64, D=40, G=2, M=100, rows=200, scale=0.1):
        return lib.fresco_attn_fwd(q, k, v, None, out, w, wsb, B, H, Lq, D, G, M, rows, scale, 0.0, None)

    assert attn(q=None) == EINVAL and attn(out=None) == EINVAL and attn(w=None) == EINVAL
    assert attn(B=5) == EINVAL            # batch not divisible by the number of key groups
    assert attn(scale=0.0) == EINVAL and attn(M=0) == EINVAL and attn(Lq=0) == EINVAL
    assert attn(wsb=ws - 1) == EWS
    assert attn(D=24, wsb=1 << 30) == EUNSUP    # head dims are instantiated for 8,16,32,40,64,80,96,128
    assert lib.fresco_attn_fwd_ld(p, p, p, None, p, p, ws, 4, 8, 64, 40, 2, 100, 200, 0.1, 0.0, 300, 320, None) == EINVAL
    # temporal pass: a trajectory's rows beyond the LDS, head dim without an instantiation, bad sharding
    assert lib.fresco_temporal_attn(p, p, p, p, p, p, 2, 2000, 64, 8, 40, 0.1, None) == EUNSUP
    assert lib.fresco_temporal_attn(p, p, p, p, p, p, 2, 8, 64, 8, 24, 0.1, None) == EUNSUP
    assert lib.fresco_temporal_attn(p, p, p, None, p, p, 2, 8, 64, 8, 40, 0.1, None) == EINVAL
    assert lib.fresco_temporal_attn_packed(None, p, p, 2, 8, 64, 8, 40, 0.1, None) == EINVAL
    assert lib.fresco_temporal_pack(p, p, p, p, p, 2, 4, 0, 64, 320, 3, 320, 320, 320, None) == EINVAL  # 64 % 3
    assert lib.fresco_temporal_unpack(p, p, None, 2, 4, 0, 64, 320, 2, None) == EINVAL
    # warp chain needs two frames; dilate needs an odd kernel; AdaIN needs >= 2 elements per row and a known dtype
    assert lib.fresco_warp_fuse_chain(p, p, p, p, p, p, p, p, 2, 1, 4, 8, 8, None) == EUNSUP
    assert lib.fresco_dilate(p, p, 1, 8, 8, 4, None) == EINVAL
    assert lib.fresco_adain(p, p, p, 4, 1, 1e-5, 1.0, 0, None) == EINVAL
    assert lib.fresco_adain(p, p, p, 4, 16, 1e-5, 1.0, 7, None) == EUNSUP
    assert lib.fresco_flow_warp(p, p, p, 1, 1, 4, 4, 1, None) == EINVAL   # in-place warp is refused
    # feature optimisation: flows without occlusions, no active term, workspace too small
    need = lib.fresco_opt_workspace_bytes(2, 4, 16, 8, 8, 1, 1)
    run = lambda fwd_flow=p, bwd_flow=p, fwd_occ=p, bwd_occ=p, target=p, wsb=need, iw=100.0: lib.fresco_opt_run(
        p, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, p, wsb, 2, 4, 16, 8, 8, iw, 3, 0.2, 0.9, 0.999, 1e-8, None)
    assert run(fwd_occ=None) == EINVAL
    assert run(fwd_flow=None, bwd_flow=None, fwd_occ=None, bwd_occ=None, target=None) == EINVAL
    assert run(fwd_flow=None, bwd_flow=None, fwd_occ=None, bwd_occ=None, iw=0.0) == EINVAL
    assert run(wsb=need - 1) == EWS


def test_standin_unet_has_the_sd15_shapes():
    """tools/standin_unet.py (full-step measurement only): parameter counts and the tensors entering the four
    up-blocks must be SD-1.5's, and the hook of apply_FRESCO_opt must see them."""
    import sys
    import types

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from standin_unet import ControlNet, UNet
    import fresco_amd

    unet = UNet().eval()
    assert round(sum(p.numel() for p in unet.parameters()) / 1e6, 1) == 859.5
    assert round(sum(p.numel() for p in ControlNet().parameters()) / 1e6, 1) == 361.3
    assert len(unet.fresco_self_attentions()) == 6
    assert [a.to_q.in_features for a in unet.fresco_self_attentions()] == [640] * 3 + [320] * 3
    pipe = types.SimpleNamespace(unet=unet)
    fresco_amd.apply_FRESCO_opt(pipe)  # = disable_FRESCO_opt: collects the decoder features, never optimises
    x, ctx = torch.randn(2, 4, 8, 8), torch.randn(2, 77, 768)
    with torch.no_grad():
        out = unet(x, 900, ctx, return_dict=False)
    assert tuple(out[0].shape) == (2, 4, 8, 8)
    assert [tuple(t.shape[1:]) for t in out[1:]] == [(1280, 1, 1), (1280, 2, 2), (1280, 4, 4), (640, 8, 8)]


def test_processor_projects_only_the_selected_keys_in_cross_frame_only_mode(monkeypatch):
    """Host plumbing of FRESCOAttnProcessor2_0 (no GPU: the operators are recorders): with the temporal pass off, K and V
    are projected for the rows the cross-frame pass gathers only -- in the order it enumerates them (reference
    src/diffusion_hacked.py:225-247) -- and the kernel is addressed without a row table; with the temporal pass on, all
    rows are projected and the row table is passed."""
    import fresco_amd
    from fresco_amd import ops

    N, chunk, HW, C, heads = 3, 2, 8, 16, 2
    attn = torch.nn.Module()
    for n in ("to_q", "to_k", "to_v"):
        setattr(attn, n, torch.nn.Linear(C, C, bias=False))
    attn.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Identity()])
    attn.heads, attn.spatial_norm, attn.group_norm, attn.norm_cross = heads, None, None, False
    attn.residual_connection, attn.rescale_output_factor = False, 1.0
    calls = []

    def rec_attention(q, k, v, h, scale, **kw):
        calls.append(("attn", tuple(q.shape), tuple(k.shape), tuple(v.shape),
                      {a: (b.clone() if torch.is_tensor(b) else b) for a, b in kw.items() if a != "workspace"}))
        return torch.zeros_like(q)

    def rec_temporal(q, k, v, fm, tm, h, scale, ch):
        calls.append(("temporal", tuple(k.shape)))
        return torch.zeros_like(q)

    monkeypatch.setattr(ops, "attention", rec_attention)
    monkeypatch.setattr(ops, "temporal_attention", rec_temporal)
    ctrl = fresco_amd.AttentionControl()
    proc = fresco_amd.FRESCOAttnProcessor2_0(chunk, ctrl)
    x = torch.randn(chunk * N, HW, C)
    mask = torch.zeros(N, HW, dtype=torch.bool)
    mask[0] = True
    mask[1, [2, 5]] = True
    mask[2, [7]] = True
    rows = mask.reshape(-1).nonzero().squeeze(1)
    ctrl.enable_cfattn([mask])
    with torch.no_grad():
        proc(attn, x)
    (tag, qs, ks, vs, kw), = calls
    M = int(mask.sum())
    assert tag == "attn" and qs == (chunk * N, HW, C) and ks == vs == (chunk, M, C)
    assert kw["n_groups"] == chunk and kw["M"] == M and kw["group_rows"] == M and kw.get("kv_rows") is None
    # same rows, same order, same arithmetic as projecting everything and gathering afterwards
    k_all = attn.to_k(x).view(chunk, N * HW, C)[:, rows]
    calls.clear()
    real_k = []
    monkeypatch.setattr(ops, "attention", lambda q, k, v, h, s, **kw: (real_k.append(k), torch.zeros_like(q))[1])
    with torch.no_grad():
        proc(attn, x)
    assert torch.allclose(real_k[0].float(), k_all.detach().half().float(), atol=2e-3)
    # temporal pass on: every row is projected, the cross-frame pass gets the row table
    monkeypatch.setattr(ops, "attention", rec_attention)
    ctrl.enable_interattn(dict(fwd_mappings=[torch.arange(HW).repeat(N, 1).unsqueeze(1)],
                               bwd_mappings=[torch.arange(HW).repeat(N, 1).unsqueeze(1)],
                               interattn_masks=[torch.ones(HW, 1, N, N, dtype=torch.bool)]))
    calls.clear()
    with torch.no_grad():
        proc(attn, x)
    assert [c[0] for c in calls] == ["attn", "temporal"]
    assert calls[0][2] == (chunk * N, HW, C) and torch.equal(calls[0][4]["kv_rows"].long(), rows)
    assert calls[1][1] == (chunk * N, HW, C)


def test_bench_workload_helpers():
    """bench.py's synthetic FRESCO parameters are valid inputs (trajectory maps = permutations with their inverses, frame 0
    of the cross-frame mask all True, symmetric trajectory masks with a True diagonal), the schedule is the reference's
    15-step mix (src/pipe_FRESCO.py:166-174), and the fabric-byte arithmetic of the frame-sharded run is what DESIGN.md
    section 6 states."""
    import bench

    assert len(bench.SCHEDULE) == 15 and bench.SCHEDULE.count("full") == 1
    assert bench.SCHEDULE.count("cf_temporal") == 7 and bench.SCHEDULE.count("cf") == 7
    N, side = 4, 8
    HW = side * side
    fwd, bwd, tmask, cfm = bench.synth_params(N, side, torch.Generator().manual_seed(0), 0.1)
    assert fwd.shape == bwd.shape == (N, 1, HW) and tmask.shape[0] == HW and cfm.shape == (N, HW)
    ar = torch.arange(HW)
    for f in range(N):
        assert torch.equal(fwd[f, 0].sort().values, ar)            # a permutation ...
        assert torch.equal(fwd[f, 0][bwd[f, 0]], ar)               # ... and its inverse
    assert bool(cfm[0].all()) and cfm.dtype == torch.bool
    tm = tmask.reshape(HW, N, N)
    assert torch.equal(tm, tm.transpose(1, 2)) and bool(tm.diagonal(dim1=1, dim2=2).all())
    # per-step fabric bytes a rank receives, 8 frames 512^2 on 8 ranks, M_rest = selected rows of the fullest rank
    out = bench.collective_bytes_per_step(8, 512, 8, {"L2": 10, "L3": 40})
    cf3 = 2 * 4096 * 2 * 320 * 2 + 2 * 7 * 40 * 2 * 320 * 2
    a2a3 = 7 / 8 * (2 * 1 * 4096) * (4 * 320) * 2
    assert out["L3"] == dict(cross_frame=cf3, temporal_all_to_all=int(a2a3))
    cf2 = 2 * 1024 * 2 * 640 * 2 + 2 * 7 * 10 * 2 * 640 * 2
    a2a2 = 7 / 8 * (2 * 1 * 1024) * (4 * 640) * 2
    assert out["per_step_mean"] == int(3 * (cf3 + a2a3 * 8 / 15) + 3 * (cf2 + a2a2 * 8 / 15))
    # one GPU: nothing crosses the fabric
    assert bench.collective_bytes_per_step(8, 512, 1, {"L2": 0, "L3": 0})["per_step_mean"] == 0
    # the PMC summary the roofline's `traffic` is read from parses to a byte count
    t = bench.pmc_traffic_bytes("attn_flash_kernelILi40")
    assert t is None or (isinstance(t["bytes"], int) and t["bytes"] > 0)


def test_attention_mask_forms_and_rejections():
    """host half of the attention_mask side path (processor._mask_bias / _masked_attention, reference
    src/diffusion_hacked.py:192-196): accepted mask forms become one additive (B, heads, Lk) fp32 bias; query-dependent
    masks and mismatched shapes are rejected before any kernel is reached."""
    import fresco_amd
    import synth

    proc = fresco_amd.FRESCOAttnProcessor2_0.__new__(fresco_amd.FRESCOAttnProcessor2_0)  # no library needed here
    attn = synth.FakeAttn(64, 8)
    B, Lk = 3, 10
    add = torch.zeros(B, Lk)
    add[:, -2:] = -10000.0
    for m in (add, add[:, None, :], add[:, None, None, :].expand(B, 8, 1, Lk)[:, 0]):
        b = proc._mask_bias(attn, m, Lk, B)
        assert b.shape == (B, 8, Lk) and b.dtype == torch.float32
        assert torch.equal(b[:, 3], add)
    keep = torch.ones(B, Lk, dtype=torch.bool)
    keep[:, 0] = False
    b = proc._mask_bias(attn, keep, Lk, B)
    assert float(b[0, 0, 0]) == -6.0e4 and float(b[0, 0, 1]) == 0.0
    assert float(proc._mask_bias(attn, torch.full((B, Lk), float("-inf")), Lk, B).min()) == -6.0e4  # kept inside fp16 range
    with pytest.raises(NotImplementedError):
        proc._mask_bias(attn, torch.zeros(B, 7, Lk), Lk, B)  # one row per query
    q = torch.zeros(B, 5, 64)
    with pytest.raises(ValueError):
        proc._masked_attention(q, torch.zeros(B, Lk + 1, 64), torch.zeros(B, Lk + 1, 64), 8, 0.3, b)  # Lk mismatch
    with pytest.raises(NotImplementedError):  # head dim 128: no larger kernel head dim to pad into
        proc._masked_attention(torch.zeros(B, 5, 1024), torch.zeros(B, Lk, 1024), torch.zeros(B, Lk, 1024), 8, 0.1, b)
