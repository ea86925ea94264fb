#!/usr/bin/env python
"""Headline benchmark: FRESCO hot-path denoising steps / second on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): 8 synthetic frames at 512^2, SD-1.5 decoder
shapes, FRESCO attention only (no feature optimisation).  One STEP = the FRESCO work of one UNet pass =
3 x up_blocks.2 (HW=1024, C=640, D=80) + 3 x up_blocks.3 (HW=4096, C=320, D=40) calls of
FRESCOAttnProcessor2_0 (q/k/v/out projections included, as the processor owns them), with the attention
mode of that denoising step.  Steps cycle through the reference's 15-step schedule (20 DDPM steps,
num_warmup_steps=5; src/pipe_FRESCO.py:166-174): 1 x spatial+cross-frame+temporal, 7 x cross-frame+
temporal, 7 x cross-frame only.  Inputs are resident in HBM before the timed region.

N > 1 GPUs: the SAME 8-frame batch is sharded by frame (strong scaling).  Per layer call the ranks exchange frame 0's
fused K|V rows (broadcast) and the other frames' selected rows (all-gather) for the cross-frame pass, and the temporal
pass runs trajectory-sharded between two all-to-alls (fresco_amd/dist.py); `collective_bytes_received_per_rank` in the
JSON line states the fabric bytes.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, attn_flash_kernel<40> on the
up_blocks.3 cross-frame pass: algorithmic flop 4*B*HW*M*C per launch / mean HIP-event duration of those
launches, measured in a second, instrumented replay of the same K steps (so `value` is not perturbed).
`cpu_baseline` times the oracle (a CPU port of the reference algorithm) on rank 0's host cores on one
call per (layer, mode) and weights them by the schedule; `torch_gpu_baseline` times the same port as plain
PyTorch ops on the same GPU (the "reference PyTorch path" of the north star).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

# (multi-process GPU work on these hosts needs dmabuf IPC; set before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCHEDULE = ["full"] + ["cf_temporal"] * 7 + ["cf"] * 7
PEAK_F16_DENSE = 2.5e15  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense fp16/bf16 MFMA
LAYERS = [("L2", 640, 16)] * 3 + [("L3", 320, 8)] * 3  # (name, channels, downscale)


def synth_params(N, side, gen, occ_p):
    """Synthetic per-batch FRESCO parameters at one feature scale (no oracle involved):
    trajectory maps = per-frame cyclic shift (3,-2)*f px + 5 % random transpositions (permutations),
    trajectory masks with Bernoulli(0.1) broken links, cross-frame mask = frame 0 + Bernoulli(occ_p)."""
    HW = side * side
    ys, xs = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
    fwd = []
    for f in range(N):
        m = (((ys - 2 * f) % side) * side + ((xs + 3 * f) % side)).reshape(-1)
        if f > 0:
            nswap = HW // 20
            a = torch.randperm(HW, generator=gen)[: 2 * nswap]
            i, j = a[:nswap], a[nswap:]
            mi, mj = m[i].clone(), m[j].clone()
            m[i], m[j] = mj, mi
        fwd.append(m)
    fwd = torch.stack(fwd, 0).unsqueeze(1)
    bwd = torch.argsort(fwd, dim=2)
    tmask = torch.ones(HW, N, N, dtype=torch.bool)
    for i in range(N - 1):
        broken = torch.rand(HW, generator=gen) < 0.1
        one = torch.ones(N, N, dtype=torch.bool)
        one[: i + 1, i + 1:] = False
        one[i + 1:, : i + 1] = False
        tmask[broken] &= one
    cf = torch.rand(N, HW, generator=gen) < occ_p
    cf[0] = True
    return fwd, bwd, tmask.unsqueeze(1), cf


def build_workload(N, R, device, seed=0):
    import synth

    gen = torch.Generator().manual_seed(seed)
    # the projection weights come from torch's GLOBAL generator (nn.Linear's default init): seed it, or every rank of a
    # multi-process run projects with its own weights (found by the sharded-vs-single-GPU check of round 5: deltas of 2e-2)
    torch.manual_seed(seed)
    B = 2 * N
    params = {}
    for down in (8, 16):
        params[down] = synth_params(N, R // down, gen, 0.004)
    layers = []
    for name, C, down in LAYERS:
        HW = (R // down) ** 2
        attn = synth.FakeAttn(C, 8).to(device).half()
        hidden = torch.randn(B, HW, C, generator=gen).half()
        ref = (hidden.float() + 0.1 * torch.randn(B, HW, C, generator=gen)).half()
        layers.append(dict(name=name, C=C, HW=HW, down=down, attn=attn, hidden=hidden.to(device),
                           ref=ref.to(device), hidden_cpu=hidden, ref_cpu=ref))
    return layers, params


def make_processor(layers, params, device, shard=None):
    import fresco_amd

    ctrl = fresco_amd.AttentionControl()
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, ctrl)
    if shard is not None:
        proc.shard = shard
    refs = [l["ref"] for l in layers]
    paras = dict(fwd_mappings=[params[8][0].to(device), params[16][0].to(device)],
                 bwd_mappings=[params[8][1].to(device), params[16][1].to(device)],
                 interattn_masks=[params[8][2].to(device), params[16][2].to(device)])
    masks = [params[8][3].to(device), params[16][3].to(device)]
    return proc, ctrl, refs, paras, masks


def set_mode(ctrl, mode, refs, paras, masks):
    ctrl.disable_controller()
    ctrl.stored_attn["decoder_attn"] = refs
    if mode == "full":
        ctrl.enable_intraattn()
    if mode in ("full", "cf_temporal"):
        ctrl.enable_interattn(paras)
    ctrl.enable_cfattn(masks)


def run_step(proc, ctrl, layers, mode, refs, paras, masks):
    set_mode(ctrl, mode, refs, paras, masks)
    out = None
    for l in layers:
        out = proc(l["attn"], l["hidden_local"])
    return out


def cpu_baseline(layers, params, N, reps=3):
    """Oracle (CPU port of the reference algorithm, fp32) per (layer kind, mode): one untimed warm-up call, then the MEAN of
    `reps` timed calls (1 warm-up + >= 3 repetitions)."""
    from oracle import fresco_oracle as O

    O.USE_TORCH_SDPA = True  # dense passes through torch's fused CPU SDPA, as the reference does
    t_mode = {}
    sample = []
    warmed = False
    for mode in ("full", "cf_temporal", "cf"):
        tot = 0.0
        for l in (layers[0], layers[3]):
            a = l["attn"]
            W = [a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out[0].weight]
            W = [w.detach().float().cpu() for w in W]
            bo = a.to_out[0].bias.detach().float().cpu()
            fwd, _, tm, cfm = params[l["down"]]
            kw = dict(use_cf=True, cf_mask=cfm)
            if mode in ("full", "cf_temporal"):
                kw.update(fwd_map=fwd[:, 0], tmask=tm[:, 0])
            if mode == "full":
                kw.update(ref=l["ref_cpu"].float())
            x = l["hidden_cpu"].float()
            dts = []
            with torch.no_grad():
                for rep in range(reps + (0 if warmed else 1)):  # the very first call also warms the thread pool / allocator
                    t0 = time.perf_counter()
                    O.fresco_attention(x, W[0], W[1], W[2], W[3], bo, 8, **kw)
                    dts.append(time.perf_counter() - t0)
            if not warmed:
                dts, warmed = dts[1:], True
            tot += 3 * sum(dts) / len(dts)
        t_mode[mode] = tot
        sample.append("%s %.2fs" % (mode, tot))
    O.USE_TORCH_SDPA = False
    step_s = sum(t_mode[m] for m in SCHEDULE) / len(SCHEDULE)
    return dict(value=1.0 / step_s, unit="denoising-steps/sec", cores=torch.get_num_threads(), kind="port",
                sample="oracle.fresco_attention (fp32, torch CPU, dense passes via torch SDPA): 1 warm-up call, then the mean of "
                       "%d timed calls per (layer kind L2/L3, mode) at N=%d frames, x3 layers each, weighted by the 15-step "
                       "schedule: %s" % (reps, N, ", ".join(sample)))


def pmc_traffic_bytes(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/*.csv,
    written by tools/pmc_attn.sh): FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
    MI355X_MICROARCH.md section HBM) and both counters are in KiB.  None if no profile is committed."""
    import csv
    import glob

    import re

    def order(path):  # rNN_pmc_attn_vMM.csv -> (NN, MM): the latest round / kernel version wins
        m = re.search(r"r(\d+)_pmc_attn_v(\d+)\.csv$", path)
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_attn_*.csv")), key=order)
    if not files:
        return None
    fetch = write = None
    for r in csv.DictReader(open(files[-1])):
        if kernel_substr in r["kernel"]:
            if r["counter"] == "FETCH_SIZE":
                fetch = float(r["per_dispatch"])
            elif r["counter"] == "WRITE_SIZE":
                write = float(r["per_dispatch"])
    if fetch is None or write is None:
        return None
    return dict(bytes=int((2.0 * fetch + write) * 1024), source=os.path.basename(files[-1]),
                note="(2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch; separate --pmc passes")


def torch_gpu_baseline(layers, params, N, device, ours):
    """The reference PyTorch path on the SAME GPU: oracle/torch_path.py issues, op for op, what
    FRESCOAttnProcessor2_0.__call__ issues to PyTorch (clones, materialised K/V repeat, the dense eye mask
    handed to SDPA, rearrange+gather round trips, masked SDPA over 2*HW length-N problems, the two
    torch.cuda.empty_cache() calls) -- the baseline the north-star speed-up target is stated against.
    `lean_port` times oracle.fresco_attention instead (same numbers from fewer, leaner torch ops and no
    empty_cache), an upper bound on what plain PyTorch ops reach.  One call per (layer kind, mode), x3
    layers, schedule-weighted.  `ours(mode, layer)` returns our output for the same call: the largest
    |ours - baseline| over all (layer kind, mode) pairs at full size is reported as max_abs_delta."""
    from oracle import fresco_oracle as O
    from oracle import torch_path as TP

    O.USE_TORCH_SDPA = True
    t_seq, t_lean, delta = {}, {}, {}
    for mode in ("full", "cf_temporal", "cf"):
        tot_seq = tot_lean = 0.0
        for l in (layers[0], layers[3]):
            a = l["attn"]
            W = [a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out[0].weight]
            fwd, _, tm, cfm = params[l["down"]]
            kw = dict(use_cf=True, cf_mask=cfm.to(device))
            if mode in ("full", "cf_temporal"):
                kw.update(fwd_map=fwd[:, 0].to(device), tmask=tm[:, 0].to(device))
            if mode == "full":
                kw.update(ref=l["ref"])
            with torch.no_grad():
                for fn, acc in ((TP.processor_call, "seq"), (O.fresco_attention, "lean")):
                    dts = []
                    for rep in range(4):  # 1 warm-up run, then the FASTEST of three (the baseline gets the benefit of
                        torch.cuda.synchronize()  # the doubt: its empty_cache() calls make single samples jump 5x)
                        t0 = time.perf_counter()
                        y = fn(l["hidden"], W[0], W[1], W[2], W[3], a.to_out[0].bias, 8, **kw)
                        torch.cuda.synchronize()
                        dts.append(time.perf_counter() - t0)
                    dt = min(dts[1:])
                    if acc == "seq":
                        tot_seq += 3 * dt
                        key = "%s/%s" % ("L2" if l["down"] == 16 else "L3", mode)
                        delta[key] = float((ours(mode, l).float() - y.float()).abs().max())
                    else:
                        tot_lean += 3 * dt
                    del y
        t_seq[mode], t_lean[mode] = tot_seq, tot_lean
    O.USE_TORCH_SDPA = False
    n = len(SCHEDULE)
    seq_s = sum(t_seq[m] for m in SCHEDULE) / n
    lean_s = sum(t_lean[m] for m in SCHEDULE) / n
    fmt = lambda t: ", ".join("%s %.1f ms" % (m, 1e3 * v) for m, v in t.items())
    return dict(value=round(1.0 / seq_s, 3), unit="denoising-steps/sec", kind="port",
                timing="1 warm-up + 3 timed calls per (layer kind, mode), the FASTEST of the three counts (our own `value` "
                       "is a mean over the timed steps): the comparison errs in the baseline's favour",
                sample="oracle/torch_path.processor_call on fp16 cuda tensors = the reference's op sequence "
                       "(diffusion_hacked.py:201-385) incl. its empty_cache() calls; one call per (layer kind, "
                       "mode), x3 layers, schedule-weighted: " + fmt(t_seq),
                lean_port=dict(value=round(1.0 / lean_s, 3),
                               sample="oracle.fresco_attention (torch SDPA, no clones / masks / empty_cache): "
                                      + fmt(t_lean)),
                max_abs_delta=dict(worst=round(max(delta.values()), 6),
                                   per_call={k: round(v, 6) for k, v in delta.items()},
                                   note="|ours - torch path| on the fp16 outputs of the same full-size call "
                                        "(outputs are O(0.3); both sides round to fp16)"))


def cfg2b_large_mask(layers, N, R, device, lib):
    """SURVEY 8d's second mask setting at the up_blocks.3 shapes: Bernoulli(0.5) blocks of 32x32 px occluded in the
    frames after the first -> M ~ (1 + 0.5 (N-1)) HW cross-frame keys, the regime real clips push towards and where
    the packed key images stop fitting an XCD's L2.  Returns the flash kernel's mean HIP-event time and fraction."""
    import fresco_amd.ops as ops

    l = layers[3]
    side = R // 8
    HW = side * side
    gen = torch.Generator().manual_seed(7)
    nb = max(side // 4, 1)  # 32 px = 4 tokens at 1/8 scale
    blocks = torch.rand(N, nb, nb, generator=gen) < 0.5
    mask = blocks.repeat_interleave(side // nb, 1).repeat_interleave(side // nb, 2).reshape(N, HW)
    mask[0] = True
    rows = mask.reshape(-1).nonzero().squeeze(1).to(torch.int32).to(device)
    a = l["attn"]
    with torch.no_grad():
        q, k, v = ops.linear(l["hidden"], [a.to_q.weight, a.to_k.weight, a.to_v.weight])
        M = rows.numel()
        kw = dict(kv_rows=rows, n_groups=2, M=M, group_rows=N * HW)
        for _ in range(2):
            ops.attention(q, k, v, 8, 1.0 / math.sqrt(40), **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_enable(64)
        for _ in range(5):
            ops.attention(q, k, v, 8, 1.0 / math.sqrt(40), **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_disable()
    t = [ms for tag, d, ms in read_prof(lib, 64) if tag == 1]
    mean_s = sum(t) / len(t) * 1e-3
    flop = 4.0 * 2 * N * HW * M * 320
    return dict(workload="up_blocks.3 cross-frame pass, block-occlusion mask: M = %d keys (%.2f x HW)" % (M, M / HW),
                flash_avg_us=round(mean_s * 1e6, 1), algorithmic_tflops=round(flop / mean_s / 1e12, 1),
                frac_of_mfma_peak=round(flop / mean_s / PEAK_F16_DENSE, 4))


def cfg2c_large_logits(layers, N, R, device, lib, rows):
    """The dominant launch (up_blocks.3 cross-frame pass, the bench's key set) with large logits: q, k ~ N(0,1) per channel
    (Cauchy-Schwarz logit bound c|q||k| up to ~23 in log2 units, softmax weights spread over ten orders of magnitude).
    By the kernel's decision rule (restated on the CPU in tools/flash_regime.py) every wave still folds the scale into Q
    (limit 24 since round 3, margin measured by tools/fold_margin.py) and ~2 % keep the running-max search after tile 0;
    the same instruction stream runs ~12 % slower on these operands than on the bench's own (data-dependent clock).
    Reported as `roofline_worst_regime` because the bench's activations (random projections of N(0,1) hidden states,
    bound ~7, near-uniform softmax) are the kernel's best case; a trained checkpoint's sit in between."""
    import fresco_amd.ops as ops

    HW = (R // 8) ** 2
    gen = torch.Generator(device=device).manual_seed(11)
    q, k, v = (torch.randn(2 * N, HW, 320, generator=gen, device=device, dtype=torch.float16) for _ in range(3))
    M = rows.numel()
    kw = dict(kv_rows=rows, n_groups=2, M=M, group_rows=N * HW)
    with torch.no_grad():
        for _ in range(2):
            ops.attention(q, k, v, 8, 1.0 / math.sqrt(40), **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_enable(64)
        for _ in range(5):
            ops.attention(q, k, v, 8, 1.0 / math.sqrt(40), **kw)
        torch.cuda.synchronize()
        lib.fresco_prof_disable()
    t = [ms for tag, d, ms in read_prof(lib, 64) if tag == 1]
    mean_s = sum(t) / len(t) * 1e-3
    flop = 4.0 * 2 * N * HW * M * 320
    return dict(workload="up_blocks.3 cross-frame pass, M = %d keys, q, k ~ N(0,1) per channel (logit bound up to ~23 log2 units)" % M,
                flash_avg_us=round(mean_s * 1e6, 1), algorithmic_tflops=round(flop / mean_s / 1e12, 1),
                frac_of_mfma_peak=round(flop / mean_s / PEAK_F16_DENSE, 4))


def collective_bytes_per_step(N, R, world, M_rest):
    """Fabric bytes one rank RECEIVES per hot-path step of the schedule mix (frame-sharded run, fresco_amd/dist.py):
    cross-frame exchange on every layer call (broadcast of frame 0's fused K|V rows + all-gather of the other frames'
    selected rows, padded to the largest rank's count), trajectory all-to-all (q|k|v out, result back) while the
    temporal pass is on (8 of 15 steps)."""
    out = {}
    tot = 0.0
    for name, C, down in (("L2", 640, 16), ("L3", 320, 8)):
        HW = (R // down) ** 2
        n_loc = N // world
        cf = 2 * HW * 2 * C * 2 * (0 if world == 1 else 1)                       # frame 0, both CFG halves, K|V, fp16
        cf += 2 * (world - 1) * M_rest[name] * 2 * C * 2                         # padded selected rows of the other ranks
        a2a = (world - 1) / world * (2 * n_loc * HW) * (3 * C + C) * 2           # q|k|v out + result back
        out[name] = dict(cross_frame=int(cf), temporal_all_to_all=int(a2a))
        tot += 3 * (cf + a2a * 8.0 / 15.0)
    out["per_step_mean"] = int(tot)
    # optimize_feature (configs[2]; not part of this bench's step): neighbour-only halo exchange of the temporal term, two
    # (chunk, C, h, w) fp32 slabs per Adam iteration and rank whatever the world size (one on two ranks with one frame each;
    # rounds 1-5 all-gathered 2 * world slabs) -- x 20 iterations x 4 decoder planes per denoising step
    if world > 1:
        halo = {}
        per_step = 0
        for name, C, down in (("up0", 1280, 64), ("up1", 1280, 32), ("up2", 1280, 16), ("up3", 640, 8)):
            slab = 2 * C * (R // down) ** 2 * 4
            n = 1 if (N // world == 1 and world == 2) else 2
            halo[name] = dict(slabs_per_iteration=n, bytes_per_iteration=n * slab)
            per_step += 20 * n * slab
        halo["per_step_20_iterations"] = per_step
        out["optimize_feature_halo"] = halo
    return out


def read_prof(lib, cap):
    tags = (ctypes.c_int * cap)()
    dims = (ctypes.c_int * (4 * cap))()
    ms = (ctypes.c_float * cap)()
    n = lib.fresco_prof_read(cap, tags, dims, ms)
    return [(tags[i], tuple(dims[4 * i: 4 * i + 4]), ms[i]) for i in range(n)]


def timed_workload(N, R, device, shard_args, steps, warmup, barrier, max_over_ranks):
    """K timed steps of another batch shape on the same ranks (the N > 1 bench line's `cfg5` leg): the workload the
    frame-parallel claim of BASELINE.json rests on, next to the strong-scaling `value` of the headline batch."""
    layers, params = build_workload(N, R, device)
    shard = None
    if shard_args is not None:
        from fresco_amd.dist import FrameShard

        shard = FrameShard(N, 2, *shard_args)
    proc, ctrl, refs, paras, masks = make_processor(layers, params, device, shard)
    for l in layers:
        if shard is not None:
            sel = shard.local_batch_index().to(device)
            l["hidden_local"] = l["hidden"].index_select(0, sel).contiguous()
            l["ref_local"] = l["ref"].index_select(0, sel).contiguous()
        else:
            l["hidden_local"], l["ref_local"] = l["hidden"], l["ref"]
    refs = [l["ref_local"] for l in layers]
    with torch.no_grad():
        for mode in sorted(set(SCHEDULE)):
            run_step(proc, ctrl, layers, mode, refs, paras, masks)
        for s in range(warmup):
            run_step(proc, ctrl, layers, SCHEDULE[s % len(SCHEDULE)], refs, paras, masks)
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            run_step(proc, ctrl, layers, SCHEDULE[s % len(SCHEDULE)], refs, paras, masks)
        barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    # parity beside the timing (one GPU): the leg's own up_blocks.3 / up_blocks.2 calls at the FULL batch (B = 2N) against the
    # reference's op sequence on the same tensors (oracle/torch_path.py: checker only, after the timed region)
    parity = None
    if shard is None:
        try:
            from oracle import torch_path as TP
            parity = {}
            with torch.no_grad():
                for mode in ("cf_temporal", "cf"):
                    set_mode(ctrl, mode, refs, paras, masks)
                    for l in (layers[0], layers[3]):
                        a = l["attn"]
                        fwd, _, tm, cfm = params[l["down"]]
                        kw = dict(use_cf=True, cf_mask=cfm.to(device))
                        if mode == "cf_temporal":
                            kw.update(fwd_map=fwd[:, 0].to(device), tmask=tm[:, 0].to(device))
                        ours = proc(a, l["hidden_local"])
                        want = TP.processor_call(l["hidden"], a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out[0].weight,
                                                 a.to_out[0].bias, 8, **kw)
                        parity["%s/%s" % ("L2" if l["down"] == 16 else "L3", mode)] = round(
                            float((ours.float() - want.float()).abs().max()), 6)
                        del ours, want
            parity = dict(max_abs_delta=max(parity.values()), per_call=parity, bar=1e-3,
                          note="|ours - the reference's op sequence| on the fp16 outputs of the leg's own calls at B = %d" % (2 * N))
        except Exception as e:  # noqa: BLE001 -- recorded, not fatal
            parity = dict(error="%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""))
    return dict(workload="cfg5's batch (BASELINE.json configs[4]): %d frames %dx%d, %s" % (N, R, R, "frame-sharded" if shard is not None else "one GPU"),
                parity=parity,
                steps=steps, value=round(steps / dt, 3), unit="denoising-steps/sec", ms_per_step=round(1e3 * dt / steps, 4),
                scaling="strong",
                note="companion of `value` on the batch whose per-frame work is 16x config 2's; the same leg runs at every "
                     "--gpus N (N = 1 included), so value(N) / value(1) is this workload's own scaling curve")


def rank_census(rank, world, device, backend):
    """every rank reports who it is: the line then proves that N distinct processes on N distinct devices took part"""
    import torch.distributed as dist

    me = dict(rank=rank, device=str(device), device_name=torch.cuda.get_device_name(device), pid=os.getpid(),
              uuid=str(getattr(torch.cuda.get_device_properties(device), "uuid", "")))
    seen = [None] * world
    dist.all_gather_object(seen, me)
    return dict(backend=backend, ranks_seen=len(seen), distinct_devices=len({(r["device"], r["uuid"]) for r in seen}),
                ranks=seen)


def sharded_vs_single(layers, params, N, device, shard, proc, ctrl, refs, paras, masks, rank):
    """One step per attention mode, frame-sharded over the ranks, against the SAME step evaluated on ONE GPU (rank 0 runs the
    whole batch through an unsharded processor): max |delta| over rank 0's frames of every layer call.  Every rank
    takes part in the sharded step (its collectives); only rank 0 evaluates the single-GPU form."""
    import fresco_amd

    out = {}
    sel = shard.local_batch_index().to(device)
    with torch.no_grad():
        for mode in ("full", "cf_temporal", "cf"):
            set_mode(ctrl, mode, list(refs), paras, masks)
            got = [proc(l["attn"], l["hidden_local"]) for l in layers]
            if rank == 0:
                c1 = fresco_amd.AttentionControl()
                p1 = fresco_amd.FRESCOAttnProcessor2_0(2, c1)
                # (the sharded branch projects K | V of the exchanged rows with fresco_linear; the comparator does the same --
                # the fused projection + pack of the single-GPU path rounds K / V identically but accumulates in another
                # order, 1 fp16 ulp of the output -- so that this check stays an EXACT test of the exchange logic)
                p1.fuse_kv_pack = False
                set_mode(c1, mode, [l["ref"] for l in layers], paras, masks)
                worst = 0.0
                for l, g in zip(layers, got):
                    full = p1(l["attn"], l["hidden"])
                    worst = max(worst, float((full.index_select(0, sel).float() - g.float()).abs().max()))
                out[mode] = round(worst, 6)
    return out


def exchange_timing(layers, params, N, device, shard, reps=10):
    """HIP-event time of the exchanges alone, per layer kind: the cross-frame exchange (launch -> wait) in the form the
    run uses, and the temporal pass's two all-to-alls around an empty kernel slot (pack -> all-to-all -> all-to-all ->
    unpack is timed as the two collectives only).  A bad scaling curve is then diagnosable from the JSON alone."""
    import torch.distributed as dist

    res = {}
    for l in (layers[0], layers[3]):
        C, HW = l["C"], l["HW"]
        mask = params[l["down"]][3].to(device)
        kv_loc = torch.randn(shard.B_loc, HW, 2 * C, device=device, dtype=torch.float16)
        plan = shard.cf_plan(mask, HW, device)
        Pw = HW // shard.world
        send = torch.randn(shard.world, shard.n_loc, shard.chunk, Pw, 3 * C, device=device, dtype=torch.float16)
        back = torch.randn(shard.world, shard.n_loc, shard.chunk, Pw, C, device=device, dtype=torch.float16)

        def cf():
            _, works = shard.exchange_cf(kv_loc, plan)
            for w in works:
                w.wait()

        def a2a():
            shard.all_to_all(send)
            shard.all_to_all(back)

        t = {}
        for name, fn in (("cross_frame_exchange_us", cf), ("temporal_all_to_all_pair_us", a2a)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[name] = round(1e3 * e0.elapsed_time(e1) / reps, 1)
        t["cross_frame_bytes_received"] = int(((HW if shard.rank else 0) + (shard.world - 1) * plan["Rmax"]) * shard.chunk * 2 * C * 2)
        t["all_to_all_bytes_sent"] = int((send.numel() + back.numel()) * 2 * (shard.world - 1) / shard.world)
        res[l["name"]] = t
    # optimize_feature's halo exchange (neighbour-only, two slabs each way) at the largest decoder plane: alone, and the
    # same exchange started before / finished after a stand-in for the launches it is meant to hide under (a device copy of
    # the features: ~ the HBM traffic of one prep launch) -- `halo_overlapped_extra_us` ~ 0 means the transfer is hidden
    C0, hh = 640, layers[3]["HW"]
    side = int(round(hh ** 0.5))
    cs = torch.randn(shard.B_loc, C0, side, side, device=device, dtype=torch.float32)
    scratch = torch.empty_like(cs)

    def halo():
        shard.halo_finish(shard.halo_start(cs))

    def halo_under_copy():
        hnd = shard.halo_start(cs)
        for _ in range(4):
            scratch.copy_(cs)
        shard.halo_finish(hnd)

    def copy_only():
        for _ in range(4):
            scratch.copy_(cs)

    th = {}
    for name, fn in (("halo_exchange_us", halo), ("halo_under_copies_us", halo_under_copy), ("copies_alone_us", copy_only)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        th[name] = round(1e3 * e0.elapsed_time(e1) / reps, 1)
    th["halo_bytes_received"] = int((1 if (shard.n_loc == 1 and shard.world == 2) else 2) * shard.chunk * C0 * hh * 4)
    res["optimize_feature_halo_640x%dx%d" % (side, side)] = th
    # slowest rank per entry
    for name in res:
        for k in ("cross_frame_exchange_us", "temporal_all_to_all_pair_us", "halo_exchange_us", "halo_under_copies_us",
                  "copies_alone_us"):
            if k not in res[name]:
                continue
            tt = torch.tensor([res[name][k]], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            res[name][k] = round(float(tt.item()), 1)
    hk = "optimize_feature_halo_640x%dx%d" % (side, side)
    res[hk]["halo_overlapped_extra_us"] = round(res[hk]["halo_under_copies_us"] - res[hk]["copies_alone_us"], 1)
    res["form"] = "grouped point-to-point" if shard.p2p_exchange else "broadcast + all-gather"
    res["note"] = "max over ranks of the mean of %d back-to-back exchanges, nothing overlapping them" % reps
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the timed block of --steps steps (each bracketed by barrier + synchronize, max over "
                         "ranks); `value` is steps / the MEDIAN block time, min / max / every block in the line")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true",
                    help="skip the auxiliary legs (cfg2b, cfg2c): only the timed steps and their instrumented replay "
                         "run, so that a rocprofv3 --stats average of the flash kernel is over the bench's own launches")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    # FRESCO_BENCH_BACKEND=gloo + FRESCO_BENCH_ONE_GPU=1: functional test of the multi-process path on a
    # single-GPU box (all ranks on device 0, collectives staged through the host) -- not a measurement
    one_gpu = os.environ.get("FRESCO_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("FRESCO_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import fresco_amd
    from fresco_amd import _lib

    lib = _lib.load()
    N, R = args.frames, args.res
    if N % world != 0:
        raise SystemExit("frames (%d) must be divisible by the number of GPUs (%d)" % (N, world))
    layers, params = build_workload(N, R, device)
    shard = None
    if world > 1:
        from fresco_amd.dist import FrameShard

        shard = FrameShard(N, 2, rank, world)
    proc, ctrl, refs, paras, masks = make_processor(layers, params, device, shard)
    n_loc = N // world
    for l in layers:
        if world > 1:
            sel = shard.local_batch_index().to(device)
            l["hidden_local"] = l["hidden"].index_select(0, sel).contiguous()
            l["ref_local"] = l["ref"].index_select(0, sel).contiguous()
        else:
            l["hidden_local"], l["ref_local"] = l["hidden"], l["ref"]
    refs = [l["ref_local"] for l in layers]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_eager(k0, k):
        with torch.no_grad():
            for s in range(k0, k0 + k):
                run_step(proc, ctrl, layers, SCHEDULE[s % len(SCHEDULE)], refs, paras, masks)

    # Setup, not a step: one untimed call sequence per attention mode, so that the per-batch index tables (row lists of
    # the cross-frame masks, trajectory-map checks -- constants of a batch of frames, built on first use) and every code
    # object exist before the W warm-up steps; with W < 9 the warm-up alone never reaches the cross-frame-only mode.
    with torch.no_grad():
        for mode in sorted(set(SCHEDULE)):
            run_step(proc, ctrl, layers, mode, refs, paras, masks)
    torch.cuda.synchronize()

    run_eager(0, args.warmup)
    # The timed block: EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks -- repeated --reps
    # times back to back.  The boxes of the pool differ by +- 5-8 % and a single 20-step sample cannot carry the headline
    # (VERDICT r05): `value` = steps / the MEDIAN block; every block, min and max are in the line.  host_s: the time this
    # rank's host needed to ISSUE the block (before the closing synchronize): N > 1 runs are host-bound when it nears dt.
    block_s, host_s = [], []
    for _ in range(max(args.reps, 1)):
        barrier()
        t0 = time.perf_counter()
        run_eager(0, args.steps)
        t_issue = time.perf_counter() - t0
        barrier()
        block_s.append(max_over_ranks(time.perf_counter() - t0))
        host_s.append(t_issue)
    dt = sorted(block_s)[len(block_s) // 2] if len(block_s) % 2 else 0.5 * sum(sorted(block_s)[len(block_s) // 2 - 1:len(block_s) // 2 + 1])
    host_ms_per_step = 1e3 * sorted(host_s)[len(host_s) // 2] / args.steps
    launch_mode = "eager"
    host_all = [host_ms_per_step]
    if world > 1:
        th = torch.zeros(world, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        th[rank] = host_ms_per_step
        dist.all_reduce(th, op=dist.ReduceOp.SUM)
        host_all = [float(v) for v in th.tolist()]

    # instrumented replay of the same K steps: per-launch HIP-event durations of the dominant kernel
    cap = args.steps * 64 + 64
    lib.fresco_prof_enable(cap)
    run_eager(0, args.steps)
    torch.cuda.synchronize()
    lib.fresco_prof_disable()
    recs = read_prof(lib, cap)
    B_loc = 2 * n_loc
    HW3 = (R // 8) ** 2
    M3 = int(params[8][3].sum())
    dom = [ms for tag, d, ms in recs if tag == 1 and d == (B_loc * 8, HW3, M3, 40)]
    roofline = None
    # (the committed PMC passes profile the single-GPU launch: no traffic figure for a frame shard's smaller launch)
    pmc = pmc_traffic_bytes("attn_flash_kernelILi40") if world == 1 else None
    if dom:
        flop = 4.0 * B_loc * HW3 * M3 * 320
        mean_s = sum(dom) / len(dom) * 1e-3
        ach = flop / mean_s
        roofline = dict(bound="mfma", kernel="attn_flash_kernel<40> (up_blocks.3 cross-frame pass)",
                        achieved=round(ach / 1e12, 2), peak=PEAK_F16_DENSE / 1e12, unit="TFLOP/s",
                        frac=round(ach / PEAK_F16_DENSE, 4),
                        # HBM bytes per launch from the committed PMC passes (a number, or null without a profile)
                        traffic=(pmc or {}).get("bytes"), traffic_source=(pmc or {}).get("source"),
                        traffic_note=(pmc or {}).get("note"),
                        algorithmic_bytes_per_launch=int(2 * B_loc * HW3 * 320 * 2 + 2 * 2 * M3 * 320 * 2),
                        launches=len(dom),
                        executed_flop_per_launch=flop * 1.4,
                        note="achieved / frac count ALGORITHMIC flop against the 2.5 PFLOP/s dense fp16 spec peak; the kernel "
                             "executes 1.40 x that (head dim 40 padded to 48 in QK^T and to 64 rows in PV), and its MFMA strand "
                             "ALONE (28 MFMAs per 64 x 64 block back to back, nothing else) takes ~314 us on these boxes, i.e. "
                             "the clock they sustain under matrix load caps this launch at ~0.45 (profiles/r03_attn_pipe.txt). "
                             "INPUT REGIME: the bench's activations (random projections of N(0,1) hidden states) give logit "
                             "bounds ~7 and a near-uniform softmax, the kernel's best case; `roofline_worst_regime` is the same "
                             "launch on N(0,1) q, k",
                        avg_launch_us=round(mean_s * 1e6, 2), algorithmic_flop_per_launch=flop)
    by_tag = {}
    for tag, d, ms in recs:
        key = {1: "attn_flash", 2: "kv_pack", 3: "temporal", 10: "linear"}.get(tag, str(tag)) + str(list(d))
        by_tag.setdefault(key, []).append(ms)
    kernels_us = {k: round(1e3 * sum(v) / len(v), 2) for k, v in sorted(by_tag.items())}

    if rank == 0:
        res = {
            "metric": "denoising-steps/sec (%d-frame batch, %d^2), FRESCO hot-path step" % (N, R),
            "value": round(args.steps / dt, 3),
            "unit": "denoising-steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "timing": dict(repetitions=len(block_s), statistic="median of the timed blocks (each = --steps steps, barrier + "
                           "synchronize on both sides, max over ranks)",
                           ms_per_step_all=[round(1e3 * b / args.steps, 4) for b in block_s],
                           ms_per_step_min=round(1e3 * min(block_s) / args.steps, 4),
                           ms_per_step_max=round(1e3 * max(block_s) / args.steps, 4),
                           value_min=round(args.steps / max(block_s), 3), value_max=round(args.steps / min(block_s), 3)),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d frames %dx%d, SD-1.5 decoder shapes, FRESCO attention only (3x up_blocks.2 "
                            "HW=%d C=640 + 3x up_blocks.3 HW=%d C=320 processor calls per step, projections "
                            "included), 15-step schedule 1x spatial+cf+temporal / 7x cf+temporal / 7x cf"
                            % ({(8, 512): "cfg2 (BASELINE.json configs[1])", (16, 512): "cfg4's batch (configs[3]: 16 frames)",
                                (32, 768): "cfg5's batch (configs[4]: 32 frames at 768^2)"}.get((N, R), "custom batch"),
                               N, R, R, (R // 16) ** 2, HW3),
                "cross_frame_keys_M": {"L3": M3, "L2": int(params[16][3].sum())},
                "parallelism": ("frame-shard x%d: frame 0's K|V (broadcast) + the masked rows of the other frames (all-gather) to "
                                "every rank, trajectory all-to-all for the temporal pass (RCCL); the grouped point-to-point form "
                                "of the exchange is timed beside it (`p2p_exchange`)" % world)
                               if world > 1 else "single GPU",
            },
            "roofline": roofline,
            "kernel_avg_us": kernels_us,
            "launch_mode": launch_mode,
            "ms_per_step_by_mode": {"eager": round(1e3 * dt / args.steps, 4)},
            # what every rank's HOST needed to issue one step (median block, before the closing synchronize): when it nears
            # ms_per_step the run is host-bound, not kernel- or fabric-bound
            "host_issue_ms_per_step": dict(per_rank=[round(v, 4) for v in host_all], max=round(max(host_all), 4)),
        }
        if world > 1:
            M_rest = {}
            for name, down in (("L2", 16), ("L3", 8)):
                m = params[down][3]
                cnt = [int(m[max(r * n_loc, 1):(r + 1) * n_loc].sum()) for r in range(world)]
                M_rest[name] = max(cnt)
            res["collective_bytes_received_per_rank"] = collective_bytes_per_step(N, R, world, M_rest)
            # per layer call: ONE grouped point-to-point launch for the cross-frame exchange (round 3: 1 broadcast + 1
            # all-gather), + 2 all-to-alls while the temporal pass is on (8 of 15 steps); 6 layer calls per step
            staged = backend != "nccl"
            cf_coll = sum(3 * (1 if (shard.p2p_exchange and not staged) else 1 + (1 if M_rest[n_] > 0 else 0))
                          for n_ in ("L2", "L3"))
            res["collectives_per_step"] = dict(cross_frame=cf_coll, temporal_all_to_all=12,
                                               schedule_mean=round(cf_coll + 12 * 8.0 / 15.0, 1))
        if world == 1 and not args.no_aux:
            res["cfg2b"] = cfg2b_large_mask(layers, N, R, device, lib)
            rows3 = params[8][3].reshape(-1).nonzero().squeeze(1).to(torch.int32).to(device)
            res["cfg2c_large_logits"] = cfg2c_large_logits(layers, N, R, device, lib, rows3)
            c2c = res["cfg2c_large_logits"]
            res["roofline_worst_regime"] = dict(bound="mfma", kernel=roofline["kernel"] if roofline else None,
                                                achieved=c2c["algorithmic_tflops"], peak=PEAK_F16_DENSE / 1e12, unit="TFLOP/s",
                                                frac=c2c["frac_of_mfma_peak"], avg_launch_us=c2c["flash_avg_us"],
                                                regime=c2c["workload"])
        if not args.no_cpu_baseline and world == 1:
            def ours(mode, l):
                set_mode(ctrl, mode, [l["ref_local"]], paras, masks)
                with torch.no_grad():
                    return proc(l["attn"], l["hidden_local"])

            res["torch_gpu_baseline"] = torch_gpu_baseline(layers, params, N, device, ours)
            res["speedup_vs_torch_gpu"] = round(res["value"] / res["torch_gpu_baseline"]["value"], 2)
            # BASELINE.md publishes no number for this metric (`published: {}`); the judge's round-3 instruction is to
            # report the ratio against the reference's own op sequence on the same GPU here
            res["vs_baseline"] = res["speedup_vs_torch_gpu"]
            res["vs_baseline_note"] = ("value / torch_gpu_baseline.value: the reference's PyTorch op sequence (oracle/torch_path.py, "
                                       "kind 'port', pinned against the reference's goldens) timed on this GPU in this run; "
                                       "BASELINE.md holds no published number.  north_star target 10x: not met")
            res["speedup_vs_torch_gpu_lean_port"] = round(
                res["value"] / res["torch_gpu_baseline"]["lean_port"]["value"], 2)
            res["cpu_baseline"] = cpu_baseline(layers, params, N)
            # auxiliary, NOT part of `value`: what BASELINE.json's third config adds to every denoising step
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_opt

            import bench_gmflow

            # auxiliary, NOT part of `value`: SURVEY 8 row f3 (flows + occlusions + masks + mappings, once per batch of frames)
            res["f3_gmflow"] = bench_gmflow.measure(N=N, R=R, dev=device)
            res["cfg3"] = bench_opt.measure(N=N, R=R, dev=device)
            # auxiliary, NOT part of `value`: SURVEY 8d's "(ii) full step" (the hot path inside a stand-in SD-1.5 UNet +
            # ControlNet; everything outside the six FRESCO layers is PyTorch's own code) and the metric's second half,
            # max latent delta vs the reference's op sequence over a K-step denoising loop
            if (N, R) == (8, 512) and os.environ.get("FRESCO_BENCH_FULL_STEP", "1") == "1":
                import bench_full_step

                fs = bench_full_step.measure(N=N, R=R, dev=device)
                res["latent_delta"] = fs.pop("latent_delta", None)
                res["full_step"] = fs
            # the step of BASELINE.json's configs[2] (attention + feature optimisation + warp): what a denoising step costs
            # on the 15 of 20 steps the pipeline optimises features on (run_fresco.py:232)
            tot_ms = res["ms_per_step"] + res["cfg3"]["ms_per_step"]
            res["cfg3_step"] = dict(metric="denoising-steps/sec with feature optimisation on (configs[2])",
                                    value=round(1e3 / tot_ms, 3), ms_per_step=round(tot_ms, 3),
                                    attention_ms=res["ms_per_step"], feature_opt_ms=res["cfg3"]["ms_per_step"],
                                    vs_torch_gpu=round((1e3 / res["torch_gpu_baseline"]["value"] +
                                                        res["cfg3"]["torch_gpu_baseline"]["ms_per_step"]) / tot_ms, 2))
            # the keyframe pipeline's own mix of the two step kinds (run_fresco.py:232 optimises features on
            # timesteps[:end_opt_step]; the SDEdit loop of src/pipe_FRESCO.py:166 runs timesteps[num_warmup_steps:]):
            # config_carturn.yaml:21-23 = 20 steps, 5 skipped, end_opt_step 15 -> 10 optimised + 5 plain steps executed;
            # without the warm-up skip (num_warmup_steps = 0) 15 + 5
            att, opt = res["ms_per_step"], res["cfg3"]["ms_per_step"]
            t_att = 1e3 / res["torch_gpu_baseline"]["value"]
            t_att_lean = 1e3 / res["torch_gpu_baseline"]["lean_port"]["value"]
            t_opt = res["cfg3"]["torch_gpu_baseline"]["ms_per_step"]

            def mix(n_opt, n_plain):
                ours = (n_opt * (att + opt) + n_plain * att) / (n_opt + n_plain)
                ref = (n_opt * (t_att + t_opt) + n_plain * t_att) / (n_opt + n_plain)
                lean = (n_opt * (t_att_lean + t_opt) + n_plain * t_att_lean) / (n_opt + n_plain)
                return dict(optimised_steps=n_opt, plain_steps=n_plain, ms_per_step=round(ours, 3),
                            value=round(1e3 / ours, 3), torch_gpu_ms_per_step=round(ref, 3),
                            vs_torch_gpu=round(ref / ours, 2), vs_torch_gpu_lean_port=round(lean / ours, 2))

            res["pipeline_weighted"] = dict(
                metric="denoising-steps/sec of the FRESCO hot path over the keyframe pipeline's own step mix",
                config_carturn=mix(10, 5), no_warmup_skip=mix(15, 5),
                where_the_north_star_is_met="steps with feature optimisation (configs[2]): %.1fx the reference's op sequence; "
                                            "attention-only steps (configs[1], the headline `value`): %.2fx (%.2fx vs the lean "
                                            "port) -- the >= 10x target is met on the former and NOT on the latter"
                                            % (res["cfg3_step"]["vs_torch_gpu"], res["speedup_vs_torch_gpu"],
                                               res["speedup_vs_torch_gpu_lean_port"]))
            res["vs_baseline_range"] = dict(vs_reference_op_sequence=res["speedup_vs_torch_gpu"],
                                            vs_lean_port=res["speedup_vs_torch_gpu_lean_port"],
                                            note="the reference's op sequence includes its two torch.cuda.empty_cache() calls "
                                                 "per layer call; the lean port drops them and the redundant clones")
        elif world == 1:
            res["cpu_baseline"] = None
    # ---- watchdog for everything below (started BEFORE the cfg5 leg: a rank that throws inside it would otherwise leave its
    # peers in a collective with nobody watching).  A hang prints the line as far as it got and exits; the watchdog's line
    # and the normal one are mutually exclusive.
    want_graph = os.environ.get("FRESCO_BENCH_GRAPH", "0") == "1"
    # (opt-in like the graph leg: a leg that has never run on RCCL must not be able to cost the driver's scale run its line --
    # a hang would hold the JSON back until the watchdog fires)
    want_p2p = world > 1 and backend == "nccl" and os.environ.get("FRESCO_BENCH_P2P", "0") == "1"
    fenced = want_graph or want_p2p or world > 1
    import threading

    out_lock = threading.Lock()
    printed = [False]
    stage = ["cfg5 leg"]

    def emit():
        with out_lock:
            if rank == 0 and not printed[0]:
                print(json.dumps(res), flush=True)
            printed[0] = True

    def bail():
        if rank == 0:
            res["optional_legs"] = dict(status="did not finish within the watchdog time", stage=stage[0])
        emit()
        os._exit(0)  # (a hung collective cannot be torn down from here; the eager result above is complete)

    dog = None
    if fenced:
        dog = threading.Timer(float(os.environ.get("FRESCO_BENCH_GRAPH_TIMEOUT", "120")) + (60.0 if world > 1 else 0.0), bail)
        dog.daemon = True
        dog.start()

    def all_min(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    # ---- cfg5 leg (every world size, every rank takes part): 32 frames x 768^2 sharded over the same ranks -- the workload
    # the frame-parallel claim of BASELINE.json rests on (per-frame work 16x config 2's).  It runs at N = 1 too, so that a
    # SCALE record of the default command carries this workload's own curve next to config 2's strong-scaling `value`.
    if (N, R) == (8, 512) and 32 % world == 0 and not args.no_aux:
        ok5 = 1.0
        try:
            c5 = timed_workload(32, 768, device, (rank, world) if world > 1 else None, max(args.steps // 2, 2), 1, barrier,
                                max_over_ranks)
        except Exception as e:  # noqa: BLE001 -- an auxiliary leg must not take the measurement with it
            c5 = dict(error="%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""))
            ok5 = 0.0
        # every rank agrees on the outcome before any rank moves on to the next collective (a rank that threw while its
        # peers sit in an all-to-all never gets past this line: the watchdog then prints the eager result and exits)
        if all_min(ok5) != 1.0 and "error" not in c5:
            c5 = dict(error="failed on another rank")
        if rank == 0:
            res["cfg5"] = c5
    # ---- optional legs, all AFTER the result above is complete and all fenced by one watchdog (a hang prints the eager
    # line and exits; the watchdog's line and the normal one are mutually exclusive):
    #  (o)  N > 1: the run proves itself -- every rank reports who it is (`rank_census`), one step per attention mode is
    #       evaluated sharded AND on one GPU and compared (`sharded_vs_single_gpu_max_abs_delta`), and the exchanges are
    #       timed alone per layer kind (`exchange_timing`).  Each in its own try / except: a failure is recorded in the
    #       line, it does not take the measurement with it.
    #  (i)  N > 1 on RCCL, OPT-IN (FRESCO_BENCH_P2P=1): the cross-frame exchange as ONE grouped launch of point-to-point
    #       transfers (FrameShard.p2p_exchange = True): its outputs must equal the broadcast + all-gather
    #       form's bit for bit (same rows, same kernels) and its K steps are timed beside `value`.  The grouped form has
    #       never run on RCCL on the build boxes, hence opt-in in the library and fenced here.
    #  (ii) hipGraph replay, OPT-IN (FRESCO_BENCH_GRAPH=1) until it has run once on RCCL: one graph per attention mode,
    #       captured after the eager measurement and replayed over the same K steps.  One GPU: replay is 3 % SLOWER than
    #       eager (the step is GPU-bound, eager launches run ahead).  N > 1: host time (0.84 ms per step) exceeds a
    #       rank's share of the kernels, so replay is what a production loop would use -- but capture with collectives
    #       inside is unverified.  A failure on ANY rank (agreed through an all-reduce) keeps the eager result.
    # `value` / `ms_per_step` are ALWAYS the eager figures of the default exchange form.
    if fenced:
        stage[0] = "optional legs"
        if world > 1:
            def leg(name, fn):
                stage[0] = name
                try:
                    return fn()
                except Exception as e:  # noqa: BLE001 -- recorded, not fatal
                    return dict(error="%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""))

            census = leg("rank census", lambda: rank_census(rank, world, device, backend))
            parity = leg("sharded vs single GPU", lambda: sharded_vs_single(layers, params, N, device, shard, proc, ctrl,
                                                                            refs, paras, masks, rank))
            xch = leg("exchange timing", lambda: exchange_timing(layers, params, N, device, shard))
            if rank == 0:
                res["rank_census"] = census
                res["sharded_vs_single_gpu_max_abs_delta"] = dict(
                    per_mode=parity, bar=1e-3,
                    note="one step per attention mode, sharded over the ranks, vs the same step on ONE GPU (rank 0, unsharded "
                         "processor): max |delta| over rank 0's frames of all six layer calls")
                res["exchange_timing"] = xch
        if want_p2p:
            stage[0] = "p2p exchange: parity"
            err = None
            same = 1.0
            try:
                with torch.no_grad():
                    for mode in ("full", "cf_temporal", "cf"):
                        shard.p2p_exchange = False
                        set_mode(ctrl, mode, list(refs), paras, masks)
                        a = [proc(l["attn"], l["hidden_local"]) for l in layers]
                        shard.p2p_exchange = True
                        set_mode(ctrl, mode, list(refs), paras, masks)
                        b = [proc(l["attn"], l["hidden_local"]) for l in layers]
                        same = min(same, 1.0 if all(torch.equal(x, y) for x, y in zip(a, b)) else 0.0)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            ok = all_min(0.0 if err else 1.0)
            same = all_min(same)
            p2p = dict(outputs_equal_collective_form=bool(same == 1.0))
            if ok == 1.0:
                stage[0] = "p2p exchange: timing"
                shard.p2p_exchange = True
                run_eager(0, args.warmup)
                barrier()
                t0 = time.perf_counter()
                run_eager(0, args.steps)
                barrier()
                dt_p = max_over_ranks(time.perf_counter() - t0)
                p2p.update(status="ok", value=round(args.steps / dt_p, 3), ms_per_step=round(1e3 * dt_p / args.steps, 4),
                           exchange_timing=exchange_timing(layers, params, N, device, shard),
                           note="same K steps with FrameShard.p2p_exchange = True (one grouped launch per layer call); not `value`")
                if rank == 0:
                    res["ms_per_step_by_mode"]["eager_p2p_exchange"] = p2p["ms_per_step"]
            else:
                p2p.update(status="failed on some rank%s" % (": " + err if err else ""))
            shard.p2p_exchange = False
            if rank == 0:
                res["p2p_exchange"] = p2p
        if want_graph:
            stage[0] = "graph capture"
            graphs, err = {}, None
            try:
                with torch.no_grad():
                    for mode in sorted(set(SCHEDULE)):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, capture_error_mode="thread_local"):
                            run_step(proc, ctrl, layers, mode, refs, paras, masks)
                        graphs[mode] = g
            except Exception as e:  # noqa: BLE001 -- any capture failure means eager
                err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            torch.cuda.synchronize()
            ok = all_min(0.0 if err else 1.0)
            if ok == 1.0:
                stage[0] = "graph replay"

                def run_graph(k0, k):
                    for s_ in range(k0, k0 + k):
                        graphs[SCHEDULE[s_ % len(SCHEDULE)]].replay()

                run_graph(0, args.warmup)
                barrier()
                t0 = time.perf_counter()
                run_graph(0, args.steps)
                barrier()
                dt_g = max_over_ranks(time.perf_counter() - t0)
                if rank == 0:
                    res["ms_per_step_by_mode"]["graph"] = round(1e3 * dt_g / args.steps, 4)
                    res["graph_replay"] = dict(status="ok", value=round(args.steps / dt_g, 3), ms_per_step=round(1e3 * dt_g / args.steps, 4),
                                               note="hipGraph replay of the same K steps, one graph per attention mode; not `value`")
            elif rank == 0:
                res["graph_replay"] = dict(status="capture failed on some rank%s" % (": " + err if err else ""))
        elif rank == 0 and world > 1:
            res["graph_replay"] = dict(status="not attempted (opt-in: FRESCO_BENCH_GRAPH=1)")
        if rank == 0 and world > 1 and not want_p2p:
            res["p2p_exchange"] = dict(status="not attempted (opt-in: FRESCO_BENCH_P2P=1)")
        dog.cancel()
        emit()
    elif rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
