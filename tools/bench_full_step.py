"""SURVEY.md 8d's second number: one FULL denoising step = the FRESCO hot path embedded in a stand-in SD-1.5 UNet +
ControlNet (tools/standin_unet.py: diffusers' module tree and shapes, random fp16 weights) + classifier-free-guidance
combine + a DDPM update, 8 frames x 512^2 (batch 16 with CFG), one MI355X.  Everything outside the hot path is
PyTorch's own conv / GEMM / SDPA code, exactly as with the real model; the number says how much of a real step the
hot path is, and what the step costs with (a) stock attention everywhere, (b) the reference's PyTorch op sequence in the
six FRESCO layers (oracle/torch_path.py, the `torch_gpu_baseline` of bench.py), (c) fresco_amd's processor, (d) (c) +
feature optimisation / warp at the four up-block inputs (config 3).

    python tools/bench_full_step.py [frames] [res]
"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402  (synthetic FRESCO parameters of the headline bench)
import bench_opt  # noqa: E402
import fresco_amd  # noqa: E402
from fresco_amd import ops  # noqa: E402
from standin_unet import ControlNet, UNet  # noqa: E402


class TorchPathProcessor:
    """the reference's op sequence (oracle/torch_path.processor_call) behind the processor call protocol"""

    def __init__(self, ctrl_state):
        self.s = ctrl_state

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, **kw):
        from oracle import torch_path as TP
        s = self.s
        hw = hidden_states.shape[1]
        i = 0 if hw == s["hw"][0] else 1
        ref = s["refs"].pop(0) if s["mode"] == "full" else None
        temporal = s["mode"] in ("full", "cf_temporal")
        return TP.processor_call(hidden_states, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                 attn.to_out[0].weight, attn.to_out[0].bias, attn.heads, ref=ref, use_cf=True,
                                 cf_mask=s["masks"][i], fwd_map=s["fwd"][i][:, 0] if temporal else None,
                                 bwd_map=s["bwd"][i][:, 0] if temporal else None,
                                 tmask=s["tmask"][i][:, 0] if temporal else None)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return 1e3 * sum(t) / len(t)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, lat = 2 * N, R // 8
    unet = UNet().to(dev).half().eval()
    cnet = ControlNet().to(dev).half().eval()
    g = torch.Generator().manual_seed(0)
    latents = torch.randn(N, 4, lat, lat, generator=g).half().to(dev)
    ctx = torch.randn(B, 77, 768, generator=g).half().to(dev)
    cond = torch.rand(B, 3, R, R, generator=g).half().to(dev)
    params = {d: bench.synth_params(N, R // d, g, 0.004) for d in (8, 16)}
    refs = [torch.randn(B, (R // 16) ** 2, 640, generator=g).half().to(dev) for _ in range(3)] + \
           [torch.randn(B, (R // 8) ** 2, 320, generator=g).half().to(dev) for _ in range(3)]
    paras = dict(fwd_mappings=[params[8][0].to(dev), params[16][0].to(dev)],
                 bwd_mappings=[params[8][1].to(dev), params[16][1].to(dev)],
                 interattn_masks=[params[8][2].to(dev), params[16][2].to(dev)])
    masks = [params[8][3].to(dev), params[16][3].to(dev)]
    fresco_layers = unet.fresco_self_attentions()
    stock = [a.processor for a in fresco_layers]

    def one_step(t=900):
        with torch.no_grad():
            x = torch.cat([latents] * 2)  # classifier-free guidance: batch (cfg_half, frame)
            down, mid = cnet(x, t, ctx, cond)
            out = unet(x, t, ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid,
                       return_dict=False)[0]
            eu, et = out.chunk(2)
            eps = eu + 7.5 * (et - eu)
            return latents - 0.1 * eps  # stand-in for the scheduler's elementwise update (pipe_FRESCO.step: ours in fresco_amd.step)

    res = dict(workload="full denoising step: stand-in SD-1.5 UNet + ControlNet (random fp16 weights), %d frames %dx%d, "
                        "CFG batch %d" % (N, R, R, B))
    # (a) stock attention everywhere
    res["stock_attention_ms"] = round(timed(one_step), 2)

    # (c) fresco_amd processor on the six decoder self-attentions, per attention mode of the schedule
    ctrl = fresco_amd.AttentionControl()
    proc = fresco_amd.FRESCOAttnProcessor2_0(2, ctrl)
    for a in fresco_layers:
        a.processor = proc
    ours = {}
    for mode in ("full", "cf_temporal", "cf"):
        def run(mode=mode):
            bench.set_mode(ctrl, mode, list(refs), paras, masks)
            return one_step()
        ours[mode] = timed(run)
    sched = bench.SCHEDULE
    res["fresco_amd_ms"] = {k: round(v, 2) for k, v in ours.items()}
    res["fresco_amd_schedule_mean_ms"] = round(sum(ours[m] for m in sched) / len(sched), 2)

    # (b) the reference's PyTorch op sequence in the same six layers
    st = dict(mode="cf", refs=[], masks=masks, hw=[(R // 8) ** 2, (R // 16) ** 2],
              fwd=paras["fwd_mappings"], bwd=paras["bwd_mappings"], tmask=paras["interattn_masks"])
    tproc = TorchPathProcessor(st)
    for a in fresco_layers:
        a.processor = tproc
    theirs = {}
    for mode in ("full", "cf_temporal", "cf"):
        def run(mode=mode):
            st["mode"], st["refs"] = mode, list(refs)
            return one_step()
        theirs[mode] = timed(run, reps=2)
    res["reference_torch_path_ms"] = {k: round(v, 2) for k, v in theirs.items()}
    res["reference_torch_path_schedule_mean_ms"] = round(sum(theirs[m] for m in sched) / len(sched), 2)
    res["full_step_speedup_vs_reference_path"] = round(res["reference_torch_path_schedule_mean_ms"]
                                                       / res["fresco_amd_schedule_mean_ms"], 2)

    # (d) ours + feature optimisation / warp at the four up-block inputs (config 3: every layer, 20 iterations)
    for a in fresco_layers:
        a.processor = proc
    flows, occs, sal = bench_opt._inputs(N, R, dev, g)
    targets = []
    for (C, h) in bench_opt.LAYERS:
        targets.append(ops.gram_target(torch.randn(B, C, h * R // 512, h * R // 512, generator=g).to(dev)))
    pipe = types.SimpleNamespace(unet=unet)
    fresco_amd.apply_FRESCO_opt(pipe, steps=torch.tensor([900]), layers=[0, 1, 2, 3], flows=flows, occs=occs,
                                correlation_matrix=targets, saliency=sal)

    def run_opt():
        bench.set_mode(ctrl, "cf_temporal", list(refs), paras, masks)
        return one_step(900)
    res["fresco_amd_with_optimisation_ms"] = round(timed(run_opt, reps=2), 2)
    fresco_amd.disable_FRESCO_opt(pipe)
    for a, p in zip(fresco_layers, stock):
        a.processor = p
    res["note"] = ("the hot-path step of bench.py (2.2 ms at 8 x 512^2) is the difference between fresco_amd_ms and what "
                   "the same six layers cost otherwise; everything else in these numbers is PyTorch / MIOpen / hipBLASLt "
                   "running a random-weight model of SD-1.5's shapes")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
