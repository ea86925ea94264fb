"""SURVEY.md 8d's second number and the metric's second half, on a stand-in model (TEST / BENCH infrastructure):

(1) one FULL denoising step = the FRESCO hot path embedded in a stand-in SD-1.5 UNet + ControlNet (tools/standin_unet.py:
    diffusers' module tree and shapes, random fp16 weights) + classifier-free-guidance combine + the DDPM update,
    8 frames x 512^2 (batch 16 with CFG), one MI355X, timed with (a) stock attention everywhere, (b) the reference's
    PyTorch op sequence in the six FRESCO layers (oracle/torch_path.py), (c) fresco_amd's processor, (d) (c) + feature
    optimisation / warp at the four up-block inputs (config 3);
(2) `latent_delta`: BASELINE.json's "max latent delta vs ref" over the denoising loop of src/pipe_FRESCO.py:166-228 --
    K steps of the reference's schedule (spatial+cf+temporal, then cf+temporal, then cf) from IDENTICAL initial latents,
    weights, FRESCO parameters and per-step noise, once with fresco_amd's processor + fresco_amd.step and once with the
    reference's op sequence (oracle/torch_path.processor_call + a plain-torch restatement of step(), :14-77) in the same
    six layers.  Feature optimisation is off (it is chaotic by construction).  Reported for fp16
    latents (the pipeline's dtype: ONE fp16 ulp of a latent in [2, 4) is already 1.95e-3, so any last-bit disagreement of
    the two UNet outputs shows up as >= 1 ulp) and with the scheduler arithmetic kept in fp32 (isolates the hot path);
    the yardstick beside it is the reference path against ITSELF with the six layers evaluated in fp32 (its own fp16
    rounding noise through the same stand-in network).

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
from standin_unet import ControlNet, UNet, reinit_unit_gain  # noqa: E402


class TorchPathProcessor:
    """the reference's op sequence (oracle/torch_path.processor_call) behind the processor call protocol.
    compute_dtype = torch.float32: the same sequence with the six layers evaluated in fp32 (noise yardstick)."""

    def __init__(self, ctrl_state, compute_dtype=None):
        self.s = ctrl_state
        self.dt = compute_dtype

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, **kw):
        from oracle import torch_path as TP
        s = self.s
        hw = hidden_states.shape[1]
        i = 0 if hw == s["hw"][0] else 1
        ref = s["refs"].pop(0) if s["mode"] == "full" else None
        temporal = s["mode"] in ("full", "cf_temporal")
        cast = (lambda t: t) if self.dt is None else (lambda t: None if t is None else t.to(self.dt))
        out = TP.processor_call(cast(hidden_states), cast(attn.to_q.weight), cast(attn.to_k.weight), cast(attn.to_v.weight),
                                cast(attn.to_out[0].weight), cast(attn.to_out[0].bias), attn.heads, ref=cast(ref), use_cf=True,
                                cf_mask=s["masks"][i], fwd_map=s["fwd"][i][:, 0] if temporal else None,
                                bwd_map=s["bwd"][i][:, 0] if temporal else None,
                                tmask=s["tmask"][i][:, 0] if temporal else None)
        return out.to(hidden_states.dtype)


class Sched:
    """what step() reads of diffusers' DDPMScheduler with SD-1.5's config (scaled-linear betas, 1000 training steps,
    set_timesteps(20), steps_offset 1): previous_timestep, alphas_cumprod, one"""

    def __init__(self):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.one = torch.tensor(1.0)
        self.timesteps = [951 - 50 * i for i in range(20)]

    def previous_timestep(self, t):
        return t - 50


def reference_step(sched, model_output, timestep, sample, generator):
    """src/pipe_FRESCO.py:14-77 in the reference's own torch ops (no background smoothing: saliency None), on whatever
    dtype `sample` has: 0-dim fp32 coefficients times fp16 tensors stay fp16, as in the reference."""
    prev_t = sched.previous_timestep(timestep)
    a_t = sched.alphas_cumprod[timestep]
    a_prev = sched.alphas_cumprod[prev_t] if prev_t >= 0 else sched.one
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    prev = (a_prev ** 0.5 * cur_b) / b_t * x0 + cur_a ** 0.5 * b_prev / b_t * sample
    var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
    var = (var ** 0.5) * torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                     dtype=model_output.dtype)
    return prev + var


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


class Harness:
    """stand-in UNet + ControlNet (initialised ON the GPU from a fixed seed) + the synthetic FRESCO parameters of bench.py"""

    def __init__(self, N=8, R=512, dev="cuda", seed=0, init="default"):
        """init: "default" = torch's default initialisation (rounds 4-5: a decoder with an input -> output gain of tens);
        "unit_gain" = standin_unet.reinit_unit_gain (variance-preserving: a perturbation of a layer output reaches the UNet
        output with gain ~ 1, as a trained network's would)"""
        self.N, self.R, self.dev = N, R, torch.device(dev)
        self.init = init
        B, lat = 2 * N, R // 8
        torch.manual_seed(seed)
        with torch.device(self.dev):
            self.unet = UNet().half().eval()
            self.cnet = ControlNet().half().eval()
        if init == "unit_gain":
            reinit_unit_gain(self.unet, seed + 1)
            reinit_unit_gain(self.cnet, seed + 2)
        self.perturb = None      # (layer index, amplitude): the gain probe adds +-amplitude to that FRESCO layer's output
        self.last_unet_out = None
        g = torch.Generator().manual_seed(seed)
        self.g = g
        lat0 = torch.randn(1, 4, lat, lat, generator=g).repeat(N, 1, 1, 1)  # repeat_noise: one initial latent for all frames (:152-153)
        self.latents = lat0.half().to(self.dev)
        self.ctx = torch.randn(B, 77, 768, generator=g).half().to(self.dev)
        self.cond = torch.rand(B, 3, R, R, generator=g).half().to(self.dev)
        params = {d: bench.synth_params(N, R // d, g, 0.004) for d in (8, 16)}
        self.refs = [torch.randn(B, (R // 16) ** 2, 640, generator=g).half().to(self.dev) for _ in range(3)] + \
                    [torch.randn(B, (R // 8) ** 2, 320, generator=g).half().to(self.dev) for _ in range(3)]
        d = self.dev
        self.paras = dict(fwd_mappings=[params[8][0].to(d), params[16][0].to(d)],
                          bwd_mappings=[params[8][1].to(d), params[16][1].to(d)],
                          interattn_masks=[params[8][2].to(d), params[16][2].to(d)])
        self.masks = [params[8][3].to(d), params[16][3].to(d)]
        self.layers = self.unet.fresco_self_attentions()
        self.stock = [a.processor for a in self.layers]
        self.ctrl = fresco_amd.AttentionControl()
        self.proc = fresco_amd.FRESCOAttnProcessor2_0(2, self.ctrl)
        self.st = dict(mode="cf", refs=[], masks=self.masks, hw=[(R // 8) ** 2, (R // 16) ** 2],
                       fwd=self.paras["fwd_mappings"], bwd=self.paras["bwd_mappings"], tmask=self.paras["interattn_masks"])
        self.sched = Sched()
        self.pipe = types.SimpleNamespace(unet=self.unet, scheduler=self.sched)

    def use(self, kind):
        p = {"ours": self.proc, "ref": TorchPathProcessor(self.st), "ref32": TorchPathProcessor(self.st, torch.float32)}.get(kind)
        for i, (a, s) in enumerate(zip(self.layers, self.stock)):
            base = p if p is not None else s
            a.processor = base if self.perturb is None or self.perturb[0] != i else _Perturbed(base, self.perturb[1])
        self.kind = kind

    def set_mode(self, mode):
        if self.kind == "ours":
            bench.set_mode(self.ctrl, mode, list(self.refs), self.paras, self.masks)
        else:
            self.st["mode"], self.st["refs"] = mode, list(self.refs)

    def eps(self, latents, t):
        """controlnet + unet + classifier-free guidance (pipe_FRESCO.py:176-214)"""
        x = torch.cat([latents] * 2)
        down, mid = self.cnet(x, t, self.ctx, self.cond)
        out = self.unet(x, t, self.ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid,
                        return_dict=False)[0]
        self.last_unet_out = out
        eu, et = out.chunk(2)
        return eu + 7.5 * (et - eu)

    @torch.no_grad()
    def loop(self, kind, modes, fp32_latents=False, noise_seed=1234):
        """K = len(modes) steps of pipe_FRESCO.py:166-228 from self.latents; returns the latents after every step"""
        self.use(kind)
        gen = torch.Generator(device=self.dev).manual_seed(noise_seed)
        lat = self.latents.float() if fp32_latents else self.latents.clone()
        traj = []
        for i, mode in enumerate(modes):
            t = self.sched.timesteps[i]
            self.set_mode(mode)
            e = self.eps(lat.half(), t)
            if fp32_latents:
                e = e.float()
            if kind == "ours":
                lat = fresco_amd.step(self.pipe, e, t, lat, gen)[0]
            else:
                lat = reference_step(self.sched, e, t, lat, gen)
            traj.append(lat.float().clone())
        return traj


class _Perturbed:
    """gain probe: a processor whose output is moved by +-amp per element (fixed pseudo-random signs); amp = "ulp": ONE
    element (the middle one of batch row 0) is moved to the next fp16 number -- the smallest deviation from the
    reference's layer output that an implementation which is not bit-identical to it can have"""

    def __init__(self, base, amp):
        self.base, self.amp = base, amp

    def __call__(self, attn, hidden_states, *a, **kw):
        out = self.base(attn, hidden_states, *a, **kw)
        if self.amp == "ulp":
            out = out.clone()
            flat = out.view(-1)
            i = out[0].numel() // 2 + 7
            flat[i:i + 1] = torch.nextafter(flat[i:i + 1], flat[i:i + 1] * 2 + 1)
            return out
        g = torch.Generator(device=out.device).manual_seed(4321)
        sgn = torch.randint(0, 2, out.shape, generator=g, device=out.device).to(out.dtype) * 2 - 1
        return out + self.amp * sgn


LOOP_MODES = ["full", "cf_temporal", "cf_temporal", "cf", "cf", "cf"]


@torch.no_grad()
def measure_repeatability(h, mode="cf"):
    """The same call twice, nothing changed, per processor kind -- and once more with PyTorch asked for deterministic
    convolution / GEMM algorithms (torch.backends.cudnn.deterministic: MIOpen on ROCm).  fresco_amd's kernels are run-to-run
    bit-identical (tests/test_gpu_opt.py, test_gpu_attention.py); whatever `ours` shows here comes from the PyTorch kernels
    of the stand-in network around the six layers."""
    t = h.sched.timesteps[0]
    lat = h.latents.float()

    def run(kind):
        h.use(kind)
        h.set_mode(mode)
        e = h.eps(lat.half(), t).float()
        gen = torch.Generator(device=h.dev).manual_seed(99)
        return h.last_unet_out.float().clone(), reference_step(h.sched, e, t, lat, gen)

    out = {}
    for det in (False, True):
        prev = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
        if det:
            torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
        try:
            for kind in ("ref", "ours"):
                a, la = run(kind)
                b, lb = run(kind)
                out["%s%s" % (kind, "_deterministic_algorithms" if det else "")] = dict(
                    unet_output_max_abs_delta=round(float((a - b).abs().max()), 7),
                    unet_output_elements_changed=int((a != b).sum()),
                    latent_max_abs_delta_one_step=round(float((la - lb).abs().max()), 7))
        finally:
            torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = prev
    h.use("stock")
    out["unet_output_elements"] = int(a.numel())
    out["note"] = ("identical inputs, identical processors, run twice: max |run 2 - run 1| of the UNet output and of the latents "
                   "after one step at t = 951.  Non-zero for `ours` (whose six layers are bit-reproducible) = the stand-in "
                   "network's own PyTorch kernels are not run-to-run deterministic on this GPU")
    return out


@torch.no_grad()
def measure_gain(h, amp=1e-3, mode="cf"):
    """How far the stand-in network carries a perturbation of ONE FRESCO layer's output (Next #4 i): the
    reference's op sequence twice on identical inputs, the second time with +-amp added to every element of layer k's output
    -> max |delta| of the UNet output (before classifier-free guidance), of the guided eps, and of the latents after one
    step at t = 951, each divided by amp.  amp = 1e-3: far enough above the fp16 grid of O(1) activations (4.9e-4) to get
    through, small enough to stay linear; fp32 latents on both sides."""
    t = h.sched.timesteps[0]
    lat = h.latents.float()

    def run():
        h.use("ref")
        h.set_mode(mode)
        e = h.eps(lat.half(), t).float()
        gen = torch.Generator(device=h.dev).manual_seed(99)
        return h.last_unet_out.float().clone(), e.clone(), reference_step(h.sched, e, t, lat, gen)

    h.perturb = None
    u0, e0, l0 = run()
    # the floor under everything below: the SAME call a second time (nothing perturbed).  PyTorch's convolution / GEMM kernels
    # around the six layers are not all run-to-run deterministic on this GPU (atomics, split-K): whatever this shows is a
    # deviation the reference path has from ITSELF
    u0b, e0b, l0b = run()
    repeat = dict(unet_output_max_abs_delta=round(float((u0b - u0).abs().max()), 7),
                  unet_output_elements_changed=int((u0b != u0).sum()), unet_output_elements=int(u0.numel()),
                  guided_eps_max_abs_delta=round(float((e0b - e0).abs().max()), 7),
                  latent_max_abs_delta_one_step=round(float((l0b - l0).abs().max()), 7))
    per_layer = []
    for k in range(len(h.layers)):
        h.perturb = (k, amp)
        u1, e1, l1 = run()
        per_layer.append(dict(layer="up_blocks.%d.attentions.%d" % (2 + k // 3, k % 3),
                              unet_output_gain=round(float((u1 - u0).abs().max()) / amp, 3),
                              unet_output_gain_rms=round(float((u1 - u0).pow(2).mean().sqrt()) / amp, 4),
                              guided_eps_gain=round(float((e1 - e0).abs().max()) / amp, 3),
                              latent_gain_one_step=round(float((l1 - l0).abs().max()) / amp, 3)))
    # the smallest possible deviation: ONE element of ONE layer output moved by one fp16 ulp
    one_ulp = []
    for k in (0, len(h.layers) - 1):
        h.perturb = (k, "ulp")
        u1, e1, l1 = run()
        one_ulp.append(dict(layer="up_blocks.%d.attentions.%d" % (2 + k // 3, k % 3),
                            unet_output_max_abs_delta=round(float((u1 - u0).abs().max()), 7),
                            unet_output_elements_changed=int((u1 != u0).sum()),
                            guided_eps_max_abs_delta=round(float((e1 - e0).abs().max()), 7),
                            latent_max_abs_delta_one_step=round(float((l1 - l0).abs().max()), 7)))
    h.perturb = None
    h.use("stock")
    worst = max(per_layer, key=lambda r: r["latent_gain_one_step"])
    return dict(amplitude=amp, per_layer=per_layer, reference_run_to_run=dict(
                    repeat, note="the reference op sequence in the six layers, the stand-in UNet + ControlNet around them, "
                                 "identical inputs, run twice: max |run 2 - run 1|"),
                one_fp16_ulp_in_one_element=dict(
                    probes=one_ulp,
                    note="ONE element (of 21 M / 10 M) of one FRESCO layer's output moved to the next fp16 number, everything "
                         "else identical: what the fp16 network itself makes of the smallest deviation an implementation that is "
                         "not bit-identical to the reference can have.  The UNet's own fp16 output grid is 9.8e-4 in [1, 2): one "
                         "flipped rounding there is 9.8e-4 x the guidance factor x the step factor in the latents"), worst_layer=worst["layer"], unet_output_gain=worst["unet_output_gain"],
                guided_eps_gain=worst["guided_eps_gain"], latent_gain_one_step=worst["latent_gain_one_step"],
                cfg_factor=round(worst["guided_eps_gain"] / max(worst["unet_output_gain"], 1e-12), 3),
                step_factor=round(worst["latent_gain_one_step"] / max(worst["guided_eps_gain"], 1e-12), 4),
                note="max |delta| / amplitude for a +-amplitude perturbation of every element of one FRESCO layer's output "
                     "(reference op sequence on both sides, identical inputs, step at t = 951): the stand-in network's own "
                     "amplification, then classifier-free guidance (eu + 7.5 (et - eu)), then the DDPM update")


@torch.no_grad()
def measure_eps_level(h, mode="cf_temporal"):
    """Step 1 of the loop, identical inputs on both sides: |ours - reference op sequence| of the UNet output BEFORE the
    classifier-free-guidance combine, of the guided eps, and of the latents after the update (Next #4 iii)"""
    t = h.sched.timesteps[0]
    lat = h.latents.float()
    got = {}
    for kind in ("ours", "ref", "ref32"):
        h.use(kind)
        h.set_mode(mode)
        e = h.eps(lat.half(), t).float()
        gen = torch.Generator(device=h.dev).manual_seed(99)
        got[kind] = (h.last_unet_out.float().clone(), e.clone(), reference_step(h.sched, e, t, lat, gen))
    h.use("stock")
    d = lambda a, b, i: float((got[a][i] - got[b][i]).abs().max())
    return dict(mode=mode,
                ours_vs_reference=dict(unet_output=round(d("ours", "ref", 0), 7), guided_eps=round(d("ours", "ref", 1), 7),
                                       latent_after_step=round(d("ours", "ref", 2), 7)),
                reference_own_fp16_noise=dict(unet_output=round(d("ref", "ref32", 0), 7), guided_eps=round(d("ref", "ref32", 1), 7),
                                              latent_after_step=round(d("ref", "ref32", 2), 7)),
                unet_output_abs_max=round(float(got["ref"][0].abs().max()), 3),
                note="fp32 scheduler arithmetic, the SAME update on all sides (isolates the six layers); "
                     "reference_own_fp16_noise = the reference op sequence vs itself with the six layers in fp32")


def measure_latent_delta(h, modes=LOOP_MODES, deterministic=False):
    """max |latent(ours) - latent(reference op sequence)| after every step, fp16 and fp32 latents, + the reference path's
    own fp16 noise (same op sequence with the six layers in fp32) as the yardstick.
    deterministic=True: the whole measurement with torch.backends.cudnn.deterministic (MIOpen's deterministic convolution
    algorithms): round 6 found that the stand-in network's PyTorch kernels are NOT run-to-run reproducible by default on this
    GPU (two identical runs of the reference path are 2.4e-3 apart at the UNet output, 0.015 in the latents after one step)
    and ARE with this switch -- only then does a difference between two paths measure the paths and not the library."""
    prev = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
    if deterministic:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        return _measure_latent_delta(h, modes, deterministic)
    finally:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = prev


def _measure_latent_delta(h, modes, deterministic):
    out = {}
    for tag, f32 in (("fp16_latents", False), ("fp32_latents", True)):
        ours = h.loop("ours", modes, f32)
        ref = h.loop("ref", modes, f32)
        ref32 = h.loop("ref32", modes, f32)
        d = [float((a - b).abs().max()) for a, b in zip(ours, ref)]
        dn = [float((a - b).abs().max()) for a, b in zip(ref, ref32)]
        d32 = [float((a - b).abs().max()) for a, b in zip(ours, ref32)]
        mean = float((ours[-1] - ref[-1]).abs().mean())
        out[tag] = dict(max_abs_delta_per_step=[round(v, 6) for v in d], max_abs_delta=round(max(d), 6),
                        ratio_to_reference_own_fp16_noise_per_step=[round(a / max(b, 1e-12), 3) for a, b in zip(d, dn)],
                        mean_abs_delta_last_step=round(mean, 8),
                        differing_elements_last_step=round(float((ours[-1] != ref[-1]).double().mean()), 6),
                        reference_own_fp16_noise_per_step=[round(v, 6) for v in dn],
                        ours_vs_reference_with_fp32_layers_per_step=[round(v, 6) for v in d32],
                        latent_abs_max=round(float(ref[-1].abs().max()), 3))
    out["steps"] = len(modes)
    out["modes"] = modes
    out["init"] = h.init
    out["pytorch_algorithms"] = "deterministic (torch.backends.cudnn.deterministic)" if deterministic else "default"
    out["standin_gain"] = measure_gain(h)
    if not deterministic:
        out["run_to_run"] = measure_repeatability(h)
    out["eps_level_step1"] = measure_eps_level(h)
    out["bar"] = ("per step: delta <= 1.5 x the reference path's own fp16 noise + 1e-3 (tests/test_gpu_latent_delta.py); the north "
                  "star's absolute 1e-3 holds per layer call (torch_gpu_baseline.max_abs_delta), not over a loop: the reference op "
                  "sequence against itself with the six layers in fp32 is 1.3e-2 apart after ONE step on this network")
    out["note"] = ("ours = FRESCOAttnProcessor2_0 + fresco_amd.step; reference = oracle/torch_path.processor_call + the "
                   "reference's step() in torch ops; same stand-in UNet + ControlNet (random fp16 weights), latents, "
                   "FRESCO parameters and noise; feature optimisation off.  fp16 latents quantise to 9.8e-4 in [1, 2) and "
                   "1.95e-3 in [2, 4): a last-bit difference of the UNet output is >= 1 ulp there; the fp32-latent rows keep "
                   "the scheduler arithmetic in fp32 on both sides.  `reference_own_fp16_noise` = the reference op sequence "
                   "against itself with the six layers in fp32")
    h.use("stock")
    return out


def measure(N=8, R=512, dev="cuda", with_opt=True, with_delta=True, verbose=False):
    h = Harness(N, R, dev)
    B = 2 * N
    g = h.g

    def one_step(t=901):
        with torch.no_grad():
            e = h.eps(h.latents, t)
            return fresco_amd.step(h.pipe, e, t, h.latents, None)[0]

    res = dict(workload="full denoising step: stand-in SD-1.5 UNet + ControlNet (random fp16 weights; everything outside "
                        "the six FRESCO layers is PyTorch / MIOpen / hipBLASLt), %d frames %dx%d, CFG batch %d, CFG combine "
                        "+ fresco_amd.step" % (N, R, R, B))
    h.use("stock")
    res["stock_attention_ms"] = round(timed(one_step), 2)
    sched = bench.SCHEDULE
    for kind, key, reps in (("ours", "fresco_amd", 3), ("ref", "reference_torch_path", 2)):
        h.use(kind)
        t_mode = {}
        for mode in ("full", "cf_temporal", "cf"):
            def run(mode=mode):
                h.set_mode(mode)
                return one_step()
            t_mode[mode] = timed(run, reps=reps)
        res[key + "_ms"] = {k: round(v, 2) for k, v in t_mode.items()}
        res[key + "_schedule_mean_ms"] = round(sum(t_mode[m] for m in sched) / len(sched), 2)
    res["full_step_speedup_vs_reference_path"] = round(res["reference_torch_path_schedule_mean_ms"]
                                                       / res["fresco_amd_schedule_mean_ms"], 2)
    res["hot_path_share_ms"] = dict(
        ours=round(res["fresco_amd_schedule_mean_ms"] - res["stock_attention_ms"], 2),
        reference=round(res["reference_torch_path_schedule_mean_ms"] - res["stock_attention_ms"], 2),
        note="difference to the same step with stock SDPA in the six layers (which itself costs ~2 ms there)")
    if with_opt:
        # (d) ours + feature optimisation / warp at the four up-block inputs (config 3: every layer, 20 iterations)
        h.use("ours")
        flows, occs, sal = bench_opt._inputs(N, R, h.dev, g)
        targets = []
        for (C, hh) in bench_opt.LAYERS:
            targets.append(ops.gram_target(torch.randn(B, C, hh * R // 512, hh * R // 512, generator=g).to(h.dev)))
        fresco_amd.apply_FRESCO_opt(h.pipe, steps=torch.tensor([901]), layers=[0, 1, 2, 3], flows=flows, occs=occs,
                                    correlation_matrix=targets, saliency=sal)

        def run_opt():
            h.set_mode("cf_temporal")
            return one_step(901)
        res["fresco_amd_with_optimisation_ms"] = round(timed(run_opt, reps=2), 2)
        fresco_amd.disable_FRESCO_opt(h.pipe)
        if "forward" in h.unet.__dict__:
            del h.unet.__dict__["forward"]
        del targets
    if with_delta:
        res["latent_delta"] = measure_latent_delta(h)
        det = measure_latent_delta(h, deterministic=True)
        res["latent_delta"]["deterministic_algorithms"] = {k: det[k] for k in ("fp32_latents", "fp16_latents", "standin_gain",
                                                                                "eps_level_step1", "pytorch_algorithms")}
        # the same six steps on a UNIT-GAIN stand-in (variance-preserving initialisation): does the north star's absolute
        # 1e-3 hold over the loop when the network itself does not amplify?
        del h
        torch.cuda.empty_cache()
        h = Harness(N, R, dev, init="unit_gain")
        ug = measure_latent_delta(h)
        ugd = measure_latent_delta(h, deterministic=True)
        res["latent_delta"]["unit_gain_standin"] = dict(
            fp32_latents=ug["fp32_latents"], fp16_latents=ug["fp16_latents"], standin_gain=ug["standin_gain"],
            run_to_run=ug["run_to_run"],
            deterministic_algorithms={k: ugd[k] for k in ("fp32_latents", "fp16_latents", "standin_gain", "eps_level_step1")},
            eps_level_step1=ug["eps_level_step1"],
            note="tools/standin_unet.reinit_unit_gain: N(0, 1 / fan_in) weights, residual branches damped to 0.3 -- the same "
                 "module tree, the same six steps, the same comparison")
    h.use("stock")
    if verbose:
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    print(json.dumps(measure(N, R)))
