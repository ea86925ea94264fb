"""Row n1 of VERDICT r04 / the second half of BASELINE.json's metric: max latent delta vs the reference path over the
denoising loop of src/pipe_FRESCO.py:166-228 -- six steps of the reference's schedule (spatial + cross-frame + temporal,
cross-frame + temporal, cross-frame only) on a stand-in SD-1.5 UNet + ControlNet (tools/standin_unet.py: diffusers'
module tree and shapes, random fp16 weights), 8 frames x 512^2, from identical latents / weights / FRESCO parameters /
per-step noise: fresco_amd's processor + fresco_amd.step against the reference's own op sequence
(oracle/torch_path.processor_call + step() restated in torch ops) in the same six layers.  Feature optimisation off.

What the bar can be.  The north star's 1e-3 holds per CALL (tests/test_gpu_fullsize.py: every element of every layer call
within 1e-3 of the fp32 oracle; bench.py `torch_gpu_baseline.max_abs_delta` 6e-5 against the reference's op sequence).
Over a LOOP it is not a property an fp16 path can have -- MEASURED in round 6 (tools/bench_full_step.py, MI355X), not argued:
  * the factors: a +-1e-3 perturbation of one FRESCO layer's output reaches the UNet output with gain 2.4 - 3.9 (default
    AND variance-preserving initialisation of the stand-in: rounds 4-5 guessed "~50"), classifier-free guidance multiplies
    it by 9 - 11 (eu + 7.5 (et - eu)), the update at t = 951 by 0.55: x 15 - 23 into the latents after one step;
  * the grid: the UNet output is fp16, 9.8e-4 apart in [1, 2) -- ONE flipped rounding there is 9.8e-4 x 10 x 0.55 = 5.4e-3
    in the latents;
  * the smallest possible deviation: ONE element (of 10 M) of the first FRESCO layer's output moved by ONE fp16 ulp,
    everything else identical and PyTorch's kernels forced deterministic, changes 110 812 of the 262 144 UNet outputs and
    the latents by 8.7e-3 after one step (the later cross-frame layers spread it, every fp16 op re-rounds it);
  * the library: with PyTorch's DEFAULT algorithms the stand-in network is not run-to-run reproducible on this GPU --
    the reference path against ITSELF, run twice, is 2.4e-3 apart at the UNet output and 0.015 in the latents (0.0 with
    torch.backends.cudnn.deterministic); fresco_amd's own kernels are bit-reproducible;
  * the yardstick: the reference's op sequence against itself with the six layers in fp32 (its own fp16 rounding noise,
    ~6e-5 per layer call) is 0.013 apart after one step and 0.05 - 0.07 after six; ours vs the reference: 0.013 - 0.015 and
    0.05 - 0.07, in default and in deterministic mode, on both initialisations.
So an absolute 1e-3 over the loop would need layer outputs BIT-IDENTICAL to the reference's -- a property the reference
does not have against itself across PyTorch versions, GPUs, or (default algorithms) two runs.  The test states what IS
attainable and what a drop-in needs: after every step our latents are no further from the reference's than the reference's
own fp16 noise (x 1.5, + 1e-3), for fp16 latents (the pipeline's dtype) and with the scheduler arithmetic in fp32, with
default and with deterministic PyTorch algorithms, on the default-initialised AND on the unit-gain stand-in; and in
deterministic mode two runs of OUR path are identical bit for bit."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _assert_ratio_bar(r, what):
    for tag in ("fp32_latents", "fp16_latents"):
        r_ = r[tag]
        for step, (d, n, d32) in enumerate(zip(r_["max_abs_delta_per_step"], r_["reference_own_fp16_noise_per_step"],
                                               r_["ours_vs_reference_with_fp32_layers_per_step"])):
            assert d <= 1.5 * n + 1e-3, (what, tag, step, d, n)
            assert d32 <= 1.5 * n + 1e-3, (what, tag, step, d32, n)


@pytest.mark.parametrize("init", ["default", "unit_gain"])
def test_latent_delta_with_deterministic_pytorch_algorithms(init):
    """the same six steps with torch.backends.cudnn.deterministic (the stand-in network is then run-to-run reproducible:
    a difference between two paths measures the paths, not the library), on both initialisations of the stand-in"""
    import bench_full_step as B
    h = B.Harness(8, 512, "cuda", init=init)
    r = B.measure_latent_delta(h, deterministic=True)
    g = r["standin_gain"]
    print("%s init, deterministic algorithms: latent delta per step (fp32 scheduler) %s | reference's own fp16 noise %s | "
          "network gain %.2f x guidance %.2f x step %.3f | one fp16 ulp in one element of the first layer -> latents %.2e"
          % (init, r["fp32_latents"]["max_abs_delta_per_step"], r["fp32_latents"]["reference_own_fp16_noise_per_step"],
             g["unet_output_gain"], g["cfg_factor"], g["step_factor"],
             g["one_fp16_ulp_in_one_element"]["probes"][0]["latent_max_abs_delta_one_step"]))
    # the harness is reproducible in this mode (reference path twice: identical) ...
    assert g["reference_run_to_run"]["latent_max_abs_delta_one_step"] == 0.0
    # ... so is our path (the six layers are bit-reproducible kernels)
    prev = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        a = h.loop("ours", B.LOOP_MODES[:2], True)
        b = h.loop("ours", B.LOOP_MODES[:2], True)
    finally:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = prev
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    _assert_ratio_bar(r, init)
    # the measured factors (the claims of the docstring / DESIGN section 2): a modest network gain, not "~50"
    assert 1.0 < g["unet_output_gain"] < 10.0 and 5.0 < g["cfg_factor"] < 15.0 and 0.4 < g["step_factor"] < 0.7


def test_latent_delta_over_six_denoising_steps():
    import bench_full_step as B
    h = B.Harness(8, 512, "cuda")
    r = B.measure_latent_delta(h)
    f32, f16 = r["fp32_latents"], r["fp16_latents"]
    print("latent delta, fp32 scheduler arithmetic: per step %s | reference's own fp16 noise %s"
          % (f32["max_abs_delta_per_step"], f32["reference_own_fp16_noise_per_step"]))
    print("latent delta, fp16 latents: per step %s | reference's own fp16 noise %s | |latent| max %.2f, %.3f %% of the "
          "elements differ after the last step" % (f16["max_abs_delta_per_step"], f16["reference_own_fp16_noise_per_step"],
                                                  f16["latent_abs_max"], 100 * f16["differing_elements_last_step"]))
    for tag, r_ in (("fp32", f32), ("fp16", f16)):
        for step, (d, n, d32) in enumerate(zip(r_["max_abs_delta_per_step"], r_["reference_own_fp16_noise_per_step"],
                                               r_["ours_vs_reference_with_fp32_layers_per_step"])):
            assert d <= 1.5 * n + 1e-3, (tag, step, d, n)
            assert d32 <= 1.5 * n + 1e-3, (tag, step, d32, n)
    assert f32["mean_abs_delta_last_step"] < 0.25 * f32["max_abs_delta"]
